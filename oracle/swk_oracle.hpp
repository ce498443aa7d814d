// swk_oracle.hpp — CPU ORACLE for swarmkit's task-placement hot path.
//
// ┌──────────────────────────────────────────────────────────────────────────┐
// │ TEST INFRASTRUCTURE. Not product code. Only tests/, __graft_entry__.smoke │
// │ and bench.py's cpu_baseline leg may build, load or call this.             │
// └──────────────────────────────────────────────────────────────────────────┘
//
// A single-threaded restatement of the reference algorithm (Go, moby/swarmkit
// manager/scheduler) with every unspecified iteration order fixed ("canonical
// order"):
//   * nodes iterate in ascending node index (index = first time the node ID was
//     seen by the scheduler; an ID keeps its index for ever),
//   * tasks inside a group, groups inside a tick and one-off tasks iterate in
//     the order they were enqueued,
//   * decision-tree branches iterate in the order they were first created,
//   * heap operations are Go's container/heap algorithms (stdlib, go 1.25).
// That is ONE legal execution of the Go program (Go leaves map order open).
//
// PARITY STATUS: the Go toolchain is absent from the build container, so the
// reference itself cannot be run. The oracle is pinned by the reference's own
// tests re-encoded as known-answer tests (tests/test_oracle_*.py; list in
// SURVEY.md §8c). Tie order among equal-score nodes is unpinned by any
// reference test ("parity unpinned" for ties; the reference tests accept any
// tied node).
//
// Reference files followed (under /root/reference/):
//   manager/scheduler/scheduler.go   (event handlers 254-396, tick 429-488,
//                                     taskFitNode 646-690, scheduleTaskGroup 694-748,
//                                     scheduleNTasksOnSubtree 772-825,
//                                     scheduleNTasksOnNodes 844-924, noSuitableNode 928-971)
//   manager/scheduler/nodeset.go     (tree 50-124)
//   manager/scheduler/nodeheap.go, decision_tree.go
//   manager/scheduler/pipeline.go, filter.go, nodeinfo.go
//   manager/constraint/constraint.go
//   api/genericresource/{validate,helpers,resource_management}.go
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace orc {

// api/types.proto:510-539
enum TaskState : int {
    TaskStateNew = 0, TaskStatePending = 64, TaskStateAssigned = 192, TaskStateAccepted = 256,
    TaskStatePreparing = 320, TaskStateReady = 384, TaskStateStarting = 448, TaskStateRunning = 512,
    TaskStateCompleted = 576, TaskStateShutdown = 640, TaskStateFailed = 704, TaskStateRejected = 768,
    TaskStateRemove = 800, TaskStateOrphaned = 832
};
// api/types.proto:200-220, api/specs.proto:31-44, api/types.proto:959-965
enum { NodeStatusUnknown = 0, NodeStatusDown = 1, NodeStatusReady = 2, NodeStatusDisconnected = 3 };
enum { NodeAvailabilityActive = 0, NodeAvailabilityPause = 1, NodeAvailabilityDrain = 2 };
enum { NodeRoleWorker = 0, NodeRoleManager = 1 };
enum { MountTypeBind = 0, MountTypeVolume = 1, MountTypeTmpfs = 2, MountTypeNamedPipe = 3, MountTypeCluster = 4 };
enum { PublishModeIngress = 0, PublishModeHost = 1 };

struct GenericResource {   // api.GenericResource oneof {Named, Discrete}
    bool named = false;
    std::string kind;
    std::string svalue;    // named
    int64_t ivalue = 0;    // discrete
};
using GenericList = std::vector<GenericResource>;

struct Resources {         // api.Resources (api/types.proto:68-82)
    int64_t nano_cpus = 0;
    int64_t memory_bytes = 0;
    bool generic_nil = true;   // Go nil slice vs empty slice matters in removeTask / HasEnough
    GenericList generic;
};

struct Platform { std::string arch, os; };
struct Plugin { std::string type, name; };

struct Node {              // the api.Node field subset the path reads
    std::string id;
    uint64_t meta_version = 0;
    // Spec
    bool labels_nil = true;
    std::map<std::string, std::string> labels;
    int availability = NodeAvailabilityActive;
    // Status
    int state = NodeStatusUnknown;
    std::string addr;
    int role = NodeRoleWorker;
    // Description
    bool has_description = false;
    std::string hostname;
    bool has_platform = false;
    Platform platform;
    bool has_resources = false;
    Resources resources;
    bool has_engine = false;
    bool engine_labels_nil = true;
    std::map<std::string, std::string> engine_labels;
    std::vector<Plugin> plugins;
    // Description.CSIInfo (api/types.proto NodeCSIInfo): the node's topology per CSI plugin (volumes.go:272-278)
    struct CSIInfo { std::string plugin_name; bool has_topology = false; std::map<std::string, std::string> segments; };
    std::vector<CSIInfo> csi;
};

// ---- CSI volumes: api.Volume, the field subset volumes.go / topology.go read -----------------------------------------------
enum { VolumeScopeSingleNode = 0, VolumeScopeMultiNode = 1 };
enum { VolumeSharingNone = 0, VolumeSharingReadOnly = 1, VolumeSharingOneWriter = 2, VolumeSharingAll = 3 };
enum { VolumeAvailabilityActive = 0, VolumeAvailabilityPause = 1, VolumeAvailabilityDrain = 2 };
struct Topology { std::map<std::string, std::string> segments; };
// api.VolumePublishStatus (types.proto:1337-1385): the two fields freeVolumes reads
enum { VolumePendingPublish = 0, VolumePublished = 1, VolumePendingNodeUnpublish = 2, VolumePendingUnpublish = 3 };
struct PublishStatus { std::string node_id; int state = VolumePendingPublish; };
struct Volume {
    std::string id, name, group, driver_name;   // ID, Spec.Annotations.Name, Spec.Group, Spec.Driver.Name
    int scope = VolumeScopeSingleNode, sharing = VolumeSharingNone, availability = VolumeAvailabilityActive;
    bool has_volume_info = false;
    std::string volume_id;                       // VolumeInfo.VolumeID ("" = not created by the plugin yet: the scheduler ignores it)
    std::vector<Topology> accessible;            // VolumeInfo.AccessibleTopology
    std::vector<PublishStatus> publish_status;   // PublishStatus
};
using VolumePtr = std::shared_ptr<Volume>;
struct VolumeAttachment { std::string id, source, target; };

struct PortConfig { int protocol = 0; uint32_t published_port = 0; int publish_mode = 0; };
struct Mount {
    int type = 0;
    bool read_only = false;
    std::string source, target;
    bool has_driver_config = false;   // VolumeOptions != nil && DriverConfig != nil
    std::string driver_name;
};
struct NetworkAttachment { bool has_network = false; bool has_driver_state = false; std::string driver_name; };
struct Preference { bool is_spread = false; std::string descriptor; };

struct Task {              // the api.Task field subset the path reads
    std::string id, service_id, node_id;
    mutable uint32_t service_key_cache = 0xFFFFFFFFu;   // service_key(service_id), filled on first use (task_service_key)
    bool has_spec_version = false;
    uint64_t spec_version = 0;
    int desired_state = TaskStateNew;
    int state = TaskStateNew;          // Status.State
    std::string message, err;          // Status.Message / Status.Err
    // Spec.Resources.Reservations
    bool has_reservations = false;
    Resources reservations;
    // Spec.Placement
    bool has_placement = false;
    std::vector<std::string> constraints;
    std::vector<Preference> preferences;
    std::vector<Platform> platforms;
    uint64_t max_replicas = 0;
    bool has_log_driver = false;
    std::string log_driver;
    bool has_container = false;
    std::vector<Mount> mounts;
    std::vector<NetworkAttachment> networks;
    bool has_endpoint = false;
    std::vector<PortConfig> ports;
    GenericList assigned_generic;      // AssignedGenericResources
    std::vector<VolumeAttachment> volumes;   // Volumes (written by the path: scheduler.go:677,872)
    std::vector<VolumeAttachment> chosen_prefix;   // harness only (not a field of api.Task): what chooseTaskVolumes had picked in front of a mount that found no volume
};
using TaskPtr = std::shared_ptr<Task>;
using NodePtr = std::shared_ptr<Node>;

// ---- manager/orchestrator/constraintenforcer/constraint_enforcer.go ---------------------------
// rejectNoncompliantTasks (constraint_enforcer.go:65-196) for ONE node: the ids of the tasks the enforcer would set to
// REJECTED. `tasks` = store.FindTasks(ByNodeID) in the order given (canonical: ascending task id; the reference itself
// calls the order nondeterministic, :104-109). `service_placement` maps a ServiceID to the CURRENT service spec's
// placement constraints (services[t.ServiceID] != nil, :152-161); a task of an unknown service uses its own spec.
struct ServicePlacement { bool has_placement = false; std::vector<std::string> constraints; };
std::vector<std::string> enforce_node(const Node& node, const std::vector<TaskPtr>& tasks,
                                      const std::map<std::string, ServicePlacement>& service_placement);
bool generic_has_resource(const GenericResource& res, const GenericList& resources);   // validate.go:54-85

// ---- manager/constraint/constraint.go -------------------------------------
struct Constraint {
    std::string key; int op = 0; std::string exp;   // op: 0 ==, 1 !=
    // which case of NodeMatches' switch (constraint.go:109-200) the key selects, and the label name behind a label prefix: pure
    // functions of `key`, worked out at the first node instead of at every node (-1: not yet)
    mutable int kind = -1;
    mutable std::string label;
};
bool constraint_parse(const std::vector<std::string>& env, std::vector<Constraint>* out, std::string* err);
bool constraint_match(const Constraint& c, const std::string& what);
bool node_matches(const std::vector<Constraint>& cs, const Node& n);
bool equal_fold(const std::string& a, const std::string& b);

// ---- api/genericresource ----------------------------------------------------
bool generic_has_enough(const Resources& node_avail, const GenericResource& task_res, bool* err);
void generic_claim(GenericList* node_avail, GenericList* task_assigned, const GenericList& reservations);
void generic_consume(GenericList* node_avail, const GenericList& res);
void generic_reclaim(GenericList* node_avail, const GenericList& task_assigned, const GenericList& node_res);
void generic_reclaim_resources(GenericList* node_avail, const GenericList& task_assigned);   // resource_management.go:87-117
void generic_sanitize(const GenericList& node_res, GenericList* node_avail);                  // resource_management.go:119-153

// ---- manager/scheduler/nodeinfo.go -------------------------------------------
struct HostPortSpec {
    int protocol; uint32_t port;
    bool operator<(const HostPortSpec& o) const { return protocol != o.protocol ? protocol < o.protocol : port < o.port; }
};
struct VersionedService {
    std::string service_id; uint64_t spec_version;   // zero Version when the task has none
    bool operator<(const VersionedService& o) const {
        return service_id != o.service_id ? service_id < o.service_id : spec_version < o.spec_version;
    }
};

// Go's NodeInfo is a VALUE type holding pointers/maps: copies share Tasks,
// ActiveTasksCountByService, AvailableResources, usedHostPorts, recentFailures but
// NOT ActiveTasksCount / lastCleanup / the *api.Node pointer slot. The shared_ptr
// members reproduce exactly that aliasing.
// ActiveTasksCountByService (nodeinfo.go:20): Go's map[string]int. The service ids are interned once (process-wide table,
// service_key) and the per-node map is a flat list keyed by the small integer — the same contents and the same zero value for
// a missing key as the Go map, without hashing a string three times per node and task in nodeLess / tree (which was most of
// the oracle's time per pair). Entries are never removed, like Go's `m[k]--`.
uint32_t service_key(const std::string& service_id);
const std::string& service_of_key(uint32_t key);
struct SvcCounts {
    std::vector<std::pair<uint32_t, int64_t>> v;
    int64_t get(uint32_t key) const {
        for (const auto& kv : v)
            if (kv.first == key) return kv.second;
        return 0;
    }
    int64_t& ref(uint32_t key) {
        for (auto& kv : v)
            if (kv.first == key) return kv.second;
        v.emplace_back(key, 0);
        return v.back().second;
    }
};

struct NodeInfo {
    NodePtr node;
    std::shared_ptr<std::map<std::string, TaskPtr>> tasks;
    int64_t active_tasks_count = 0;
    std::shared_ptr<SvcCounts> by_service;
    std::shared_ptr<Resources> available;
    std::shared_ptr<std::map<HostPortSpec, int>> used_ports;
    std::shared_ptr<std::map<VersionedService, std::vector<int64_t>>> recent_failures;
    int64_t last_cleanup = 0;

    bool valid() const { return bool(node); }
    int64_t svc_count(const std::string& s) const;   // by service id
    int64_t svc_count_key(uint32_t key) const;        // by interned id (hot paths intern once per task)
    bool add_task(const TaskPtr& t);
    bool remove_task(const Task& t);
    void task_failed(int64_t now, const Task& t);
    int64_t count_recent_failures(int64_t now, const Task& t) const;
    int64_t count_recent_failures_key(int64_t now, const VersionedService& vs) const;
    void cleanup_failures(int64_t now);
};
NodeInfo new_node_info(const NodePtr& n, const std::vector<TaskPtr>& tasks, const Resources& avail, int64_t now);

// ---- topology.go / volumes.go ---------------------------------------------------------
struct NodeInfo;
bool is_in_topology(bool has_top, const std::map<std::string, std::string>& top, const std::vector<Topology>& accessible);   // topology.go:23-47
// volumeSet (volumes.go:19-316). The reference walks a group's volumes in Go map order (volumes.go:250): canonical order here = the
// order in which the volumes were first added.
class VolumeSet {
  public:
    struct Usage { std::string node_id; bool read_only = false; };
    struct Info {
        VolumePtr volume;                          // volumeInfo.volume: the object of the FIRST addOrUpdateVolume (volumes.go:71 updates a copy)
        VolumePtr store;                           // the latest object: stands for the store's copy, which freeVolumes reads (volumes.go:189)
        std::map<std::string, Usage> tasks;        // task id -> usage
        std::map<std::string, int> nodes;          // node id -> reference count
        uint64_t order = 0;                        // creation ordinal (canonical group order)
    };
    void add_or_update(const VolumePtr& v);        // volumes.go:62-82
    void remove(const std::string& id);            // volumes.go:85-95
    // chooseTaskVolumes, volumes.go:101-140: false + *err = the reference's error string
    bool choose_task_volumes(const Task& task, const NodeInfo& node, std::vector<VolumeAttachment>* out, std::string* err, std::vector<VolumeAttachment>* prefix = nullptr);
    void reserve_task_volumes(const Task& task);   // volumes.go:144-154
    void reserve(const std::string& volume_id, const std::string& task_id, const std::string& node_id, bool read_only);   // :156-167
    void release(const std::string& volume_id, const std::string& task_id);                                              // :169-187
    std::string is_available_on_node(const Mount& mount, const NodeInfo& node) const;                                     // :223-257
    bool check_volume(const std::string& id, const NodeInfo& node, bool read_only) const;                                 // :261-316
    // freeVolumes, volumes.go:181-221: every PUBLISHED status of a volume on a node whose reference count is zero becomes
    // PENDING_NODE_UNPUBLISH. Returns (volume id, the nodes whose status changed) for the volumes that changed, by volume id (the reference
    // walks a Go map and writes one store update per volume: the order carries no meaning). The volume object kept here stands for the
    // store's copy (store.GetVolume, :189): it is updated, as the store's is — a second call reports nothing new.
    std::vector<std::pair<std::string, std::vector<std::string>>> free_volumes();
    const Info* info(const std::string& id) const { auto it = volumes_.find(id); return it == volumes_.end() ? nullptr : &it->second; }
    size_t size() const { return volumes_.size(); }

  private:
    std::map<std::string, Info> volumes_;
    std::map<std::string, std::vector<std::string>> by_group_;   // group -> volume ids in creation order
    std::map<std::string, std::string> by_name_;
    uint64_t next_order_ = 0;
};

// ---- filter.go / pipeline.go -----------------------------------------------------
enum FilterId { F_READY = 0, F_RESOURCE, F_PLUGIN, F_CONSTRAINT, F_PLATFORM, F_HOSTPORT, F_MAXREPLICAS, F_VOLUMES, F_COUNT };

struct Pipeline {
    struct Entry { bool enabled = false; int64_t failure_count = 0; };
    Entry checklist[F_COUNT];
    const Task* t = nullptr;
    const VolumeSet* vs = nullptr;   // VolumesFilter.vs (filter.go:383): nil = the filter is never enabled (a pipeline built outside Run)
    std::vector<Constraint> constraints;
    uint64_t process_calls = 0;   // instrumentation (pair evaluations), not reference state

    void set_task(const Task* task);
    bool process(const NodeInfo& n);
    std::string explain() const;
    bool check(int f, const NodeInfo& n) const;
};
std::string filter_explain(int f, int64_t nodes);

// ---- nodeheap.go / decision_tree.go ----------------------------------------------
using NodeLess = std::function<bool(const NodeInfo&, const NodeInfo&)>;
struct NodeMaxHeap {
    std::vector<NodeInfo> nodes;
    NodeLess less_func;
    int length = 0;
};
struct DecisionTree {
    int64_t tasks = 0;
    bool has_next = false;                                                 // next != nil
    std::vector<std::pair<std::string, std::unique_ptr<DecisionTree>>> next;   // first-created order
    NodeMaxHeap heap;
    DecisionTree* child(const std::string& v);
    std::vector<NodeInfo>& ordered_nodes(const std::function<bool(const NodeInfo&)>& meets);
};

struct Decision {
    TaskPtr old_task, new_task;
};

struct ServiceRec { bool has_spec_version = false; uint64_t spec_version = 0; };

// ---- scheduler.go ------------------------------------------------------------------
class Scheduler {
  public:
    int64_t now = 1'000'000'000'000LL;   // oracle clock (ns); tests move it explicitly

    // event handlers (scheduler.go:254-396, Run loop 175-237)
    void create_or_update_node(const NodePtr& n);
    void delete_node(const std::string& id);
    bool create_task(const TaskPtr& t);
    bool setup_task(const TaskPtr& t);   // setupTasksList, scheduler.go:88-124: a task of the store at scheduler start
    bool update_task(const TaskPtr& t);
    bool delete_task_event(const TaskPtr& t);
    void update_volume(const VolumePtr& v);   // EventUpdateVolume (scheduler.go:200-213) and setupTasksList's volumes (:70-81)
    const VolumeSet& volumes() const { return volumes_; }
    std::vector<std::pair<std::string, std::vector<std::string>>> free_volumes() { return volumes_.free_volumes(); }   // tick's deferred store.Batch(s.volumes.freeVolumes), scheduler.go:501
    Scheduler() { pipeline.vs = &volumes_; }   // Run appends the VolumesFilter (scheduler.go:132)
    Scheduler(const Scheduler&) = delete;
    Scheduler& operator=(const Scheduler&) = delete;
    void set_service(const std::string& id, const ServiceRec& rec) { services_[id] = rec; }
    void delete_service(const std::string& id) { services_.erase(id); }

    // tick() (scheduler.go:429-488) with an always-succeeding store commit.
    std::vector<Decision> tick();
    // processPreassignedTasks (scheduler.go:398-426)
    std::vector<Decision> process_preassigned();

    // nodeSet surface (nodeset.go:18-48)
    bool node_info(const std::string& id, NodeInfo* out) const;
    size_t node_count() const;
    std::vector<std::string> node_ids() const;
    DecisionTree tree(const std::string& service_id, const std::vector<Preference>& prefs, int max_assignments,
                      const std::function<bool(const NodeInfo&)>& meets, const NodeLess& less);

    Pipeline pipeline;
    uint64_t nodeless_calls = 0;

    // test hook: inject a NodeInfo verbatim (nodeset_test.go builds NodeInfo literals)
    void add_or_update_node_info(const NodeInfo& ni);

  private:
    struct Slot { bool present = false; NodeInfo info; };
    std::vector<Slot> slots_;                               // canonical node order
    std::unordered_map<std::string, size_t> slot_of_;
    std::set<size_t> free_slots_;                           // slots of removed nodes, lowest first
    std::map<std::string, ServiceRec> services_;
    VolumeSet volumes_;

    // insertion-ordered id -> task maps (Go maps with canonical iteration order)
    struct OrderedTasks {
        std::vector<std::pair<std::string, TaskPtr>> items;
        std::unordered_map<std::string, size_t> pos;
        void put(const std::string& id, const TaskPtr& t);
        void erase(const std::string& id);
        bool has(const std::string& id) const { return pos.count(id) != 0; }
        TaskPtr get(const std::string& id) const;
        size_t live() const { return pos.size(); }
        void compact();
    };
    OrderedTasks unassigned_, pending_preassigned_;
    std::map<std::string, bool> preassigned_;
    std::map<std::string, TaskPtr> all_tasks_;

    void ns_add_or_update(const NodeInfo& ni);
    void ns_update(const NodeInfo& ni);
    void enqueue(const TaskPtr& t) { unassigned_.put(t->id, t); }
    bool delete_task(const Task& t);
    TaskPtr task_fit_node(const TaskPtr& t, const std::string& node_id);
    void schedule_task_group(OrderedTasks& group, std::vector<Decision>& decisions,
                             std::unordered_map<std::string, size_t>& decided);
    int schedule_n_on_subtree(int n, OrderedTasks& group, DecisionTree* tree, std::vector<Decision>& decisions,
                              std::unordered_map<std::string, size_t>& decided, const NodeLess& less);
    int schedule_n_on_nodes(int n, OrderedTasks& group, std::vector<NodeInfo>& nodes, std::vector<Decision>& decisions,
                            std::unordered_map<std::string, size_t>& decided, const NodeLess& less);
    void no_suitable_node(OrderedTasks& group, std::vector<Decision>& decisions,
                          std::unordered_map<std::string, size_t>& decided);
    static void put_decision(std::vector<Decision>& decisions, std::unordered_map<std::string, size_t>& decided,
                             const std::string& id, Decision d);
};

}  // namespace orc
