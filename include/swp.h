/* swp.h — C ABI of libswp.so, the MI355X-native batch task-placement engine.
 *
 * Drop-in boundary for swarmkit's manager/scheduler hot path. The reference has no FFI: the seam
 * is Go-internal (Scheduler.nodeSet / Scheduler.pipeline, scheduler.go:32-51). Each entry point
 * below names the reference interface it replaces (paths under /root/reference/). A thin cgo shim
 *  keeps nodeSet's Go map as the source of truth for the non-numeric fields and
 * mirrors every mutator call into the engine.
 *
 * Conventions
 *   - returns 0 (SWP_OK) or a negative SWP_E*; no C++ exception crosses this boundary;
 *   - every buffer is caller-allocated, caller-owned and only read/written during the call
 *     (cgo pointer rules: the engine retains no caller pointer);
 *   - the engine owns all device memory; one engine per Scheduler; NOT thread-safe: the caller
 *     (the single scheduler goroutine, scheduler.go:175-237) serialises calls;
 *   - strings never reach the device: they are interned to dense uint32 ids (0 = "" / absent);
 *   - all structs are little-endian PODs with natural alignment; sizes are asserted in swp_abi_check().
 *   - There is NO CPU implementation behind this ABI: without a gfx950 device swp_create()
 *     fails with SWP_ENODEVICE and nothing else is usable.
 */
#ifndef SWP_H
#define SWP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWP_ABI_VERSION 2   /* 2: swp_config lost `window` (unused since round 3), the shard set arrived, swp_volume_upsert restates addOrUpdateVolume to the letter,
                             * a failed volume choice reports the prefix it had chosen */

enum {
    SWP_OK = 0,
    SWP_EINVAL = -1,       /* bad argument / unknown id */
    SWP_ENOTFOUND = -2,    /* errNodeNotFound, nodeset.go:12 */
    SWP_ENOMEM = -3,
    SWP_EHIP = -4,         /* HIP runtime error; swp_last_error() has the text */
    SWP_EUNSUPPORTED = -5, /* feature kept on the Go path (CSI volumes, generic reservations of whole task groups, ...) */
    SWP_ERANGE = -6,       /* value outside the engine's documented limits */
    SWP_ENODEVICE = -7     /* no gfx950 device: the engine has no CPU fallback */
};

typedef struct swp_engine swp_engine;

/* ------------------------------------------------------------------------------------------ */
/* configuration                                                                               */
typedef struct {
    int32_t  device;        /* HIP device ordinal */
    uint32_t resolver_threads; /* 0 = auto (256 or 1024 by node count) */
    uint32_t flags;         /* SWP_CFG_* */
    /* node-range shard owned by this engine for the sharded scan; [0,0) = all */
    uint32_t shard_rank, shard_count;
    uint32_t reserved[3];
} swp_config;

#define SWP_CFG_PROFILE 1u   /* record hipEvent timings per kernel class into swp_stats_t */

/* ------------------------------------------------------------------------------------------ */
/* interning — replaces every string compare on the path (constraint.go:90,109-203 EqualFold;   */
/* filter.go:283-306 platform strings; filter.go:179-202 plugin names; nodeinfo.go map keys)    */
enum {
    SWP_SPACE_NODE_ID = 0,   /* api.Node.ID → dense node index (also the canonical scan order) */
    SWP_SPACE_SERVICE = 1,   /* api.Task.ServiceID */
    SWP_SPACE_LABEL_KEY = 2, /* label names, case-sensitive (constraint.go:180-182,195-196) */
    SWP_SPACE_FOLDED = 3,    /* values compared with strings.EqualFold: label values, node.id,
                                node.hostname, node.platform.os/arch constraint operands */
    SWP_SPACE_OS = 4,        /* Platform.OS, case-sensitive (filter.go:302) */
    SWP_SPACE_ARCH = 5,      /* Platform.Architecture after x86_64→amd64, aarch64→arm64 (filter.go:285-299) */
    SWP_SPACE_PLUGIN = 6,    /* "<Type>\0<Name>" (filter.go:179-202) */
    SWP_SPACE_RAW = 7,       /* label values as written, case-sensitive: decision-tree branches (nodeset.go:84-101) */
    SWP_SPACE_GENERIC_KIND = 8, /* GenericResource kinds, case-sensitive (api/genericresource/helpers.go Kind()) */
    SWP_SPACE_VOLUME = 9,    /* api.Volume.ID -> dense volume index; ascending index = the order a group's volumes are tried in
                              * (volumes.go:250 ranges over a Go map: any order is legal, this one is the oracle's) */
    SWP_SPACE_VOLUME_GROUP = 10, /* VolumeSpec.Group */
    SWP_SPACE_CSI = 11,      /* CSI plugin names, topology subdomains and segments: compared as written (volumes.go:273, topology.go:35) */
    SWP_SPACE_COUNT = 12
};
/* id 0 is reserved for the empty string in every space except NODE_ID (node index 0 is a node). */
int swp_intern(swp_engine*, int space, const char* utf8, size_t len, uint32_t* id_out);
/* reverse lookup, for tests / debugging: copies up to cap bytes, returns the full length or <0 */
int swp_intern_lookup(swp_engine*, int space, uint32_t id, char* out, size_t cap);

/* ------------------------------------------------------------------------------------------ */
/* node rows — NodeInfo numeric mirror (nodeinfo.go:28-44) plus the api.Node fields the filters */
/* read                                                                  */
#define SWP_NODE_READY        0x001u  /* Status.State==READY && Spec.Availability==ACTIVE (filter.go:41-44) */
#define SWP_NODE_HAS_DESC     0x002u  /* Description != nil */
#define SWP_NODE_HAS_PLATFORM 0x004u  /* Description.Platform != nil */
#define SWP_NODE_HAS_ENGINE   0x008u  /* Description.Engine != nil */
#define SWP_NODE_HAS_LABELS   0x010u  /* Spec.Annotations.Labels != nil */
#define SWP_NODE_HAS_ELABELS  0x020u  /* Description.Engine.Labels != nil */
#define SWP_NODE_MANAGER      0x040u  /* Role == MANAGER (constraint.go:148 compares Role.String()) */
#define SWP_NODE_HAS_LOGPLUG  0x080u  /* Engine.Plugins lists at least one Type=="Log" (filter.go:169-175) */
#define SWP_NODE_IP_VALID     0x100u  /* net.ParseIP(Status.Addr) != nil */
#define SWP_NODE_IP_V4        0x200u  /* address is IPv4 (or IPv4-mapped) */

typedef struct {
    uint32_t node;        /* NODE_ID id */
    uint32_t flags;       /* SWP_NODE_* */
    int64_t  cpu;         /* AvailableResources.NanoCPUs   (may be negative, scheduler.go:378-379) */
    int64_t  mem;         /* AvailableResources.MemoryBytes */
    uint32_t total;       /* ActiveTasksCount */
    uint32_t os;          /* SWP_SPACE_OS id   (platform filter) */
    uint32_t arch;        /* SWP_SPACE_ARCH id (platform filter) */
    uint32_t os_fold;     /* SWP_SPACE_FOLDED id of Platform.OS            (node.platform.os constraint) */
    uint32_t arch_fold;   /* SWP_SPACE_FOLDED id of Platform.Architecture  (node.platform.arch) */
    uint32_t hostname_fold; /* SWP_SPACE_FOLDED id of Description.Hostname (node.hostname) */
    uint32_t id_fold;     /* SWP_SPACE_FOLDED id of Node.ID                (node.id) */
    uint32_t reserved;
    uint8_t  ip[16];      /* Status.Addr as 16-byte address (v4 as ::ffff:a.b.c.d) */
    uint64_t version;     /* Meta.Version.Index — echoed by swp_node_get for the stale check, scheduler.go:540 */
} swp_node_row;           /* 80 bytes */

typedef struct { uint32_t key; uint32_t value; uint32_t raw; } swp_kv;   /* (LABEL_KEY id, FOLDED id, RAW id) */

/* nodeSet.alloc, nodeset.go:18-20: drop every node row and all derived state */
int swp_reset(swp_engine*, uint32_t n_nodes_hint);
/* nodeSet.addOrUpdateNode / updateNode (nodeset.go:33-44) as called from createOrUpdateNode
 * (scheduler.go:368-396) and buildNodeSet (:973-990). Labels/plugins replace the previous ones. */
int swp_node_upsert(swp_engine*, const swp_node_row* row,
                    const swp_kv* node_labels, uint32_t n_node_labels,
                    const swp_kv* engine_labels, uint32_t n_engine_labels,
                    const uint32_t* plugins, uint32_t n_plugins);
/* numeric-only fast path of swp_node_upsert: update flags/cpu/mem/total of an existing node
 * (availability flips, resource reconciliation) without touching labels/plugins */
int swp_node_update_dynamic(swp_engine*, uint32_t node, uint32_t flags, int64_t cpu, int64_t mem, uint32_t total);
/* nodeSet.remove, nodeset.go:46-48: the node leaves the set AND its index goes back to the pool — the next node id that is new to
 * swp_intern(SWP_SPACE_NODE_ID, ...) is given the LOWEST free index (the index space, i.e. the width of every bitmap row, is bounded
 * by the nodes alive at once). The canonical tie order among equal-score nodes is the index order, so a recycled index places its
 * new node where the old one stood; Go's map iteration order is unspecified, the CPU oracle recycles its slots by the same rule.
 * A caller must therefore forget the index of a removed node (intern its id again if it comes back). */
int swp_node_remove(swp_engine*, uint32_t node);
/* Bulk forms for bursts of node events (a drain round touches 10 % of the cluster: scheduler.go:368-396 once per node): one
 * call instead of one per node. Same semantics and error behaviour as the single-node calls, applied in array order; the
 * first failing row stops the call (rows before it stay applied) and is named in swp_last_error. */
typedef struct {
    uint32_t node, flags;      /* SWP_NODE_* as in swp_node_row.flags */
    int64_t  cpu, mem;         /* AvailableResources */
    uint32_t total, reserved;  /* ActiveTasksCount */
} swp_node_dynamic;            /* 32 bytes */
int swp_node_update_dynamic_many(swp_engine*, const swp_node_dynamic* rows, uint32_t n);
int swp_node_get_many(swp_engine*, const uint32_t* nodes, uint32_t n, swp_node_row* out);
/* nodeSet.nodeInfo, nodeset.go:23-29: SWP_ENOTFOUND <-> errNodeNotFound */
int swp_node_get(swp_engine*, uint32_t node, swp_node_row* out);
/* NodeInfo.ActiveTasksCountByService[service] = count (nodeinfo.go:32) */
int swp_node_set_svc_count(swp_engine*, uint32_t node, uint32_t service, uint32_t count);
int swp_node_get_svc_count(swp_engine*, uint32_t node, uint32_t service, uint32_t* count_out);
/* countRecentFailures(now, t) for (service, specVersion) on this node (nodeinfo.go:206-221), as
 * evaluated by the shim at the `now` of the coming batch (scheduler.go:706). spec_version is 0
 * for tasks without one (the zero api.Version). */
int swp_node_set_failures(swp_engine*, uint32_t node, uint32_t service, uint64_t spec_version, uint32_t count);
/* usedHostPorts insert/delete (nodeinfo.go:78-84,139-145). protocol: TCP 0 / UDP 1 / SCTP 2 */
int swp_node_port(swp_engine*, uint32_t node, uint32_t protocol, uint32_t port, int set);

/* Generic resources (api/genericresource). What the placement decision reads of a node's AvailableResources.Generic is ONE number
 * per kind: genericresource.HasEnough (validate.go:24-52) looks at the entries of the kind — none: not enough, whatever the
 * request; the first one Discrete: its Value; Named: how many there are — and ResourceFilter.Check (filter.go:86-91) compares
 * that with the task's Discrete reservation. The caller keeps the LIST (which named values a task is given — Claim,
 * resource_management.go:11-72 — is its bookkeeping; so are Reclaim and sanitize after a node update, :75-153) and mirrors the
 * counts: after every change of a node's list it passes the kinds that are present with their count; kinds not named are absent
 * (count 0: a request of at least 1 never fits, as for a kind the node does not offer). Inside a batch the engine does the
 * arithmetic of Claim itself: count -= request, an entry that reaches 0 is gone (helpers.go:87-111 remove()). */
typedef struct {
    uint32_t kind;       /* SWP_SPACE_GENERIC_KIND id */
    uint32_t reserved;
    int64_t  value;      /* a node's count / a task's Discrete reservation; 1 <= value < 2^31 */
} swp_generic;           /* 16 bytes */
/* replaces the node's counts (n == 0: the node offers nothing). A kind listed twice: SWP_EINVAL.
 * ONE count per kind stands for the node's list only while the list holds the kind ONCE as a Discrete entry or as Named entries with
 * distinct values. Reclaim + sanitize (resource_management.go:75-153) can leave a kind behind twice — a node's description changed
 * under a running task that then went away —, and for such a list HasEnough (first entry) and ConsumeNodeResources (every entry) no
 * longer amount to "count -= request": the count is right for the next request, not for a second one inside the same batch. The caller
 * keeps tasks that reserve such a kind on its own path until the list is regular again (shim/go/swp_cgo.go irregularKinds, csrc/
 * swp_generic.hpp irregular_kinds; swp_sched.cpp hands the whole tick back: every line Deferred). */
int swp_node_set_generic(swp_engine*, uint32_t node, const swp_generic* counts, uint32_t n);
int swp_node_get_generic(swp_engine*, uint32_t node, uint32_t kind, int64_t* count_out);

/* ------------------------------------------------------------------------------------------ */
/* CSI volumes: VolumesFilter (filter.go:382-441), volumeSet (volumes.go), IsInTopology (topology.go:23-47).
 * The engine holds, per volume, what checkVolume reads: availability, access mode, the accessible topologies, and how it is in use —
 * the number of tasks holding it, how many of them write, and the node they sit on. The caller keeps the maps behind those numbers
 * (task -> usage, volumes.go:31-46) and sets them with swp_volume_set_usage whenever they change outside a batch; inside a batch the
 * engine reserves for the tasks it places (scheduler.go:857-874) and swp_batch_fetch leaves the numbers as the batch made them. */
typedef struct { uint32_t key; uint32_t value; } swp_seg;          /* one (subdomain, segment) pair: SWP_SPACE_CSI ids */
typedef struct {
    uint32_t plugin;        /* NodeCSIInfo.PluginName: SWP_SPACE_CSI id */
    uint32_t has_topology;  /* AccessibleTopology != nil */
    uint32_t seg_off, n_seg;/* its segments in the array passed along */
} swp_csi;
/* Description.CSIInfo of a node (replaces what was set; n == 0: none). checkVolume takes the FIRST entry of the volume's driver. */
int swp_node_set_csi(swp_engine*, uint32_t node, const swp_csi* infos, uint32_t n, const swp_seg* segs, uint32_t n_segs);
enum { SWP_VOL_SCOPE_SINGLE_NODE = 0, SWP_VOL_SCOPE_MULTI_NODE = 1 };
enum { SWP_VOL_SHARING_NONE = 0, SWP_VOL_SHARING_READ_ONLY = 1, SWP_VOL_SHARING_ONE_WRITER = 2, SWP_VOL_SHARING_ALL = 3 };
typedef struct {
    uint32_t group;         /* SWP_SPACE_VOLUME_GROUP id of Spec.Group ("" is a group like any other) */
    uint32_t driver;        /* Spec.Driver.Name: SWP_SPACE_CSI id */
    uint32_t scope, sharing;/* Spec.AccessMode */
    uint32_t active;        /* Spec.Availability == ACTIVE (volumes.go:265) */
    uint32_t n_topologies;  /* VolumeInfo.AccessibleTopology: topology t = segs[topo_off[t] .. topo_off[t + 1]) */
} swp_volume;
/* addOrUpdateVolume (volumes.go:62-82): `volume` is the SWP_SPACE_VOLUME index. To the letter: for a volume the set holds already the
 * reference assigns the new object to a COPY of its map entry (:71), so what checkVolume reads — availability, access mode, driver,
 * topologies — stays what the FIRST call said; the call only adds the volume to the group the new object names (byGroup is never pruned,
 * :74-78: a volume whose Group changes is a member of both). The usage is kept. */
int swp_volume_upsert(swp_engine*, uint32_t volume, const swp_volume* v, const uint32_t* topo_off, const swp_seg* segs);
#define SWP_PIN_NONE 0xFFFFFFFFu   /* nobody uses the volume */
#define SWP_PIN_MANY 0xFFFFFFFEu   /* its users sit on more than one node */
typedef struct {
    uint32_t n_tasks;       /* len(volumeInfo.tasks) */
    uint32_t n_writers;     /* ... of them with readOnly == false (hasWriter, volumes.go:320-327) */
    uint32_t pin;           /* the node index all usages share, SWP_PIN_NONE, SWP_PIN_MANY (volumes.go:284-292) */
    uint32_t reserved;
} swp_volume_usage;
int swp_volume_set_usage(swp_engine*, uint32_t volume, const swp_volume_usage* u);
int swp_volume_get_usage(swp_engine*, uint32_t volume, swp_volume_usage* out);
#define SWP_NO_VOLUME 0xFFFFFFFFu
#define SWP_MAX_MOUNTS 8
typedef struct {
    uint32_t is_group;          /* Source starts with "group:" (volumes.go:229) */
    uint32_t ref;               /* the SWP_SPACE_VOLUME_GROUP id of the rest, or the SWP_SPACE_VOLUME index byName[Source] gives; SWP_NO_VOLUME: neither exists */
    uint32_t read_only;         /* Mount.ReadOnly: what checkVolume is asked with */
    uint32_t reserve_read_only; /* ReadOnly of the LAST mount of the task with this mount's (Source, Target): what reserveTaskVolumes records (volumes.go:144-154) */
} swp_mount;
/* VolumesFilter.SetTask (filter.go:392-422): the task's MountTypeCluster mounts in spec order, at most SWP_MAX_MOUNTS (SWP_ERANGE). The set id
 * goes into bits 8-31 of swp_task_desc.flags (SWP_TASK_MOUNTS); 0 = the filter is disabled. */
int swp_mount_set(swp_engine*, const swp_mount* mounts, uint32_t n, uint32_t* id_out);
/* isVolumeAvailableOnNode for every mount of the set on one node, in order, each seeing the ones before it (chooseTaskVolumes,
 * volumes.go:101-140): out[i] = the volume index for mount i. Returns SWP_OK with *n_out = the set's size, or *n_out = 0 when a mount
 * finds no volume (the reference's "cannot find volume to satisfy mount"; *failed_mount = its position; out[] then holds what the mounts
 * in front of it chose and SWP_NO_VOLUME from the failing one on: the caller's volumeSet counts that prefix, see swp_batch_attachments).
 * Reserves nothing. */
int swp_choose_volumes(swp_engine*, uint32_t mount_set, uint32_t node, uint32_t* out /* [SWP_MAX_MOUNTS] */, uint32_t* n_out, uint32_t* failed_mount);

/* ------------------------------------------------------------------------------------------ */
/* task-side predicate sets (what Filter.SetTask extracts from a task, filter.go)               */
enum {   /* constraint kinds, constraint.go:109-203 */
    SWP_CK_NODE_ID = 0, SWP_CK_HOSTNAME = 1, SWP_CK_IP = 2, SWP_CK_ROLE = 3,
    SWP_CK_PLATFORM_OS = 4, SWP_CK_PLATFORM_ARCH = 5, SWP_CK_NODE_LABEL = 6, SWP_CK_ENGINE_LABEL = 7,
    SWP_CK_INVALID = 8   /* unknown key: false for both operators (constraint.go:200-203) */
};
enum { SWP_OP_EQ = 0, SWP_OP_NE = 1 };
enum { SWP_IP_SINGLE = 0, SWP_IP_CIDR = 1, SWP_IP_MALFORMED = 2 };

typedef struct {
    uint32_t kind;       /* SWP_CK_* */
    uint32_t op;         /* SWP_OP_* */
    uint32_t key;        /* LABEL_KEY id for the two label kinds */
    uint32_t value;      /* FOLDED id of the expression; for SWP_CK_ROLE: FOLDED id as well */
    uint8_t  ip[16];     /* SWP_CK_IP: address / network (already masked) */
    uint32_t ip_kind;    /* SWP_IP_* */
    uint32_t prefix_len; /* bits, counted in the 128-bit form (v4 /24 → 120) */
    uint32_t ip_is_v4;   /* the expression was written as IPv4 */
    uint32_t reserved;
} swp_constraint;        /* 48 bytes */

typedef struct { uint32_t os; uint32_t arch; } swp_platform;   /* ids in OS / ARCH space; 0 = wildcard */
typedef struct { uint32_t protocol; uint32_t port; } swp_port;

/* ConstraintFilter.SetTask (filter.go:218-232): returns the id of the de-duplicated set */
int swp_constraint_set(swp_engine*, const swp_constraint* cs, uint32_t n, uint32_t* id_out);
/* PlatformFilter.SetTask (filter.go:253-263) */
int swp_platform_set(swp_engine*, const swp_platform* ps, uint32_t n, uint32_t* id_out);
/* PluginFilter.SetTask (filter.go:119-131): `required` = Volume/Network plugins that must exist;
 * log_plugin = Log plugin id or 0 (filter.go:165-175) */
int swp_plugin_set(swp_engine*, const uint32_t* required, uint32_t n, uint32_t log_plugin, uint32_t* id_out);
/* HostPortFilter.SetTask (filter.go:322-333): host-mode published ports of the task (at most 32 per task: SWP_ERANGE) */
int swp_port_set(swp_engine*, const swp_port* ports, uint32_t n, uint32_t* id_out);
/* ResourceFilter.SetTask's generic half (filter.go:61-74): the task's Reservations.Generic, all Discrete (ValidateTask,
 * validate.go:11-22), one entry per kind, each value >= 1 (a request of 0 makes selectNodeResources claim every named value of the
 * kind, resource_management.go:52-66: such a task stays on the Go path, SWP_EUNSUPPORTED; so does a kind requested twice).
 * At most 8 kinds per task (SWP_ERANGE). */
int swp_generic_set(swp_engine*, const swp_generic* items, uint32_t n, uint32_t* id_out);
/* Spread preferences that create a decision-tree level (nodeset.go:59-82): kind is SWP_CK_NODE_LABEL or
 * SWP_CK_ENGINE_LABEL, key the LABEL_KEY id of the part after the prefix; other descriptors are skipped by
 * the caller exactly as the reference skips them. */
typedef struct { uint32_t kind; uint32_t key; } swp_spread;
int swp_spread_set(swp_engine*, const swp_spread* levels, uint32_t n, uint32_t* id_out);

#define SWP_TASK_RES_ENABLED 0x1u   /* ResourceFilter.SetTask returned true (filter.go:61-74) */
#define SWP_TASK_MOUNTS_SHIFT 8     /* flags >> 8 = the task's mount set (swp_mount_set), 0 = no MountTypeCluster mounts */
#define SWP_TASK_MOUNTS(set) ((uint32_t)(set) << SWP_TASK_MOUNTS_SHIFT)

typedef struct {
    uint32_t service;        /* SERVICE id */
    uint32_t flags;          /* SWP_TASK_* */
    int64_t  cpu;            /* Reservations.NanoCPUs */
    int64_t  mem;            /* Reservations.MemoryBytes */
    uint32_t constraint_set; /* 0 = ConstraintFilter disabled */
    uint32_t platform_set;   /* 0 = PlatformFilter disabled */
    uint32_t plugin_set;     /* 0 = PluginFilter disabled */
    uint32_t port_set;       /* 0 = HostPortFilter disabled */
    uint64_t max_replicas;   /* 0 = MaxReplicasFilter disabled (filter.go:363-370) */
    uint64_t spec_version;   /* SpecVersion.Index (0 when nil) — selects the failure bucket */
    uint32_t spread_set;     /* 0 = no spread preferences (Placement.Preferences, nodeset.go:59-82) */
    uint32_t generic_set;    /* 0 = no generic reservations (swp_generic_set); such a task needs SWP_TASK_RES_ENABLED */
} swp_task_desc;             /* 64 bytes */

/* ------------------------------------------------------------------------------------------ */
/* the hot path                                                                                */
#define SWP_NFILTERS 8   /* Ready, Resource, Plugin, Constraint, Platform, HostPort, MaxReplicas, Volumes */

/* Whole-tick batch for singleton groups == tick() over one-off tasks (scheduler.go:429-488 with
 * scheduleTaskGroup :694-748 on a group of one, nodeSet.tree nodeset.go:50-124, Pipeline.Process
 * pipeline.go:56-68, nodeLess :708-735) INCLUDING the residual update (nodeinfo.go:108-154).
 * Tasks are placed in array order; each placement is visible to the next task.
 *   out_node[i]      node index or -1 ("no suitable node")
 *   out_fail_hist[i] per-filter first-failure counts over all nodes at the moment task i was
 *                    tried (what Pipeline.Explain reads, pipeline.go:84-103); only written for
 *                    tasks with out_node[i] == -1; may be NULL. */
int swp_schedule_batch(swp_engine*, const swp_task_desc* tasks, uint32_t n_tasks,
                       int32_t* out_node, uint32_t* out_fail_hist /* [n_tasks][SWP_NFILTERS] */);

/* tick() over task GROUPS (tasks sharing ServiceID + SpecVersion, scheduler.go:438-466): for each group, in
 * array order, scheduleTaskGroup with k = sizes[g] (scheduler.go:694-748): nodeSet.tree with a max-heap of k
 * per leaf (nodeset.go:50-124, container/heap order reproduced), scheduleNTasksOnSubtree (:772-825), the fill
 * loop scheduleNTasksOnNodes (:844-924) incl. the residual update. One descriptor per group (all tasks of a
 * group are identical for the filters, scheduler.go:696-702). Also the path of one-off tasks that carry spread
 * preferences (a group of one). (SURVEY.md 8b sketched this entry as swp_scan_groups — candidates out, tree walk and
 * fill loop on the host; they run on the device as well, so the call returns placements.)
 *   out_node       Σ sizes entries, group after group, tasks in canonical (enqueue) order; -1 = left over
 *   out_fail_hist  [n_groups][SWP_NFILTERS]: Pipeline counters as noSuitableNode would read them
 *                  (scheduler.go:929), written for groups with left-over tasks */
int swp_schedule_groups(swp_engine*, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups,
                        int32_t* out_node, uint32_t* out_fail_hist);
/* The same for a call with groups whose tasks have cluster mounts (SWP_TASK_MOUNTS): out_att[i * SWP_MAX_MOUNTS + m] = the volume chosen for
 * mount m of the call's i-th task on its node (chooseTaskVolumes, scheduler.go:857-872), SWP_NO_VOLUME as for swp_batch_attachments. The
 * volumes' usage numbers are left as the call made them. */
int swp_schedule_groups_volumes(swp_engine*, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups,
                                int32_t* out_node, uint32_t* out_fail_hist, uint32_t* out_att);

/* The same in three steps so that a caller (bench.py) can time the device pass alone:
 *   prepare: de-duplicate predicate sets, upload descriptors and per-service state
 *   run:     kernels only (class bitmaps → scan → resolve/commit → explain); asynchronous
 *   fetch:   wait, copy results back, fold the placements into the host-side node mirror */
typedef struct swp_batch swp_batch;
int swp_batch_prepare(swp_engine*, const swp_task_desc* tasks, uint32_t n_tasks, swp_batch** out);
/* swp_batch_prepare for a caller that knows which tasks share a descriptor (tasks of one service spec do): task i carries
 * templates[template_of_task[i]]. The per-task pass is an array lookup instead of the hash of 64 bytes that finds the same out. */
int swp_batch_prepare_templates(swp_engine*, const swp_task_desc* templates, uint32_t n_templates, const uint32_t* template_of_task,
                                uint32_t n_tasks, swp_batch** out);
int swp_batch_run(swp_engine*, swp_batch*);
int swp_batch_fetch(swp_engine*, swp_batch*, int32_t* out_node, uint32_t* out_fail_hist);
/* copy the device results of the last run back WITHOUT folding them into the host mirror (replay
 * benchmarking: run → results → swp_state_restore → run …) */
int swp_batch_results(swp_engine*, swp_batch*, int32_t* out_node, uint32_t* out_fail_hist);
/* The volumes the batch chose for the cluster mounts of `n` of its tasks (chooseTaskVolumes on the node each was placed on,
 * scheduler.go:857-872): out[i * SWP_MAX_MOUNTS + m] = the SWP_SPACE_VOLUME index for mount m of task tasks[i]; SWP_NO_VOLUME for
 * every mount of a task that was not placed or has no mounts. A task for which mount k found no volume is assigned WITHOUT attachments
 * (as the reference does) — its row holds the volumes mounts 0 .. k-1 had chosen by then and SWP_NO_VOLUME from k on: a row is a full
 * set of attachments iff none of the task's mounts reads SWP_NO_VOLUME. (The prefix matters to the caller's books: chooseTaskVolumes
 * reserves what it picks and releases each volume ONCE, so a volume picked for m mounts of the task leaves m - 1 counts on the node
 * whether the choice succeeds or not, volumes.go:104-131,162-178.) After swp_batch_fetch / swp_batch_results. */
int swp_batch_attachments(swp_engine*, swp_batch*, const uint32_t* tasks, uint32_t n, uint32_t* out);
void swp_batch_free(swp_engine*, swp_batch*);
/* Device-side snapshot / restore of all mutable node state (cpu, mem, total, per-service counts,
 * host ports) so that a benchmark can replay the same batch from the same state. */
int swp_state_save(swp_engine*);
int swp_state_restore(swp_engine*);

/* ------------------------------------------------------------------------------------------ */
/* node-range shards: the nodeSet split over several engines — one per GPU of a  */
/* node, or several on one GPU — each owning a contiguous range of the canonical node order      */
/* (shard g's nodes all precede shard g+1's). The scan nodeSet.tree does over ALL nodes           */
/* (nodeset.go:57-120) becomes: every shard proposes, for a block of pending tasks, its best       */
/* candidates against its current state; the proposals are exchanged (RCCL all-gather between     */
/* ranks, plain pointers inside one process); every participant merges them with the SAME          */
/* deterministic rule and learns which tasks of the block are decided; the owner of each picked    */
/* node applies the placement (NodeInfo.addTask, nodeinfo.go:108-154).                             */
/*                                                                                                 */
/* Exactness: inside a batch a node's key (scheduler.go:708-735) only grows and feasibility only    */
/* shrinks, so a task's best node against the block's snapshot stays its best node until an         */
/* earlier task of the same block takes it. A proposal therefore lists the first SWP_SHARD_CAND      */
/* non-empty 64-node words of the task's survivors at its minimum level, in node order (a wave's     */
/* ballot IS such a word); the merge walks the shards in range order at the global minimum level     */
/* and gives the task the first listed node no earlier task of the block took. The block is cut — the remaining tasks are proposed again against the new        */
/* state — when a list runs out (a taken node sits one level up and may precede what was not         */
/* listed) or when a task must fall back to its service's exception list (nodes where the service    */
/* already runs / that failed it: their order changes with every placement of that service), unless  */
/* it is the first task of the block, for which the snapshot IS the sequential state. The first      */
/* task of a block is always decided, so every round makes progress.                                 */
#define SWP_SHARD_CAND 4
typedef struct {
    uint32_t level;          /* ActiveTasksCount of the shard's best plain candidates; 0xFFFFFFFF = none */
    uint32_t n_cand;         /* words listed; bit 31: the shard has more nodes of that level after the last word listed */
    uint32_t word[SWP_SHARD_CAND]; /* shard-local word indices (node = 64 * word + bit), ascending */
    uint64_t bits[SWP_SHARD_CAND]; /* the level's survivors inside each word */
    uint64_t exc_hi;         /* best exception-list candidate: (failure class << 32 | svcCount); ~0 = none */
    uint64_t exc_lo;         /*                                (ActiveTasksCount << 32 | shard-local node) */
    uint32_t exc_entry;      /* its entry in the shard's exception list */
    uint32_t flags;          /* bit 0: the task does not count on its node (DesiredState > COMPLETED, nodeinfo.go:131-134): its node stays
                              * on its level, so the merge ends the block behind its pick */
} swp_proposal;              /* 80 bytes */
typedef struct {
    int32_t  shard;          /* owner of the picked node; -1 = no suitable node on any shard */
    uint32_t node;           /* shard-local node index */
    uint32_t entry;          /* exception-list entry the pick came from, 0xFFFFFFFF for a plain node */
    uint32_t reserved;
} swp_shard_pick;            /* 16 bytes */
/* per-batch device state to pristine + predicate class bitmaps (what swp_batch_run does before its first window) */
int swp_shard_begin(swp_engine*, swp_batch*);
/* proposals of tasks [j0, j0+count) against the shard's current state; out = host array[count]; blocks until done */
int swp_shard_propose(swp_engine*, swp_batch*, uint32_t j0, uint32_t count, swp_proposal* out);
/* the merge (pure function, no engine): proposals[g] = shard g's array[count], shards in range order.
 * picks[0 .. *accepted) are decided; the next block starts at j0 + *accepted (>= 1 whenever count >= 1). */
int swp_shard_merge(const swp_proposal* const* proposals, const uint32_t* shard_first_node, uint32_t n_shards, uint32_t count,
                    swp_shard_pick* picks, uint32_t* accepted);
/* apply the decided prefix: the picks whose shard == swp_config.shard_rank change this engine's node rows, exception
 * bitmaps / lists, host ports and commit log; every engine counts all of them (the commit log is numbered globally) and
 * records the unplaceable tasks for its explain pass */
int swp_shard_commit(swp_engine*, swp_batch*, uint32_t j0, const swp_shard_pick* picks, uint32_t accepted);
/* end of the batch: Explain histograms of the unplaceable tasks over THIS shard's nodes (sum them over the shards),
 * shard-local node index of the tasks placed here (-1 elsewhere), and the placements folded into the host mirror */
int swp_shard_end(swp_engine*, swp_batch*, int32_t* out_node_local, uint32_t* out_fail_hist);

/* The same job with the ROUNDS ON THE DEVICE, for the deployment a Go manager is: ONE process, n engines — one per GPU of the box
 * (peer access over xGMI), or several on one GPU — engines[g] owning node range g of the canonical order, batches[g] prepared on
 * engines[g] from the SAME task list. Per round — two kernel launches per device — every shard proposes for a block of tasks over its
 * own nodes (the block resolver's propose kernel: up to 32 non-empty half-words per task), and every shard then folds the records of
 * ALL shards into one list per task in global node order, walks the block with its matching wave — the per-task "allreduce(min-score,
 * argmin-node)" of the node-range split, done for the whole block by the wave that has to order it; every shard derives the same
 * picks from the same bytes, so no pick travels — and applies the picks that landed in its own range. No host work inside a round: the
 * call enqueues rounds on the engines' streams (events order them across devices) and reads a header every few dozen rounds. At most 8
 * shards. Generic reservations and cluster mounts are carried (a placed task's volumes travel behind its owner's next proposals; the
 * volume table is replicated on every shard).
 *   out_shard[i]    owner of task i's node, -1 = no suitable node
 *   out_node[i]     shard-LOCAL node index on that engine
 *   out_fail_hist   [n_tasks][SWP_NFILTERS], summed over the shards; may be NULL
 * The placements are folded into each engine's host mirror (as swp_batch_fetch does) unless flags has SWP_SHARD_NO_FOLD. */
#define SWP_SHARD_NO_FOLD 1u   /* leave the host mirrors alone (replay benchmarking, as swp_batch_results: run -> swp_state_restore -> run ...) */
int swp_shard_run(swp_engine* const* engines, swp_batch* const* batches, uint32_t n_shards, uint32_t flags, int32_t* out_shard, int32_t* out_node,
                  uint32_t* out_fail_hist);

/* The rank variant: ONE engine per process / GPU, the ranks of a job connected by RCCL over xGMI (librccl.so is loaded on first use;
 * the engine links nothing of it). The rounds are the same kernels — every rank proposes over its own node range, an
 * ncclAllGather on the engine's stream hands every rank the proposals of all of them (block x 224 + 144 bytes per rank and round: the
 * proposals, the trailer slots of a placed task's volumes and the rank's `dead` word — a rank whose launch failed keeps issuing the
 * stretch's collectives with that word set, every rank's kernels stand still, and all ranks leave at the next status exchange), and
 * EVERY rank folds + matches the block (the same deterministic wave everywhere: no second collective to agree on the picks) and
 * applies the picks of its own range. Bootstrap as usual with RCCL: rank 0 fills an id (swp_rccl_unique_id), the application
 * hands it to the other ranks over whatever it has (the Go manager's raft / gRPC; bench.py: torch.distributed), every rank calls
 * swp_rccl_init once per engine.
 *   shard_nodes[g]   node count of rank g's range (every rank passes the same array: the global tie order)
 *   out_node_local   this rank's node index of the tasks placed here, -1 elsewhere
 *   out_fail_hist    this rank's share of the Explain histograms (sum over the ranks), may be NULL */
#define SWP_RCCL_ID_BYTES 128
int swp_rccl_available(swp_engine*);   /* SWP_OK: librccl.so loads with every symbol the engine uses. Ranks exchange this BEFORE any of them calls swp_rccl_init (a rank that cannot would leave the others inside ncclCommInitRank) */
int swp_rccl_unique_id(swp_engine*, uint8_t id_out[SWP_RCCL_ID_BYTES]);
int swp_rccl_init(swp_engine*, const uint8_t id[SWP_RCCL_ID_BYTES], uint32_t rank, uint32_t n_ranks);
int swp_rccl_finalize(swp_engine*);   /* ncclCommDestroy; swp_destroy does not (an engine often dies with the process, after RCCL itself) */
int swp_shard_run_rank(swp_engine*, swp_batch*, const uint32_t* shard_nodes, uint32_t flags, int32_t* out_node_local, uint32_t* out_fail_hist);
/* The ranks leave swp_shard_run_rank TOGETHER: before the first round and after every stretch of rounds each rank contributes one
 * status word {int32 code, uint32 position, uint32 kernel error, uint32 rounds} to a 16-byte ncclAllGather, and every rank derives
 * the same verdict from the same words: 0 go on, 1 a rank could not take part (*who_out names it), 2 a rank's kernels reported an
 * error, 3 the positions differ (the ranks diverged). Exposed for the host layer's tests; words = n_ranks x 4 uint32. */
int swp_shard_verdict(const uint32_t* words, uint32_t n_ranks, uint32_t* who_out);
/* TASK GROUPS over node-range shards (scheduler.go:449-461: replicated services ARE groups; nodeset.go:107-120) are capacity-bound by
 * ONE GPU: swp_schedule_groups replays container/heap over all nodes in node order — one sequential machine (csrc/swp_groups.hpp) — so a
 * shard set runs a group call on a union engine of its own, and a job of ranks keeps that union on rank 0 (swarmkit_amd/shard.py
 * RankUnionGroups: every sharded batch's result enters it with swp_commit, the groups' placements travel to the owners in one broadcast,
 * every owner books its share with swp_commit). SURVEY 8e's merge of k candidates per shard was not built: which node the heap admits
 * next depends on every earlier admission (a root replacement changes what the next node is compared with), so the shards' offers would
 * have to be cut again after every admitted node. */

/* A shard SET: the same node-range split behind ONE engine handle — what a Go manager on a multi-GPU box holds instead of an engine.
 * swp_shardset_create makes n_shards engines (devices[g]: the HIP device of shard g; NULL: all on cfg->device) and returns a handle that
 * every entry point of this header takes like an engine's: the set interns node ids itself (lowest free index first, as swp_node_remove
 * documents) and owns the GLOBAL node index — node i lives on shard i / nodes_per_shard —, routes every node call to the owner
 * (nodeSet.addOrUpdateNode / remove, NodeInfo.addTask / removeTask: the incremental path between two batches, scheduler.go:254-396,
 * nodeinfo.go:66-154), replicates services, predicate sets, volumes and mount sets on every shard, and runs a batch with swp_shard_run
 * (the rounds on the device); swp_batch_fetch returns global node indices and folds every shard's placements into its mirror. What a
 * range cannot hold is refused when a node id is interned (SWP_ERANGE: n_shards x nodes_per_shard slots). swp_destroy destroys the
 * shards. The swp_shard_* / swp_rccl_* calls take the engine of ONE range, never a set (SWP_EINVAL). */
int swp_shardset_create(const swp_config* cfg, const int32_t* devices, uint32_t n_shards, uint32_t nodes_per_shard, swp_engine** out);

/* NodeInfo.addTask / removeTask for tasks the engine did not place itself (event handlers
 * scheduler.go:254-366; rollback :472-487). add_or_remove: 1 = add, 0 = remove. */
typedef struct {
    uint32_t node;
    uint32_t service;
    int64_t  cpu, mem;
    uint32_t port_set;     /* 0 = none */
    uint32_t counted;      /* DesiredState <= COMPLETED: counts toward ActiveTasksCount* (nodeinfo.go:148) */
} swp_placement;           /* 32 bytes */
int swp_commit(swp_engine*, const swp_placement* p, uint32_t n, int add_or_remove);

/* Pipeline.Process on ONE (task, node) pair == taskFitNode's check (scheduler.go:646-654).
 * *first_fail = -1 on pass, else the index of the first failing filter. */
int swp_check_node(swp_engine*, const swp_task_desc* task, uint32_t node, int32_t* first_fail);

/* constraintenforcer.rejectNoncompliantTasks (manager/orchestrator/constraintenforcer/constraint_enforcer.go:65-196)
 * over MANY nodes in one call: the enforcer's start-up sweep (Run, :45-52) or a burst of EventUpdateNode. Per node the
 * reference walks the node's tasks in store order: skip by desired / observed state (:118-126), reject when the node
 * no longer matches the placement constraints of the task's CURRENT service spec (or of the task itself when the service
 * is gone, :152-168), else account the reservation against Description.Resources in that order and reject what no
 * longer fits (:172-184). The caller lists only ACTIVE nodes (:70-72) and resolves which constraint list applies;
 * an unparsable list is passed as constraint_set 0 (`constraints, _ := constraint.Parse`, :163). Tasks that carry
 * AssignedGenericResources (:188-199) stay on the Go path: do not pass their node.
 *   out_reject[i] = 1 when task i would be set to REJECTED, else 0. */
typedef struct {
    uint32_t node;          /* NODE_ID id; must be present in the engine's nodeSet mirror */
    uint32_t first_task;    /* index of the node's first task in `tasks` */
    uint32_t n_tasks;
    uint32_t reserved;
    int64_t  cpu, mem;      /* node.Description.Resources (0,0 when nil, :101-106) */
} swp_enforce_node;         /* 32 bytes */
#define SWP_ENF_RESERVATIONS 0x1u   /* t.Spec.Resources != nil && Reservations != nil (:172) */
typedef struct {
    int64_t  cpu, mem;          /* Reservations */
    uint32_t constraint_set;    /* swp_constraint_set id, 0 = no (parsable) constraints */
    uint32_t flags;             /* SWP_ENF_* */
    uint32_t desired_state;     /* api.TaskState numeric values (NEW=0 … ASSIGNED=192 … COMPLETE=576 …) */
    uint32_t state;             /* Status.State */
} swp_enforce_task;             /* 32 bytes */
int swp_enforce(swp_engine*, const swp_enforce_node* nodes, uint32_t n_nodes, const swp_enforce_task* tasks, uint32_t n_tasks,
                uint8_t* out_reject);

/* constraint.NodeMatches(service.constraints, node) for EVERY (constraint set, node) pair: the global orchestrator's
 * reconciliation sweeps (manager/orchestrator/global/global.go:306, :440, :513 — one NodeMatches per global service
 * and node). out_bitmaps is [n_sets][n_words] with n_words = ceil(node slots / 64) as reported by swp_stats; bit i of
 * word w = node index 64w+i. Set 0 (no constraints) matches every present node (NodeMatches(nil) == true). */
int swp_node_matches(swp_engine*, const uint32_t* constraint_sets, uint32_t n_sets, uint64_t* out_bitmaps, uint32_t n_words);

/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t batches, tasks, placed, infeasible;
    uint64_t pair_evals;        /* Σ tasks × present nodes */
    uint64_t verify_retries;    /* resolver candidates rejected by the freshness re-check */
    uint64_t slow_path_tasks;   /* tasks resolved through the per-service exception list */
    uint64_t rebase_events;
    uint64_t generic_tasks;     /* tasks that left the resolver's hot-level fast path */
    uint64_t resolver_spins;    /* resolver wave polls while waiting for staged rows */
    uint32_t n_nodes, n_words, last_windows, last_static_classes;
    float    ms_classes, ms_scan, ms_resolve, ms_explain, ms_total;   /* last batch, SWP_CFG_PROFILE */
    uint32_t scan_launches, resolve_launches;
    uint32_t last_resolver;     /* resolver kernel of the last batch: 105 = k_resolve5 exact mode, 5 = k_resolve5 over the scan's rows,
                                   3 = k_resolve3, 2 = k_resolve2, 1 = k_resolve1, 0 = k_resolve */
    /* node-range shard protocol, last batch (since swp_shard_begin); the times need SWP_CFG_PROFILE */
    float    ms_propose, ms_apply;          /* Σ k_propose / k_shard_apply launch durations */
    uint32_t propose_launches, propose_tasks;   /* launches and Σ tasks proposed (a task cut off a block is proposed again) */
    uint32_t waterfill_tasks;   /* tasks placed as runs of identical tasks (k_waterfill), since swp_create */
    uint32_t scan_tasks;        /* last batch: tasks the scan resolver decided (k_scan / k_scanb: stretches without plain candidates); scan_launches
                                   counts those stretches and ms_scan their time (SWP_CFG_PROFILE) */
} swp_stats_t;

int swp_create(const swp_config*, swp_engine** out);
void swp_destroy(swp_engine*);
int swp_stats(swp_engine*, swp_stats_t* out);
const char* swp_strerror(int code);
const char* swp_last_error(swp_engine*);   /* engine may be NULL: last swp_create failure */
/* sizeof() of every ABI struct, so that a binding can assert its own layout */
int swp_abi_check(uint32_t* sizes, uint32_t n);   /* order: config,node_row,kv,constraint,platform,port,task_desc,placement,stats,spread,generic */

#ifdef __cplusplus
}
#endif
#endif /* SWP_H */
