// swp_engine.hip — host runtime behind include/swp.h (libswp.so).
//
// Owns: string interning, the host mirror of the nodeSet's numeric state, predicate-set
// de-duplication, device memory, kernel sequencing. There is NO CPU placement path in here:
// every placement decision is taken by the kernels in swp_device.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <dlfcn.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <tuple>
#include <queue>
#include <mutex>
#include <numeric>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/swp.h"
#include "swp_device.hpp"
#include "swp_groups.hpp"
#include "swp_launch.hpp"
#include "swp_resolve6.hpp"
#include "swp_resolve7.hpp"
#include "swp_scan.hpp"
#include "swp_shard.hpp"
#include "swp_waterfill.hpp"

using namespace swpdev;

hipError_t swpdev::ensure_big_lds(const void* fn, int device) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, device})) return hipSuccess;
    hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (r == hipSuccess) done.insert({fn, device});
    return r;
}


namespace {

thread_local std::string g_create_error;

// Device allocations of freed batches are kept for the next batch (a tick prepares and frees one batch after the other: ~50
// hipMalloc / hipFree pairs per tick otherwise). Per device, keyed by capacity; bounded.
struct DevPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void*> free_blocks;   // (device, capacity) -> block
    size_t held = 0;
    static constexpr size_t LIMIT = (size_t)8 << 30;
    void* take(int dev, size_t want, size_t* cap) {
        std::lock_guard<std::mutex> g(mu);
        auto it = free_blocks.lower_bound({dev, want});
        if (it == free_blocks.end() || it->first.first != dev || it->first.second > want + want / 2 + 4096) return nullptr;
        void* p = it->second;
        *cap = it->first.second;
        held -= *cap;
        free_blocks.erase(it);
        return p;
    }
    bool give(int dev, size_t cap, void* p) {
        std::lock_guard<std::mutex> g(mu);
        if (held + cap > LIMIT) return false;
        free_blocks.emplace(std::make_pair(dev, cap), p);
        held += cap;
        return true;
    }
    void drain(int dev) {   // swp_destroy: nothing of this device stays behind
        std::lock_guard<std::mutex> g(mu);
        for (auto it = free_blocks.begin(); it != free_blocks.end();) {
            if (it->first.first == dev) {
                (void)hipFree(it->second);
                held -= it->first.second;
                it = free_blocks.erase(it);
            } else
                ++it;
        }
    }
};
inline DevPool& dev_pool() {
    static DevPool* p = new DevPool();   // (never destroyed: frees at process exit would race the HIP runtime's own teardown)
    return *p;
}

// SWP_HOST_PROF=1: wall time of the host-side sections of the calls a churn round makes, summed per section, printed when an engine is
// destroyed (tools: where a round's host milliseconds go; nothing is measured without the variable)
struct HostProf {
    bool on = getenv("SWP_HOST_PROF") != nullptr;
    std::mutex mu;
    std::map<std::string, std::pair<double, uint64_t>> acc;
    std::map<std::string, double> longest;
    void add(const char* name, double ms) {
        std::lock_guard<std::mutex> g(mu);
        auto& a = acc[name];
        a.first += ms;
        a.second += 1;
        double& l = longest[name];
        if (ms > l) l = ms;
    }
    void print() {
        if (!on) return;
        std::lock_guard<std::mutex> g(mu);
        for (auto& kv : acc)   // (the longest call apart: a script's first batch places every task, its rounds a tenth of them)
            fprintf(stderr, "[swp host] %-44s %9.3f ms over %7llu calls | longest %8.3f ms, the others %8.2f us each\n", kv.first.c_str(), kv.second.first, (unsigned long long)kv.second.second,
                    longest[kv.first], kv.second.second > 1 ? 1e3 * (kv.second.first - longest[kv.first]) / (kv.second.second - 1) : 0.0);
        acc.clear();
        longest.clear();
    }
};
static HostProf& host_prof() { static HostProf p; return p; }
struct HostSpan {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit HostSpan(const char* n) : name(host_prof().on ? n : nullptr) { if (name) t0 = std::chrono::steady_clock::now(); }
    void next(const char* n) {
        if (!name) return;
        const auto t1 = std::chrono::steady_clock::now();
        host_prof().add(name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        name = n;
        t0 = t1;
    }
    ~HostSpan() { next(nullptr); }
};

struct DevBuf {   // owns one device allocation: movable, not copyable
    void* p = nullptr;
    size_t cap = 0;
    int dev = -1;
    bool view = false;   // p lies inside another DevBuf's allocation (a batch's upload arena): nothing of its own to give back
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), dev(o.dev), view(o.view) { o.p = nullptr; o.cap = 0; o.view = false; }
    ~DevBuf() { release(); }
    void release() {
        if (p && !view && !dev_pool().give(dev, cap, p)) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        view = false;
    }
    void point_into(void* q, size_t bytes) {   // (what was here before goes back first)
        release();
        p = q;
        cap = bytes;
        view = true;
    }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        release();
        size_t want = bytes < 256 ? 256 : bytes;
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
        if ((p = dev_pool().take(dev, want, &cap))) return hipSuccess;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipErrorOutOfMemory) {   // the blocks cached for reuse are the first thing to give back: drain them and try once more
            (void)hipGetLastError();
            dev_pool().drain(dev);
            e = hipMalloc(&p, want);
        }
        if (e == hipSuccess) cap = want;
        else p = nullptr;
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Pinned host blocks of freed batches, kept for the next batch: hipHostMalloc of a batch's 64 MB of task records costs more than filling them
struct PinPool {
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;   // capacity -> block
    size_t held = 0;
    static constexpr size_t LIMIT = (size_t)1 << 30;
    void* take(size_t want, size_t* cap) {
        std::lock_guard<std::mutex> g(mu);
        auto it = free_blocks.lower_bound(want);
        if (it == free_blocks.end() || it->first > 2 * want + 65536) return nullptr;
        void* p = it->second;
        *cap = it->first;
        held -= *cap;
        free_blocks.erase(it);
        return p;
    }
    bool give(size_t cap, void* p) {
        std::lock_guard<std::mutex> g(mu);
        if (held + cap > LIMIT) return false;
        free_blocks.emplace(cap, p);
        held += cap;
        return true;
    }
};
inline PinPool& pin_pool() {
    static PinPool* p = new PinPool();   // (never destroyed, as the device pool)
    return *p;
}

struct PinBuf {   // owns one pinned host allocation (async copies in both directions without a staging pass)
    void* p = nullptr;
    size_t cap = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    PinBuf(PinBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    ~PinBuf() { release(); }
    void release() {
        if (p && !pin_pool().give(cap, p)) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        release();
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        if ((p = pin_pool().take(want, &cap))) return hipSuccess;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        else p = nullptr;
        return e;
    }
};

// A batch's per-task records in pinned memory: NOT initialised by resize (every element is written by the builder), uploaded without a
// staging pass, and the block goes back to the pool with the batch
template <class T>
struct PinVec {
    PinBuf buf;
    size_t n = 0;
    hipError_t resize(size_t k) {
        hipError_t e = buf.reserve(k * sizeof(T));
        if (e == hipSuccess) n = k;
        return e;
    }
    T& operator[](size_t i) { return static_cast<T*>(buf.p)[i]; }
    const T& operator[](size_t i) const { return static_cast<const T*>(buf.p)[i]; }
    T* data() { return static_cast<T*>(buf.p); }
    const T* data() const { return static_cast<const T*>(buf.p); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};

struct Interner {
    std::unordered_map<std::string, uint32_t> map;
    std::vector<std::string> strs;
    bool zero_is_empty = true;
    void init(bool zero_empty) {
        zero_is_empty = zero_empty;
        map.clear();
        strs.clear();
        free_ids = decltype(free_ids)();
        freed.clear();
        if (zero_empty) {
            strs.emplace_back();
            map.emplace(std::string(), 0u);
        }
    }
    // ids handed back (the node space only: swp_node_remove): the next NEW string takes the LOWEST free id, so that the node index
    // space — the width of every bitmap row — is bounded by the number of nodes alive at once, not by the nodes ever seen
    std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> free_ids;
    std::vector<char> freed;   // [id] the id is on the free heap
    uint32_t get(const std::string& s) {
        auto it = map.find(s);
        if (it != map.end()) return it->second;
        uint32_t id;
        if (!free_ids.empty()) {
            id = free_ids.top();
            free_ids.pop();
            freed[id] = 0;
            strs[id] = s;
        } else {
            id = (uint32_t)strs.size();
            strs.push_back(s);
        }
        map.emplace(s, id);
        return id;
    }
    void release(uint32_t id) {
        if (id >= strs.size()) return;
        if (freed.size() < strs.size()) freed.resize(strs.size(), 0);
        if (freed[id]) return;
        auto it = map.find(strs[id]);
        if (it != map.end() && it->second == id) map.erase(it);
        strs[id].clear();
        freed[id] = 1;
        free_ids.push(id);
    }
};

// strings.EqualFold canonicalisation for the FOLDED space. Expressions are limited by the value
// regexp (constraint.go:23-26) to ASCII plus the two runes RE2 folds onto ASCII letters
// (U+212A KELVIN SIGN, U+017F LONG S), so mapping exactly those orbits onto lower-case ASCII and
// leaving every other byte alone decides EqualFold(exp, value) by byte equality.
std::string fold_canon(const char* s, size_t len) {
    std::string out;
    out.reserve(len);
    for (size_t i = 0; i < len;) {
        unsigned char c = (unsigned char)s[i];
        if (c < 0x80) {
            out.push_back((c >= 'A' && c <= 'Z') ? char(c + 32) : char(c));
            ++i;
        } else if (c == 0xE2 && i + 2 < len && (unsigned char)s[i + 1] == 0x84 && (unsigned char)s[i + 2] == 0xAA) {
            out.push_back('k');
            i += 3;
        } else if (c == 0xC5 && i + 1 < len && (unsigned char)s[i + 1] == 0xBF) {
            out.push_back('s');
            i += 2;
        } else {
            out.push_back(char(c));
            ++i;
        }
    }
    return out;
}

// uint32 -> uint32 as a vector sorted by key: the per-node service counts (a dozen entries) and the per-service node counts (a hundred):
// a binary search and a short memmove per change instead of a tree / hash node, and batch preparation reads a service's nodes in order
struct FlatMap32 {
    typedef std::pair<uint32_t, uint32_t> Ent;
    std::vector<Ent> v;
    typedef std::vector<Ent>::iterator iterator;
    typedef std::vector<Ent>::const_iterator const_iterator;
    iterator begin() { return v.begin(); }
    iterator end() { return v.end(); }
    const_iterator begin() const { return v.begin(); }
    const_iterator end() const { return v.end(); }
    size_t size() const { return v.size(); }
    bool empty() const { return v.empty(); }
    void clear() { v.clear(); }
    iterator lower(uint32_t k) { return std::lower_bound(v.begin(), v.end(), k, [](const Ent& e, uint32_t key) { return e.first < key; }); }
    const_iterator lower(uint32_t k) const { return std::lower_bound(v.begin(), v.end(), k, [](const Ent& e, uint32_t key) { return e.first < key; }); }
    iterator find(uint32_t k) { auto it = lower(k); return it != v.end() && it->first == k ? it : v.end(); }
    const_iterator find(uint32_t k) const { auto it = lower(k); return it != v.end() && it->first == k ? it : v.end(); }
    uint32_t& operator[](uint32_t k) {
        auto it = lower(k);
        if (it == v.end() || it->first != k) it = v.insert(it, Ent(k, 0u));
        return it->second;
    }
    void erase(uint32_t k) { auto it = find(k); if (it != v.end()) v.erase(it); }
};

struct HostNode {
    bool present = false;
    swp_node_row row{};
    std::vector<swp_kv> labels, elabels;
    std::vector<uint32_t> plugins;
    FlatMap32 svc;                                                // service -> ActiveTasksCountByService
    std::map<std::pair<uint32_t, uint64_t>, uint32_t> fails;      // (service, specVersion) -> recent failures
    std::set<uint64_t> ports;                                     // protocol<<32 | port
    std::vector<std::pair<uint32_t, int64_t>> gen;                // (GENERIC_KIND id, count) of AvailableResources.Generic, counts >= 1, by kind
    struct Csi { uint32_t plugin, has_topology; std::vector<swp_seg> segs; };
    std::vector<Csi> csi;                                         // Description.CSIInfo (swp_node_set_csi)
};

// service -> (node -> count): the engine's index of where a service runs (the exception lists of a batch are read off it). Service ids are
// interned, so small: a vector by id — a placement's update is then an index, and the loops that book thousands of placements can
// prefetch it (an unordered_map was a dependent miss per placement); ids beyond the dense range (a caller's own numbering) live in a map
struct SvcNodes {
    static constexpr uint32_t DENSE_MAX = 1u << 20;
    std::vector<FlatMap32> dense;
    std::unordered_map<uint32_t, FlatMap32> sparse;
    FlatMap32& operator[](uint32_t s) {
        if (s >= DENSE_MAX) return sparse[s];
        if (s >= dense.size()) dense.resize(std::max<size_t>((size_t)s + 1, std::min<size_t>(dense.size() * 2, DENSE_MAX)));
        return dense[s];
    }
    const FlatMap32* find(uint32_t s) const {
        if (s < dense.size()) return &dense[s];
        if (s < DENSE_MAX) return nullptr;
        auto it = sparse.find(s);
        return it == sparse.end() ? nullptr : &it->second;
    }
    void clear() { dense.clear(); sparse.clear(); }
    void prefetch(uint32_t s) const { if (s < dense.size()) __builtin_prefetch(&dense[s]); }
    void prefetch_entries(uint32_t s) const { if (s < dense.size() && !dense[s].v.empty()) __builtin_prefetch(dense[s].v.data()); }
};

struct HostVolume {   // swp_volume_upsert / swp_volume_set_usage
    bool present = false;
    swp_volume spec{};
    std::vector<uint32_t> groups;   // byGroup memberships: spec.group first, then every group a later addOrUpdateVolume named (never pruned: volumes.go:74-78)
    std::vector<uint32_t> topo_off{0};
    std::vector<swp_seg> segs;
    swp_volume_usage use{0, 0, SWP_PIN_NONE, 0};
};

inline uint64_t port_key(uint32_t proto, uint32_t port) { return ((uint64_t)proto << 32) | port; }
inline uint64_t svcver_key(uint32_t svc, uint64_t ver) { return ((uint64_t)svc << 40) ^ ver * 0x9E3779B97F4A7C15ull; }

}  // namespace

struct swp_batch {
    uint32_t T = 0;
    std::vector<swp_task_desc> tasks;      // the descriptors: one per task, or the caller's templates with task_tmpl naming each task's
    std::vector<uint32_t> task_tmpl;       // (swp_batch_prepare_templates) [T] index into `tasks`; empty: tasks[i] is task i's
    const swp_task_desc& desc(uint32_t i) const { return task_tmpl.empty() ? tasks[i] : tasks[task_tmpl[i]]; }
    PinVec<RTask> rt;
    std::vector<uint32_t> svc_global;      // batch-local service -> SERVICE id
    std::vector<uint32_t> list_off;        // [n_svc+1]
    std::vector<uint32_t> list_node0, list_svc0, list_fail0;   // pristine lists
    std::vector<uint32_t> list_cnt0;                            // [n_svc] entries in use at the front of a service's range
    uint32_t n_list0 = 0, max_list = 0;    // entries the exception lists start with (all services); the longest list's length, free slots included
    std::vector<uint32_t> prow, pnode;     // scatter sources for portmap
    std::vector<uint64_t> port_keys;       // batch-local port -> (proto,port)
    std::vector<uint32_t> pset_off, pset_ids;
    std::vector<uint32_t> con_off, plat_off, plug_off, plug_req;
    std::vector<DevConstraint> cons;
    std::vector<uint2> plats;
    std::vector<uint4> triples;
    uint32_t n_con = 0, n_plat = 0, n_plug = 0, n_sc = 0, n_svc = 0, n_ports = 0;
    uint32_t n_nodes_prepared = 0;         // e->n_nodes when the batch was prepared: its bitmap rows are sized for that
    bool ran = false;
    int64_t unit_cpu = 1, unit_mem = 1;   // k_resolve5: gcd of the batch's reservations (RTask.kc / km count these units)
    bool units_ok = false;                // every reservation fits 2^30 units
    // k_resolve5 exact mode: the distinct cpu / memory reservations of the batch as demand classes (RTask.flags carries the
    // class indices); usable when they fit R5's row budget
    std::vector<int32_t> thr;
    uint32_t n_dc = 0, n_dm = 0;
    bool exact_ok = false;
    // k_resolve6 (block resolver): the same classes with the thresholds in raw units (NanoCPUs / bytes); usable whenever the
    // class indices fit the 8 bits RTask.flags has for each
    std::vector<int64_t> thr64;
    bool classes_ok = false;
    // generic reservations (filter.go:86-91): the distinct (kind, value) pairs of the batch's tasks are more demand-class rows,
    // rg[r] = {count[kind_r] >= value_r}, sorted by (kind, value); a task names the rows of its set
    bool has_generic = false;
    std::vector<uint32_t> tg;                 // [T] batch-local generic set of the task, 0 = none
    std::vector<uint32_t> gs_off, gs_row;     // CSR: batch-local set -> rows
    std::vector<uint32_t> rg_kind, rg_k0, rg_k1;   // per row: kind id, first / one-past-last row of the same kind
    std::vector<int32_t> rg_val;
    DevBuf d_tg, d_gs_off, d_gs_row, d_rg_kind, d_rg_k0, d_rg_k1, d_rg_val, d_rg;
    std::vector<uint32_t> tmpl;   // [T] the first task with this task's descriptor (identical tasks: swp_resolve6.hpp R6Args.tmpl)
    DevBuf d_tmpl;
    // tasks with cluster mounts (swp_volumes.hpp): csi_of[task] = its index among them (0xFFFFFFFF: none), csi_set[that] = its mount set
    std::vector<uint32_t> csi_of, csi_set, csi_task;
    DevBuf d_csi_of, d_csi_set, d_vrows, d_att;
    std::vector<uint32_t> h_att;   // [csi tasks][SWP_MAX_MOUNTS] after fetch / results

    DevBuf d_up;    // the upload arena: every small table of the batch in ONE allocation, filled by ONE copy from h_up (upload_batch)
    PinBuf h_up;
    DevBuf d_rt, d_out, d_hist, d_X, d_list_node, d_list_svc, d_list_fail, d_list_node0, d_list_svc0, d_list_fail0, d_list_off;
    DevBuf d_hmat, d_emat;   // the scan resolver's (service, node) matrices, allocated when a stretch first goes to it
    DevBuf d_prow, d_pnode, d_portmap, d_pset_off, d_pset_ids;
    DevBuf d_con_off, d_cons, d_plat_off, d_plats, d_plug_off, d_plug_req, d_triples;
    DevBuf d_con, d_plat, d_plug, d_sc, d_log_node, d_log_task, d_log_prev, d_last, d_inf_task, d_inf_pos, d_ctl;
    DevBuf d_seg_off, d_seg_len, d_ent_ci, d_ent_scpu, d_ent_smem, d_seg_alloc;   // explain pass: per-node commit segments
    uint32_t x_ninf = 0;             // unplaceable tasks of the last run (their list is in hx_in)
    DevBuf d_xrows;                  // their Explain rows, gathered for the copy to the host
    PinBuf hx_rows;
    DevBuf d_xpack, d_xdiff, d_xnr;  // explain pass by group: one packed upload (groups, entries, orders) + the difference planes
    PinBuf hx_in, hx_pack;           // its host staging: the unplaceable list as it comes back; the packed upload
    std::vector<uint32_t> xg_of;     // [T] explain group of the task (0xFFFFFFFF: per-task pass), fixed when the batch is built
    std::vector<XGroup> xg_proto;    // the groups (off / cnt / doff filled per run; pad = reservation pair)
    uint32_t xg_pairs = 0;
    std::vector<uint32_t> hx_fill, hx_pair_cnt, hx_compact;
    DevBuf d_qres;                         // k_resolve5: [n_nodes][2] residuals in resource units
    DevBuf d_thr;                          // k_resolve5 exact mode: thresholds of the demand-class rows
    DevBuf d_trows;                        // k_resolve6, task-rows mode: [block][n_words]
    DevBuf d_thr64, d_planes6, d_rr6, d_blk6;   // k_resolve6: raw thresholds, level planes, demand-class rows, control block
    // segments of the batch: runs of identical tasks (k_waterfill) and the stretches between them (the resolvers)
    struct Seg { uint32_t j0, n; bool run; };
    std::vector<Seg> segs;
    DevBuf d_wf;                           // k_waterfill scratch: [3][n_nodes] u32
    // node-range shard protocol (swp_shard_*): commits / unplaceable tasks of ALL shards so far, this shard's results
    bool shard_open = false, shard_apply_timed = false;
    uint32_t shard_ncommit = 0, shard_ninf = 0;
    std::vector<int32_t> shard_out;        // [T] shard-local node of the tasks placed here, else -1
    std::vector<ShardPickDev> shard_mine;  // upload sources of the last commit (alive until the stream has consumed them)
    std::vector<ShardInfDev> shard_infs;
    DevBuf d_prop, d_picks, d_infs;
    DevBuf d_cmask, d_crank, d_cidx;       // the block resolver's compact index of a round (k_r6_compact)
    // a batch of a shard SET (swp_shardset.hpp): one part per shard, prepared from the same task list; the last run's results
    bool is_set = false;
    std::vector<swp_batch*> parts;
    std::vector<int32_t> set_shard, set_node;   // [T] owner shard (-1: no suitable node) and its local node index
    std::vector<uint32_t> set_hist;             // [T][SWP_NFILTERS] summed over the shards
    int32_t set_single = -1;                    // >= 0: only that shard holds nodes, its own batch path ran
};

// A shard set (swp_shardset.hpp): the engines of one process behind one handle. The set owns the global node index space.
struct ShardSet {
    std::vector<swp_engine*> sh;   // owned
    uint32_t cap = 0;              // node slots per shard: global index i = shard i / cap, local index i % cap
    Interner nodes;                // SWP_SPACE_NODE_ID of the set (lowest free index first, as every engine's)
    uint32_t hi = 0;               // highest global index handed out so far + 1
    // The UNION engine (owned, on shard 0's device): one more replica of every replicated table, and — only while a call for task
    // GROUPS runs — the whole nodeSet by global index, copied from the shards' mirrors (swp_shardset.hpp ss::schedule_groups).
    swp_engine* uni = nullptr;
    std::vector<swp_engine*> all;  // sh + uni: who a replicated call goes to
    bool broken = false;           // a shard holds a node id under another index than the set's (ss::intern): refused until swp_reset
};

struct swp_engine {
    swp_config cfg{};
    ShardSet* set = nullptr;   // != nullptr: this handle is a shard SET (swp_shardset_create): no device state of its own, every call is routed
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    Interner spaces[SWP_SPACE_COUNT];
    uint32_t role_worker = 0, role_manager = 0;

    std::vector<HostNode> nodes;
    uint32_t n_nodes = 0;   // highest node index in use + 1
    uint32_t n_present = 0;
    SvcNodes svc_nodes;                                    // service -> node -> count (>0), nodes ascending
    std::unordered_map<uint32_t, std::unordered_set<uint32_t>> fail_nodes;            // service -> nodes with failure records
    std::unordered_map<uint64_t, std::unordered_set<uint32_t>> port_nodes;            // (proto,port) -> nodes

    // predicate sets, de-duplicated by content
    std::vector<std::vector<swp_constraint>> con_sets{1};
    std::vector<std::vector<swp_platform>> plat_sets{1};
    struct PlugSet { std::vector<uint32_t> required; uint32_t log = 0; };
    std::vector<PlugSet> plug_sets{1};
    std::vector<std::vector<swp_port>> port_sets{1};
    std::vector<std::vector<swp_spread>> spread_sets{1};
    std::vector<std::vector<swp_generic>> gen_sets{1};
    std::vector<std::vector<swp_mount>> mount_sets{1};
    std::map<std::string, uint32_t> con_index, plat_index, plug_index, port_index, spread_index, gen_index, mount_index;

    // CSI volumes (swp_volumes.hpp): host mirror and its device tables
    std::vector<HostVolume> volumes;          // by SWP_SPACE_VOLUME index
    bool vol_static_dirty = true;             // a volume's spec, a node's CSI info or the node set changed: tables + topology bitmaps again
    bool vol_dyn_dirty = true;                // usage numbers changed on the host
    uint32_t dev_vol_words = 0;
    DevBuf d_vflags, d_vdyn, d_vT, d_grp_off, d_grp_vol, d_ms_off, d_ms_mount;
    DevBuf d_ncsi_off, d_ncsi, d_ncsi_seg, d_vdriver, d_vtopo_off, d_topo_off, d_vseg;
    bool has_volumes() const { return volumes.size() > 0 || mount_sets.size() > 1; }

    // label columns of attr[][]: 0 id, 1 hostname, 2 os, 3 arch, 4.. labels
    std::map<uint32_t, uint32_t> node_label_col, engine_label_col;
    uint32_t n_cols = 4;

    // device node arrays
    uint32_t ncap = 0;
    DevBuf d_flags, d_cpu, d_mem, d_total, d_os, d_arch, d_attr, d_ip, d_plug_off, d_plug_ids, d_ready, d_valid;
    DevBuf d_save_cpu, d_save_mem, d_save_total;
    DevBuf d_gcnt, d_save_gcnt;      // [dev_gkinds][ncap] int32: generic counts per (kind, node), part of the dynamic state
    uint32_t dev_gkinds = 0;
    bool dev_static_dirty = true;    // flags/os/arch/attr/ip/plugins need a re-upload
    bool dev_dynamic_dirty = true;   // cpu/mem/total need a re-upload
    // Rows that swp_node_update_dynamic changed since the last flush (a drain flips one flag word, a correction moves one node's
    // residuals): flush_nodes scatters exactly those — one pinned block, one copy, k_scatter_rows — instead of rebuilding and uploading
    // every array, as long as they are few and nothing else is pending. `dev_flags_dirty`: some flag word changed, so a fallback to the
    // whole-array path must carry the flags along (they are not one of the "dynamic" arrays).
    std::vector<uint32_t> dirty_rows;
    bool dev_flags_dirty = false;
    PinBuf rows_stage;
    DevBuf d_rows;
    hipEvent_t rows_ev = nullptr;    // the last scatter's copy has read rows_stage
    uint32_t dev_cols = 0;

    // saved host state for swp_state_restore
    struct Saved {
        bool valid = false;
        std::vector<HostNode> nodes;
        decltype(svc_nodes) svc_nodes_;
        decltype(port_nodes) port_nodes_;
    } saved;

    // RCCL communicator of this rank (swp_rccl_init); librccl.so is loaded on first use
    void* rccl_comm = nullptr;
    uint32_t rccl_rank = 0, rccl_ranks = 0;

    std::vector<hipEvent_t> ev_scan;   // SWP_CFG_PROFILE: a pair per stretch the scan resolver took (run_blocks); ev_scan_used of them in this batch
    uint32_t ev_scan_used = 0, scan_tasks_batch = 0, scan_stretches_batch = 0;
    bool r6_compact_hint = false;   // the last batch's rounds used a compact index (re-placements after a drain): the next one starts with it
    swp_stats_t stats{};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_pool;   // per-launch kernel timing (SWP_CFG_PROFILE)
    bool host_dirty_since_save = true;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        last_error = buf;
        return code;
    }
};

#define HIPCHECK(e, call)                                                                             \
    do {                                                                                              \
        hipError_t _r = (call);                                                                       \
        if (_r != hipSuccess) return (e)->fail(SWP_EHIP, "%s: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

namespace {

uint32_t n_words_of(uint32_t n) { return (n + 63) / 64; }

// a node index that swp_node_remove handed back to the pool and no swp_intern has taken again: a caller that still holds it is stale
// (the next new node id will be given exactly this index)
bool node_index_released(const swp_engine* e, uint32_t node) {
    const Interner& sp = e->spaces[SWP_SPACE_NODE_ID];
    return node < sp.freed.size() && sp.freed[node] != 0;
}

template <class T>
int upload(swp_engine* e, DevBuf& b, const PinVec<T>& v, size_t min_elems = 1);
template <class T>
int upload(swp_engine* e, DevBuf& b, const std::vector<T>& v, size_t min_elems = 1) {
    size_t bytes = std::max(v.size(), min_elems) * sizeof(T);
    HIPCHECK(e, b.reserve(bytes));
    if (!v.empty()) HIPCHECK(e, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, e->stream));
    return SWP_OK;
}
template <class T>
int upload(swp_engine* e, DevBuf& b, const PinVec<T>& v, size_t min_elems) {
    size_t bytes = std::max(v.size(), min_elems) * sizeof(T);
    HIPCHECK(e, b.reserve(bytes));
    if (!v.empty()) HIPCHECK(e, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, e->stream));
    return SWP_OK;
}

void engine_reset_nodes(swp_engine* e) {
    e->nodes.clear();
    e->n_nodes = 0;
    e->n_present = 0;
    e->svc_nodes.clear();
    e->fail_nodes.clear();
    e->port_nodes.clear();
    e->dev_static_dirty = true;
    e->dev_dynamic_dirty = true;
    e->saved.valid = false;
    for (HostVolume& v : e->volumes) v.use = swp_volume_usage{0, 0, SWP_PIN_NONE, 0};   // (usage names node indices: none of them survives)
    e->vol_static_dirty = e->vol_dyn_dirty = true;
}

uint32_t label_value(const std::vector<swp_kv>& kv, uint32_t key) {
    for (const swp_kv& x : kv)
        if (x.key == key) return x.value;
    return 0;
}

// Push the host mirror of the node rows to the device (bulk; node counts are ≤ ~1e5 and a tick
// is preceded by few mutations, so whole-array uploads of the dirty group are the simple choice).
int flush_volumes(swp_engine* e);
int flush_nodes(swp_engine* e) {
    if (e->dev_static_dirty) e->vol_static_dirty = true;   // the node set or a node's CSI info may have changed: the topology bitmaps follow
    uint32_t N = e->n_nodes;
    uint32_t need_cap = std::max<uint32_t>(64, ((N + 63) / 64) * 64);
    if (e->dev_gkinds != (uint32_t)e->spaces[SWP_SPACE_GENERIC_KIND].strs.size()) e->dev_dynamic_dirty = true;   // a new kind: a new row
    bool grow = need_cap > e->ncap || e->dev_cols != e->n_cols;
    if (grow) {
        e->ncap = std::max(need_cap, e->ncap);
        e->dev_static_dirty = e->dev_dynamic_dirty = true;
    }
    uint32_t cap = e->ncap;
    if (e->dev_static_dirty) {
        std::vector<uint32_t> flags(cap, 0), os(cap, 0), arch(cap, 0), attr((size_t)e->n_cols * cap, 0), ip((size_t)cap * 4, 0);
        std::vector<uint32_t> poff(cap + 1, 0), pids;
        for (uint32_t n = 0; n < N; ++n) {
            const HostNode& h = e->nodes[n];
            poff[n] = (uint32_t)pids.size();
            if (!h.present) continue;
            flags[n] = (h.row.flags & 0x7FFFFFFFu) | DEV_VALID;
            os[n] = h.row.os;
            arch[n] = h.row.arch;
            attr[0 * (size_t)cap + n] = h.row.id_fold;
            attr[1 * (size_t)cap + n] = h.row.hostname_fold;
            attr[2 * (size_t)cap + n] = h.row.os_fold;
            attr[3 * (size_t)cap + n] = h.row.arch_fold;
            for (auto& kc : e->node_label_col) attr[(size_t)kc.second * cap + n] = label_value(h.labels, kc.first);
            for (auto& kc : e->engine_label_col) attr[(size_t)kc.second * cap + n] = label_value(h.elabels, kc.first);
            for (int q = 0; q < 4; ++q)
                ip[(size_t)n * 4 + q] = ((uint32_t)h.row.ip[4 * q] << 24) | ((uint32_t)h.row.ip[4 * q + 1] << 16) | ((uint32_t)h.row.ip[4 * q + 2] << 8) | h.row.ip[4 * q + 3];
            pids.insert(pids.end(), h.plugins.begin(), h.plugins.end());
        }
        for (uint32_t n = N; n <= cap; ++n) poff[n] = (uint32_t)pids.size();
        int rc;
        if ((rc = upload(e, e->d_flags, flags))) return rc;
        if ((rc = upload(e, e->d_os, os))) return rc;
        if ((rc = upload(e, e->d_arch, arch))) return rc;
        if ((rc = upload(e, e->d_attr, attr))) return rc;
        if ((rc = upload(e, e->d_ip, ip))) return rc;
        if ((rc = upload(e, e->d_plug_off, poff))) return rc;
        if ((rc = upload(e, e->d_plug_ids, pids))) return rc;
        HIPCHECK(e, e->d_ready.reserve((size_t)(cap / 64) * 8));
        HIPCHECK(e, e->d_valid.reserve((size_t)(cap / 64) * 8));
        HIPCHECK(e, hipStreamSynchronize(e->stream));   // host vectors die at scope exit
        e->dev_cols = e->n_cols;
        e->dev_static_dirty = false;
        e->dev_flags_dirty = false;
    }
    if (!e->dev_dynamic_dirty && !e->dirty_rows.empty()) {
        if (e->dirty_rows.size() * 4 > (size_t)N) e->dev_dynamic_dirty = true;   // most of the node set: the whole arrays are cheaper
        else {
            const size_t R = e->dirty_rows.size();
            if (e->rows_ev) HIPCHECK(e, hipEventSynchronize(e->rows_ev));   // (the block is free again: long since, as a rule)
            else HIPCHECK(e, hipEventCreateWithFlags(&e->rows_ev, hipEventDisableTiming));
            HIPCHECK(e, e->rows_stage.reserve(R * sizeof(DevRow)));
            DevRow* rows = static_cast<DevRow*>(e->rows_stage.p);
            for (size_t i = 0; i < R; ++i) {   // (a node listed twice carries its latest row twice)
                const uint32_t n = e->dirty_rows[i];
                const HostNode& h = e->nodes[n];
                DevRow& r = rows[i];
                r.node = n; r.pad = 0;
                r.flags = h.present ? ((h.row.flags & 0x7FFFFFFFu) | DEV_VALID) : 0u;
                r.total = h.present ? h.row.total : 0u;
                r.cpu = h.present ? h.row.cpu : 0;
                r.mem = h.present ? h.row.mem : 0;
            }
            HIPCHECK(e, e->d_rows.reserve(R * sizeof(DevRow)));   // (stream order: the last scatter's kernel is through before this copy lands)
            HIPCHECK(e, hipMemcpyAsync(e->d_rows.p, rows, R * sizeof(DevRow), hipMemcpyHostToDevice, e->stream));
            HIPCHECK(e, hipEventRecord(e->rows_ev, e->stream));
            hipLaunchKernelGGL(k_scatter_rows, dim3(((uint32_t)R + 255) / 256), dim3(256), 0, e->stream, (uint32_t)R, e->d_rows.as<DevRow>(), e->d_flags.as<uint32_t>(),
                               e->d_cpu.as<long long>(), e->d_mem.as<long long>(), e->d_total.as<uint32_t>());
            HIPCHECK(e, hipGetLastError());
            e->dirty_rows.clear();
            e->dev_flags_dirty = false;
        }
    }
    if (e->dev_flags_dirty) {   // a flag word changed and the rows go up as whole arrays: the flags with them
        std::vector<uint32_t> flags(cap, 0);
        for (uint32_t n = 0; n < N; ++n)
            if (e->nodes[n].present) flags[n] = (e->nodes[n].row.flags & 0x7FFFFFFFu) | DEV_VALID;
        int rc;
        if ((rc = upload(e, e->d_flags, flags))) return rc;
        HIPCHECK(e, hipStreamSynchronize(e->stream));
        e->dev_flags_dirty = false;
    }
    if (e->dev_dynamic_dirty) {
        e->dirty_rows.clear();
        std::vector<int64_t> cpu(cap, 0), mem(cap, 0);
        std::vector<uint32_t> total(cap, 0);
        for (uint32_t n = 0; n < N; ++n) {
            const HostNode& h = e->nodes[n];
            if (!h.present) continue;
            cpu[n] = h.row.cpu;
            mem[n] = h.row.mem;
            total[n] = h.row.total;
        }
        int rc;
        if ((rc = upload(e, e->d_cpu, cpu))) return rc;
        if ((rc = upload(e, e->d_mem, mem))) return rc;
        if ((rc = upload(e, e->d_total, total))) return rc;
        const uint32_t K = (uint32_t)e->spaces[SWP_SPACE_GENERIC_KIND].strs.size();   // kind ids are 1 .. K - 1
        std::vector<int32_t> gcnt;
        if (K > 1) {
            gcnt.assign((size_t)K * cap, 0);
            for (uint32_t n = 0; n < N; ++n) {
                const HostNode& h = e->nodes[n];
                if (!h.present) continue;
                for (const auto& kv : h.gen) gcnt[(size_t)kv.first * cap + n] = (int32_t)kv.second;
            }
            if ((rc = upload(e, e->d_gcnt, gcnt))) return rc;
        }
        e->dev_gkinds = K;
        HIPCHECK(e, hipStreamSynchronize(e->stream));
        e->dev_dynamic_dirty = false;
    }
    return flush_volumes(e);
}

// The volume tables on the device (swp_volumes.hpp): flags, groups, mount sets, the node CSI table and the volumes' topologies — and from
// the last two the topology bitmaps (k_vol_topology) — whenever a volume's spec, a node or a mount set changed; the usage numbers whenever
// the host changed them.
VolView vol_view(swp_engine* e) {
    VolView v{};
    if (!e->has_volumes()) return v;
    v.n_vol = (u32)e->volumes.size();
    v.n_words = e->dev_vol_words;
    v.vflags = e->d_vflags.as<u32>();
    v.vdyn = e->d_vdyn.as<VolDyn>();
    v.T = e->d_vT.as<u64>();
    v.grp_off = e->d_grp_off.as<u32>();
    v.grp_vol = e->d_grp_vol.as<u32>();
    v.ms_off = e->d_ms_off.as<u32>();
    v.ms_mount = e->d_ms_mount.as<VolMount>();
    return v;
}
int flush_volumes(swp_engine* e) {
    if (!e->has_volumes()) return SWP_OK;
    const uint32_t N = e->n_nodes, Wn = n_words_of(N), V = (uint32_t)e->volumes.size();
    int rc;
    if (e->vol_static_dirty || e->dev_vol_words != Wn) {
        std::vector<uint32_t> vflags(V, 0), vdriver(V, 0), vtopo_off(V + 1, 0), topo_off(1, 0), vseg;
        const uint32_t G = (uint32_t)e->spaces[SWP_SPACE_VOLUME_GROUP].strs.size();
        std::vector<std::vector<uint32_t>> members(G);
        for (uint32_t v = 0; v < V; ++v) {
            const HostVolume& hv = e->volumes[v];
            vtopo_off[v] = (uint32_t)topo_off.size() - 1;
            if (!hv.present) continue;
            vflags[v] = (hv.spec.active ? VOL_ACTIVE : 0u) | (hv.spec.scope == SWP_VOL_SCOPE_MULTI_NODE ? VOL_MULTI : 0u) | ((hv.spec.sharing & 3u) << VOL_SHARING_SHIFT);
            vdriver[v] = hv.spec.driver;
            for (uint32_t gq : hv.groups)
                if (gq < G) members[gq].push_back(v);
            for (uint32_t t = 0; t < hv.spec.n_topologies; ++t) {
                for (uint32_t q = hv.topo_off[t]; q < hv.topo_off[t + 1]; ++q) { vseg.push_back(hv.segs[q].key); vseg.push_back(hv.segs[q].value); }
                topo_off.push_back((uint32_t)vseg.size() / 2);
            }
        }
        vtopo_off[V] = (uint32_t)topo_off.size() - 1;
        std::vector<uint32_t> grp_off(G + 1, 0), grp_vol;
        for (uint32_t g = 0; g < G; ++g) {
            grp_off[g] = (uint32_t)grp_vol.size();
            grp_vol.insert(grp_vol.end(), members[g].begin(), members[g].end());   // ascending volume index = creation order
        }
        grp_off[G] = (uint32_t)grp_vol.size();
        std::vector<uint32_t> ms_off(e->mount_sets.size() + 1, 0);
        std::vector<swp_mount> ms;
        for (size_t q = 0; q < e->mount_sets.size(); ++q) {
            ms_off[q] = (uint32_t)ms.size();
            ms.insert(ms.end(), e->mount_sets[q].begin(), e->mount_sets[q].end());
        }
        ms_off[e->mount_sets.size()] = (uint32_t)ms.size();
        std::vector<uint32_t> ncsi_off(N + 1, 0), ncsi, nseg;
        for (uint32_t n = 0; n < N; ++n) {
            ncsi_off[n] = (uint32_t)ncsi.size() / 4;
            const HostNode& h = e->nodes[n];
            if (!h.present) continue;
            for (const HostNode::Csi& c : h.csi) {
                ncsi.push_back(c.plugin);
                ncsi.push_back(c.has_topology);
                ncsi.push_back((uint32_t)nseg.size() / 2);
                ncsi.push_back((uint32_t)c.segs.size());
                for (const swp_seg& sg : c.segs) { nseg.push_back(sg.key); nseg.push_back(sg.value); }
            }
        }
        ncsi_off[N] = (uint32_t)ncsi.size() / 4;
        if ((rc = upload(e, e->d_vflags, vflags))) return rc;
        if ((rc = upload(e, e->d_vdriver, vdriver))) return rc;
        if ((rc = upload(e, e->d_vtopo_off, vtopo_off))) return rc;
        if ((rc = upload(e, e->d_topo_off, topo_off))) return rc;
        if ((rc = upload(e, e->d_vseg, vseg, 2))) return rc;
        if ((rc = upload(e, e->d_grp_off, grp_off))) return rc;
        if ((rc = upload(e, e->d_grp_vol, grp_vol))) return rc;
        if ((rc = upload(e, e->d_ms_off, ms_off))) return rc;
        if ((rc = upload(e, e->d_ms_mount, ms))) return rc;
        if ((rc = upload(e, e->d_ncsi_off, ncsi_off))) return rc;
        if ((rc = upload(e, e->d_ncsi, ncsi, 4))) return rc;
        if ((rc = upload(e, e->d_ncsi_seg, nseg, 2))) return rc;
        HIPCHECK(e, e->d_vT.reserve((size_t)std::max<uint32_t>(V, 1) * std::max<uint32_t>(Wn, 1) * 8));
        VolTopoArgs ta{};
        ta.n_nodes = N;
        ta.n_words = Wn;
        ta.n_vol = V;
        ta.node_csi_off = e->d_ncsi_off.as<u32>();
        ta.csi = e->d_ncsi.as<u32>();
        ta.csi_seg = e->d_ncsi_seg.as<u32>();
        ta.vol_driver = e->d_vdriver.as<u32>();
        ta.vol_topo_off = e->d_vtopo_off.as<u32>();
        ta.topo_off = e->d_topo_off.as<u32>();
        ta.vol_seg = e->d_vseg.as<u32>();
        ta.T = e->d_vT.as<u64>();
        hipError_t r = launch_vol_topology(ta, e->stream);
        if (r != hipSuccess) return e->fail(SWP_EHIP, "k_vol_topology launch: %s", hipGetErrorString(r));
        HIPCHECK(e, hipStreamSynchronize(e->stream));   // host vectors die at scope exit
        e->dev_vol_words = Wn;
        e->vol_static_dirty = false;
        e->vol_dyn_dirty = true;
    }
    if (e->vol_dyn_dirty) {
        std::vector<swp_volume_usage> dyn(V);
        for (uint32_t v = 0; v < V; ++v) dyn[v] = e->volumes[v].use;
        if ((rc = upload(e, e->d_vdyn, dyn))) return rc;
        HIPCHECK(e, hipStreamSynchronize(e->stream));
        e->vol_dyn_dirty = false;
    }
    return SWP_OK;
}

NodeView node_view(swp_engine* e) {
    NodeView nv{};
    nv.n_nodes = e->n_nodes;
    nv.n_words = n_words_of(e->n_nodes);
    nv.ncap = e->ncap;
    nv.flags = e->d_flags.as<uint32_t>();
    nv.cpu = e->d_cpu.as<long long>();
    nv.mem = e->d_mem.as<long long>();
    nv.total = e->d_total.as<uint32_t>();
    nv.os = e->d_os.as<uint32_t>();
    nv.arch = e->d_arch.as<uint32_t>();
    nv.attr = e->d_attr.as<uint32_t>();
    nv.ip = e->d_ip.as<uint32_t>();
    nv.plug_off = e->d_plug_off.as<uint32_t>();
    nv.plug_ids = e->d_plug_ids.as<uint32_t>();
    nv.role_worker = e->role_worker;
    nv.role_manager = e->role_manager;
    return nv;
}

template <class T>
std::string bytes_of(const T* p, size_t n) { return std::string(reinterpret_cast<const char*>(p), n * sizeof(T)); }

// ---- batch construction -------------------------------------------------------------------------
// tmpl_idx != nullptr: `descs` holds n_tmpl TEMPLATES and task i carries descs[tmpl_idx[i]] (swp_batch_prepare_templates): the caller has
// done the de-duplication, the per-task pass is an array lookup instead of a hash of 64 bytes
int build_batch(swp_engine* e, const swp_task_desc* descs, uint32_t T, swp_batch* b, const uint32_t* weights = nullptr, const uint32_t* tmpl_idx = nullptr, uint32_t n_tmpl = 0) {
    struct TaskView {   // tasks[i] whatever the form
        const swp_task_desc* d;
        const uint32_t* ix;
        const swp_task_desc& operator[](uint32_t i) const { return ix ? d[ix[i]] : d[i]; }
    } tasks{descs, tmpl_idx};
    HostSpan sp("build_batch: descriptors -> records");
    if (tmpl_idx)
        for (uint32_t i = 0; i < T; ++i)
            if (tmpl_idx[i] >= n_tmpl) return e->fail(SWP_EINVAL, "task %u names template %u of %u", i, tmpl_idx[i], n_tmpl);
    const bool dbg_prep = getenv("SWP_DEBUG_PREPARE") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!dbg_prep) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[swp]   build_batch %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
        tp = now;
    };
    b->T = T;
    if (tmpl_idx) {   // (no 64-byte copy per task: a million tasks are 64 MB)
        b->tasks.assign(descs, descs + n_tmpl);
        b->task_tmpl.assign(tmpl_idx, tmpl_idx + T);
    } else {
        b->tasks.assign(descs, descs + T);
        b->task_tmpl.clear();
    }
    HIPCHECK(e, b->rt.resize(T));
    std::unordered_map<uint32_t, uint32_t> con_local, plat_local, plug_local, pset_local, svc_local;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> sc_local;
    std::vector<uint32_t> con_ids{0}, plat_ids{0}, plug_ids{0}, pset_ids_global;
    std::vector<uint64_t> svc_ver;          // per local service: spec version used for failure lookup
    std::vector<uint32_t> svc_ntasks;
    std::vector<uint32_t> task_rank(T);
    std::unordered_map<uint64_t, uint32_t> port_local;
    // Tasks of one service carry the same descriptor: everything below that depends on the descriptor alone is done once per DISTINCT
    // descriptor (tmpl_of[i] = the first task with task i's descriptor) and copied to the others.
    struct DescKey {
        uint64_t w[8];
        bool operator==(const DescKey& o) const { return std::memcmp(w, o.w, sizeof w) == 0; }
    };
    struct DescHash {
        size_t operator()(const DescKey& k) const {
            uint64_t h = 0x9E3779B97F4A7C15ull;
            for (uint64_t x : k.w) h = (h ^ x) * 0xFF51AFD7ED558CCDull + (h >> 29);
            return (size_t)h;
        }
    };
    static_assert(sizeof(swp_task_desc) == sizeof(DescKey), "descriptor key");
    // descriptor -> the first task that carries it: open addressing over (hash, task) pairs — one probe and one 64-byte compare per
    // task in the common case, no node allocations (1M tasks over 10k services: the map lookups were most of the preparation)
    struct Slot { uint64_t h; uint32_t first, pad; };
    std::vector<Slot> table((size_t)1 << 12, Slot{0, 0, 0});
    size_t table_used = 0;
    auto desc_hash = [](const swp_task_desc& d) {
        DescKey key;
        std::memcpy(key.w, &d, sizeof key);
        return (uint64_t)DescHash()(key) | 1ull;   // 0 = empty slot
    };
    auto same_desc = [&](uint32_t a, uint32_t c) { return std::memcmp(&tasks[a], &tasks[c], sizeof(swp_task_desc)) == 0; };   // (TaskView: references into the caller's array)
    std::vector<uint32_t> tmpl_of(T), firsts, first_of_tmpl(tmpl_idx ? n_tmpl : 0, 0xFFFFFFFFu);
    mark("descriptors kept, per-task arrays");

    for (uint32_t i = 0; i < T; ++i) {
        const swp_task_desc& d = tasks[i];
        {
            uint32_t hit = 0xFFFFFFFFu;
            uint64_t h = 0;
            size_t at = 0;
            if (tmpl_idx) {   // the caller's templates: the first task of each
                uint32_t& f = first_of_tmpl[tmpl_idx[i]];
                if (f == 0xFFFFFFFFu) f = i;
                else hit = f;
                at = SIZE_MAX;
            } else if (i && same_desc(i, i - 1)) hit = tmpl_of[i - 1];   // (service-major batches: a run of one descriptor)
            else {
                h = desc_hash(d);
                const size_t mask = table.size() - 1;
                for (at = (size_t)(h >> 7) & mask; table[at].h; at = (at + 1) & mask)
                    if (table[at].h == h && same_desc(table[at].first, i)) {
                        hit = table[at].first;
                        break;
                    }
            }
            if (hit != 0xFFFFFFFFu) {
                tmpl_of[i] = hit;
                const uint32_t sv = b->rt[hit].svc;
                task_rank[i] = svc_ntasks[sv];
                svc_ntasks[sv] += weights ? weights[i] : 1u;
                continue;
            }
            if (at != SIZE_MAX) table[at] = Slot{h, i, 0};
            if (at != SIZE_MAX && ++table_used * 2 > table.size()) {   // grow: re-insert by the stored hashes
                std::vector<Slot> bigger(table.size() * 4, Slot{0, 0, 0});
                const size_t m2 = bigger.size() - 1;
                for (const Slot& sl : table)
                    if (sl.h) {
                        size_t q = (size_t)(sl.h >> 7) & m2;
                        while (bigger[q].h) q = (q + 1) & m2;
                        bigger[q] = sl;
                    }
                table.swap(bigger);
            }
            tmpl_of[i] = i;
            firsts.push_back(i);
        }
        if (d.constraint_set >= e->con_sets.size() || d.platform_set >= e->plat_sets.size() || d.plugin_set >= e->plug_sets.size() ||
            d.port_set >= e->port_sets.size())
            return e->fail(SWP_EINVAL, "task %u references an unknown predicate set", i);
        if (d.service >= e->spaces[SWP_SPACE_SERVICE].strs.size()) return e->fail(SWP_EINVAL, "task %u: unknown service id %u", i, d.service);
        if (d.spread_set >= e->spread_sets.size()) return e->fail(SWP_EINVAL, "task %u references an unknown spread set", i);
        if (d.spread_set && !weights) return e->fail(SWP_EUNSUPPORTED, "task %u has spread preferences: schedule it through swp_schedule_groups", i);
        if (d.generic_set >= e->gen_sets.size()) return e->fail(SWP_EINVAL, "task %u references an unknown generic set", i);
        if ((d.flags >> SWP_TASK_MOUNTS_SHIFT) >= e->mount_sets.size()) return e->fail(SWP_EINVAL, "task %u references an unknown mount set", i);
        // the exactness argument (feasibility only shrinks inside a batch) needs non-negative reservations; the API layer
        // rejects negative ones (manager/controlapi validateResources), a task that carries them stays on the Go path
        if (d.cpu < 0 || d.mem < 0) return e->fail(SWP_EUNSUPPORTED, "task %u has a negative resource reservation", i);
        RTask& r = b->rt[i];
        std::memset(&r, 0, sizeof r);
        r.cpu = d.cpu;
        r.mem = d.mem;
        r.flags = (d.flags & SWP_TASK_RES_ENABLED) ? RT_RES : 0u;
        if (d.flags & 0x2u) r.flags |= RT_UNCOUNTED;
        auto local = [](std::unordered_map<uint32_t, uint32_t>& m, std::vector<uint32_t>& ids, uint32_t g) -> uint32_t {
            if (g == 0) return 0u;
            auto it = m.find(g);
            if (it != m.end()) return it->second;
            uint32_t l = (uint32_t)ids.size();
            ids.push_back(g);
            m.emplace(g, l);
            return l;
        };
        r.cls_con = local(con_local, con_ids, d.constraint_set);
        r.cls_plat = local(plat_local, plat_ids, d.platform_set);
        r.cls_plug = local(plug_local, plug_ids, d.plugin_set);
        auto key = std::make_tuple(r.cls_con, r.cls_plat, r.cls_plug);
        auto sit = sc_local.find(key);
        if (sit == sc_local.end()) {
            sit = sc_local.emplace(key, (uint32_t)b->triples.size()).first;
            b->triples.push_back(uint4{r.cls_con, r.cls_plat, r.cls_plug, 0u});
        }
        r.sc = sit->second;
        if (d.port_set) {
            r.flags |= RT_PORTS;
            auto pit = pset_local.find(d.port_set);
            if (pit == pset_local.end()) {
                pit = pset_local.emplace(d.port_set, (uint32_t)pset_ids_global.size()).first;
                pset_ids_global.push_back(d.port_set);
            }
            r.pset = pit->second;
        }
        if (d.max_replicas) {
            r.flags |= RT_MAXREP;
            r.maxrep = d.max_replicas;
        }
        auto vit = svc_local.find(d.service);
        if (vit == svc_local.end()) {
            vit = svc_local.emplace(d.service, (uint32_t)b->svc_global.size()).first;
            b->svc_global.push_back(d.service);
            svc_ver.push_back(d.spec_version);
            svc_ntasks.push_back(0);
        } else if (svc_ver[vit->second] != d.spec_version) {
            return e->fail(SWP_EUNSUPPORTED, "one batch mixes spec versions of service %u (one-off tasks have none)", d.service);
        }
        r.svc = vit->second;
        task_rank[i] = svc_ntasks[r.svc];
        svc_ntasks[r.svc] += weights ? weights[i] : 1u;
    }
    // resource units of the round resolver: residual fits(need) <=> need/unit <= floor(residual/unit) when every need is a
    // multiple of the unit, so the kernel can keep exact residuals as 32-bit counts in LDS
    {
        int64_t gc = 0, gm = 0;
        for (uint32_t i : firsts) {
            gc = std::gcd(gc, b->rt[i].cpu);
            gm = std::gcd(gm, b->rt[i].mem);
        }
        b->unit_cpu = gc ? gc : 1;
        b->unit_mem = gm ? gm : 1;
        b->units_ok = true;
        for (uint32_t i : firsts) {
            const int64_t kc = b->rt[i].cpu / b->unit_cpu, km = b->rt[i].mem / b->unit_mem;
            if (kc >= R5_QLIM_HOST || km >= R5_QLIM_HOST) { b->units_ok = false; break; }
            b->rt[i].kc = (uint32_t)kc;
            b->rt[i].km = (uint32_t)km;
        }
        // demand classes: ResourceFilter (filter.go:77-84) becomes membership in two bitmap rows per task — the distinct cpu /
        // memory reservations of the batch, ascending; RTask.flags carries the task's two row indices. k_resolve5's exact mode
        // keeps the rows in LDS (thresholds in resource units, at most r5_max_rows() of them), k_resolve6 in global memory.
        b->exact_ok = b->classes_ok = false;
        b->thr.clear();
        b->thr64.clear();
        b->n_dc = b->n_dm = 0;
        {
            std::set<int64_t> sc_, sm_;
            for (uint32_t i : firsts)
                if (b->rt[i].flags & RT_RES) {
                    sc_.insert(b->rt[i].cpu);
                    sm_.insert(b->rt[i].mem);
                }
            if (sc_.size() <= RT_DCLS_MASK && sm_.size() <= RT_DCLS_MASK) {
                std::unordered_map<int64_t, uint32_t> ic, im;
                for (int64_t v : sc_) { ic[v] = (uint32_t)b->thr64.size(); b->thr64.push_back(v); }
                b->n_dc = (uint32_t)sc_.size();
                for (int64_t v : sm_) { im[v] = (uint32_t)b->thr64.size() - b->n_dc; b->thr64.push_back(v); }
                b->n_dm = (uint32_t)sm_.size();
                for (uint32_t i : firsts)
                    if (b->rt[i].flags & RT_RES) b->rt[i].flags |= (ic[b->rt[i].cpu] << RT_DC_SHIFT) | (im[b->rt[i].mem] << RT_DM_SHIFT);
                b->classes_ok = true;
                if (b->units_ok && b->n_dc + b->n_dm <= r5_max_rows()) {   // the same order in resource units (division by the gcd is monotone)
                    for (uint32_t c = 0; c < b->n_dc; ++c) b->thr.push_back((int32_t)(b->thr64[c] / b->unit_cpu));
                    for (uint32_t c = 0; c < b->n_dm; ++c) b->thr.push_back((int32_t)(b->thr64[b->n_dc + c] / b->unit_mem));
                    b->exact_ok = true;
                }
            }
        }
    }
    mark("templates + first records");
    for (uint32_t i = 0; i < T; ++i)   // every other task of a descriptor gets the record of the first one (its list slot follows below)
        if (tmpl_of[i] != i) b->rt[i] = b->rt[tmpl_of[i]];
    mark("records copied to the tasks");
    sp.next("build_batch: generic sets, explain groups, runs");
    // generic reservations: the distinct (kind, value) pairs become rows sorted by (kind, value); a task's set names its rows
    b->has_generic = false;
    b->tg.clear();
    b->gs_off.assign(2, 0);   // local set 0 = "none": the empty range
    b->gs_row.clear();
    b->rg_kind.clear();
    b->rg_val.clear();
    b->rg_k0.clear();
    b->rg_k1.clear();
    {
        std::set<std::pair<uint32_t, int32_t>> pairs;
        for (uint32_t i = 0; i < T; ++i)
            if (tasks[i].generic_set)
                for (const swp_generic& g : e->gen_sets[tasks[i].generic_set]) pairs.insert({g.kind, (int32_t)g.value});
        if (!pairs.empty()) {
            b->has_generic = true;
            std::map<std::pair<uint32_t, int32_t>, uint32_t> row_of;
            for (const auto& pr : pairs) {
                row_of[pr] = (uint32_t)b->rg_kind.size();
                b->rg_kind.push_back(pr.first);
                b->rg_val.push_back(pr.second);
            }
            const uint32_t R = (uint32_t)b->rg_kind.size();
            b->rg_k0.resize(R);
            b->rg_k1.resize(R);
            for (uint32_t r = 0; r < R;) {
                uint32_t q = r;
                while (q < R && b->rg_kind[q] == b->rg_kind[r]) ++q;
                for (uint32_t x = r; x < q; ++x) { b->rg_k0[x] = r; b->rg_k1[x] = q; }
                r = q;
            }
            std::unordered_map<uint32_t, uint32_t> set_local;
            b->tg.assign(T, 0);
            for (uint32_t i = 0; i < T; ++i) {
                const uint32_t gs = tasks[i].generic_set;
                if (!gs) continue;
                auto it = set_local.find(gs);
                if (it == set_local.end()) {
                    it = set_local.emplace(gs, (uint32_t)b->gs_off.size() - 1).first;
                    for (const swp_generic& g : e->gen_sets[gs]) b->gs_row.push_back(row_of[{g.kind, (int32_t)g.value}]);
                    b->gs_off.push_back((uint32_t)b->gs_row.size());
                }
                b->tg[i] = it->second;
            }
        }
    }
    mark("generic sets");
    // explain groups (k_xg_nodes): tasks that share predicate classes, reservations, port set and — with MaxReplicas — service
    b->xg_of.assign(T, 0xFFFFFFFFu);
    b->xg_proto.clear();
    {
        typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, int64_t, int64_t, uint32_t, uint32_t, uint64_t> GKey;
        std::map<GKey, uint32_t> gid;
        std::map<std::tuple<uint32_t, int64_t, int64_t>, uint32_t> pair_id;
        for (uint32_t i : firsts) {
            const RTask& r = b->rt[i];
            if (b->has_generic && b->tg[i]) continue;   // generic reservations: the per-task pass
            if (tasks[i].flags >> SWP_TASK_MOUNTS_SHIFT) continue;   // cluster mounts: the per-task pass (its VolumesFilter row)
            XGroup g{};
            g.flags = ((r.flags & RT_RES) ? XG_RES : 0u) | ((r.flags & RT_PORTS) ? XG_PORTS : 0u) | ((r.flags & RT_MAXREP) ? XG_MAXREP : 0u);
            g.cpu = (g.flags & XG_RES) ? r.cpu : 0;
            g.mem = (g.flags & XG_RES) ? r.mem : 0;
            g.cls_con = r.cls_con;
            g.cls_plat = r.cls_plat;
            g.cls_plug = r.cls_plug;
            g.pset = (g.flags & XG_PORTS) ? r.pset : 0u;
            g.svc = (g.flags & XG_MAXREP) ? r.svc : 0u;
            g.maxrep = (g.flags & XG_MAXREP) ? r.maxrep : 0u;
            const GKey k{g.cls_con, g.cls_plat, g.cls_plug, g.flags, g.cpu, g.mem, g.pset, g.svc, g.maxrep};
            auto it = gid.find(k);
            if (it == gid.end()) {
                it = gid.emplace(k, (uint32_t)b->xg_proto.size()).first;
                g.pad = pair_id.emplace(std::make_tuple(g.flags & XG_RES, g.cpu, g.mem), (uint32_t)pair_id.size()).first->second;
                b->xg_proto.push_back(g);
            }
            b->xg_of[i] = it->second;
        }
        for (uint32_t i = 0; i < T; ++i)
            if (tmpl_of[i] != i) b->xg_of[i] = b->xg_of[tmpl_of[i]];
        b->xg_pairs = (uint32_t)pair_id.size();
    }
    mark("explain groups");
    // runs of identical one-off tasks (same service, filters, reservations; only their list slot differs) are placed by
    // water-filling instead of task by task (csrc/swp_waterfill.hpp). A run must be long enough to pay for its launch, and
    // splitting the batch must not shred the round resolver's work into many launches: runs are used when they make up most
    // of the batch or one of them is long. SWP_WATERFILL=0 switches them off, =1 forces every run of 2 or more (tests).
    b->segs.clear();
    {
        const char* env_wf = getenv("SWP_WATERFILL");
        const int mode = env_wf ? atoi(env_wf) : -1;
        // (the block resolver decides a task in ~130 ns; a water-fill launch costs ~50 µs: a run pays from a few hundred tasks on)
        const uint32_t run_min = mode == 1 ? 2u : 512u;
        auto same = [&](uint32_t i, uint32_t k) { return tmpl_of[i] == tmpl_of[k]; };   // identical descriptors
        std::vector<swp_batch::Seg> segs;
        uint64_t in_runs = 0;
        uint32_t longest = 0, i = 0;
        while (i < T) {
            uint32_t k = i + 1;
            const bool can = !(b->rt[i].flags & (RT_PORTS | RT_UNCOUNTED)) && !b->has_generic && !(tasks[i].flags >> SWP_TASK_MOUNTS_SHIFT);   // (a batch with generic reservations is decided by the block resolver alone; so is a task with cluster mounts)
            while (can && k < T && same(i, k)) ++k;
            const bool run = k - i >= run_min;
            if (run) {
                in_runs += k - i;
                longest = std::max(longest, k - i);
            }
            if (!segs.empty() && !run && !segs.back().run) segs.back().n += k - i;
            else segs.push_back({i, k - i, run});
            i = k;
        }
        const bool use = mode != 0 && (mode == 1 || in_runs * 2 >= T || longest >= 4096);
        if (use && in_runs) b->segs = std::move(segs);
    }
    b->n_svc = (uint32_t)b->svc_global.size();
    b->n_sc = (uint32_t)b->triples.size();
    b->n_con = (uint32_t)con_ids.size();
    b->n_plat = (uint32_t)plat_ids.size();
    b->n_plug = (uint32_t)plug_ids.size();

    b->tmpl = std::move(tmpl_of);
    b->csi_of.clear();
    b->csi_set.clear();
    b->csi_task.clear();
    for (uint32_t i = 0; i < T; ++i)
        if (tasks[i].flags >> SWP_TASK_MOUNTS_SHIFT) {
            if (b->csi_of.empty()) b->csi_of.assign(T, 0xFFFFFFFFu);
            b->csi_of[i] = (uint32_t)b->csi_set.size();
            b->csi_set.push_back(tasks[i].flags >> SWP_TASK_MOUNTS_SHIFT);
            b->csi_task.push_back(i);
        }
    mark("runs");
    sp.next("build_batch: exception lists");
    // per-service exception lists: nodes with svcCount>0 or ≥ maxFailures recent failures
    b->list_off.assign(b->n_svc + 1, 0);
    b->list_cnt0.clear();
    b->list_node0.reserve((size_t)T + 1024);
    b->list_svc0.reserve((size_t)T + 1024);
    b->list_fail0.reserve((size_t)T + 1024);
    std::vector<uint32_t> init_cnt(b->n_svc, 0);
    for (uint32_t s = 0; s < b->n_svc; ++s) {
        uint32_t g = b->svc_global[s];
        b->list_off[s] = (uint32_t)b->list_node0.size();
        auto emit = [&](uint32_t node, uint32_t cnt, uint32_t fails) {
            b->list_node0.push_back(node);
            b->list_svc0.push_back(cnt);
            b->list_fail0.push_back(fails);
        };
        const FlatMap32* sn = e->svc_nodes.find(g);
        auto fn = e->fail_nodes.find(g);
        if (fn == e->fail_nodes.end() || fn->second.empty()) {   // the common case: the service's nodes, already in node order
            // (svc_nodes names present nodes only — swp_node_remove takes a node out of every service's entry — so the node records,
            // a miss each, are not looked at: 90 000 entries per churn round)
            if (sn)
                for (const auto& kv : *sn)
                    if (kv.second > 0) emit(kv.first, kv.second, 0u);
        } else {
            std::map<uint32_t, std::pair<uint32_t, uint32_t>> ent;   // node -> (svc, fail), node-ordered
            if (sn)
                for (const auto& kv : *sn)
                    if (kv.second > 0 && e->nodes[kv.first].present) ent[kv.first].first = kv.second;
            for (uint32_t n : fn->second) {
                if (!e->nodes[n].present) continue;
                auto fit = e->nodes[n].fails.find({g, svc_ver[s]});
                if (fit != e->nodes[n].fails.end() && fit->second >= MAX_FAILURES) ent[n].second = fit->second;
            }
            for (auto& kv : ent) {
                if (kv.second.first == 0 && kv.second.second == 0) continue;
                emit(kv.first, kv.second.first, kv.second.second);
            }
        }
        init_cnt[s] = (uint32_t)b->list_node0.size() - b->list_off[s];
        b->list_cnt0.push_back(init_cnt[s]);
        b->list_node0.insert(b->list_node0.end(), svc_ntasks[s], LIST_EMPTY);   // one free entry per task of the service
        b->list_svc0.insert(b->list_svc0.end(), svc_ntasks[s], 0u);
        b->list_fail0.insert(b->list_fail0.end(), svc_ntasks[s], 0u);
    }
    b->list_off[b->n_svc] = (uint32_t)b->list_node0.size();
    b->n_list0 = b->max_list = 0;
    for (uint32_t s = 0; s < b->n_svc; ++s) {
        b->n_list0 += init_cnt[s];
        b->max_list = std::max(b->max_list, b->list_off[s + 1] - b->list_off[s]);
    }
    for (uint32_t i = 0; i < T; ++i) b->rt[i].slot = b->list_off[b->rt[i].svc] + init_cnt[b->rt[i].svc] + task_rank[i];

    mark("exception lists + slots");
    sp.next("build_batch: ports, class tables");
    // host ports
    b->pset_off.assign(1, 0);
    for (uint32_t gs : pset_ids_global) {
        for (const swp_port& p : e->port_sets[gs]) {
            uint64_t k = port_key(p.protocol, p.port);
            auto it = port_local.find(k);
            if (it == port_local.end()) {
                it = port_local.emplace(k, (uint32_t)b->port_keys.size()).first;
                b->port_keys.push_back(k);
                auto pn = e->port_nodes.find(k);
                if (pn != e->port_nodes.end())
                    for (uint32_t n : pn->second)
                        if (e->nodes[n].present) {
                            b->prow.push_back(it->second);
                            b->pnode.push_back(n);
                        }
            }
            b->pset_ids.push_back(it->second);
        }
        b->pset_off.push_back((uint32_t)b->pset_ids.size());
    }
    b->n_ports = (uint32_t)b->port_keys.size();

    // class tables (row 0 = "filter disabled")
    b->con_off.assign(2, 0);   // row 0 ("filter disabled") is the empty range [0,0)
    for (uint32_t l = 1; l < con_ids.size(); ++l) {
        for (const swp_constraint& c : e->con_sets[con_ids[l]]) {
            DevConstraint dc{};
            dc.kind = c.kind;
            dc.op = c.op;
            dc.value = c.value;
            dc.col = 0;
            if (c.kind == SWP_CK_NODE_LABEL) {
                auto it = e->node_label_col.find(c.key);
                if (it == e->node_label_col.end()) return e->fail(SWP_EINVAL, "label column missing (internal)");
                dc.col = it->second;
            } else if (c.kind == SWP_CK_ENGINE_LABEL) {
                auto it = e->engine_label_col.find(c.key);
                if (it == e->engine_label_col.end()) return e->fail(SWP_EINVAL, "engine label column missing (internal)");
                dc.col = it->second;
            }
            for (int q = 0; q < 4; ++q)
                dc.ip[q] = ((uint32_t)c.ip[4 * q] << 24) | ((uint32_t)c.ip[4 * q + 1] << 16) | ((uint32_t)c.ip[4 * q + 2] << 8) | c.ip[4 * q + 3];
            dc.ip_kind = c.ip_kind;
            dc.prefix_len = c.prefix_len;
            dc.ip_is_v4 = c.ip_is_v4;
            b->cons.push_back(dc);
        }
        b->con_off.push_back((uint32_t)b->cons.size());
    }
    b->plat_off.assign(2, 0);
    for (uint32_t l = 1; l < plat_ids.size(); ++l) {
        for (const swp_platform& p : e->plat_sets[plat_ids[l]]) b->plats.push_back(uint2{p.os, p.arch});
        b->plat_off.push_back((uint32_t)b->plats.size());
    }
    b->plug_off.assign(2, 0);
    for (uint32_t l = 1; l < plug_ids.size(); ++l) {
        const swp_engine::PlugSet& ps = e->plug_sets[plug_ids[l]];
        b->plug_req.push_back(ps.log);
        b->plug_req.insert(b->plug_req.end(), ps.required.begin(), ps.required.end());
        b->plug_off.push_back((uint32_t)b->plug_req.size());
    }
    return SWP_OK;
}

int upload_batch(swp_engine* e, swp_batch* b) {
    const uint32_t Wn = n_words_of(e->n_nodes);
    const uint32_t T = b->T;
    int rc;
    if ((rc = upload(e, b->d_rt, b->rt))) return rc;
    // Every other table goes up through ONE pinned block and ONE copy into ONE device allocation (round 6: two dozen copies from pageable
    // vectors, 10-20 us each whatever their size, were a sixth of a churn round's preparation); the DevBufs become views into the arena.
    struct Up { DevBuf* d; const void* src; size_t bytes, room, off; };
    std::vector<Up> ups;
    auto stage = [&](DevBuf& d, const auto& v) -> int {
        typedef typename std::decay<decltype(v)>::type::value_type V;
        ups.push_back(Up{&d, v.data(), v.size() * sizeof(V), std::max<size_t>(v.size(), 1) * sizeof(V), 0});
        return SWP_OK;
    };
    if ((rc = stage(b->d_tmpl, b->tmpl))) return rc;
    if (!b->csi_set.empty()) {
        if ((rc = stage(b->d_csi_of, b->csi_of))) return rc;
        if ((rc = stage(b->d_csi_set, b->csi_set))) return rc;
        HIPCHECK(e, b->d_vrows.reserve(b->csi_set.size() * (size_t)std::max<uint32_t>(Wn, 1) * 8));
        HIPCHECK(e, b->d_att.reserve(b->csi_set.size() * (size_t)SWP_MAX_MOUNTS * 4));
    }
    if ((rc = stage(b->d_thr, b->thr))) return rc;
    if ((rc = stage(b->d_thr64, b->thr64))) return rc;
    if (b->has_generic) {
        if ((rc = stage(b->d_tg, b->tg))) return rc;
        if ((rc = stage(b->d_gs_off, b->gs_off))) return rc;
        if ((rc = stage(b->d_gs_row, b->gs_row))) return rc;
        if ((rc = stage(b->d_rg_kind, b->rg_kind))) return rc;
        if ((rc = stage(b->d_rg_k0, b->rg_k0))) return rc;
        if ((rc = stage(b->d_rg_k1, b->rg_k1))) return rc;
        if ((rc = stage(b->d_rg_val, b->rg_val))) return rc;
    }
    if ((rc = stage(b->d_list_node0, b->list_node0))) return rc;
    if ((rc = stage(b->d_list_svc0, b->list_svc0))) return rc;
    if ((rc = stage(b->d_list_fail0, b->list_fail0))) return rc;
    if ((rc = stage(b->d_list_off, b->list_off))) return rc;
    if ((rc = stage(b->d_prow, b->prow))) return rc;
    if ((rc = stage(b->d_pnode, b->pnode))) return rc;
    if ((rc = stage(b->d_pset_off, b->pset_off))) return rc;
    if ((rc = stage(b->d_pset_ids, b->pset_ids))) return rc;
    if ((rc = stage(b->d_con_off, b->con_off))) return rc;
    if ((rc = stage(b->d_cons, b->cons))) return rc;
    if ((rc = stage(b->d_plat_off, b->plat_off))) return rc;
    if ((rc = stage(b->d_plats, b->plats))) return rc;
    if ((rc = stage(b->d_plug_off, b->plug_off))) return rc;
    if ((rc = stage(b->d_plug_req, b->plug_req))) return rc;
    if ((rc = stage(b->d_triples, b->triples))) return rc;
    {
        size_t total = 0;
        for (Up& u : ups) { u.off = total; total += (u.room + 255) & ~(size_t)255; }
        if (b->h_up.p) HIPCHECK(e, hipStreamSynchronize(e->stream));   // (a batch uploaded again: the first copy may still be reading the block)
        HIPCHECK(e, b->d_up.reserve(std::max<size_t>(total, 256)));
        HIPCHECK(e, b->h_up.reserve(std::max<size_t>(total, 256)));
        for (Up& u : ups) {
            if (u.bytes) std::memcpy(static_cast<char*>(b->h_up.p) + u.off, u.src, u.bytes);
            u.d->point_into(static_cast<char*>(b->d_up.p) + u.off, (u.room + 255) & ~(size_t)255);
        }
        if (total) HIPCHECK(e, hipMemcpyAsync(b->d_up.p, b->h_up.p, total, hipMemcpyHostToDevice, e->stream));
    }
    size_t L = std::max<size_t>(b->list_node0.size(), 1);
    HIPCHECK(e, b->d_list_node.reserve(L * 4));
    HIPCHECK(e, b->d_list_svc.reserve(L * 4));
    HIPCHECK(e, b->d_list_fail.reserve(L * 4));
    HIPCHECK(e, b->d_out.reserve((size_t)T * 4));
    HIPCHECK(e, b->d_hist.reserve((size_t)T * 8 * 4));
    HIPCHECK(e, b->d_X.reserve((size_t)std::max<uint32_t>(b->n_svc, 1) * Wn * 8));
    HIPCHECK(e, b->d_portmap.reserve((size_t)std::max<uint32_t>(b->n_ports, 1) * Wn * 8));
    HIPCHECK(e, b->d_con.reserve((size_t)b->n_con * Wn * 8));
    HIPCHECK(e, b->d_plat.reserve((size_t)b->n_plat * Wn * 8));
    HIPCHECK(e, b->d_plug.reserve((size_t)b->n_plug * Wn * 8));
    HIPCHECK(e, b->d_sc.reserve((size_t)b->n_sc * Wn * 8));
    HIPCHECK(e, b->d_log_node.reserve((size_t)T * 4));
    HIPCHECK(e, b->d_log_task.reserve((size_t)T * 4));
    HIPCHECK(e, b->d_log_prev.reserve((size_t)T * 4));
    HIPCHECK(e, b->d_ent_ci.reserve((size_t)T * 4));
    HIPCHECK(e, b->d_ent_scpu.reserve((size_t)T * 8));
    HIPCHECK(e, b->d_ent_smem.reserve((size_t)T * 8));
    HIPCHECK(e, b->d_seg_alloc.reserve(64));
    HIPCHECK(e, b->d_last.reserve((size_t)std::max<uint32_t>(e->n_nodes, 1) * 4));
    HIPCHECK(e, b->d_inf_task.reserve((size_t)T * 4));
    HIPCHECK(e, b->d_inf_pos.reserve((size_t)T * 4));
    HIPCHECK(e, b->d_ctl.reserve(sizeof(Ctl)));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return SWP_OK;
}

int run_classes(swp_engine* e, swp_batch* b) {
    const uint32_t N = e->n_nodes, Wn = n_words_of(N);
    NodeView nv = node_view(e);
    dim3 blk(256);
    uint32_t gx = (N + 255) / 256;
    hipLaunchKernelGGL(k_ready, dim3(gx), blk, 0, e->stream, nv, e->d_ready.as<u64>(), e->d_valid.as<u64>());
    if (b->n_con > 1)
        hipLaunchKernelGGL(k_constraint_classes, dim3(gx, b->n_con - 1), blk, 0, e->stream, nv, b->d_con_off.as<uint32_t>(),
                           b->d_cons.as<DevConstraint>(), b->d_con.as<u64>());
    if (b->n_plat > 1)
        hipLaunchKernelGGL(k_platform_classes, dim3(gx, b->n_plat - 1), blk, 0, e->stream, nv, b->d_plat_off.as<uint32_t>(),
                           b->d_plats.as<uint2>(), b->d_plat.as<u64>());
    if (b->n_plug > 1)
        hipLaunchKernelGGL(k_plugin_classes, dim3(gx, b->n_plug - 1), blk, 0, e->stream, nv, b->d_plug_off.as<uint32_t>(),
                           b->d_plug_req.as<uint32_t>(), b->d_plug.as<u64>());
    hipLaunchKernelGGL(k_static_combine, dim3((Wn + 255) / 256, b->n_sc), blk, 0, e->stream, Wn, b->n_sc, b->d_triples.as<uint4>(),
                       e->d_ready.as<u64>(), b->d_con.as<u64>(), b->d_plat.as<u64>(), b->d_plug.as<u64>(), b->d_sc.as<u64>());
    HIPCHECK(e, hipGetLastError());
    return SWP_OK;
}

// explain pass over the batch's unplaceable tasks (first failing filter per node at the task's moment, rebuilt from the commit log)
int run_explain(swp_engine* e, swp_batch* b, uint32_t n_inf) {
    const uint32_t N = e->n_nodes, Wn = n_words_of(N);
    hipStream_t st = e->stream;
    {
        ExplainArgs xa{};
        xa.n_nodes = N;
        xa.n_words = Wn;
        xa.n_inf = n_inf;
        xa.inf_task = b->d_inf_task.as<uint32_t>();
        xa.inf_pos = b->d_inf_pos.as<uint32_t>();
        xa.rt = b->d_rt.as<RTask>();
        xa.valid = e->d_valid.as<u64>();
        xa.ready = e->d_ready.as<u64>();
        xa.con = b->d_con.as<u64>();
        xa.plat = b->d_plat.as<u64>();
        xa.plug = b->d_plug.as<u64>();
        xa.cpu = e->d_cpu.as<long long>();
        xa.mem = e->d_mem.as<long long>();
        xa.portmap = b->d_portmap.as<u64>();
        xa.pset_off = b->d_pset_off.as<uint32_t>();
        xa.pset_ids = b->d_pset_ids.as<uint32_t>();
        xa.list_node = b->d_list_node.as<uint32_t>();
        xa.list_svc = b->d_list_svc.as<uint32_t>();
        xa.list_off = b->d_list_off.as<uint32_t>();
        xa.log_task = b->d_log_task.as<uint32_t>();
        xa.log_prev = b->d_log_prev.as<int32_t>();
        xa.last = b->d_last.as<int32_t>();
        if (b->has_generic) {
            xa.n_rg = (u32)b->rg_kind.size();
            xa.gstride = e->ncap;
            xa.gcnt = e->d_gcnt.as<int32_t>();
            xa.tg = b->d_tg.as<uint32_t>();
            xa.gs_off = b->d_gs_off.as<uint32_t>();
            xa.gs_row = b->d_gs_row.as<uint32_t>();
            xa.rg_kind = b->d_rg_kind.as<uint32_t>();
            xa.rg_val = b->d_rg_val.as<int32_t>();
        }
        if (!b->csi_set.empty()) {
            xa.csi_of = b->d_csi_of.as<uint32_t>();
            xa.vrows = b->d_vrows.as<u64>();
        }
        xa.hist = b->d_hist.as<uint32_t>();
        // per-node commit segments (sorted, suffix sums) for the residual-at-the-moment lookups
        HIPCHECK(e, b->d_seg_off.reserve((size_t)N * 4));
        HIPCHECK(e, b->d_seg_len.reserve((size_t)N * 4));
        HIPCHECK(e, hipMemsetAsync(b->d_seg_alloc.p, 0, 4, st));
        SegArgs sg{};
        sg.n_nodes = N;
        sg.rt = b->d_rt.as<RTask>();
        sg.log_task = b->d_log_task.as<uint32_t>();
        sg.log_prev = b->d_log_prev.as<int32_t>();
        sg.last = b->d_last.as<int32_t>();
        sg.alloc = b->d_seg_alloc.as<uint32_t>();
        sg.seg_off = b->d_seg_off.as<uint32_t>();
        sg.seg_len = b->d_seg_len.as<uint32_t>();
        sg.ent_ci = b->d_ent_ci.as<uint32_t>();
        sg.ent_scpu = b->d_ent_scpu.as<long long>();
        sg.ent_smem = b->d_ent_smem.as<long long>();
        hipLaunchKernelGGL(k_chain_segments, dim3((N + 255) / 256), dim3(256), 0, st, sg);
        xa.seg_off = sg.seg_off;
        xa.seg_len = sg.seg_len;
        xa.ent_ci = sg.ent_ci;
        xa.ent_scpu = sg.ent_scpu;
        xa.ent_smem = sg.ent_smem;
        // Which unplaceable tasks are explained BY GROUP (k_xg_nodes / k_xg_maxrep / k_xg_write) and which need the per-task pass was
        // fixed when the batch was built (xg_of). The list of unplaceable tasks comes back once (4 + 4 bytes each), is dealt into the
        // groups by a counting sort that keeps the list's order (moments ascend), and goes up again as ONE packed buffer together
        // with the groups, their order by reservation pair and the chunks of that order (pinned staging both ways).
        b->x_ninf = n_inf;
        HIPCHECK(e, b->hx_in.reserve((size_t)n_inf * 8));
        uint32_t* h_task = static_cast<uint32_t*>(b->hx_in.p);
        uint32_t* h_pos = h_task + n_inf;
        HIPCHECK(e, hipMemcpyAsync(h_task, b->d_inf_task.p, (size_t)n_inf * 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(e, hipMemcpyAsync(h_pos, b->d_inf_pos.p, (size_t)n_inf * 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(e, hipStreamSynchronize(st));
        const char* env_xg = getenv("SWP_EXPLAIN_GROUPS");
        const bool use_groups = !(env_xg && atoi(env_xg) == 0);
        const uint32_t NG = (uint32_t)b->xg_proto.size(), NONE = 0xFFFFFFFFu;
        std::vector<uint32_t>&fill = b->hx_fill, &pair_cnt = b->hx_pair_cnt, &compact = b->hx_compact;
        fill.assign(NG + 1, 0);   // group -> its entries, then -> the next free one
        uint32_t n_fast = 0, n_slow = 0;
        for (uint32_t q = 0; q < n_inf; ++q) {
            const uint32_t g = use_groups ? b->xg_of[h_task[q]] : NONE;
            if (g == NONE) ++n_slow;
            else {
                ++fill[g];
                ++n_fast;
            }
        }
        if (getenv("SWP_DEBUG_EXPLAIN")) fprintf(stderr, "explain: %u unplaceable, %u by group (%u groups known), %u per task\n", n_inf, n_fast, NG, n_slow);
        uint32_t ng = 0, nm = 0;
        pair_cnt.assign(b->xg_pairs + 1, 0);
        for (uint32_t g = 0; g < NG; ++g)
            if (fill[g]) {
                ++ng;
                nm += (b->xg_proto[g].flags & XG_MAXREP) ? 1u : 0u;
                ++pair_cnt[b->xg_proto[g].pad];
            }
        // chunks: groups of one pair, few enough per chunk that the launch still fills the device
        const uint32_t bx = (N + 256 * XG_NPT - 1) / (256 * XG_NPT);
        const uint32_t csize = std::max<uint32_t>(1, std::min<uint32_t>(32, (uint32_t)(((uint64_t)ng * bx) / 2048)));
        uint32_t nc = 0, nw = 0;
        for (uint32_t p = 0; p < b->xg_pairs; ++p) nc += (pair_cnt[p] + csize - 1) / csize;
        for (uint32_t g = 0; g < NG; ++g) nw += (fill[g] + XG_WCH - 1) / XG_WCH;   // k_xg_write's chunks
        auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
        const size_t o_g = 0, o_pos = al(o_g + (size_t)ng * sizeof(XGroup)), o_task = al(o_pos + (size_t)n_fast * 4), o_mr = al(o_task + (size_t)n_fast * 4),
                     o_ord = al(o_mr + (size_t)nm * 4), o_ch = al(o_ord + (size_t)ng * 4), o_st = al(o_ch + (size_t)nc * 8), o_sp = al(o_st + (size_t)n_slow * 4),
                     o_wch = al(o_sp + (size_t)n_slow * 4), total = al(o_wch + (size_t)nw * 8);
        HIPCHECK(e, b->hx_pack.reserve(total));
        HIPCHECK(e, b->d_xpack.reserve(total));
        char* hp = static_cast<char*>(b->hx_pack.p);
        char* dp = static_cast<char*>(b->d_xpack.p);
        XGroup* groups = reinterpret_cast<XGroup*>(hp + o_g);
        uint32_t *xpos = reinterpret_cast<uint32_t*>(hp + o_pos), *xtask = reinterpret_cast<uint32_t*>(hp + o_task), *mr = reinterpret_cast<uint32_t*>(hp + o_mr),
                 *gorder = reinterpret_cast<uint32_t*>(hp + o_ord), *slow_task = reinterpret_cast<uint32_t*>(hp + o_st), *slow_pos = reinterpret_cast<uint32_t*>(hp + o_sp);
        uint2* chunks = reinterpret_cast<uint2*>(hp + o_ch);
        uint2* wchunks = reinterpret_cast<uint2*>(hp + o_wch);
        if (n_fast) {
            compact.assign(NG, NONE);
            uint32_t off = 0, gi = 0, mi = 0, wi = 0;
            for (uint32_t g = 0; g < NG; ++g) {
                const uint32_t cnt = fill[g];
                if (!cnt) continue;
                XGroup x = b->xg_proto[g];
                x.off = off;
                x.cnt = cnt;
                x.doff = off + gi;   // cnt + 1 difference slots per group
                if (x.flags & XG_MAXREP) mr[mi++] = gi;
                groups[gi] = x;
                for (uint32_t s0 = 0; s0 < cnt; s0 += XG_WCH) wchunks[wi++] = make_uint2(gi, s0);
                compact[g] = gi++;
                fill[g] = off;
                off += cnt;
            }
            // the groups by pair (counting sort), then the chunks of that order
            uint32_t run = 0;
            for (uint32_t p = 0; p <= b->xg_pairs; ++p) {
                const uint32_t c = pair_cnt[p];
                pair_cnt[p] = run;
                run += c;
            }
            for (uint32_t g = 0; g < NG; ++g)
                if (compact[g] != NONE) gorder[pair_cnt[b->xg_proto[g].pad]++] = compact[g];
            uint32_t ci = 0;
            for (uint32_t q = 0; q < ng;) {
                uint32_t r = q + 1;
                const uint32_t pr = groups[gorder[q]].pad;
                while (r < ng && r - q < csize && groups[gorder[r]].pad == pr) ++r;
                chunks[ci++] = make_uint2(q, r - q);
                q = r;
            }
            nc = ci;
        }
        uint32_t si = 0;
        for (uint32_t q = 0; q < n_inf; ++q) {
            const uint32_t g = use_groups ? b->xg_of[h_task[q]] : NONE;
            if (g == NONE) {
                slow_task[si] = h_task[q];
                slow_pos[si++] = h_pos[q];
            } else {
                const uint32_t at = fill[g]++;
                xtask[at] = h_task[q];
                xpos[at] = h_pos[q];
            }
        }
        HIPCHECK(e, hipMemcpyAsync(dp, hp, total, hipMemcpyHostToDevice, st));
        if (n_fast) {
            const uint32_t dstride = n_fast + ng;
            HIPCHECK(e, b->d_xdiff.reserve((size_t)XG_PLANES * dstride * 4));
            HIPCHECK(e, b->d_xnr.reserve(256));
            HIPCHECK(e, hipMemsetAsync(b->d_xdiff.p, 0, (size_t)XG_PLANES * dstride * 4, st));
            HIPCHECK(e, hipMemsetAsync(b->d_xnr.p, 0, 4, st));
            XGArgs ga{};
            ga.n_nodes = N;
            ga.n_words = Wn;
            ga.n_groups = ng;
            ga.dstride = dstride;
            ga.g = reinterpret_cast<const XGroup*>(dp + o_g);
            ga.gorder = reinterpret_cast<const uint32_t*>(dp + o_ord);
            ga.chunks = reinterpret_cast<const uint2*>(dp + o_ch);
            ga.xpos = reinterpret_cast<const uint32_t*>(dp + o_pos);
            ga.xtask = reinterpret_cast<const uint32_t*>(dp + o_task);
            ga.rt = xa.rt;
            ga.valid = xa.valid;
            ga.ready = xa.ready;
            ga.con = xa.con;
            ga.plat = xa.plat;
            ga.plug = xa.plug;
            ga.cpu = xa.cpu;
            ga.mem = xa.mem;
            ga.portmap = xa.portmap;
            ga.pset_off = xa.pset_off;
            ga.pset_ids = xa.pset_ids;
            ga.list_node = xa.list_node;
            ga.list_svc = xa.list_svc;
            ga.list_off = xa.list_off;
            ga.log_task = xa.log_task;
            ga.seg_off = xa.seg_off;
            ga.seg_len = xa.seg_len;
            ga.ent_ci = xa.ent_ci;
            ga.ent_scpu = xa.ent_scpu;
            ga.ent_smem = xa.ent_smem;
            ga.mr = reinterpret_cast<const uint32_t*>(dp + o_mr);
            ga.wchunks = reinterpret_cast<const uint2*>(dp + o_wch);
            ga.diff = b->d_xdiff.as<int32_t>();
            ga.notready = b->d_xnr.as<uint32_t>();
            ga.hist = xa.hist;
            const uint32_t GY = 32768u;   // grid.y <= 65535
            for (uint32_t c0 = 0; c0 < nc; c0 += GY) {
                XGArgs gc = ga;
                gc.chunks += c0;
                // (the ReadyFilter counter is taken by the blocks of chunk 0 of the FIRST launch only)
                if (c0) gc.notready = b->d_xnr.as<uint32_t>() + 1;
                hipLaunchKernelGGL(k_xg_nodes, dim3(bx, std::min<uint32_t>(GY, nc - c0)), dim3(256), 0, st, gc);
            }
            for (uint32_t m0 = 0; m0 < nm; m0 += GY) {
                XGArgs gc = ga;
                gc.mr += m0;
                hipLaunchKernelGGL(k_xg_maxrep, dim3(8, std::min<uint32_t>(GY, nm - m0)), dim3(256), 0, st, gc);
            }
            hipLaunchKernelGGL(k_xg_write, dim3(nw), dim3(256), 0, st, ga);
        }
        if (n_slow) {
            xa.inf_task = reinterpret_cast<const uint32_t*>(dp + o_st);
            xa.inf_pos = reinterpret_cast<const uint32_t*>(dp + o_sp);
            uint32_t done = 0;
            while (done < n_slow) {   // grid.y ≤ 65535 blocks of EX_TCH tasks
                uint32_t chunk = std::min<uint32_t>(n_slow - done, 32768u * EX_TCH);
                ExplainArgs xc = xa;
                xc.inf_task += done;
                xc.inf_pos += done;
                xc.n_inf = chunk;
                hipLaunchKernelGGL(k_explain, dim3((N + 255) / 256, (chunk + EX_TCH - 1) / EX_TCH), dim3(256), 0, st, xc);
                done += chunk;
            }
        }
        HIPCHECK(e, hipGetLastError());
    }
    return SWP_OK;
}

// per-batch device state back to pristine (exception lists and bitmaps, host ports, commit log, results) + the predicate
// class bitmaps: what a run does before its first resolver launch
int batch_begin(swp_engine* e, swp_batch* b) {
    const uint32_t N = e->n_nodes, Wn = n_words_of(N), T = b->T;
    b->x_ninf = 0;
    hipStream_t st = e->stream;
    size_t L = b->list_node0.size();
    if (L) {
        HIPCHECK(e, hipMemcpyAsync(b->d_list_node.p, b->d_list_node0.p, L * 4, hipMemcpyDeviceToDevice, st));
        HIPCHECK(e, hipMemcpyAsync(b->d_list_svc.p, b->d_list_svc0.p, L * 4, hipMemcpyDeviceToDevice, st));
        HIPCHECK(e, hipMemcpyAsync(b->d_list_fail.p, b->d_list_fail0.p, L * 4, hipMemcpyDeviceToDevice, st));
    }
    HIPCHECK(e, hipMemsetAsync(b->d_X.p, 0, (size_t)std::max<uint32_t>(b->n_svc, 1) * Wn * 8, st));
    HIPCHECK(e, hipMemsetAsync(b->d_portmap.p, 0, (size_t)std::max<uint32_t>(b->n_ports, 1) * Wn * 8, st));
    HIPCHECK(e, hipMemsetAsync(b->d_last.p, 0xFF, (size_t)N * 4, st));
    HIPCHECK(e, hipMemsetAsync(b->d_ctl.p, 0, sizeof(Ctl), st));
    HIPCHECK(e, hipMemsetAsync(b->d_hist.p, 0, (size_t)T * 8 * 4, st));
    HIPCHECK(e, hipMemsetAsync(b->d_out.p, 0xFF, (size_t)T * 4, st));   // -1 = no suitable node unless a commit says otherwise
    if (!b->csi_set.empty()) {   // tasks with cluster mounts: no attachment yet, no node passes the VolumesFilter until a round says so
        if (int rcv = flush_volumes(e)) return rcv;   // (a usage set since swp_batch_prepare)
        HIPCHECK(e, hipMemsetAsync(b->d_att.p, 0xFF, b->csi_set.size() * (size_t)SWP_MAX_MOUNTS * 4, st));
        HIPCHECK(e, hipMemsetAsync(b->d_vrows.p, 0, b->csi_set.size() * (size_t)Wn * 8, st));
    }
    if (b->n_list0 && b->n_svc)   // X rows from the lists as they start (d_list_node0: the batch's own copy, whatever the rounds did to d_list_node)
        hipLaunchKernelGGL(k_scatter_lists, dim3(std::min<uint32_t>(64u, (b->max_list + 255u) / 256u), b->n_svc), dim3(256), 0, st, b->d_list_off.as<uint32_t>(),
                           b->d_list_node0.as<uint32_t>(), Wn, b->d_X.as<u64>());
    if (!b->prow.empty())
        hipLaunchKernelGGL(k_scatter_bits, dim3(((uint32_t)b->prow.size() + 255) / 256), dim3(256), 0, st, (uint32_t)b->prow.size(),
                           b->d_prow.as<uint32_t>(), b->d_pnode.as<uint32_t>(), Wn, b->d_portmap.as<u64>());
    return run_classes(e, b);
}

// The block resolver's argument record for (engine, batch): reserves the buffers it names. task_rows: ResourceFilter rows per task of
// the block instead of per demand class (swp_resolve6.hpp k_r6_taskrows).
int r6_args_for(swp_engine* e, swp_batch* b, uint32_t r6_block, bool r6_task_rows, uint32_t dbg_bits, R6Args* out) {
    const uint32_t N = e->n_nodes, Wn = n_words_of(N);
    const uint32_t r6_nrr = r6_task_rows ? 0u : b->n_dc + b->n_dm;
    HIPCHECK(e, b->d_planes6.reserve((size_t)R6_NP * Wn * 8));
    HIPCHECK(e, b->d_rr6.reserve((size_t)std::max<uint32_t>(r6_nrr, 1) * Wn * 8));
    if (r6_task_rows) HIPCHECK(e, b->d_trows.reserve((size_t)r6_block * Wn * 8));
    HIPCHECK(e, b->d_blk6.reserve(sizeof(Blk6)));
    HIPCHECK(e, b->d_prop.reserve(r7_send_bytes(r6_block)));   // (the shard drivers send two trailer slots behind the block: swp_resolve7.hpp)
    if (Wn <= R6_COMPACT_MAX_WORDS) {   // the compact index of a round (k_r6_compact); run_blocks switches it on
        HIPCHECK(e, b->d_cmask.reserve((size_t)Wn * 8));
        HIPCHECK(e, b->d_crank.reserve((size_t)Wn * 4));
        HIPCHECK(e, b->d_cidx.reserve((size_t)r6_compact_cap(Wn) * 4));
    }
    R6Args ra{};
    ra.n_nodes = N;
    ra.n_words = Wn;
    ra.xs = Wn;
    ra.block = r6_block;
    ra.n_dc = r6_task_rows ? 0u : b->n_dc;
    ra.n_dm = r6_task_rows ? 0u : b->n_dm;
    ra.task_rows = r6_task_rows ? 1u : 0u;
    ra.trows = r6_task_rows ? b->d_trows.as<u64>() : nullptr;
    ra.dbg = dbg_bits;
    ra.valid = e->d_valid.as<u64>();
    ra.sc = b->d_sc.as<u64>();
    ra.X = b->d_X.as<u64>();
    ra.rt = b->d_rt.as<RTask>();
    ra.cpu = e->d_cpu.as<long long>();
    ra.mem = e->d_mem.as<long long>();
    ra.total = e->d_total.as<uint32_t>();
    ra.list_node = b->d_list_node.as<uint32_t>();
    ra.list_svc = b->d_list_svc.as<uint32_t>();
    ra.list_fail = b->d_list_fail.as<uint32_t>();
    ra.list_off = b->d_list_off.as<uint32_t>();
    ra.portmap = b->d_portmap.as<u64>();
    ra.pset_off = b->d_pset_off.as<uint32_t>();
    ra.pset_ids = b->d_pset_ids.as<uint32_t>();
    ra.out_node = b->d_out.as<int32_t>();
    ra.log_node = b->d_log_node.as<uint32_t>();
    ra.log_task = b->d_log_task.as<uint32_t>();
    ra.log_prev = b->d_log_prev.as<int32_t>();
    ra.last = b->d_last.as<int32_t>();
    ra.inf_task = b->d_inf_task.as<uint32_t>();
    ra.inf_pos = b->d_inf_pos.as<uint32_t>();
    ra.ctl = b->d_ctl.as<Ctl>();
    ra.planes = b->d_planes6.as<u64>();
    ra.rr = b->d_rr6.as<u64>();
    ra.thr = b->d_thr64.as<long long>();
    ra.blk = b->d_blk6.as<Blk6>();
    ra.prop = b->d_prop.as<R6Prop>();
    ra.cbase = e->d_ready.as<u64>();
    ra.cmask = b->d_cmask.as<u64>();
    ra.crank = b->d_crank.as<uint32_t>();
    ra.cidx = b->d_cidx.as<uint32_t>();
    if (!b->csi_set.empty()) {
        ra.csi_of = b->d_csi_of.as<uint32_t>();
        ra.csi_set = b->d_csi_set.as<uint32_t>();
        ra.vrows = b->d_vrows.as<u64>();
        ra.att = b->d_att.as<uint32_t>();
        ra.vol = vol_view(e);
    }
    {   // SWP_R6_TWINS=0: every list starts at its level's first candidate (A/B runs)
        const char* env_tw = getenv("SWP_R6_TWINS");
        ra.tmpl = (env_tw && atoi(env_tw) == 0) ? nullptr : b->d_tmpl.as<uint32_t>();
    }
    if (b->has_generic) {
        HIPCHECK(e, b->d_rg.reserve((size_t)b->rg_kind.size() * Wn * 8));
        ra.n_rg = (u32)b->rg_kind.size();
        ra.gstride = e->ncap;
        ra.gcnt = e->d_gcnt.as<int32_t>();
        ra.rg = b->d_rg.as<u64>();
        ra.tg = b->d_tg.as<uint32_t>();
        ra.gs_off = b->d_gs_off.as<uint32_t>();
        ra.gs_row = b->d_gs_row.as<uint32_t>();
        ra.rg_kind = b->d_rg_kind.as<uint32_t>();
        ra.rg_val = b->d_rg_val.as<int32_t>();
        ra.rg_k0 = b->d_rg_k0.as<uint32_t>();
        ra.rg_k1 = b->d_rg_k1.as<uint32_t>();
    }
    *out = ra;
    return SWP_OK;
}

int batch_run_impl(swp_engine* e, swp_batch* b);
// Every failure exit of the device pass leaves the device node rows (cpu / mem / total) possibly half-updated — the resolvers commit
// while they run — so the untouched host mirror is uploaded again before the next device call (as swp_schedule_groups does).
int batch_run(swp_engine* e, swp_batch* b) {
    const int rc = batch_run_impl(e, b);
    if (rc != SWP_OK) e->dev_dynamic_dirty = true;
    return rc;
}

int batch_run_impl(swp_engine* e, swp_batch* b) {
    if (e->n_nodes != b->n_nodes_prepared)
        return e->fail(SWP_EINVAL, "the nodeSet grew from %u to %u node slots since swp_batch_prepare: prepare the batch again", b->n_nodes_prepared, e->n_nodes);
    const uint32_t N = e->n_nodes, Wn = n_words_of(N), T = b->T;
    if (N == 0 || T == 0) { b->ran = true; return SWP_OK; }
    const bool prof = (e->cfg.flags & SWP_CFG_PROFILE) != 0;
    hipStream_t st = e->stream;
    if (prof) HIPCHECK(e, hipEventRecord(e->ev[0], st));
    e->ev_scan_used = 0;
    e->scan_tasks_batch = e->scan_stretches_batch = 0;

    int rc = batch_begin(e, b);
    if (rc) return rc;
    if (prof) HIPCHECK(e, hipEventRecord(e->ev[1], st));

    const size_t lds_budget = 160 * 1024 - 512;
    // Two resolver families. k_resolve5, the ROUND resolver: everything it decides from lives in one workgroup's LDS (<= ~12 000
    // nodes, the batch's distinct reservations as at most r5_max_rows() demand-class rows, residuals as 32-bit counts of the
    // batch's resource units) — one launch per batch. k_resolve6, the BLOCK resolver: bitmap rows in global memory, lists built by
    // the whole chip — everything else. Test / debugging knobs, read once per batch: SWP_RESOLVER=5|6 forces a family
    // (tests/test_engine_resolvers.py), SWP_DBG bit 16 switches the in-kernel section timers on.
    const char* env_res = getenv("SWP_RESOLVER");
    const char* env_dbg = getenv("SWP_DBG");
    const uint32_t dbg_bits = env_dbg ? (uint32_t)atoi(env_dbg) : 0u;
    // (without the knob: the block resolver. Round 4 measured it ahead of the round resolver at EVERY batch size on 10 000 nodes — 0.21 vs
    // 0.54 ms for 256 tasks, 1.1 vs 1.5 ms for 8 192, 12.8 vs 14.9 ms for 100 000 — and twice as fast on the churn rounds, whose ~9 000
    // re-placements all aim at the few emptied nodes (profiles/r04_*). The round resolver stays as the family that needs no bitmaps:
    // SWP_RESOLVER=5, and the fall-back below.)
    int variant = env_res ? atoi(env_res) : 6;
    if (variant != 5 && variant != 6) return e->fail(SWP_EINVAL, "SWP_RESOLVER=%d: the resolver families are 5 (round) and 6 (block)", variant);
    if (!b->csi_set.empty()) variant = 6;   // tasks with cluster mounts: the block resolver knows the volumes
    const size_t r5_lds = r5_lds_size(N, Wn, b->exact_ok ? b->n_dc + b->n_dm : 0u);
    bool r5_ok = b->exact_ok && b->units_ok && r5_supports(Wn) && r5_lds <= lds_budget && !b->has_generic && b->csi_set.empty();
    for (uint32_t n = 0; r5_ok && n < N; ++n) {
        const HostNode& h = e->nodes[n];
        if (!h.present) continue;
        const int64_t qc = h.row.cpu / b->unit_cpu, qm = h.row.mem / b->unit_mem;
        if (qc >= R5_QLIM_HOST || qc <= -R5_QLIM_HOST || qm >= R5_QLIM_HOST || qm <= -R5_QLIM_HOST) r5_ok = false;
    }
    // k_resolve6 (block resolver): lists built by the whole chip from bitmap rows in global memory, matched by one wave. For node
    // sets beyond k_resolve5's LDS; needs the demand classes and two candidate buffers of the propose kernel in LDS (≈ 650k nodes).
    // SWP_RESOLVER=6 forces it at any size; SWP_R6_BLOCK sets the tasks per round.
    const char* env_blk = getenv("SWP_R6_BLOCK");
    // (without the knob: as many 64-task groups as the commit kernel's LDS holds next to the TK row, at most R6_BLOCK_DEFAULT_CAP tasks)
    uint32_t r6_block = std::min<uint32_t>(r6_block_max(), std::max<uint32_t>(1u, env_blk ? (uint32_t)atoi(env_blk) : R6_BLOCK_DEFAULT_CAP));
    // demand-class rows patched by every commit while the batch has few distinct reservations; rows per task of the block, rebuilt
    // every round from the exact residuals, when it has many (a commit would cross too many thresholds) — no limit then
    const char* env_tr = getenv("SWP_R6_TASKROWS");
    const bool r6_task_rows = env_tr ? atoi(env_tr) != 0 : (!b->classes_ok || b->n_dc + b->n_dm > 128);
    const uint32_t r6_nrr = r6_task_rows ? 0u : b->n_dc + b->n_dm;
    // (the commit kernel stages the block's lists in LDS next to the TK row: a very large node set gets a smaller block)
    while (r6_block > 64 && r6_commit_lds_size(Wn, r6_block, r6_nrr) > lds_budget) r6_block = (r6_block - 1u) / 64u * 64u;
    const bool r6_ok = r6_propose_lds_size(Wn) <= lds_budget && r6_commit_lds_size(Wn, r6_block, r6_nrr) <= lds_budget && Wn <= 32768u;   // (half-word indices of 16 bits in the commit kernel's LDS)
    if (variant == 5 && !r5_ok) variant = 6;
    if (variant == 6 && !r6_ok) {
        if (r5_ok) variant = 5;   // (a node set whose rows leave the block resolver no LDS but fits the round resolver's: cannot happen with today's limits)
        else return e->fail(SWP_ERANGE, "node count %u exceeds the block resolver's LDS (shard the node set)", N);
    }
    if (variant == 5) {
        HIPCHECK(e, b->d_qres.reserve((size_t)N * 8));
        hipLaunchKernelGGL(k_units, dim3((N + 255) / 256), dim3(256), 0, st, N, e->d_cpu.as<long long>(), e->d_mem.as<long long>(), (long long)b->unit_cpu,
                           (long long)b->unit_mem, b->d_qres.as<int32_t>());
    }
    uint32_t wi = 0;   // resolver stretches launched so far (profiling slots)
    uint64_t r6_rounds = 0;
    // k_resolve6 over the stretch [start, end): build the bitmaps from the node rows as they are, then rounds of propose + commit.
    // The device advances on its own (the position lives in the control block); the host only learns every so many rounds how far it is.
    auto run_blocks = [&](uint32_t start, uint32_t end) -> int {
        if (prof)
            while (e->ev_pool.size() < (size_t)4 * (wi + 1)) {
                hipEvent_t x;
                HIPCHECK(e, hipEventCreate(&x));
                e->ev_pool.push_back(x);
            }
        R6Args ra{};
        if (int rc6 = r6_args_for(e, b, r6_block, r6_task_rows, dbg_bits, &ra)) return rc6;
        if (prof) {
            HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 0], st));
            HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 1], st));
            HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 2], st));
        }
        Blk6 hb{};
        hb.pos = start;
        hb.end = end;
        HIPCHECK(e, hipMemcpyAsync(b->d_blk6.p, &hb, sizeof hb, hipMemcpyHostToDevice, st));
        hipError_t r = launch_r6_build(ra, st);
        if (r != hipSuccess) return e->fail(SWP_EHIP, "k_r6 build launch: %s", hipGetErrorString(r));
        uint32_t pos = start, chunk = std::min<uint32_t>(16u, (end - start + 255u) / 256u + 1u);   // a short stretch does not pay for empty rounds
        uint32_t rounds_seen = 0, scan_len = 2048, scanned = 0, skipped_seen = 0;
        bool after_scan = false;   // the rounds since the last scan stretch: four of them say whether the tasks still have no plain candidates
        // The compact index (swp_resolve6.hpp, R6Args.compact): one more small launch per round, worth it when the level the tasks aim at is a
        // sparse set of nodes (re-placements after a drain): the matcher then stops at an emptied half-word every few tasks and the rounds are
        // cut by exhausted lists after a fraction of their block. Switched on by that symptom (a stop per <= 12 decided tasks; ordinary
        // batches have one per 30-40), off again when the kernel found no such level in a whole chunk; the next batch starts the way this
        // one ended if the index was in use then.
        // SWP_R6_COMPACT=0 never, 1 from the first round on (tests, A/B runs).
        const char* env_cpt = getenv("SWP_R6_COMPACT");
        const bool cpt_ok = Wn <= R6_COMPACT_MAX_WORDS && r6_commit_lds_size(Wn, r6_block, r6_nrr, true) <= lds_budget && b->csi_set.empty() && !(env_cpt && atoi(env_cpt) == 0);   // (no smaller blocks for it)
        bool cpt = cpt_ok && ((env_cpt && atoi(env_cpt) != 0) || e->r6_compact_hint);
        // (the index of the NEXT round built at the end of k_r6_commit_c instead of by a launch of its own: SWP_R6_COMPACT_FUSED=0 for A/B runs)
        const char* env_cf = getenv("SWP_R6_COMPACT_FUSED");
        const bool cpt_fused = !(env_cf && atoi(env_cf) == 0);
        bool cpt_ever = false;          // some chunk of this batch had rounds with an index
        // (round 6: a churn round's batch ENDS on tasks that aim at a level every node has — no index — and the hint used to say "the last
        // chunk had one": every batch then began with sixteen rounds without it. Starting with the previous batch's small BLOCK as well was
        // measured and lost: device 3.39 -> 3.62 ms a round, same box, three runs each.)
        uint32_t exh_seen = 0, crounds_seen = 0, stops_seen = 0;
        const char* env_scan = getenv("SWP_SCAN");   // 0: never hand a stretch to the scan resolver (tests, A/B runs)
        const bool scan_ok = N <= scan_max_nodes() && !(env_scan && atoi(env_scan) == 0) && b->csi_set.empty();   // (the scan resolver knows no volumes)
        // The batched instance of the scan (k_scanb) decides ~4 tasks a barrier and answers an unplaceable task's twins without a look: where
        // it can run, a stretch goes to it after four poor rounds and two probing rounds follow it; the one-task-a-barrier instance
        // (~1 us a task) has to be worth more: eight poor rounds, four probes
        const bool scan_fast = scan_ok && ra.n_rg == 0 && ra.csi_of == nullptr && scan_batched_fits(N, b->n_svc, b->n_sc);
        while (pos < end) {
            ra.compact = cpt ? (cpt_fused ? 2u : 1u) : 0u;
            r = launch_r6_rounds(ra, chunk, st, e->device);
            if (r != hipSuccess) return e->fail(SWP_EHIP, "k_r6 round launch: %s", hipGetErrorString(r));
            HIPCHECK(e, hipMemcpyAsync(&hb, b->d_blk6.p, sizeof hb, hipMemcpyDeviceToHost, st));
            HIPCHECK(e, hipStreamSynchronize(st));
            if (hb.error == ERR_GROUP_RANGE) return e->fail(SWP_ERANGE, "a node's key left its range (>= 256 recent failures or >= 2^24 tasks of one service on a node)");
            if (hb.error != ERR_NONE) return e->fail(SWP_ERANGE, "per-node task-count spread exceeds the %d level planes of the block resolver", R6_NP);
            if (hb.pos <= pos) return e->fail(SWP_EHIP, "block resolver made no progress at task %u", pos);   // a round decides its first task at least
            // The rounds of this chunk decided only a handful of tasks each: the tasks here have no plain candidates (their services run on
            // every node that could take them) and a round ends behind the first of them. The scan resolver decides such a stretch one task
            // after the other at ~1 us each (swp_scan.hpp); afterwards the bitmaps are rebuilt and the rounds are tried again.
            const uint32_t used = hb.rounds - rounds_seen;   // rounds that found work
            const double recent = (double)(hb.pos - pos) / (double)std::max<uint32_t>(used, 1);
            rounds_seen = hb.rounds;
            pos = hb.pos;
            if (dbg_bits & 32) fprintf(stderr, "[swp] chunk: %u rounds, %.1f tasks each, at %u of %u | block %u compact %u csize %u clevel %u base %u maxrel %u\n", used, recent, pos, end, ra.block, ra.compact, hb.csize, hb.clevel, hb.base, hb.maxrel);
            {
                const uint32_t exh = hb.cut_exhausted - exh_seen, cr = hb.crounds - crounds_seen, stops = hb.reseats - stops_seen;
                exh_seen = hb.cut_exhausted;
                crounds_seen = hb.crounds;
                stops_seen = hb.reseats;
                if (cpt && cr) cpt_ever = true;
                if (cpt_ok && !(env_cpt && atoi(env_cpt) != 0)) {
                    if (!cpt && used >= 4 && 2 * exh >= used && recent < 0.4 * ra.block && (double)stops * 12.0 >= recent * used) cpt = true;
                    else if (cpt && used >= 4 && cr == 0) cpt = false;
                }
            }
            if (scan_ok && used >= (scan_fast ? (after_scan ? 2u : 4u) : (after_scan ? 4u : 8u)) && recent < 8.0 && end - pos >= 64) {
              scan_again:
                const uint32_t upto = std::min<uint64_t>(end, (uint64_t)pos + scan_len);
                ScanArgs sa{};
                sa.a = ra;
                sa.j0 = pos;
                sa.j1 = upto;
                sa.n_svc = b->n_svc;
                HIPCHECK(e, b->d_hmat.reserve((size_t)b->n_svc * N * 4));
                HIPCHECK(e, b->d_emat.reserve((size_t)b->n_svc * N * 4));
                sa.hmat = b->d_hmat.as<uint32_t>();
                sa.emat = b->d_emat.as<uint32_t>();
                sa.n_sc = b->n_sc;
                bool node_local = ra.n_rg == 0 && ra.csi_of == nullptr;   // (k_scanb: every input of a task in LDS, every effect on ONE node)
                for (uint32_t j = pos; j < upto && node_local; ++j)
                    if (b->rt[j].flags & RT_PORTS) node_local = false;
                if (prof) {
                    while (e->ev_scan.size() < (size_t)e->ev_scan_used + 2) {
                        hipEvent_t x;
                        HIPCHECK(e, hipEventCreate(&x));
                        e->ev_scan.push_back(x);
                    }
                    HIPCHECK(e, hipEventRecord(e->ev_scan[e->ev_scan_used], st));
                }
                r = launch_scan(sa, st, e->device, node_local);
                if (prof) {
                    HIPCHECK(e, hipEventRecord(e->ev_scan[e->ev_scan_used + 1], st));
                    e->ev_scan_used += 2;
                }
                e->scan_tasks_batch += upto - pos;
                e->scan_stretches_batch += 1;
                if (r == hipSuccess) r = launch_r6_build(ra, st);   // the rounds go on from the rows as the scan left them
                if (r != hipSuccess) return e->fail(SWP_EHIP, "k_scan launch: %s", hipGetErrorString(r));
                scanned += upto - pos;
                scan_len = std::min<uint32_t>(scan_len * 2, 1u << 20);   // still no plain candidates afterwards: the next stretch is twice as long
                chunk = scan_fast ? 2 : 4;   // (round 6: sixteen rounds of ~40 us between two stretches were a sixth of the dense batch)
                after_scan = true;
                if (node_local && scan_fast && (upto < end || (dbg_bits & 16))) {
                    // Most of the stretch was answered without a look (k_scanb: identical tasks had found no node — a saturated cluster's
                    // backlog): the kernel does that at a hundred tasks a microsecond, a round of the block resolver at ten. On with it.
                    HIPCHECK(e, hipMemcpyAsync(&hb, b->d_blk6.p, sizeof hb, hipMemcpyDeviceToHost, st));
                    HIPCHECK(e, hipStreamSynchronize(st));
                    const uint32_t sk = hb.scan_skipped - skipped_seen;
                    skipped_seen = hb.scan_skipped;
                    if (upto < end && 2 * (uint64_t)sk >= upto - pos) {
                        pos = upto;
                        goto scan_again;
                    }
                }
                pos = upto;
                continue;
            }
            after_scan = false;
            // as many rounds as the rest needs at the pace so far, and a few more: a round past the end costs two empty launches
            const double pace = std::max(1.0, recent);
            chunk = (uint32_t)std::min<double>(4096.0, (double)(end - pos) / pace * 1.05 + 4.0);
            if (scan_ok) chunk = std::min<uint32_t>(chunk, scan_fast ? (recent < 16.0 ? 8u : recent < 64.0 ? 16u : 256u) : (recent < 16.0 ? 16u : recent < 64.0 ? 48u : 256u));   // (look again soon: rounds that hit such a stretch decide one task each)
            // the block follows the pace: rounds that are cut after a few dozen tasks (re-placements that all aim at the few emptied nodes)
            // need not propose and stage hundreds of lists each; rounds that fill their block get the next size up
            if (!env_blk) {
                if (recent > 0.4 * ra.block) ra.block = std::min<uint32_t>(r6_block, ra.block * 2u);
                else ra.block = std::min<uint32_t>(r6_block, std::max<uint32_t>(128u, ((uint32_t)(2.0 * recent) + 63u) / 64u * 64u));
                chunk = std::min<uint32_t>(chunk, ra.block < r6_block ? 64u : 4096u);   // (look again before long while the block is small)
            }
        }
        if ((dbg_bits & 16) && scanned) fprintf(stderr, "[swp] k_scan decided %u tasks of [%u, %u), %u of them without a look (an identical task had found no node), the others %.2f to a barrier\n", scanned, start, end, hb.scan_skipped, (double)(scanned - hb.scan_skipped) / std::max(1u, hb.scan_batches));
#ifdef SWP_SCAN_PROF
        if ((dbg_bits & 16) && scanned) {
            Ctl pc{};
            (void)hipMemcpy(&pc, b->d_ctl.p, sizeof pc, hipMemcpyDeviceToHost);
            const double nb_ = std::max(1u, hb.scan_batches);
            fprintf(stderr, "[swp] k_scanb cycles per batch (wave 0; %u batches): read + evaluate %.0f, reduce + atomic %.0f, barrier %.0f, accept %.0f, apply %.0f, window work %.0f\n", hb.scan_batches,
                    pc.cyc[0] / nb_, pc.cyc[1] / nb_, pc.cyc[2] / nb_, pc.cyc[3] / nb_, pc.cyc[4] / nb_, pc.cyc[5] / nb_);
        }
#endif
        if ((dbg_bits & 16) && hb.crounds) fprintf(stderr, "[swp] %u of the %u rounds with a compact index | of the cuts at an exhausted list: %u full lists, %u lists in compact positions, %u lists of one entry\n", hb.crounds, hb.rounds, hb.dbg_cut[0], hb.dbg_cut[1], hb.dbg_cut[2]);
        e->r6_compact_hint = cpt_ever;
        r6_rounds += hb.rounds;
        if (prof) HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 3], st));
        ++wi;
        if (dbg_bits & 16)
            fprintf(stderr, "[swp] k_resolve6 tasks [%u, %u): %u rounds of %u (%.1f decided each) | cut by an exhausted list %u, an exception-list task %u, an uncounted task %u\n", start, end,
                    hb.rounds, r6_block, (double)(end - start) / std::max<uint32_t>(hb.rounds, 1), hb.cut_exhausted, hb.cut_exception, hb.cut_uncounted);
        if (dbg_bits & 16) {
            const double rr_ = std::max<uint32_t>(hb.rounds, 1);
            fprintf(stderr, "[swp] k_r6_commit shader cycles per round: prologue %.0f, matching (wave 0) %.0f (list loads %.0f, walks %.0f), the others' wait for it %.0f, apply %.0f | %.1f matcher stops at an emptied half-word per round\n",
                    hb.cyc[0] * 64.0 / rr_, hb.cyc[1] * 64.0 / rr_, hb.cyc_load * 64.0 / rr_, hb.cyc_walk * 64.0 / rr_, hb.cyc[2] * 64.0 / rr_, hb.cyc[3] * 64.0 / rr_, hb.reseats / rr_);
            fprintf(stderr, "[swp] ... of the list loads: waiting for a group's lists %.0f, its head records %.0f, seating %.0f (%.1f steps of the seating loop, stops included) per round\n", hb.cyc_g[0] * 64.0 / rr_,
                    hb.cyc_g[1] * 64.0 / rr_, hb.cyc_g[2] * 64.0 / rr_, hb.cyc_g[3] / rr_);
        }
        return SWP_OK;
    };
    // one stretch of tasks through the chosen family: the round resolver decides it in ONE launch
    auto run_stretch = [&](uint32_t start, uint32_t end, int variant) -> int {
        if (variant == 6) return run_blocks(start, end);
        if (prof)
            while (e->ev_pool.size() < (size_t)4 * (wi + 1)) {
                hipEvent_t x;
                HIPCHECK(e, hipEventCreate(&x));
                e->ev_pool.push_back(x);
            }
        ResolveArgs ra{};
        ra.n_nodes = N;
        ra.n_words = Wn;
        ra.j0 = start;
        ra.count = end - start;
        ra.dbg = dbg_bits;
        ra.xs = Wn;
        ra.valid = e->d_valid.as<u64>();
        ra.X = b->d_X.as<u64>();
        ra.rt = b->d_rt.as<RTask>();
        ra.cpu = e->d_cpu.as<long long>();
        ra.mem = e->d_mem.as<long long>();
        ra.total = e->d_total.as<uint32_t>();
        ra.list_node = b->d_list_node.as<uint32_t>();
        ra.list_svc = b->d_list_svc.as<uint32_t>();
        ra.list_fail = b->d_list_fail.as<uint32_t>();
        ra.list_off = b->d_list_off.as<uint32_t>();
        ra.portmap = b->d_portmap.as<u64>();
        ra.pset_off = b->d_pset_off.as<uint32_t>();
        ra.pset_ids = b->d_pset_ids.as<uint32_t>();
        ra.out_node = b->d_out.as<int32_t>();
        ra.log_node = b->d_log_node.as<uint32_t>();
        ra.log_task = b->d_log_task.as<uint32_t>();
        ra.log_prev = b->d_log_prev.as<int32_t>();
        ra.last = b->d_last.as<int32_t>();
        ra.inf_task = b->d_inf_task.as<uint32_t>();
        ra.inf_pos = b->d_inf_pos.as<uint32_t>();
        ra.ctl = b->d_ctl.as<Ctl>();
        ra.qres = b->d_qres.as<int32_t>();
        ra.unit_cpu = b->unit_cpu;
        ra.unit_mem = b->unit_mem;
        ra.sc = b->d_sc.as<u64>();
        ra.thr = b->d_thr.as<int32_t>();
        ra.n_dc = b->n_dc;
        ra.n_dm = b->n_dm;
        if (prof) {
            HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 0], st));
            HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 1], st));
            HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 2], st));
        }
        const hipError_t r = launch_resolve5(ra, r5_lds, st, e->device);
        if (r != hipSuccess) return e->fail(SWP_EHIP, "k_resolve5 launch: %s", hipGetErrorString(r));
        if (prof) HIPCHECK(e, hipEventRecord(e->ev_pool[4 * wi + 3], st));
        ++wi;
        return SWP_OK;
    };
    if (b->segs.empty()) rc = run_stretch(0, T, variant);
    else {
        HIPCHECK(e, b->d_wf.reserve((size_t)3 * N * 4));
        for (const swp_batch::Seg& sg : b->segs) {
            if (!sg.run) {
                if ((rc = run_stretch(sg.j0, sg.j0 + sg.n, variant))) break;
                continue;
            }
            WaterArgs wa{};
            wa.n_nodes = N;
            wa.n_words = Wn;
            wa.xs = Wn;
            wa.j0 = sg.j0;
            wa.count = sg.n;
            wa.rt = b->d_rt.as<RTask>();
            wa.sc = b->d_sc.as<u64>();
            wa.cpu = e->d_cpu.as<long long>();
            wa.mem = e->d_mem.as<long long>();
            wa.total = e->d_total.as<uint32_t>();
            wa.X = b->d_X.as<u64>();
            wa.list_node = b->d_list_node.as<uint32_t>();
            wa.list_svc = b->d_list_svc.as<uint32_t>();
            wa.list_fail = b->d_list_fail.as<uint32_t>();
            wa.list_off = b->d_list_off.as<uint32_t>();
            wa.out_node = b->d_out.as<int32_t>();
            wa.log_node = b->d_log_node.as<uint32_t>();
            wa.log_task = b->d_log_task.as<uint32_t>();
            wa.log_prev = b->d_log_prev.as<int32_t>();
            wa.last = b->d_last.as<int32_t>();
            wa.inf_task = b->d_inf_task.as<uint32_t>();
            wa.inf_pos = b->d_inf_pos.as<uint32_t>();
            wa.ctl = b->d_ctl.as<Ctl>();
            wa.qres = variant == 5 ? b->d_qres.as<int32_t>() : nullptr;
            wa.ps = b->d_wf.as<uint32_t>();
            wa.cap = wa.ps + N;
            wa.ent = wa.cap + N;
            hipError_t r = launch_waterfill(wa, st);
            if (r != hipSuccess) return e->fail(SWP_EHIP, "k_waterfill launch: %s", hipGetErrorString(r));
            e->stats.waterfill_tasks += sg.n;
        }
    }
    if (rc) return rc;

    // explain pass: needs the number of unplaceable tasks (one small D2H, once per batch)
    Ctl ctl{};
    HIPCHECK(e, hipMemcpyAsync(&ctl, b->d_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, st));
    HIPCHECK(e, hipStreamSynchronize(st));
    if (ctl.error == ERR_LEVEL_RANGE && variant == 5 && r6_ok && ctl.resume < T) {
        // The per-node task-count spread outgrew the 8 level planes the round resolver keeps in LDS (255 levels). It stopped cleanly
        // after task `resume` - 1 and every later launch returned at once: the block resolver (16 planes in global memory) carries
        // on from there — runs of identical tasks included, task by task.
        const uint32_t zero = 0;
        HIPCHECK(e, hipMemcpyAsync((char*)b->d_ctl.p + offsetof(Ctl, error), &zero, 4, hipMemcpyHostToDevice, st));
        variant = 6;
        rc = run_stretch(ctl.resume, T, 6);
        if (rc) return rc;
        HIPCHECK(e, hipMemcpyAsync(&ctl, b->d_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, st));
        HIPCHECK(e, hipStreamSynchronize(st));
    }
    if (prof) HIPCHECK(e, hipEventRecord(e->ev[2], st));
    if (ctl.error != ERR_NONE) return e->fail(SWP_ERANGE, "per-node task-count spread exceeds the resolvers' level planes");
    if (ctl.ninf && (rc = run_explain(e, b, ctl.ninf))) return rc;
    if (prof) {
        HIPCHECK(e, hipEventRecord(e->ev[3], st));
        HIPCHECK(e, hipEventSynchronize(e->ev[3]));
        float a = 0, c = 0, d = 0, t = 0;
        (void)hipEventElapsedTime(&a, e->ev[0], e->ev[1]);
        (void)hipEventElapsedTime(&c, e->ev[1], e->ev[2]);
        (void)hipEventElapsedTime(&d, e->ev[2], e->ev[3]);
        (void)hipEventElapsedTime(&t, e->ev[0], e->ev[3]);
        float scan_sum = 0, res_sum = 0;
        for (uint32_t q = 0; q < wi; ++q) {
            float x = 0, y = 0;
            (void)hipEventElapsedTime(&x, e->ev_pool[4 * q + 0], e->ev_pool[4 * q + 1]);
            (void)hipEventElapsedTime(&y, e->ev_pool[4 * q + 2], e->ev_pool[4 * q + 3]);
            scan_sum += x;
            res_sum += y;
        }
        for (uint32_t q = 0; q + 1 < e->ev_scan_used; q += 2) {   // the stretches the scan resolver took (k_scan_fill + k_scan_lists + k_scan / k_scanb each)
            float x = 0;
            (void)hipEventElapsedTime(&x, e->ev_scan[q], e->ev_scan[q + 1]);
            scan_sum += x;
        }
        e->stats.ms_classes = a;
        e->stats.ms_scan = scan_sum;      // Σ over windows of k_scan launch durations
        e->stats.ms_resolve = b->segs.empty() ? res_sum : c;   // Σ of the resolver launches; with runs of identical tasks in the batch: the whole phase (k_waterfill launches + the stretches between them)
        e->stats.ms_explain = d;
        e->stats.ms_total = t;
    }
    e->stats.verify_retries += ctl.verify_retries;
    e->stats.slow_path_tasks += ctl.slow_tasks;
    e->stats.rebase_events += ctl.rebases;
    e->stats.generic_tasks += ctl.generic_tasks;
    e->stats.resolver_spins += ctl.spin_waits;
    if ((dbg_bits & 16) && variant == 5)
        fprintf(stderr, "[swp] k_resolve5: rounds %llu (full %llu, cut by class %llu, cut by an exhausted list %llu) | commits %u inf %u generic-path tasks %llu retries %llu\n",
                ctl.cyc[0], ctl.cyc[1], ctl.cyc[2], ctl.cyc[3], ctl.ncommit, ctl.ninf, ctl.generic_tasks, ctl.verify_retries);
    if ((dbg_bits & 16) && variant == 5) {
        auto lo = [](unsigned long long v) { return (double)(v & 0xFFFFFFFFull) * 64.0; };
        auto hi = [](unsigned long long v) { return (double)(v >> 32) * 64.0; };
        const double r = ctl.cyc[0] ? (double)ctl.cyc[0] : 1.0;
        fprintf(stderr, "[swp] k_resolve5 shader cycles per round: wave0 match %.0f barrier1 %.0f commit %.0f barrier2 %.0f | wave1 list %.0f barrier1 %.0f idle %.0f barrier2 %.0f | cut/refill/generic %.0f\n",
                lo(ctl.cyc[4]) / r, hi(ctl.cyc[4]) / r, lo(ctl.cyc[5]) / r, hi(ctl.cyc[5]) / r, lo(ctl.cyc[6]) / r, hi(ctl.cyc[6]) / r, lo(ctl.cyc[7]) / r,
                hi(ctl.cyc[7]) / r, (double)ctl.pad1 * 64.0 / r);
        fprintf(stderr, "[swp] k_resolve5 matcher per round (cycles): list load %.0f, matching loop %.0f of which inside the scalar loop %.0f over %.1f entries\n", (double)ctl.m_cyc[0] * 64.0 / r, (double)ctl.m_cyc[1] * 64.0 / r,
                (double)ctl.m_cyc[2] * 64.0 / r, (double)ctl.m_cyc[3] / r);
fprintf(stderr, "[swp] k_resolve5 lister wave 1 per round (cycles): prologue %.0f | per 4 tasks: row wait %.0f, level search %.0f, entries %.0f, validation %.0f\n",
                (double)ctl.l_cyc[0] * 64.0 / r, (double)ctl.l_cyc[1] * 64.0 / r, (double)ctl.l_cyc[2] * 64.0 / r, (double)ctl.l_cyc[3] * 64.0 / r, (double)ctl.l_cyc[4] * 64.0 / r);
        fprintf(stderr, "[swp] k_resolve5 phase-1 work per wave and round (cycles):");
        for (int w = 0; w < 16; ++w) fprintf(stderr, " %.0f", (double)ctl.wave_cyc[w] * 64.0 / r);
        fprintf(stderr, "\n");
    }
    else if (dbg_bits & 16)
        fprintf(stderr, "[swp] resolver cycles (100MHz ticks): wait %llu prep %llu pick %llu generic %llu commit %llu blockend %llu | commits %u inf %u generic %llu\n",
                ctl.cyc[0], ctl.cyc[1], ctl.cyc[2], ctl.cyc[3], ctl.cyc[4], ctl.cyc[5], ctl.ncommit, ctl.ninf, ctl.generic_tasks);
    if ((dbg_bits & 16) && ctl.cyc[7])
        fprintf(stderr, "[swp] shader clock: %llu cycles / %llu x10ns => %.0f MHz\n", ctl.cyc[6], ctl.cyc[7], (double)ctl.cyc[6] / ((double)ctl.cyc[7] * 0.01));
    e->stats.scan_launches = e->scan_stretches_batch;
    e->stats.scan_tasks = e->scan_tasks_batch;
    e->stats.last_windows = wi;   // resolver launches of this batch (1 in k_resolve5's exact mode: no scan windows)
    e->stats.last_static_classes = b->n_sc;
    e->stats.resolve_launches += variant == 6 ? (uint32_t)(2 * r6_rounds) : wi;   // k_resolve6: a propose and a commit launch per round
    e->stats.last_resolver = variant == 5 ? 105u : 6u;   // 105 = k_resolve5 (its demand-class rows in LDS), 6 = k_resolve6
    b->ran = true;
    return SWP_OK;
}

// generic_set: only for placements the engine made itself (Claim's arithmetic, resource_management.go:11-72 with helpers.go:87-111:
// count -= request, an entry that reaches 0 leaves the list); the caller's own addTask / removeTask push the node's counts afterwards
void host_apply_placement(swp_engine* e, uint32_t node, uint32_t service, int64_t cpu, int64_t mem, uint32_t port_set, bool counted, bool add, uint32_t generic_set = 0) {
    e->host_dirty_since_save = true;
    HostNode& h = e->nodes[node];
    if (generic_set && add)
        for (const swp_generic& g : e->gen_sets[generic_set])
            for (size_t q = 0; q < h.gen.size(); ++q)
                if (h.gen[q].first == g.kind) {
                    h.gen[q].second -= g.value;
                    if (h.gen[q].second <= 0) h.gen.erase(h.gen.begin() + (long)q);
                    break;
                }
    if (add) {
        h.row.cpu -= cpu;
        h.row.mem -= mem;
    } else {
        h.row.cpu += cpu;
        h.row.mem += mem;
    }
    if (counted) {
        if (add) {
            h.row.total += 1;
            uint32_t c = ++h.svc[service];
            e->svc_nodes[service][node] = c;
        } else {
            h.row.total -= 1;
            uint32_t& c = h.svc[service];
            c -= 1;   // may wrap like Go's int going negative would not; counts are never negative on this path
            if (c == 0) {
                h.svc.erase(service);
                e->svc_nodes[service].erase(node);
            } else {
                e->svc_nodes[service][node] = c;
            }
        }
    }
    if (port_set)
        for (const swp_port& p : e->port_sets[port_set]) {
            uint64_t k = port_key(p.protocol, p.port);
            if (add) {
                h.ports.insert(k);
                e->port_nodes[k].insert(node);
            } else {
                h.ports.erase(k);   // removeTask deletes unconditionally (nodeinfo.go:78-84)
                e->port_nodes[k].erase(node);
            }
        }
}

// Many placements at once — a batch's results (swp_batch_fetch), a tick's groups, swp_commit's list: COUNTED tasks without host ports and
// without generic reservations (the caller sends the others through host_apply_placement one by one; bookings commute). The node records
// are updated in the order given, the ones a later placement needs on their way (prefetch); the service -> nodes index then takes every
// service's changes in ONE merge — the changes ordered by (service, node) with two counting sorts — instead of a sorted-vector insertion
// (or erasure) per placement. Round 6: 100 000 placements were 4.4-6 ms one by one.
struct BulkItem { uint32_t node, service; int64_t cpu, mem; };
struct BulkScratch { std::vector<uint32_t> cnt, ord, ord2, hist; std::vector<FlatMap32::Ent> merged; };
void host_apply_bulk(swp_engine* e, const std::vector<BulkItem>& it, bool add, BulkScratch& sc) {
    const size_t n = it.size();
    if (n == 0) return;
    e->host_dirty_since_save = true;
    uint32_t max_node = 0, max_svc = 0;
    for (const BulkItem& x : it) { max_node = std::max(max_node, x.node); max_svc = std::max(max_svc, x.service); }
    if (n < 32 || (size_t)max_node > 16 * n + 4096 || (size_t)max_svc > 16 * n + 4096) {   // a few placements, or ids the histograms below would not pay for
        for (const BulkItem& x : it) host_apply_placement(e, x.node, x.service, x.cpu, x.mem, 0, true, add);
        return;
    }
    // the node side, in the order given; cnt[i]: the node's count of the service after placement i
    sc.cnt.resize(n);
    for (size_t i = 0; i < n; ++i) {
        if (i + 16 < n) {
            __builtin_prefetch(&e->nodes[it[i + 16].node].row);
            __builtin_prefetch(&e->nodes[it[i + 16].node].svc);
        }
        if (i + 8 < n) {
            const FlatMap32& hs = e->nodes[it[i + 8].node].svc;
            if (!hs.v.empty()) __builtin_prefetch(hs.v.data());
        }
        HostNode& h = e->nodes[it[i].node];
        if (add) {
            h.row.cpu -= it[i].cpu;
            h.row.mem -= it[i].mem;
            h.row.total += 1;
            sc.cnt[i] = ++h.svc[it[i].service];
        } else {
            h.row.cpu += it[i].cpu;
            h.row.mem += it[i].mem;
            h.row.total -= 1;
            uint32_t& c = h.svc[it[i].service];
            c -= 1;   // (as host_apply_placement: counts are never negative on this path)
            sc.cnt[i] = c;
            if (c == 0) h.svc.erase(it[i].service);
        }
    }
    // order by (service, node), stable: by node, then by service (counting sorts; equal pairs stay in the order they were booked in,
    // so the LAST of a run carries the pair's final count)
    sc.ord.resize(n);
    sc.ord2.resize(n);
    sc.hist.assign((size_t)max_node + 2, 0);
    for (const BulkItem& x : it) sc.hist[x.node + 1]++;
    for (size_t q = 1; q < sc.hist.size(); ++q) sc.hist[q] += sc.hist[q - 1];
    for (size_t i = 0; i < n; ++i) sc.ord[sc.hist[it[i].node]++] = (uint32_t)i;
    sc.hist.assign((size_t)max_svc + 2, 0);
    for (const BulkItem& x : it) sc.hist[x.service + 1]++;
    for (size_t q = 1; q < sc.hist.size(); ++q) sc.hist[q] += sc.hist[q - 1];
    for (size_t j = 0; j < n; ++j) sc.ord2[sc.hist[it[sc.ord[j]].service]++] = sc.ord[j];
    // the index side: one merge per service
    for (size_t j = 0; j < n;) {
        const uint32_t svc = it[sc.ord2[j]].service;
        size_t j1 = j;
        while (j1 < n && it[sc.ord2[j1]].service == svc) ++j1;
        FlatMap32& m = e->svc_nodes[svc];
        sc.merged.clear();
        sc.merged.reserve(m.v.size() + (j1 - j));
        size_t a = 0;
        for (size_t u = j; u < j1; ++u) {
            const uint32_t node = it[sc.ord2[u]].node;
            if (u + 1 < j1 && it[sc.ord2[u + 1]].node == node) continue;
            while (a < m.v.size() && m.v[a].first < node) sc.merged.push_back(m.v[a++]);
            if (a < m.v.size() && m.v[a].first == node) ++a;
            if (sc.cnt[sc.ord2[u]]) sc.merged.emplace_back(node, sc.cnt[sc.ord2[u]]);   // (count 0: the service left the node)
        }
        while (a < m.v.size()) sc.merged.push_back(m.v[a++]);
        m.v.swap(sc.merged);
        j = j1;
    }
}

template <class Set, class Index, class Vec>
int register_set(Index& index, Vec& sets, const std::string& key, Set&& value, uint32_t* id_out) {
    auto it = index.find(key);
    if (it != index.end()) {
        *id_out = it->second;
        return SWP_OK;
    }
    uint32_t id = (uint32_t)sets.size();
    sets.push_back(std::forward<Set>(value));
    index.emplace(key, id);
    *id_out = id;
    return SWP_OK;
}

}  // namespace

// the shard set's side of every entry point (swp_shardset.hpp, included behind the extern "C" block)
namespace ss {
int create(const swp_config*, const int32_t*, uint32_t, uint32_t, swp_engine**);
void destroy(swp_engine*);
int reset(swp_engine*, uint32_t);
int intern(swp_engine*, int, const char*, size_t, uint32_t*);
int intern_lookup(swp_engine*, int, uint32_t, char*, size_t);
int node_upsert(swp_engine*, const swp_node_row*, const swp_kv*, uint32_t, const swp_kv*, uint32_t, const uint32_t*, uint32_t);
int node_update_dynamic(swp_engine*, uint32_t, uint32_t, int64_t, int64_t, uint32_t);
int node_get(swp_engine*, uint32_t, swp_node_row*);
int node_remove(swp_engine*, uint32_t);
int node_set_svc_count(swp_engine*, uint32_t, uint32_t, uint32_t);
int node_get_svc_count(swp_engine*, uint32_t, uint32_t, uint32_t*);
int node_set_failures(swp_engine*, uint32_t, uint32_t, uint64_t, uint32_t);
int node_port(swp_engine*, uint32_t, uint32_t, uint32_t, int);
int node_set_generic(swp_engine*, uint32_t, const swp_generic*, uint32_t);
int node_get_generic(swp_engine*, uint32_t, uint32_t, int64_t*);
int node_set_csi(swp_engine*, uint32_t, const swp_csi*, uint32_t, const swp_seg*, uint32_t);
int constraint_set(swp_engine*, const swp_constraint*, uint32_t, uint32_t*);
int platform_set(swp_engine*, const swp_platform*, uint32_t, uint32_t*);
int plugin_set(swp_engine*, const uint32_t*, uint32_t, uint32_t, uint32_t*);
int port_set(swp_engine*, const swp_port*, uint32_t, uint32_t*);
int spread_set(swp_engine*, const swp_spread*, uint32_t, uint32_t*);
int generic_set(swp_engine*, const swp_generic*, uint32_t, uint32_t*);
int mount_set(swp_engine*, const swp_mount*, uint32_t, uint32_t*);
int volume_upsert(swp_engine*, uint32_t, const swp_volume*, const uint32_t*, const swp_seg*);
int volume_set_usage(swp_engine*, uint32_t, const swp_volume_usage*);
int volume_get_usage(swp_engine*, uint32_t, swp_volume_usage*);
int choose_volumes(swp_engine*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*);
int batch_prepare(swp_engine*, const swp_task_desc*, uint32_t, const uint32_t*, uint32_t, bool, swp_batch**);
int batch_run(swp_engine*, swp_batch*);
int batch_collect(swp_engine*, swp_batch*, int32_t*, uint32_t*, bool);
int batch_attachments(swp_engine*, swp_batch*, const uint32_t*, uint32_t, uint32_t*);
void batch_free(swp_engine*, swp_batch*);
int schedule_groups(swp_engine*, const swp_task_desc*, const uint32_t*, uint32_t, int32_t*, uint32_t*, uint32_t*);
int state_save(swp_engine*);
int state_restore(swp_engine*);
int commit(swp_engine*, const swp_placement*, uint32_t, int);
int check_node(swp_engine*, const swp_task_desc*, uint32_t, int32_t*);
int enforce(swp_engine*, const swp_enforce_node*, uint32_t, const swp_enforce_task*, uint32_t, uint8_t*);
int node_matches(swp_engine*, const uint32_t*, uint32_t, uint64_t*, uint32_t);
int stats(swp_engine*, swp_stats_t*);
}   // namespace ss

// =================================================================================================
extern "C" {

const char* swp_strerror(int code) {
    switch (code) {
    case SWP_OK: return "ok";
    case SWP_EINVAL: return "invalid argument";
    case SWP_ENOTFOUND: return "node not found in scheduler dataset";   // errNodeNotFound, nodeset.go:12
    case SWP_ENOMEM: return "out of memory";
    case SWP_EHIP: return "HIP runtime error";
    case SWP_EUNSUPPORTED: return "unsupported on the device path";
    case SWP_ERANGE: return "value outside engine limits";
    case SWP_ENODEVICE: return "no gfx950 device (the engine has no CPU fallback)";
    }
    return "unknown error";
}

const char* swp_last_error(swp_engine* e) { return e ? e->last_error.c_str() : g_create_error.c_str(); }

int swp_abi_check(uint32_t* sizes, uint32_t n) {
    const uint32_t s[] = {sizeof(swp_config), sizeof(swp_node_row), sizeof(swp_kv), sizeof(swp_constraint), sizeof(swp_platform),
                          sizeof(swp_port), sizeof(swp_task_desc), sizeof(swp_placement), sizeof(swp_stats_t), sizeof(swp_spread), sizeof(swp_generic)};
    uint32_t m = sizeof s / sizeof s[0];
    for (uint32_t i = 0; i < n && i < m; ++i) sizes[i] = s[i];
    return (int)m;
}

int swp_create(const swp_config* cfg, swp_engine** out) {
    if (!out) return SWP_EINVAL;
    *out = nullptr;
    int count = 0;
    hipError_t r = hipGetDeviceCount(&count);
    if (r != hipSuccess || count == 0) {
        g_create_error = std::string("no HIP device: ") + (r == hipSuccess ? "device count is 0" : hipGetErrorString(r));
        return SWP_ENODEVICE;
    }
    int dev = cfg ? cfg->device : 0;
    if (dev < 0 || dev >= count) {
        g_create_error = "device ordinal out of range";
        return SWP_EINVAL;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is not gfx950: ") + prop.gcnArchName;
        return SWP_ENODEVICE;
    }
    auto e = std::make_unique<swp_engine>();
    if (cfg) e->cfg = *cfg;
    e->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_error = "hipStreamCreate failed";
        return SWP_EHIP;
    }
    for (auto& ev : e->ev)
        if (hipEventCreate(&ev) != hipSuccess) {
            g_create_error = "hipEventCreate failed";
            return SWP_EHIP;
        }
    for (int s = 0; s < SWP_SPACE_COUNT; ++s) e->spaces[s].init(s != SWP_SPACE_NODE_ID);
    e->role_worker = e->spaces[SWP_SPACE_FOLDED].get("worker");
    e->role_manager = e->spaces[SWP_SPACE_FOLDED].get("manager");
    *out = e.release();
    return SWP_OK;
}

void swp_destroy(swp_engine* e) {
    if (!e) return;
    host_prof().print();
    if (e->set) { ss::destroy(e); return; }
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    // (an RCCL communicator is NOT torn down here: engines are often destroyed while the process exits, after RCCL's own static
    // state is gone — swp_rccl_finalize is the orderly way)
    for (auto& ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_scan) (void)hipEventDestroy(ev);
    if (e->rows_ev) (void)hipEventDestroy(e->rows_ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    const int dev = e->device;
    delete e;
    dev_pool().drain(dev);   // (the allocations the engine just handed back, and whatever freed batches left)
}

int swp_reset(swp_engine* e, uint32_t n_nodes_hint) {
    if (e && e->set) return ss::reset(e, n_nodes_hint);
    if (!e) return SWP_EINVAL;
    engine_reset_nodes(e);
    e->spaces[SWP_SPACE_NODE_ID].init(false);
    e->nodes.reserve(n_nodes_hint);
    return SWP_OK;
}

int swp_intern(swp_engine* e, int space, const char* utf8, size_t len, uint32_t* id_out) {
    if (e && e->set) return ss::intern(e, space, utf8, len, id_out);
    if (!e || space < 0 || space >= SWP_SPACE_COUNT || !id_out || (!utf8 && len)) return SWP_EINVAL;
    std::string s = space == SWP_SPACE_FOLDED ? fold_canon(utf8, len) : std::string(utf8 ? utf8 : "", len);
    if (space == SWP_SPACE_ARCH) {   // filter.go:285-299
        if (s == "x86_64") s = "amd64";
        else if (s == "aarch64") s = "arm64";
    }
    *id_out = e->spaces[space].get(s);
    return SWP_OK;
}

int swp_intern_lookup(swp_engine* e, int space, uint32_t id, char* out, size_t cap) {
    if (e && e->set) return ss::intern_lookup(e, space, id, out, cap);
    if (!e || space < 0 || space >= SWP_SPACE_COUNT) return SWP_EINVAL;
    const auto& strs = e->spaces[space].strs;
    if (id >= strs.size()) return SWP_ENOTFOUND;
    const std::string& s = strs[id];
    if (out && cap) std::memcpy(out, s.data(), std::min(cap, s.size()));
    return (int)s.size();
}

int swp_node_upsert(swp_engine* e, const swp_node_row* row, const swp_kv* node_labels, uint32_t n_node_labels, const swp_kv* engine_labels,
                    uint32_t n_engine_labels, const uint32_t* plugins, uint32_t n_plugins) {
    if (e && e->set) return ss::node_upsert(e, row, node_labels, n_node_labels, engine_labels, n_engine_labels, plugins, n_plugins);
    if (!e || !row) return SWP_EINVAL;
    e->host_dirty_since_save = true;
    if (row->node >= e->spaces[SWP_SPACE_NODE_ID].strs.size()) return e->fail(SWP_EINVAL, "node id %u was never interned", row->node);
    if (node_index_released(e, row->node)) return e->fail(SWP_EINVAL, "node index %u was released by swp_node_remove: intern the node id again", row->node);
    if (row->total >= (1u << 30)) return e->fail(SWP_ERANGE, "ActiveTasksCount out of range");
    if (row->node >= e->nodes.size()) e->nodes.resize(row->node + 1);
    HostNode& h = e->nodes[row->node];
    if (!h.present) e->n_present++;
    h.present = true;
    h.row = *row;
    h.labels.assign(node_labels, node_labels + n_node_labels);
    h.elabels.assign(engine_labels, engine_labels + n_engine_labels);
    h.plugins.assign(plugins, plugins + n_plugins);
    e->n_nodes = std::max(e->n_nodes, row->node + 1);
    e->dev_static_dirty = e->dev_dynamic_dirty = true;
    return SWP_OK;
}

int swp_node_update_dynamic(swp_engine* e, uint32_t node, uint32_t flags, int64_t cpu, int64_t mem, uint32_t total) {
    if (e && e->set) return ss::node_update_dynamic(e, node, flags, cpu, mem, total);
    if (!e) return SWP_EINVAL;
    e->host_dirty_since_save = true;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    HostNode& h = e->nodes[node];
    const bool moved = h.row.flags != flags || h.row.cpu != cpu || h.row.mem != mem || h.row.total != total;
    if (h.row.flags != flags) e->dev_flags_dirty = true;   // (the flags word alone: no other static array, no volume topology, depends on it)
    h.row.flags = flags;
    h.row.cpu = cpu;
    h.row.mem = mem;
    h.row.total = total;
    if (moved && !e->dev_dynamic_dirty) e->dirty_rows.push_back(node);   // (a pending whole-array upload carries the row anyway)
    return SWP_OK;
}

int swp_node_update_dynamic_many(swp_engine* e, const swp_node_dynamic* rows, uint32_t n) {
    if (!e || (!rows && n)) return SWP_EINVAL;
    for (uint32_t i = 0; i < n; ++i) {
        const int rc = swp_node_update_dynamic(e, rows[i].node, rows[i].flags, rows[i].cpu, rows[i].mem, rows[i].total);
        if (rc) return e->fail(rc, "swp_node_update_dynamic_many: row %u (node %u)", i, rows[i].node);
    }
    return SWP_OK;
}

int swp_node_get_many(swp_engine* e, const uint32_t* nodes, uint32_t n, swp_node_row* out) {
    if (!e || ((!nodes || !out) && n)) return SWP_EINVAL;
    for (uint32_t i = 0; i < n; ++i) {
        const int rc = swp_node_get(e, nodes[i], &out[i]);
        if (rc) return e->fail(rc, "swp_node_get_many: row %u (node %u)", i, nodes[i]);
    }
    return SWP_OK;
}

int swp_node_remove(swp_engine* e, uint32_t node) {
    if (e && e->set) return ss::node_remove(e, node);
    if (!e) return SWP_EINVAL;
    e->host_dirty_since_save = true;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_OK;   // delete of an absent key is a no-op
    HostNode& h = e->nodes[node];
    for (auto& kv : h.svc) e->svc_nodes[kv.first].erase(node);
    for (auto& kv : h.fails) e->fail_nodes[kv.first.first].erase(node);
    for (uint64_t k : h.ports) e->port_nodes[k].erase(node);
    h = HostNode();
    e->n_present--;
    e->dev_static_dirty = e->dev_dynamic_dirty = true;
    for (HostVolume& v : e->volumes)   // a volume whose users sat on this node: the index will name another node — no node of the set is theirs
        if (v.use.pin == node) { v.use.pin = SWP_PIN_MANY; e->vol_dyn_dirty = true; }
    // nodeSet.remove deletes the map entry (nodeset.go:46-48): the node's index goes back to the pool, the next node id that is new to
    // swp_intern(SWP_SPACE_NODE_ID) gets the lowest free one (the canonical scan order is the index order; the oracle recycles its
    // slots by the same rule)
    e->spaces[SWP_SPACE_NODE_ID].release(node);
    return SWP_OK;
}

int swp_node_get(swp_engine* e, uint32_t node, swp_node_row* out) {
    if (e && e->set) return ss::node_get(e, node, out);
    if (!e || !out) return SWP_EINVAL;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    *out = e->nodes[node].row;
    return SWP_OK;
}

int swp_node_set_svc_count(swp_engine* e, uint32_t node, uint32_t service, uint32_t count) {
    if (e && e->set) return ss::node_set_svc_count(e, node, service, count);
    if (!e) return SWP_EINVAL;
    e->host_dirty_since_save = true;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    if (count >= (1u << 30)) return e->fail(SWP_ERANGE, "service task count out of range");
    HostNode& h = e->nodes[node];
    if (count) {
        h.svc[service] = count;
        e->svc_nodes[service][node] = count;
    } else {
        h.svc.erase(service);
        e->svc_nodes[service].erase(node);
    }
    return SWP_OK;
}

int swp_node_get_svc_count(swp_engine* e, uint32_t node, uint32_t service, uint32_t* count_out) {
    if (e && e->set) return ss::node_get_svc_count(e, node, service, count_out);
    if (!e || !count_out) return SWP_EINVAL;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    auto it = e->nodes[node].svc.find(service);
    *count_out = it == e->nodes[node].svc.end() ? 0 : it->second;
    return SWP_OK;
}

int swp_node_set_failures(swp_engine* e, uint32_t node, uint32_t service, uint64_t spec_version, uint32_t count) {
    if (e && e->set) return ss::node_set_failures(e, node, service, spec_version, count);
    if (!e) return SWP_EINVAL;
    e->host_dirty_since_save = true;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    if (count >= (1u << 30)) return e->fail(SWP_ERANGE, "failure count out of range");
    HostNode& h = e->nodes[node];
    if (count) {
        h.fails[{service, spec_version}] = count;
        e->fail_nodes[service].insert(node);
    } else {
        h.fails.erase({service, spec_version});
    }
    return SWP_OK;
}

int swp_node_port(swp_engine* e, uint32_t node, uint32_t protocol, uint32_t port, int set) {
    if (e && e->set) return ss::node_port(e, node, protocol, port, set);
    if (!e) return SWP_EINVAL;
    e->host_dirty_since_save = true;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    uint64_t k = port_key(protocol, port);
    if (set) {
        e->nodes[node].ports.insert(k);
        e->port_nodes[k].insert(node);
    } else {
        e->nodes[node].ports.erase(k);
        e->port_nodes[k].erase(node);
    }
    return SWP_OK;
}

int swp_constraint_set(swp_engine* e, const swp_constraint* cs, uint32_t n, uint32_t* id_out) {
    if (e && e->set) return ss::constraint_set(e, cs, n, id_out);
    if (!e || !id_out || (!cs && n)) return SWP_EINVAL;
    if (n == 0) { *id_out = 0; return SWP_OK; }
    for (uint32_t i = 0; i < n; ++i) {
        if (cs[i].kind > SWP_CK_INVALID || cs[i].op > SWP_OP_NE) return e->fail(SWP_EINVAL, "bad constraint kind/op");
        if (cs[i].kind == SWP_CK_NODE_LABEL && !e->node_label_col.count(cs[i].key)) {
            e->node_label_col[cs[i].key] = e->n_cols++;
            e->dev_static_dirty = true;
        }
        if (cs[i].kind == SWP_CK_ENGINE_LABEL && !e->engine_label_col.count(cs[i].key)) {
            e->engine_label_col[cs[i].key] = e->n_cols++;
            e->dev_static_dirty = true;
        }
    }
    return register_set(e->con_index, e->con_sets, bytes_of(cs, n), std::vector<swp_constraint>(cs, cs + n), id_out);
}

int swp_platform_set(swp_engine* e, const swp_platform* ps, uint32_t n, uint32_t* id_out) {
    if (e && e->set) return ss::platform_set(e, ps, n, id_out);
    if (!e || !id_out || (!ps && n)) return SWP_EINVAL;
    if (n == 0) { *id_out = 0; return SWP_OK; }
    return register_set(e->plat_index, e->plat_sets, bytes_of(ps, n), std::vector<swp_platform>(ps, ps + n), id_out);
}

int swp_plugin_set(swp_engine* e, const uint32_t* required, uint32_t n, uint32_t log_plugin, uint32_t* id_out) {
    if (e && e->set) return ss::plugin_set(e, required, n, log_plugin, id_out);
    if (!e || !id_out || (!required && n)) return SWP_EINVAL;
    swp_engine::PlugSet ps;
    ps.required.assign(required, required + n);
    ps.log = log_plugin;
    std::string key = bytes_of(&log_plugin, 1) + bytes_of(required, n);
    return register_set(e->plug_index, e->plug_sets, key, std::move(ps), id_out);
}

int swp_port_set(swp_engine* e, const swp_port* ports, uint32_t n, uint32_t* id_out) {
    if (e && e->set) return ss::port_set(e, ports, n, id_out);
    if (!e || !id_out || (!ports && n)) return SWP_EINVAL;
    if (n == 0) { *id_out = 0; return SWP_OK; }
    // the Explain pass tracks "this port was taken on the node by a LATER commit" in a 32-bit mask per task
    if (n > 32) return e->fail(SWP_ERANGE, "a task with %u host-mode ports exceeds the device limit of 32 (keep it on the Go path)", n);
    return register_set(e->port_index, e->port_sets, bytes_of(ports, n), std::vector<swp_port>(ports, ports + n), id_out);
}

int swp_spread_set(swp_engine* e, const swp_spread* levels, uint32_t n, uint32_t* id_out) {
    if (e && e->set) return ss::spread_set(e, levels, n, id_out);
    if (!e || !id_out || (!levels && n)) return SWP_EINVAL;
    if (n == 0) { *id_out = 0; return SWP_OK; }
    for (uint32_t i = 0; i < n; ++i)
        if (levels[i].kind != SWP_CK_NODE_LABEL && levels[i].kind != SWP_CK_ENGINE_LABEL) return e->fail(SWP_EINVAL, "spread level kind must be a label kind");
    return register_set(e->spread_index, e->spread_sets, bytes_of(levels, n), std::vector<swp_spread>(levels, levels + n), id_out);
}

int swp_generic_set(swp_engine* e, const swp_generic* items, uint32_t n, uint32_t* id_out) {
    if (e && e->set) return ss::generic_set(e, items, n, id_out);
    if (!e || !id_out || (!items && n)) return SWP_EINVAL;
    if (n == 0) { *id_out = 0; return SWP_OK; }
    if (n > 8) return e->fail(SWP_ERANGE, "a task reserves %u generic kinds (the engine takes 8)", n);
    std::vector<swp_generic> v(items, items + n);
    for (swp_generic& g : v) {
        g.reserved = 0;
        if (g.kind == 0 || g.kind >= e->spaces[SWP_SPACE_GENERIC_KIND].strs.size()) return e->fail(SWP_EINVAL, "unknown generic kind id %u", g.kind);
        if (g.value < 1) return e->fail(SWP_EUNSUPPORTED, "a generic reservation of %lld (a request of 0 claims every named value of the kind in the reference): the task stays on the Go path", (long long)g.value);
        if (g.value >= (1ll << 31)) return e->fail(SWP_ERANGE, "generic reservation %lld exceeds 2^31", (long long)g.value);
    }
    std::sort(v.begin(), v.end(), [](const swp_generic& a, const swp_generic& b) { return a.kind < b.kind; });
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i].kind == v[i - 1].kind) return e->fail(SWP_EUNSUPPORTED, "a task reserves generic kind %u twice: it stays on the Go path", v[i].kind);
    return register_set(e->gen_index, e->gen_sets, bytes_of(v.data(), v.size()), std::move(v), id_out);
}

// ---- CSI volumes ------------------------------------------------------------------------------------------------------------
int swp_node_set_csi(swp_engine* e, uint32_t node, const swp_csi* infos, uint32_t n, const swp_seg* segs, uint32_t n_segs) {
    if (e && e->set) return ss::node_set_csi(e, node, infos, n, segs, n_segs);
    if (!e || (!infos && n) || (!segs && n_segs)) return SWP_EINVAL;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    std::vector<HostNode::Csi> v(n);
    for (uint32_t i = 0; i < n; ++i) {
        if ((uint64_t)infos[i].seg_off + infos[i].n_seg > n_segs) return e->fail(SWP_EINVAL, "CSI info %u: segments outside the array", i);
        v[i].plugin = infos[i].plugin;
        v[i].has_topology = infos[i].has_topology ? 1u : 0u;
        v[i].segs.assign(segs + infos[i].seg_off, segs + infos[i].seg_off + infos[i].n_seg);
    }
    auto same = [](const HostNode::Csi& a, const HostNode::Csi& b) {
        return a.plugin == b.plugin && a.has_topology == b.has_topology && a.segs.size() == b.segs.size() &&
               (a.segs.empty() || std::memcmp(a.segs.data(), b.segs.data(), a.segs.size() * sizeof(swp_seg)) == 0);
    };
    HostNode& h = e->nodes[node];
    if (h.csi.size() != v.size() || !std::equal(v.begin(), v.end(), h.csi.begin(), same)) {
        h.csi = std::move(v);
        e->vol_static_dirty = true;
    }
    return SWP_OK;
}

int swp_volume_upsert(swp_engine* e, uint32_t volume, const swp_volume* v, const uint32_t* topo_off, const swp_seg* segs) {
    if (e && e->set) return ss::volume_upsert(e, volume, v, topo_off, segs);
    if (!e || !v || (v->n_topologies && (!topo_off || !segs))) return SWP_EINVAL;
    if (volume == 0 || volume >= e->spaces[SWP_SPACE_VOLUME].strs.size()) return e->fail(SWP_EINVAL, "volume id %u was never interned", volume);
    if (v->group >= e->spaces[SWP_SPACE_VOLUME_GROUP].strs.size()) return e->fail(SWP_EINVAL, "volume group id %u was never interned", v->group);
    if (v->scope > 1 || v->sharing > 3) return e->fail(SWP_EINVAL, "volume %u: unknown access mode", volume);
    if (volume >= e->volumes.size()) e->volumes.resize(volume + 1);
    HostVolume& hv = e->volumes[volume];
    if (hv.present) {
        // addOrUpdateVolume for a volume the set holds already (volumes.go:62-72): `info.volume = v` assigns to a COPY of the map's value
        // (vs.volumes is a map of structs), so the volume object checkVolume reads — availability, access mode, driver, accessible
        // topology — stays the FIRST one for ever; what the call does change is byGroup, which gains the volume under the group the new
        // object names and is never pruned (:74-78). Restated to the letter.
        if (std::find(hv.groups.begin(), hv.groups.end(), v->group) == hv.groups.end()) {
            hv.groups.push_back(v->group);
            e->vol_static_dirty = true;
        }
        return SWP_OK;
    }
    hv.present = true;
    hv.spec = *v;
    hv.groups.assign(1, v->group);
    hv.topo_off.assign(1, 0);
    hv.segs.clear();
    for (uint32_t t = 0; t < v->n_topologies; ++t) {
        if (topo_off[t + 1] < topo_off[t]) return e->fail(SWP_EINVAL, "volume %u: topology offsets must ascend", volume);
        hv.segs.insert(hv.segs.end(), segs + topo_off[t], segs + topo_off[t + 1]);
        hv.topo_off.push_back((uint32_t)hv.segs.size());
    }
    e->vol_static_dirty = true;
    return SWP_OK;
}

int swp_volume_set_usage(swp_engine* e, uint32_t volume, const swp_volume_usage* u) {
    if (e && e->set) return ss::volume_set_usage(e, volume, u);
    if (!e || !u) return SWP_EINVAL;
    if (volume >= e->volumes.size() || !e->volumes[volume].present) return SWP_ENOTFOUND;
    // (a pin with VOL_PIN_FOREIGN set names a node of ANOTHER shard of the set this engine belongs to: swp_shardset.hpp)
    if (u->pin < VOL_PIN_FOREIGN && (u->pin >= e->nodes.size() || node_index_released(e, u->pin))) return e->fail(SWP_EINVAL, "volume %u: unknown node %u", volume, u->pin);
    if (u->pin >= VOL_PIN_FOREIGN && u->pin < SWP_PIN_MANY && e->cfg.shard_count < 2) return e->fail(SWP_EINVAL, "volume %u: unknown node %u", volume, u->pin);
    e->volumes[volume].use = *u;
    e->volumes[volume].use.reserved = 0;
    e->vol_dyn_dirty = true;
    return SWP_OK;
}

int swp_volume_get_usage(swp_engine* e, uint32_t volume, swp_volume_usage* out) {
    if (e && e->set) return ss::volume_get_usage(e, volume, out);
    if (!e || !out) return SWP_EINVAL;
    if (volume >= e->volumes.size() || !e->volumes[volume].present) return SWP_ENOTFOUND;
    *out = e->volumes[volume].use;
    return SWP_OK;
}

int swp_mount_set(swp_engine* e, const swp_mount* mounts, uint32_t n, uint32_t* id_out) {
    if (e && e->set) return ss::mount_set(e, mounts, n, id_out);
    if (!e || !id_out || (!mounts && n)) return SWP_EINVAL;
    if (n == 0) { *id_out = 0; return SWP_OK; }
    if (n > SWP_MAX_MOUNTS) return e->fail(SWP_ERANGE, "a task has %u cluster mounts (the engine takes %d)", n, SWP_MAX_MOUNTS);
    std::vector<swp_mount> v(mounts, mounts + n);
    for (swp_mount& m : v) {
        m.is_group = m.is_group ? 1u : 0u;
        m.read_only = m.read_only ? 1u : 0u;
        m.reserve_read_only = m.reserve_read_only ? 1u : 0u;
        if (m.ref != SWP_NO_VOLUME) {
            if (m.is_group && m.ref >= e->spaces[SWP_SPACE_VOLUME_GROUP].strs.size()) return e->fail(SWP_EINVAL, "unknown volume group id %u", m.ref);
            if (!m.is_group && (m.ref >= e->volumes.size() || !e->volumes[m.ref].present)) return e->fail(SWP_EINVAL, "unknown volume %u", m.ref);
        }
    }
    const size_t before = e->mount_sets.size();
    const int rc = register_set(e->mount_index, e->mount_sets, bytes_of(v.data(), v.size()), std::move(v), id_out);
    if (e->mount_sets.size() != before) e->vol_static_dirty = true;
    if (!rc && *id_out >= (1u << 24)) return e->fail(SWP_ERANGE, "too many distinct mount sets");
    return rc;
}

int swp_choose_volumes(swp_engine* e, uint32_t mount_set, uint32_t node, uint32_t* out, uint32_t* n_out, uint32_t* failed_mount) {
    if (e && e->set) return ss::choose_volumes(e, mount_set, node, out, n_out, failed_mount);
    if (!e || !out || !n_out) return SWP_EINVAL;
    if (mount_set == 0 || mount_set >= e->mount_sets.size()) return e->fail(SWP_EINVAL, "unknown mount set %u", mount_set);
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    (void)hipSetDevice(e->device);
    int rc = flush_nodes(e);
    if (rc) return rc;
    for (uint32_t q = 0; q < SWP_MAX_MOUNTS; ++q) out[q] = SWP_NO_VOLUME;
    *n_out = 0;
    if (!e->has_volumes()) {   // no volume exists: the first mount fails
        if (failed_mount) *failed_mount = 0;
        return SWP_OK;
    }
    DevBuf d;
    HIPCHECK(e, d.reserve((SWP_MAX_MOUNTS + 3) * 4));
    VolChooseArgs ca{};
    ca.vol = vol_view(e);
    ca.set = mount_set;
    ca.node = node;
    ca.out = d.as<u32>();
    hipError_t r = launch_vol_choose(ca, e->stream);
    if (r != hipSuccess) return e->fail(SWP_EHIP, "k_vol_choose launch: %s", hipGetErrorString(r));
    uint32_t h[SWP_MAX_MOUNTS + 3];
    HIPCHECK(e, hipMemcpyAsync(h, d.p, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    for (uint32_t q = 0; q < SWP_MAX_MOUNTS; ++q) out[q] = h[q];
    *n_out = h[SWP_MAX_MOUNTS];
    if (failed_mount) *failed_mount = h[SWP_MAX_MOUNTS + 1];
    return SWP_OK;
}

int swp_node_set_generic(swp_engine* e, uint32_t node, const swp_generic* counts, uint32_t n) {
    if (e && e->set) return ss::node_set_generic(e, node, counts, n);
    if (!e || (!counts && n)) return SWP_EINVAL;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    std::vector<std::pair<uint32_t, int64_t>> v;
    for (uint32_t i = 0; i < n; ++i) {
        if (counts[i].kind == 0 || counts[i].kind >= e->spaces[SWP_SPACE_GENERIC_KIND].strs.size()) return e->fail(SWP_EINVAL, "unknown generic kind id %u", counts[i].kind);
        if (counts[i].value < 0 || counts[i].value >= (1ll << 31)) return e->fail(SWP_ERANGE, "generic count %lld outside [0, 2^31)", (long long)counts[i].value);
        if (counts[i].value > 0) v.emplace_back(counts[i].kind, counts[i].value);
    }
    std::sort(v.begin(), v.end());
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i].first == v[i - 1].first) return e->fail(SWP_EINVAL, "generic kind %u listed twice", v[i].first);
    if (v != e->nodes[node].gen) {
        e->nodes[node].gen = std::move(v);
        e->dev_dynamic_dirty = true;
        e->host_dirty_since_save = true;
    }
    return SWP_OK;
}

int swp_node_get_generic(swp_engine* e, uint32_t node, uint32_t kind, int64_t* count_out) {
    if (e && e->set) return ss::node_get_generic(e, node, kind, count_out);
    if (!e || !count_out) return SWP_EINVAL;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    *count_out = 0;
    for (const auto& kv : e->nodes[node].gen)
        if (kv.first == kind) *count_out = kv.second;
    return SWP_OK;
}

static int groups_apply_to_host(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, uint64_t total, const int32_t* out_node);

static int schedule_groups_impl(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, int32_t* out_node, uint32_t* out_fail_hist, uint32_t* out_att);
int swp_schedule_groups(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, int32_t* out_node,
                        uint32_t* out_fail_hist) {
    return schedule_groups_impl(e, groups, sizes, n_groups, out_node, out_fail_hist, nullptr);
}
int swp_schedule_groups_volumes(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, int32_t* out_node,
                                uint32_t* out_fail_hist, uint32_t* out_att) {
    if (!out_att && n_groups) return SWP_EINVAL;
    return schedule_groups_impl(e, groups, sizes, n_groups, out_node, out_fail_hist, out_att);
}
static int schedule_groups_impl(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, int32_t* out_node,
                                uint32_t* out_fail_hist, uint32_t* out_att) {
    if (!e || (!groups && n_groups) || (!sizes && n_groups)) return SWP_EINVAL;
    if (n_groups == 0) return SWP_OK;
    if (e->set) return ss::schedule_groups(e, groups, sizes, n_groups, out_node, out_fail_hist, out_att);
    (void)hipSetDevice(e->device);
    uint64_t total = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        if (sizes[g] == 0) return e->fail(SWP_EINVAL, "group %u is empty", g);
        total += sizes[g];
    }
    if (total >= (1ull << 31) || !out_node) return SWP_EINVAL;
    if (out_fail_hist) std::memset(out_fail_hist, 0, (size_t)n_groups * SWP_NFILTERS * 4);
    int rc = flush_nodes(e);
    if (rc) return rc;
    const uint32_t N = e->n_nodes, Wn = n_words_of(N);
    if (N == 0) {   // empty nodeSet: every task is left over, empty explanation
        for (uint64_t i = 0; i < total; ++i) out_node[i] = -1;
        return SWP_OK;
    }
    HostSpan sp("groups: build_batch + upload");
    swp_batch b;
    if ((rc = build_batch(e, groups, n_groups, &b, sizes))) return rc;
    if ((rc = flush_nodes(e))) return rc;
    if ((rc = upload_batch(e, &b))) return rc;
    hipStream_t st = e->stream;
    sp.next("groups: trees, records, class lists, buffers");

    // decision-tree topology per spread set: branch creation order = node index order (nodeset.go:57-101)
    std::map<uint32_t, uint32_t> tree_local;   // spread set -> local tree
    std::vector<uint32_t> tree_sets;
    for (uint32_t g = 0; g < n_groups; ++g)
        if (!tree_local.count(groups[g].spread_set)) {
            tree_local[groups[g].spread_set] = (uint32_t)tree_sets.size();
            tree_sets.push_back(groups[g].spread_set);
        }
    std::vector<uint32_t> tree_off{0}, tn_parent, tn_first, tn_next, tn_nchild, tn_nodes, leaf_of((size_t)tree_sets.size() * N, 0xFFFFFFFFu);
    uint32_t max_ntn = 1, max_depth = 0;
    for (size_t t = 0; t < tree_sets.size(); ++t) {
        const auto& levels = e->spread_sets[tree_sets[t]];
        const uint32_t base = (uint32_t)tn_parent.size();
        std::vector<uint32_t> last_child;   // per tnode: last child created (for sibling chaining)
        auto new_node = [&](uint32_t parent) -> uint32_t {
            uint32_t id = (uint32_t)tn_parent.size() - base;
            tn_parent.push_back(parent);
            tn_first.push_back(0xFFFFFFFFu);
            tn_next.push_back(0xFFFFFFFFu);
            tn_nchild.push_back(0);
            tn_nodes.push_back(0);
            last_child.push_back(0xFFFFFFFFu);
            return id;
        };
        new_node(0xFFFFFFFFu);
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> child_of;   // (tnode, raw value id) -> child
        for (uint32_t n = 0; n < N; ++n) {
            const HostNode& h = e->nodes[n];
            if (!h.present) continue;
            uint32_t tn = 0;
            for (const swp_spread& lv : levels) {
                uint32_t value = 0;
                if (lv.kind == SWP_CK_NODE_LABEL) {
                    if (h.row.flags & SWP_NODE_HAS_LABELS)
                        for (const swp_kv& kv : h.labels)
                            if (kv.key == lv.key) value = kv.raw;
                } else if ((h.row.flags & SWP_NODE_HAS_DESC) && (h.row.flags & SWP_NODE_HAS_ENGINE) && (h.row.flags & SWP_NODE_HAS_ELABELS)) {
                    for (const swp_kv& kv : h.elabels)
                        if (kv.key == lv.key) value = kv.raw;
                }
                auto it = child_of.find({tn, value});
                if (it == child_of.end()) {
                    uint32_t c = new_node(tn);
                    if (last_child[tn] == 0xFFFFFFFFu) tn_first[base + tn] = c;
                    else tn_next[base + last_child[tn]] = c;
                    last_child[tn] = c;
                    tn_nchild[base + tn]++;
                    it = child_of.emplace(std::make_pair(tn, value), c).first;
                }
                tn = it->second;
            }
            leaf_of[t * N + n] = tn;
            tn_nodes[base + tn]++;   // nodes of this leaf: its heap never holds more (nodeset.go:107-120)
        }
        tree_off.push_back((uint32_t)tn_parent.size());
        max_ntn = std::max<uint32_t>(max_ntn, (uint32_t)tn_parent.size() - base);
        max_depth = std::max<uint32_t>(max_depth, (uint32_t)levels.size());
    }
    std::vector<GroupRec2> recs(n_groups);
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> scls_ids;   // static classes: the distinct (plugin, constraint, platform) class triples
    std::vector<uint32_t> scls_def;
    uint32_t off = 0, att_rows = 0;   // att_rows: tasks of the groups with cluster mounts
    size_t arena_bytes = 64;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const RTask& r = b.rt[g];
        GroupRec2& q = recs[g];
        std::memset(&q, 0, sizeof q);
        q.cpu = r.cpu; q.mem = r.mem; q.flags = r.flags; q.k = sizes[g]; q.svc = r.svc; q.out_off = off; q.pset = r.pset;
        q.cls_con = r.cls_con; q.cls_plat = r.cls_plat; q.cls_plug = r.cls_plug; q.maxrep = r.maxrep;
        q.tree = tree_local[groups[g].spread_set];
        {
            auto key = std::make_tuple(q.cls_plug, q.cls_con, q.cls_plat);
            auto it = scls_ids.find(key);
            if (it == scls_ids.end()) {
                it = scls_ids.emplace(key, (uint32_t)scls_ids.size()).first;
                scls_def.push_back(q.cls_plug); scls_def.push_back(q.cls_con); scls_def.push_back(q.cls_plat);
            }
            q.scls = it->second;
        }
        q.dep_prev = g > 0 && b.rt[g - 1].svc == r.svc;
        q.mset = groups[g].flags >> SWP_TASK_MOUNTS_SHIFT;
        if (q.mset) {   // its VolumesFilter depends on the volumes every earlier group took: nothing of it is prepared ahead
            q.att_off = att_rows;
            att_rows += sizes[g];
            if (g > 0) q.dep_prev = 1;
        }
        if (groups[g].generic_set) {
            const auto& gs = e->gen_sets[groups[g].generic_set];
            if (gs.size() > G2_MAXGEN) return e->fail(SWP_ERANGE, "group %u reserves %zu generic kinds (the engine takes %d)", g, gs.size(), G2_MAXGEN);
            for (const swp_generic& x : gs) { q.gkind[q.n_gen] = x.kind; q.gval[q.n_gen++] = (int32_t)x.value; }
        }
        // heap slots the group needs: per leaf min(k, nodes of the leaf) (nodeset.go:107-120: a leaf's heap never holds more)
        uint64_t need = 0;
        for (uint32_t i = tree_off[q.tree]; i < tree_off[q.tree + 1]; ++i)
            if (tn_nchild[i] == 0) need += std::min<uint32_t>(sizes[g], tn_nodes[i]);
        q.n_slots = (uint32_t)need;
        const size_t ab = g2_arena_bytes(q.n_slots, tree_off[q.tree + 1] - tree_off[q.tree], q.n_gen, max_depth, q.k);
        if (ab > G2_ARENA_LDS) arena_bytes = std::max(arena_bytes, ab);   // this group's working set lives in global memory
        off += sizes[g];
    }
    DevBuf d_recs, d_tree_off, d_par, d_first, d_next, d_nch, d_tnn, d_leaf, d_ff, d_key, d_dense, d_tsum, d_xroot, d_xadm, d_arena, d_lcnt, d_out, d_hist;
    if ((rc = upload(e, d_recs, recs))) return rc;
    if ((rc = upload(e, d_tree_off, tree_off))) return rc;
    if ((rc = upload(e, d_par, tn_parent))) return rc;
    if ((rc = upload(e, d_first, tn_first))) return rc;
    if ((rc = upload(e, d_next, tn_next))) return rc;
    if ((rc = upload(e, d_nch, tn_nchild))) return rc;
    if ((rc = upload(e, d_tnn, tn_nodes))) return rc;
    if ((rc = upload(e, d_leaf, leaf_of))) return rc;
    if ((rc = upload(e, d_lcnt, b.list_cnt0))) return rc;
    HIPCHECK(e, d_ff.reserve((size_t)2 * N));
    HIPCHECK(e, d_key.reserve((size_t)2 * N * 8));
    DevBuf d_ccand, d_cpos, d_cmin, d_scdef, d_slist, d_scnt;   // static class lists and a group's candidate list (swp_groups.hpp)
    const uint32_t n_scls = (uint32_t)scls_ids.size();
    if ((uint64_t)n_scls * N * 4 > (64ull << 30))
        return e->fail(SWP_ERANGE, "%u distinct (plugin, constraint, platform) classes over %u nodes: the static class lists would take more than 64 GiB", n_scls, N);
    HIPCHECK(e, d_ccand.reserve((size_t)2 * N * sizeof(G2Cand)));
    HIPCHECK(e, d_cpos.reserve((size_t)2 * N * 4));
    HIPCHECK(e, d_cmin.reserve((size_t)2 * Wn * 8));
    if ((rc = upload(e, d_scdef, scls_def))) return rc;
    HIPCHECK(e, d_slist.reserve((size_t)n_scls * N * 4));
    HIPCHECK(e, d_scnt.reserve((size_t)n_scls * 4));
    DevBuf d_sbits;
    HIPCHECK(e, d_sbits.reserve((size_t)n_scls * Wn * 8));
    HIPCHECK(e, d_dense.reserve((size_t)6 * N * 4));
    HIPCHECK(e, d_tsum.reserve((size_t)2 * max_ntn * 8));
    HIPCHECK(e, d_xroot.reserve((size_t)max_ntn * 8));
    HIPCHECK(e, d_xadm.reserve((size_t)max_ntn * 4));
    HIPCHECK(e, d_arena.reserve(arena_bytes));
    HIPCHECK(e, d_out.reserve((size_t)total * 4));
    HIPCHECK(e, d_hist.reserve((size_t)n_groups * 8 * 4));
    HIPCHECK(e, hipMemsetAsync(d_dense.p, 0, (size_t)6 * N * 4, st));   // the dense service columns start (and end) all zero
    HIPCHECK(e, hipMemsetAsync(d_hist.p, 0, (size_t)n_groups * 8 * 4, st));
    HIPCHECK(e, hipMemsetAsync(d_out.p, 0xFF, (size_t)total * 4, st));
    size_t L = b.list_node0.size();
    if (L) {
        HIPCHECK(e, hipMemcpyAsync(b.d_list_node.p, b.d_list_node0.p, L * 4, hipMemcpyDeviceToDevice, st));
        HIPCHECK(e, hipMemcpyAsync(b.d_list_svc.p, b.d_list_svc0.p, L * 4, hipMemcpyDeviceToDevice, st));
        HIPCHECK(e, hipMemcpyAsync(b.d_list_fail.p, b.d_list_fail0.p, L * 4, hipMemcpyDeviceToDevice, st));
    }
    HIPCHECK(e, hipMemsetAsync(b.d_portmap.p, 0, (size_t)std::max<uint32_t>(b.n_ports, 1) * Wn * 8, st));
    HIPCHECK(e, hipMemsetAsync(b.d_ctl.p, 0, sizeof(Ctl), st));
    if (!b.prow.empty())
        hipLaunchKernelGGL(k_scatter_bits, dim3(((uint32_t)b.prow.size() + 255) / 256), dim3(256), 0, st, (uint32_t)b.prow.size(),
                           b.d_prow.as<uint32_t>(), b.d_pnode.as<uint32_t>(), Wn, b.d_portmap.as<u64>());
    if ((rc = run_classes(e, &b))) return rc;
    Groups2Args ga{};
    ga.n_nodes = N; ga.n_words = Wn; ga.n_groups = n_groups; ga.gstride = e->ncap; ga.max_ntn = max_ntn; ga.max_depth = max_depth;
    const bool gdbg = getenv("SWP_DBG") && (atoi(getenv("SWP_DBG")) & 16);
    ga.dbg = gdbg ? (16u | ((uint32_t)atoi(getenv("SWP_DBG")) & 64u)) : 0u;
    ga.g = d_recs.as<GroupRec2>();
    ga.valid = e->d_valid.as<u64>(); ga.ready = e->d_ready.as<u64>();
    ga.con = b.d_con.as<u64>(); ga.plat = b.d_plat.as<u64>(); ga.plug = b.d_plug.as<u64>();
    ga.cpu = e->d_cpu.as<long long>(); ga.mem = e->d_mem.as<long long>(); ga.total = e->d_total.as<uint32_t>();
    ga.gcnt = e->d_gcnt.as<int32_t>();
    ga.portmap = b.d_portmap.as<u64>(); ga.pset_off = b.d_pset_off.as<uint32_t>(); ga.pset_ids = b.d_pset_ids.as<uint32_t>();
    ga.list_node = b.d_list_node.as<uint32_t>(); ga.list_svc = b.d_list_svc.as<uint32_t>(); ga.list_fail = b.d_list_fail.as<uint32_t>();
    ga.list_off = b.d_list_off.as<uint32_t>(); ga.list_cnt = d_lcnt.as<uint32_t>();
    ga.tree_off = d_tree_off.as<uint32_t>(); ga.tn_parent = d_par.as<uint32_t>(); ga.tn_first = d_first.as<uint32_t>();
    ga.tn_next = d_next.as<uint32_t>(); ga.tn_nchild = d_nch.as<uint32_t>(); ga.tn_nodes = d_tnn.as<uint32_t>(); ga.leaf_of_node = d_leaf.as<uint32_t>();
    ga.ffbuf = d_ff.as<unsigned char>(); ga.keybuf = d_key.as<u64>();
    ga.ccand = d_ccand.as<G2Cand>(); ga.cpos = d_cpos.as<uint32_t>(); ga.cmin = d_cmin.as<u64>();
    ga.n_scls = n_scls; ga.scls_def = d_scdef.as<uint32_t>(); ga.slist = d_slist.as<uint32_t>(); ga.scnt = d_scnt.as<uint32_t>(); ga.sbits = d_sbits.as<u64>();
    ga.svc_dense = d_dense.as<uint32_t>(); ga.fail_dense = d_dense.as<uint32_t>() + (size_t)2 * N; ga.lpos_dense = d_dense.as<uint32_t>() + (size_t)4 * N;
    ga.tsumbuf = d_tsum.as<long long>(); ga.xroot = d_xroot.as<u64>(); ga.xadm = d_xadm.as<int32_t>(); ga.arena = d_arena.as<unsigned char>();
    ga.out_node = d_out.as<int32_t>(); ga.hist = d_hist.as<uint32_t>(); ga.ctl = b.d_ctl.as<Ctl>();
    hipEvent_t gev0 = nullptr, gev1 = nullptr;
    if (gdbg) { (void)hipEventCreate(&gev0); (void)hipEventCreate(&gev1); (void)hipEventRecord(gev0, st); }
    DevBuf d_gatt;
    if (att_rows) {
        if ((rc = flush_volumes(e))) return rc;
        HIPCHECK(e, d_gatt.reserve((size_t)att_rows * SWP_MAX_MOUNTS * 4));
        HIPCHECK(e, hipMemsetAsync(d_gatt.p, 0xFF, (size_t)att_rows * SWP_MAX_MOUNTS * 4, st));
        ga.vol = vol_view(e);
        ga.att = d_gatt.as<uint32_t>();
    }
    sp.next("groups: k_groups2 + download");
    HIPCHECK(e, launch_groups2(ga, st, e->device));
    if (gdbg) {
        (void)hipEventRecord(gev1, st);
        (void)hipEventSynchronize(gev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, gev0, gev1);
        Ctl c2{};
        (void)hipMemcpy(&c2, b.d_ctl.p, sizeof c2, hipMemcpyDeviceToHost);
        fprintf(stderr, "[swp] k_groups2 %.3f ms for %u groups | shader cycles: wait-prep %llu reset %llu admit %llu load %llu walk %llu explain %llu writeback %llu patch %llu | in walk: ordered %llu fill %llu | admission: words %llu candidates %llu heap ops %llu replay cycles %llu parallel pushes %llu load-wait cycles %llu\n", ms, n_groups,
                c2.cyc[0], c2.cyc[1], c2.cyc[2], c2.cyc[3], c2.cyc[4], c2.cyc[5], c2.cyc[6], c2.cyc[7], c2.m_cyc[0], c2.m_cyc[1], c2.l_cyc[0], c2.l_cyc[1], c2.l_cyc[2], c2.l_cyc[3], c2.l_cyc[4], c2.l_cyc[5]);
        fprintf(stderr, "[swp] k_groups2 admission, cycles by part: records %llu, whole chunks while filling %llu, flat counting %llu, flushes %llu, staging %llu, per-chunk pass %llu, replay + pipelined %llu, minima scan %llu\n",
                c2.wave_cyc[0], c2.wave_cyc[1], c2.wave_cyc[2], c2.wave_cyc[3], c2.wave_cyc[4], c2.wave_cyc[5], c2.wave_cyc[6], c2.wave_cyc[7]);
        fprintf(stderr, "[swp] k_groups2 sort section, cycles: equal-keys check %llu, rotation / lane 0's pops %llu, residual loads %llu\n", c2.wave_cyc[8], c2.wave_cyc[9], c2.wave_cyc[10]);
        fprintf(stderr, "[swp] k_groups2 batches of candidate records: %llu, %llu cycles from the first load to the last answer\n", c2.wave_cyc[14], c2.wave_cyc[13]);
        (void)hipEventDestroy(gev0); (void)hipEventDestroy(gev1);
    }
    Ctl ctl{};
    HIPCHECK(e, hipMemcpyAsync(&ctl, b.d_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, st));
    HIPCHECK(e, hipMemcpyAsync(out_node, d_out.p, (size_t)total * 4, hipMemcpyDeviceToHost, st));
    if (out_fail_hist) HIPCHECK(e, hipMemcpyAsync(out_fail_hist, d_hist.p, (size_t)n_groups * 8 * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(e, hipStreamSynchronize(st));
    if (ctl.error != ERR_NONE) {
        e->dev_dynamic_dirty = true;   // device rows may be half-updated: the host mirror (untouched) is re-uploaded
        e->vol_dyn_dirty = true;
        if (ctl.error == ERR_GROUP_HANG) return e->fail(SWP_EHIP, "the group kernel's waves lost each other (a wait exceeded its bound): nothing was applied");
        return e->fail(SWP_ERANGE, "a node's key left its range (>= 256 recent failures or >= 2^24 tasks of one service on a node)");
    }
    if (out_att) std::fill(out_att, out_att + (size_t)total * SWP_MAX_MOUNTS, SWP_NO_VOLUME);
    if (att_rows) {   // the attachments of the groups with cluster mounts, and the volumes' usage as the call left it
        std::vector<uint32_t> h((size_t)att_rows * SWP_MAX_MOUNTS);
        std::vector<swp_volume_usage> dyn(e->volumes.size());
        HIPCHECK(e, hipMemcpyAsync(h.data(), d_gatt.p, h.size() * 4, hipMemcpyDeviceToHost, st));
        if (!dyn.empty()) HIPCHECK(e, hipMemcpyAsync(dyn.data(), e->d_vdyn.p, dyn.size() * sizeof(swp_volume_usage), hipMemcpyDeviceToHost, st));
        HIPCHECK(e, hipStreamSynchronize(st));
        for (size_t v = 0; v < dyn.size(); ++v)
            if (e->volumes[v].present) e->volumes[v].use = dyn[v];
        if (out_att)
            for (uint32_t g = 0; g < n_groups; ++g)
                if (recs[g].mset)
                    std::memcpy(out_att + (size_t)recs[g].out_off * SWP_MAX_MOUNTS, h.data() + (size_t)recs[g].att_off * SWP_MAX_MOUNTS, (size_t)sizes[g] * SWP_MAX_MOUNTS * 4);
    }
    sp.next("groups: placements into the node mirror");
    return groups_apply_to_host(e, groups, sizes, n_groups, total, out_node);
}

// the host mirror follows the device: NodeInfo.addTask for every placement of the call
static int groups_apply_to_host(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, uint64_t total, const int32_t* out_node) {
    uint64_t placed = 0;
    uint32_t off = 0;
    std::vector<BulkItem> bulk;   // (a group's placements are k tasks of one service: the bulk path's merge per service is one per group)
    bulk.reserve(total);
    for (uint32_t g = 0; g < n_groups; ++g) {
        const swp_task_desc& d = groups[g];
        const int32_t* on = out_node + off;
        const uint32_t k = sizes[g];
        off += k;
        const bool plain = !d.port_set && !d.generic_set && !(d.flags & 0x2u);
        for (uint32_t i = 0; i < k; ++i) {
            if (on[i] < 0) continue;
            if ((uint32_t)on[i] >= e->nodes.size() || !e->nodes[on[i]].present) return e->fail(SWP_EHIP, "device returned an invalid node index %d", on[i]);
            if (plain) bulk.push_back(BulkItem{(uint32_t)on[i], d.service, d.cpu, d.mem});
            else host_apply_placement(e, (uint32_t)on[i], d.service, d.cpu, d.mem, d.port_set, !(d.flags & 0x2u), true, d.generic_set);
            ++placed;
        }
    }
    {
        BulkScratch scr;
        host_apply_bulk(e, bulk, true, scr);
    }
    e->stats.batches++;
    e->stats.tasks += total;
    e->stats.placed += placed;
    e->stats.infeasible += total - placed;
    e->stats.pair_evals += (uint64_t)n_groups * e->n_present;
    return SWP_OK;
}

// [T][8] Explain counters to the caller's buffer: zeros but for the unplaceable tasks, whose rows come over as one gathered block
static int download_hist(swp_engine* e, swp_batch* b, uint32_t* out_fail_hist) {
    const uint32_t T = b->T, n = b->x_ninf;
    std::memset(out_fail_hist, 0, (size_t)T * SWP_NFILTERS * 4);
    if (!n) return SWP_OK;
    HIPCHECK(e, b->d_xrows.reserve((size_t)n * 32));
    HIPCHECK(e, b->hx_rows.reserve((size_t)n * 32));
    hipLaunchKernelGGL(k_gather_rows, dim3((n * 8u + 255u) / 256u), dim3(256), 0, e->stream, b->d_inf_task.as<uint32_t>(), b->d_hist.as<uint32_t>(), b->d_xrows.as<uint32_t>(), n);
    HIPCHECK(e, hipMemcpyAsync(b->hx_rows.p, b->d_xrows.p, (size_t)n * 32, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    const uint32_t* tasks = static_cast<const uint32_t*>(b->hx_in.p);   // the unplaceable list as run_explain fetched it
    const uint32_t* rows = static_cast<const uint32_t*>(b->hx_rows.p);
    for (uint32_t q = 0; q < n; ++q) std::memcpy(out_fail_hist + (size_t)tasks[q] * SWP_NFILTERS, rows + (size_t)q * 8, 32);
    return SWP_OK;
}

int swp_batch_prepare(swp_engine* e, const swp_task_desc* tasks, uint32_t n_tasks, swp_batch** out) {
    if (e && e->set) return ss::batch_prepare(e, tasks, n_tasks, nullptr, 0, false, out);
    if (!e || !out || (!tasks && n_tasks)) return SWP_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(e->device);
    int rc = flush_nodes(e);
    if (rc) return rc;
    auto b = std::make_unique<swp_batch>();
    const bool dbg = getenv("SWP_DEBUG_PREPARE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = build_batch(e, tasks, n_tasks, b.get()))) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    if (e->dev_static_dirty && (rc = flush_nodes(e))) return rc;
    if (n_tasks && e->n_nodes && (rc = upload_batch(e, b.get()))) return rc;
    if (dbg) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[swp] swp_batch_prepare %u tasks: build %.2f ms, upload %.2f ms\n", n_tasks, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    b->n_nodes_prepared = e->n_nodes;
    *out = b.release();
    return SWP_OK;
}

int swp_batch_prepare_templates(swp_engine* e, const swp_task_desc* templates, uint32_t n_templates, const uint32_t* template_of_task, uint32_t n_tasks, swp_batch** out) {
    if (e && e->set) return ss::batch_prepare(e, templates, n_tasks, template_of_task, n_templates, true, out);
    if (!e || !out || (!templates && n_templates) || (!template_of_task && n_tasks) || (n_tasks && !n_templates)) return SWP_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(e->device);
    HostSpan sp("prepare_templates: flush_nodes");
    int rc = flush_nodes(e);
    if (rc) return rc;
    sp.next("prepare_templates: build_batch");
    auto b = std::make_unique<swp_batch>();
    if ((rc = build_batch(e, templates, n_tasks, b.get(), nullptr, template_of_task, n_templates))) return rc;
    if (e->dev_static_dirty && (rc = flush_nodes(e))) return rc;
    sp.next("prepare_templates: upload_batch");
    if (n_tasks && e->n_nodes && (rc = upload_batch(e, b.get()))) return rc;
    b->n_nodes_prepared = e->n_nodes;
    *out = b.release();
    return SWP_OK;
}

int swp_batch_run(swp_engine* e, swp_batch* b) {
    if (e && e->set) return ss::batch_run(e, b);
    if (!e || !b) return SWP_EINVAL;
    (void)hipSetDevice(e->device);
    return batch_run(e, b);
}

// the attachments of the batch's tasks with cluster mounts; with `fold` also the usage numbers as the batch left them (into the host mirror)
static int download_volumes(swp_engine* e, swp_batch* b, bool fold) {
    if (b->csi_set.empty()) return SWP_OK;
    b->h_att.assign(b->csi_set.size() * (size_t)SWP_MAX_MOUNTS, SWP_NO_VOLUME);
    HIPCHECK(e, hipMemcpyAsync(b->h_att.data(), b->d_att.p, b->h_att.size() * 4, hipMemcpyDeviceToHost, e->stream));
    std::vector<swp_volume_usage> dyn(e->volumes.size());
    if (fold && !dyn.empty()) HIPCHECK(e, hipMemcpyAsync(dyn.data(), e->d_vdyn.p, dyn.size() * sizeof(swp_volume_usage), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    if (fold)
        for (size_t v = 0; v < dyn.size(); ++v)
            if (e->volumes[v].present) e->volumes[v].use = dyn[v];
    return SWP_OK;
}

int swp_batch_attachments(swp_engine* e, swp_batch* b, const uint32_t* tasks, uint32_t n, uint32_t* out) {
    if (e && e->set) return ss::batch_attachments(e, b, tasks, n, out);
    if (!e || !b || (!tasks && n) || (!out && n)) return SWP_EINVAL;
    for (uint32_t i = 0; i < n; ++i) {
        if (tasks[i] >= b->T) return e->fail(SWP_EINVAL, "task %u is not of this batch", tasks[i]);
        const uint32_t ck = b->csi_of.empty() ? 0xFFFFFFFFu : b->csi_of[tasks[i]];
        for (uint32_t m = 0; m < SWP_MAX_MOUNTS; ++m)
            out[(size_t)i * SWP_MAX_MOUNTS + m] = (ck == 0xFFFFFFFFu || b->h_att.empty()) ? SWP_NO_VOLUME : b->h_att[(size_t)ck * SWP_MAX_MOUNTS + m];
    }
    return SWP_OK;
}

int swp_batch_fetch(swp_engine* e, swp_batch* b, int32_t* out_node, uint32_t* out_fail_hist) {
    if (e && e->set) return ss::batch_collect(e, b, out_node, out_fail_hist, true);
    if (!e || !b || (!out_node && b->T)) return SWP_EINVAL;
    if (!b->ran) return e->fail(SWP_EINVAL, "swp_batch_fetch before swp_batch_run");
    (void)hipSetDevice(e->device);
    const uint32_t T = b->T;
    if (T == 0) return SWP_OK;
    if (e->n_nodes == 0) {
        // nodeSet is empty: every task is "no suitable node" with an empty explanation
        for (uint32_t i = 0; i < T; ++i) out_node[i] = -1;
        if (out_fail_hist) std::memset(out_fail_hist, 0, (size_t)T * SWP_NFILTERS * 4);
        e->stats.batches++;
        e->stats.tasks += T;
        e->stats.infeasible += T;
        b->ran = false;
        return SWP_OK;
    }
    HostSpan sp("fetch: D2H + wait");
    HIPCHECK(e, hipMemcpyAsync(out_node, b->d_out.p, (size_t)T * 4, hipMemcpyDeviceToHost, e->stream));
    if (out_fail_hist) {
        if (int rch = download_hist(e, b, out_fail_hist)) return rch;
    }
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    if (int rcv = download_volumes(e, b, true)) return rcv;
    sp.next("fetch: placements into the node mirror");
    uint64_t placed = 0;
    {
        std::vector<BulkItem> bulk;
        bulk.reserve(T);
        for (uint32_t i = 0; i < T; ++i) {
            int32_t n = out_node[i];
            if (n < 0) continue;
            if ((uint32_t)n >= e->nodes.size() || !e->nodes[n].present) return e->fail(SWP_EHIP, "device returned an invalid node index %d for task %u", n, i);
            const swp_task_desc& d = b->desc(i);
            if (!d.port_set && !d.generic_set && !(d.flags & 0x2u)) bulk.push_back(BulkItem{(uint32_t)n, d.service, d.cpu, d.mem});
            else host_apply_placement(e, (uint32_t)n, d.service, d.cpu, d.mem, d.port_set, !(d.flags & 0x2u), true, d.generic_set);
            ++placed;
        }
        BulkScratch scr;
        host_apply_bulk(e, bulk, true, scr);
    }
    e->stats.batches++;
    e->stats.tasks += T;
    e->stats.placed += placed;
    e->stats.infeasible += T - placed;
    e->stats.pair_evals += (uint64_t)T * e->n_present;
    b->ran = false;
    return SWP_OK;
}

int swp_batch_results(swp_engine* e, swp_batch* b, int32_t* out_node, uint32_t* out_fail_hist) {
    if (e && e->set) return ss::batch_collect(e, b, out_node, out_fail_hist, false);
    if (!e || !b || (!out_node && b->T)) return SWP_EINVAL;
    if (!b->ran) return e->fail(SWP_EINVAL, "swp_batch_results before swp_batch_run");
    (void)hipSetDevice(e->device);
    if (b->T == 0 || e->n_nodes == 0) return SWP_OK;
    HIPCHECK(e, hipMemcpyAsync(out_node, b->d_out.p, (size_t)b->T * 4, hipMemcpyDeviceToHost, e->stream));
    if (out_fail_hist) {
        if (int rch = download_hist(e, b, out_fail_hist)) return rch;
    }
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    if (int rcv = download_volumes(e, b, false)) return rcv;
    if (!b->csi_set.empty()) e->vol_dyn_dirty = true;   // (the device's usage numbers moved on, the host's did not: the next flush restores them)
    return SWP_OK;
}

void swp_batch_free(swp_engine* e, swp_batch* b) {
    if ((e && e->set) || (b && b->is_set)) { ss::batch_free(e, b); return; }
    HostSpan sp("batch_free");
    if (e) {
        (void)hipSetDevice(e->device);
        if (e->stream) (void)hipStreamSynchronize(e->stream);
    }
    delete b;
}

// ------------------------------------------------------------------------------------------------------------------
// node-range shards (include/swp.h): propose / merge / commit over a block of tasks
int swp_shard_begin(swp_engine* e, swp_batch* b) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set is not a shard: swp_shard_* take the engines of the ranges");
    if (!e || !b) return SWP_EINVAL;
    (void)hipSetDevice(e->device);
    if (e->n_nodes != b->n_nodes_prepared)
        return e->fail(SWP_EINVAL, "the nodeSet grew from %u to %u node slots since swp_batch_prepare: prepare the batch again", b->n_nodes_prepared, e->n_nodes);
    if (b->has_generic) return e->fail(SWP_EUNSUPPORTED, "generic reservations are not part of the HOST-merged shard protocol (swp_shard_run / swp_shard_run_rank carry them)");
    if (!b->csi_set.empty()) return e->fail(SWP_EUNSUPPORTED, "tasks with cluster mounts are not part of the HOST-merged shard protocol (swp_shard_run / swp_shard_run_rank carry them)");
    b->shard_open = true;
    b->shard_ncommit = b->shard_ninf = 0;
    e->stats.ms_propose = e->stats.ms_apply = 0.f;
    e->stats.propose_launches = e->stats.propose_tasks = 0;
    b->shard_out.assign(b->T, -1);
    if (e->n_nodes == 0 || b->T == 0) return SWP_OK;
    int rc = batch_begin(e, b);
    if (rc) return rc;
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return SWP_OK;
}

int swp_shard_propose(swp_engine* e, swp_batch* b, uint32_t j0, uint32_t count, swp_proposal* out) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set is not a shard: swp_shard_* take the engines of the ranges");
    if (!e || !b || (!out && count)) return SWP_EINVAL;
    if (!b->shard_open) return e->fail(SWP_EINVAL, "swp_shard_propose before swp_shard_begin");
    if (j0 > b->T || count > b->T - j0) return e->fail(SWP_ERANGE, "tasks [%u, %u) are outside the batch of %u", j0, j0 + count, b->T);
    static_assert(sizeof(swp_proposal) == sizeof(Proposal), "swp_proposal layout");
    if (count == 0) return SWP_OK;
    if (e->n_nodes == 0) {   // an empty shard has nothing to offer
        for (uint32_t i = 0; i < count; ++i) {
            std::memset(&out[i], 0, sizeof out[i]);
            out[i].level = 0xFFFFFFFFu;
            out[i].exc_hi = out[i].exc_lo = ~0ull;
        }
        return SWP_OK;
    }
    (void)hipSetDevice(e->device);
    const uint32_t N = e->n_nodes, Wn = n_words_of(N);
    HIPCHECK(e, b->d_prop.reserve((size_t)count * sizeof(Proposal)));
    ProposeArgs a{};
    a.n_nodes = N;
    a.n_words = Wn;
    a.j0 = j0;
    a.count = count;
    a.xs = Wn;
    a.cpu = e->d_cpu.as<long long>();
    a.mem = e->d_mem.as<long long>();
    a.total = e->d_total.as<uint32_t>();
    a.rt = b->d_rt.as<RTask>();
    a.sc = b->d_sc.as<u64>();
    a.X = b->d_X.as<u64>();
    a.portmap = b->d_portmap.as<u64>();
    a.pset_off = b->d_pset_off.as<uint32_t>();
    a.pset_ids = b->d_pset_ids.as<uint32_t>();
    a.list_node = b->d_list_node.as<uint32_t>();
    a.list_svc = b->d_list_svc.as<uint32_t>();
    a.list_fail = b->d_list_fail.as<uint32_t>();
    a.list_off = b->d_list_off.as<uint32_t>();
    a.out = b->d_prop.as<Proposal>();
    const bool prof = (e->cfg.flags & SWP_CFG_PROFILE) != 0;
    if (prof) HIPCHECK(e, hipEventRecord(e->ev[0], e->stream));
    hipError_t r = launch_propose(a, e->stream);
    if (r != hipSuccess) return e->fail(SWP_EHIP, "k_propose launch: %s", hipGetErrorString(r));
    if (prof) HIPCHECK(e, hipEventRecord(e->ev[1], e->stream));
    HIPCHECK(e, hipMemcpyAsync(out, b->d_prop.p, (size_t)count * sizeof(Proposal), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    if (prof) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]);
        e->stats.ms_propose += ms;
        if (b->shard_apply_timed) {   // the commit that ran in front of this propose on the same stream
            (void)hipEventElapsedTime(&ms, e->ev[2], e->ev[3]);
            e->stats.ms_apply += ms;
            b->shard_apply_timed = false;
        }
    }
    e->stats.propose_launches += 1;
    e->stats.propose_tasks += count;
    return SWP_OK;
}

int swp_shard_merge(const swp_proposal* const* proposals, const uint32_t* shard_first_node, uint32_t n_shards, uint32_t count, swp_shard_pick* picks,
                    uint32_t* accepted) {
    if (!proposals || !shard_first_node || !n_shards || (!picks && count) || !accepted) return SWP_EINVAL;
    *accepted = 0;
    std::unordered_map<uint64_t, uint64_t> taken;   // (shard << 32 | local word) -> bits picked earlier in this block
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t lmin = 0xFFFFFFFFu;
        for (uint32_t g = 0; g < n_shards; ++g) lmin = std::min(lmin, proposals[g][i].level);
        swp_shard_pick pk{-1, 0, 0xFFFFFFFFu, 0};
        if (lmin == 0xFFFFFFFFu) {
            // no plain node anywhere — and there will be none later in the batch (feasibility only shrinks). The exception lists
            // decide; their order moves with every placement of the service, so only the block's first task may use them.
            uint64_t bhi = ~0ull, btot = 0, bnode = 0;
            int32_t bg = -1;
            uint32_t bent = 0, blocal = 0;
            for (uint32_t g = 0; g < n_shards; ++g) {
                const swp_proposal& p = proposals[g][i];
                if (p.exc_hi == ~0ull) continue;
                const uint64_t tot = p.exc_lo >> 32, node = (uint64_t)shard_first_node[g] + (uint32_t)p.exc_lo;
                if (bg < 0 || p.exc_hi < bhi || (p.exc_hi == bhi && (tot < btot || (tot == btot && node < bnode)))) {
                    bhi = p.exc_hi;
                    btot = tot;
                    bnode = node;
                    bg = (int32_t)g;
                    bent = p.exc_entry;
                    blocal = (uint32_t)p.exc_lo;
                }
            }
            if (bg >= 0) {
                if (i != 0) break;   // cut in front of it
                pk.shard = bg;
                pk.node = blocal;
                pk.entry = bent;
                picks[i] = pk;
                *accepted = 1;
                break;   // and right behind it
            }
            picks[i] = pk;   // no suitable node: final whatever the earlier tasks of the block did
            *accepted = i + 1;
            continue;
        }
        bool decided = false, cut = false;
        for (uint32_t g = 0; g < n_shards && !decided && !cut; ++g) {
            const swp_proposal& p = proposals[g][i];
            if (p.level != lmin) continue;
            const uint32_t nc = p.n_cand & 0x7FFFFFFFu;
            for (uint32_t c = 0; c < nc; ++c) {
                auto it = taken.find(((uint64_t)g << 32) | p.word[c]);
                const uint64_t avail = p.bits[c] & ~(it == taken.end() ? 0ull : it->second);
                if (avail) {
                    pk.shard = (int32_t)g;
                    pk.node = p.word[c] * 64u + (uint32_t)__builtin_ctzll(avail);
                    decided = true;
                    break;
                }
            }
            if (!decided && (p.n_cand & 0x80000000u)) cut = true;   // this shard has more nodes of the level that were not listed
        }
        if (!decided) break;   // every listed node is taken: propose again against the new state
        taken[((uint64_t)(uint32_t)pk.shard << 32) | (pk.node >> 6)] |= 1ull << (pk.node & 63);
        picks[i] = pk;
        *accepted = i + 1;
        if (proposals[0][i].flags & 1u) break;   // an uncounted task: its node did not move up a level, the later lists are stale about it
    }
    return SWP_OK;
}

int swp_shard_commit(swp_engine* e, swp_batch* b, uint32_t j0, const swp_shard_pick* picks, uint32_t accepted) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set is not a shard: swp_shard_* take the engines of the ranges");
    if (!e || !b || (!picks && accepted)) return SWP_EINVAL;
    if (!b->shard_open) return e->fail(SWP_EINVAL, "swp_shard_commit before swp_shard_begin");
    if (j0 > b->T || accepted > b->T - j0) return e->fail(SWP_ERANGE, "tasks [%u, %u) are outside the batch of %u", j0, j0 + accepted, b->T);
    (void)hipSetDevice(e->device);
    HIPCHECK(e, hipStreamSynchronize(e->stream));   // the previous commit's copies have left the staging vectors
    std::vector<ShardPickDev>& mine = b->shard_mine;
    std::vector<ShardInfDev>& infs = b->shard_infs;
    mine.clear();
    infs.clear();
    const uint32_t inf_base = b->shard_ninf;
    for (uint32_t i = 0; i < accepted; ++i) {
        const swp_shard_pick& p = picks[i];
        if (p.shard < 0) {
            infs.push_back(ShardInfDev{j0 + i, b->shard_ncommit});
            b->shard_ninf++;
            continue;
        }
        if ((uint32_t)p.shard == e->cfg.shard_rank) {
            if (p.node >= e->n_nodes || !e->nodes[p.node].present) return e->fail(SWP_EINVAL, "pick of task %u names node %u, which this shard does not hold", j0 + i, p.node);
            mine.push_back(ShardPickDev{j0 + i, p.node, p.entry, b->shard_ncommit});
            b->shard_out[j0 + i] = (int32_t)p.node;
        }
        b->shard_ncommit++;
    }
    if (e->n_nodes == 0 || (mine.empty() && infs.empty())) return SWP_OK;
    int rc;
    if ((rc = upload(e, b->d_picks, mine))) return rc;
    if ((rc = upload(e, b->d_infs, infs))) return rc;
    const uint32_t Wn = n_words_of(e->n_nodes);
    ShardApplyArgs a{};
    a.n_picks = (uint32_t)mine.size();
    a.n_inf = (uint32_t)infs.size();
    a.inf_base = inf_base;
    a.n_words = Wn;
    a.xs = Wn;
    a.picks = b->d_picks.as<ShardPickDev>();
    a.infs = b->d_infs.as<ShardInfDev>();
    a.rt = b->d_rt.as<RTask>();
    a.cpu = e->d_cpu.as<long long>();
    a.mem = e->d_mem.as<long long>();
    a.total = e->d_total.as<uint32_t>();
    a.X = b->d_X.as<u64>();
    a.list_node = b->d_list_node.as<uint32_t>();
    a.list_svc = b->d_list_svc.as<uint32_t>();
    a.list_fail = b->d_list_fail.as<uint32_t>();
    a.portmap = b->d_portmap.as<u64>();
    a.pset_off = b->d_pset_off.as<uint32_t>();
    a.pset_ids = b->d_pset_ids.as<uint32_t>();
    a.out_node = b->d_out.as<int32_t>();
    a.log_node = b->d_log_node.as<uint32_t>();
    a.log_task = b->d_log_task.as<uint32_t>();
    a.log_prev = b->d_log_prev.as<int32_t>();
    a.last = b->d_last.as<int32_t>();
    a.inf_task = b->d_inf_task.as<uint32_t>();
    a.inf_pos = b->d_inf_pos.as<uint32_t>();
    const bool prof = (e->cfg.flags & SWP_CFG_PROFILE) != 0;
    if (prof) HIPCHECK(e, hipEventRecord(e->ev[2], e->stream));
    hipError_t r = launch_shard_apply(a, e->stream);
    if (r != hipSuccess) return e->fail(SWP_EHIP, "k_shard_apply launch: %s", hipGetErrorString(r));
    if (prof) {
        HIPCHECK(e, hipEventRecord(e->ev[3], e->stream));
        b->shard_apply_timed = true;
    }
    return SWP_OK;   // not waited for: the next propose (same stream) is ordered behind it
}

int swp_shard_end(swp_engine* e, swp_batch* b, int32_t* out_node_local, uint32_t* out_fail_hist) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set is not a shard: swp_shard_* take the engines of the ranges");
    if (!e || !b || (!out_node_local && b->T)) return SWP_EINVAL;
    if (!b->shard_open) return e->fail(SWP_EINVAL, "swp_shard_end before swp_shard_begin");
    (void)hipSetDevice(e->device);
    b->shard_open = false;
    const uint32_t T = b->T;
    if (out_fail_hist) std::memset(out_fail_hist, 0, (size_t)T * SWP_NFILTERS * 4);
    if (T == 0) return SWP_OK;
    if (e->n_nodes && b->shard_ninf) {
        int rc = run_explain(e, b, b->shard_ninf);
        if (rc) return rc;
        if (out_fail_hist) HIPCHECK(e, hipMemcpyAsync(out_fail_hist, b->d_hist.p, (size_t)T * 8 * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHECK(e, hipStreamSynchronize(e->stream));
    }
    uint64_t placed = 0;
    for (uint32_t i = 0; i < T; ++i) {
        const int32_t n = b->shard_out[i];
        out_node_local[i] = n;
        if (n < 0) continue;
        const swp_task_desc& d = b->desc(i);
        host_apply_placement(e, (uint32_t)n, d.service, d.cpu, d.mem, d.port_set, !(d.flags & 0x2u), true);
        ++placed;
    }
    e->stats.batches++;
    e->stats.tasks += T;
    e->stats.placed += placed;
    e->stats.pair_evals += (uint64_t)T * e->n_present;
    return SWP_OK;
}

// The block follows the pace (as in the single engine's rounds): rounds that are cut after a few dozen tasks — re-placements that all aim at
// the few emptied nodes — need not propose and stage hundreds of lists on every shard; rounds that fill their block get the next size up.
// `recent`: tasks decided per round in the last stretch. The argument records carry the block; the proposals' buffer and the tails stay
// where the LARGEST block put them. Every rank derives the same size from the same agreed numbers.
static uint32_t r7_next_block(uint32_t cur, uint32_t largest, double recent) {
    if (recent > 0.4 * cur) return std::min<uint32_t>(largest, cur * 2u);
    return std::min<uint32_t>(largest, std::max<uint32_t>(128u, ((uint32_t)(2.0 * recent) + 63u) / 64u * 64u));
}

int swp_shard_run(swp_engine* const* engines, swp_batch* const* batches, uint32_t G, uint32_t flags, int32_t* out_shard, int32_t* out_node, uint32_t* out_fail_hist) {
    const auto t_begin = std::chrono::steady_clock::now();
    if (!engines || !batches || G == 0 || !engines[0] || !batches[0]) return SWP_EINVAL;
    swp_engine* e0 = engines[0];
    for (uint32_t g = 0; g < G; ++g)
        if (engines[g] && engines[g]->set) return e0->fail(SWP_EINVAL, "shard %u is a shard set: swp_shard_run takes the engines of the ranges", g);
    if (G > R7_MAXS) return e0->fail(SWP_ERANGE, "%u shards: the device-side rounds take at most %d", G, R7_MAXS);
    const uint32_t T = batches[0]->T;
    if ((!out_shard || !out_node) && T) return SWP_EINVAL;
    for (uint32_t g = 0; g < G; ++g) {
        if (!engines[g] || !batches[g] || batches[g]->T != T) return e0->fail(SWP_EINVAL, "shard %u: every shard's batch must hold the same %u tasks", g, T);
        if (engines[g]->n_nodes != batches[g]->n_nodes_prepared) return e0->fail(SWP_EINVAL, "shard %u: the nodeSet grew since swp_batch_prepare", g);
        for (uint32_t h = 0; h < g; ++h)
            if (engines[h] == engines[g]) return e0->fail(SWP_EINVAL, "shards %u and %u name the same engine", h, g);
    }
    if (out_fail_hist) std::memset(out_fail_hist, 0, (size_t)T * SWP_NFILTERS * 4);
    for (uint32_t i = 0; i < T; ++i) out_shard[i] = out_node[i] = -1;
    if (T == 0) return SWP_OK;
    const char* env_dbg = getenv("SWP_DBG");
    const uint32_t dbg_bits = env_dbg ? (uint32_t)atoi(env_dbg) : 0u;
    const char* env_blk = getenv("SWP_R6_BLOCK");
    uint32_t block = std::min<uint32_t>(r6_block_max(), std::max<uint32_t>(1u, env_blk ? (uint32_t)atoi(env_blk) : R6_BLOCK_DEFAULT_CAP));   // (as large as the commit kernel's LDS allows: below)
    const size_t lds_budget = 160 * 1024 - 512;
    {   // the commit kernel keeps the TK row of ALL shards' nodes next to the block's lists: large node sets get smaller blocks
        uint32_t hw_all = 0;
        bool tr = false;
        for (uint32_t g = 0; g < G; ++g) {
            hw_all += (engines[g]->n_nodes + 31) / 32;
            tr = tr || !batches[g]->classes_ok || batches[g]->n_dc + batches[g]->n_dm > 128;
        }
        const uint32_t nrr = tr ? 0u : batches[0]->n_dc + batches[0]->n_dm;
        while (block > 64 && r7_commit_lds_size(hw_all, block, nrr) > lds_budget) block = (block - 1u) / 64u * 64u;
    }
    // one failure anywhere leaves every shard's device rows possibly half-updated: the host mirrors (untouched) are uploaded again
    auto fail_all = [&](int rc) {
        for (uint32_t g = 0; g < G; ++g) engines[g]->dev_dynamic_dirty = true;
        return rc;
    };
    // per shard: state to pristine + class bitmaps (what swp_batch_run does first), the block resolver's bitmaps, its argument record
    std::vector<R6Args> ra(G);
    R7Args ma{};
    ma.n_shards = G;
    ma.block = block;
    ma.dbg = dbg_bits;
    uint32_t first = 0, hw = 0;
    const bool csi = !batches[0]->csi_set.empty();   // (the task list is the same on every shard)
    ma.use_trailers = csi ? 1u : 0u;
    const bool task_rows = [&] {
        const char* env_tr = getenv("SWP_R6_TASKROWS");
        if (env_tr) return atoi(env_tr) != 0;
        for (uint32_t g = 0; g < G; ++g)
            if (!batches[g]->classes_ok || batches[g]->n_dc + batches[g]->n_dm > 128) return true;
        return false;
    }();
    for (uint32_t g = 0; g < G; ++g) {
        swp_engine* e = engines[g];
        swp_batch* b = batches[g];
        (void)hipSetDevice(e->device);
        ma.hw_base[g] = hw;
        ma.first_node[g] = first;
        first += e->n_nodes;
        hw += (e->n_nodes + 31) / 32;
        for (uint32_t h = 0; h < g; ++h) {   // every shard reads every other shard's proposals: peer memory between their devices
            swp_engine* o = engines[h];
            if (o->device == e->device) continue;
            int ok01 = 0, ok10 = 0;
            (void)hipDeviceCanAccessPeer(&ok01, o->device, e->device);
            (void)hipDeviceCanAccessPeer(&ok10, e->device, o->device);
            if (!ok01 || !ok10) return e0->fail(SWP_EUNSUPPORTED, "devices %d and %d cannot access each other's memory", o->device, e->device);
            (void)hipSetDevice(o->device);
            hipError_t pr = hipDeviceEnablePeerAccess(e->device, 0);
            if (pr != hipSuccess && pr != hipErrorPeerAccessAlreadyEnabled) return e0->fail(SWP_EHIP, "hipDeviceEnablePeerAccess: %s", hipGetErrorString(pr));
            (void)hipSetDevice(e->device);
            pr = hipDeviceEnablePeerAccess(o->device, 0);
            if (pr != hipSuccess && pr != hipErrorPeerAccessAlreadyEnabled) return e0->fail(SWP_EHIP, "hipDeviceEnablePeerAccess: %s", hipGetErrorString(pr));
            (void)hipGetLastError();
        }
        if (e->n_nodes == 0) return e0->fail(SWP_EINVAL, "shard %u owns no node", g);
        const uint32_t Wn = n_words_of(e->n_nodes);
        if (r6_propose_lds_size(Wn) > lds_budget)
            return e0->fail(SWP_ERANGE, "shard %u: %u nodes exceed the block resolver's LDS", g, e->n_nodes);
        int rc = batch_begin(e, b);
        if (rc) return fail_all(rc);
        if ((rc = r6_args_for(e, b, block, task_rows, dbg_bits, &ra[g]))) return fail_all(rc);
        ra[g].tmpl = nullptr;   // a range sees only its own part of a level: its lists start at the level's first candidate
        R7Tail* tail = reinterpret_cast<R7Tail*>(ra[g].prop + block);   // behind the shard's proposals: the trailer slots
        ma.tail[g] = tail;
        HIPCHECK(e, hipMemsetAsync(tail, 0, sizeof(R7Tail), e->stream));
        if (csi) ra[g].trail_out = tail->slot;
        Blk6 hb{};
        hb.pos = 0;
        hb.end = T;
        HIPCHECK(e, hipMemcpyAsync(b->d_blk6.p, &hb, sizeof hb, hipMemcpyHostToDevice, e->stream));
        hipError_t r = launch_r6_build(ra[g], e->stream);
        if (r != hipSuccess) return fail_all(e0->fail(SWP_EHIP, "k_r6 build launch: %s", hipGetErrorString(r)));
        ma.prop[g] = ra[g].prop;
    }
    ma.hw_base[G] = ma.hw_total = hw;
    const uint32_t nrr_all = task_rows ? 0u : batches[0]->n_dc + batches[0]->n_dm;
    const size_t lds_commit = r7_commit_lds_size(hw, block, nrr_all);
    if (hw > 0xFFFFu) return fail_all(e0->fail(SWP_ERANGE, "%u nodes over all shards: the commit kernel's half-word indices are 16 bits (2^21 nodes)", first));
    if (lds_commit > lds_budget) return fail_all(e0->fail(SWP_ERANGE, "%u nodes over all shards with blocks of %u tasks exceed the commit kernel's LDS (SWP_R6_BLOCK)", first, block));
    // shards that live on one device are served by ONE propose and ONE apply launch per round, on the stream of the first of them
    struct Group { uint32_t g0, count, max_words; int device; hipStream_t stream; DevBuf d_args, d_m; hipEvent_t ev_prop = nullptr, ev_commit = nullptr; };
    std::vector<Group> groups;
    for (uint32_t g = 0; g < G; ++g) {
        if (groups.empty() || groups.back().device != engines[g]->device) {
            groups.emplace_back();
            groups.back().g0 = g;
            groups.back().count = 0;
            groups.back().max_words = 0;
            groups.back().device = engines[g]->device;
            groups.back().stream = engines[g]->stream;
        }
        groups.back().count += 1;
        groups.back().max_words = std::max(groups.back().max_words, n_words_of(engines[g]->n_nodes));
    }
    (void)hipSetDevice(e0->device);
    auto cleanup = [&] {
        for (Group& gr : groups) {
            if (gr.ev_prop) (void)hipEventDestroy(gr.ev_prop);
            if (gr.ev_commit) (void)hipEventDestroy(gr.ev_commit);
        }
    };
    auto die = [&](int rc) {
        for (uint32_t g = 0; g < G; ++g) {
            (void)hipSetDevice(engines[g]->device);
            (void)hipStreamSynchronize(engines[g]->stream);
        }
        cleanup();
        return fail_all(rc);
    };
    for (Group& gr : groups) {
        (void)hipSetDevice(gr.device);
        if (gr.d_args.reserve((size_t)gr.count * sizeof(R6Args)) != hipSuccess || gr.d_m.reserve(sizeof(R7Args)) != hipSuccess ||
            hipMemcpyAsync(gr.d_args.p, &ra[gr.g0], (size_t)gr.count * sizeof(R6Args), hipMemcpyHostToDevice, gr.stream) != hipSuccess ||
            hipMemcpyAsync(gr.d_m.p, &ma, sizeof ma, hipMemcpyHostToDevice, gr.stream) != hipSuccess ||
            hipEventCreateWithFlags(&gr.ev_prop, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&gr.ev_commit, hipEventDisableTiming) != hipSuccess)
            return die(e0->fail(SWP_EHIP, "setting up the shards of device %d: %s", gr.device, hipGetErrorString(hipGetLastError())));
    }
    // the per-engine streams have the batch set-up in flight: the group streams start behind it
    for (uint32_t g = 0; g < G; ++g) {
        (void)hipSetDevice(engines[g]->device);
        if (hipStreamSynchronize(engines[g]->stream) != hipSuccess) return die(e0->fail(SWP_EHIP, "shard %u set-up: %s", g, hipGetErrorString(hipGetLastError())));
    }
    const auto t_rounds0 = std::chrono::steady_clock::now();
    // rounds: enqueued blindly, the leader's header read every `chunk` rounds (a round past the end is a handful of empty launches)
    uint32_t pos = 0, chunk = std::min<uint32_t>(16u, (T + 255u) / 256u + 1u);
    uint64_t rounds = 0;
    uint32_t cur_block = block, rounds_seen = 0;
    Blk6 hb{};
    while (pos < T) {
        // One round = two launches per device: every shard proposes over its nodes; once the proposals of ALL devices are there (events
        // between the devices' streams; shards of one device share a stream) every shard folds + matches the block itself and applies the
        // picks of its own range. A device's next propose overwrites what the other devices' commit kernels read: it waits for them.
        const bool multi = groups.size() > 1;
        for (uint32_t r = 0; r < chunk; ++r) {
            for (size_t q = 0; q < groups.size(); ++q) {
                Group& gr = groups[q];
                (void)hipSetDevice(gr.device);
                hipError_t x = hipSuccess;
                if (multi && rounds + r > 0)
                    for (size_t o = 0; o < groups.size() && x == hipSuccess; ++o)
                        if (o != q) x = hipStreamWaitEvent(gr.stream, groups[o].ev_commit, 0);
                if (x == hipSuccess) x = launch_r7_propose(gr.d_args.as<R6Args>(), gr.count, cur_block, gr.max_words, task_rows, csi, gr.stream, gr.device);
                if (x == hipSuccess && multi) x = hipEventRecord(gr.ev_prop, gr.stream);
                if (x != hipSuccess) return die(e0->fail(SWP_EHIP, "propose on device %d: %s", gr.device, hipGetErrorString(x)));
            }
            for (size_t q = 0; q < groups.size(); ++q) {
                Group& gr = groups[q];
                (void)hipSetDevice(gr.device);
                hipError_t x = hipSuccess;
                for (size_t o = 0; multi && o < groups.size() && x == hipSuccess; ++o)
                    if (o != q) x = hipStreamWaitEvent(gr.stream, groups[o].ev_prop, 0);
                if (x == hipSuccess) x = launch_r7_commit(gr.d_args.as<R6Args>(), gr.count, gr.d_m.as<R7Args>(), lds_commit, csi, gr.g0, gr.stream, gr.device);
                if (x == hipSuccess && multi) x = hipEventRecord(gr.ev_commit, gr.stream);
                if (x != hipSuccess) return die(e0->fail(SWP_EHIP, "commit on device %d: %s", gr.device, hipGetErrorString(x)));
            }
        }
        rounds += chunk;
        (void)hipSetDevice(e0->device);
        if (hipMemcpyAsync(&hb, batches[0]->d_blk6.p, sizeof hb, hipMemcpyDeviceToHost, groups[0].stream) != hipSuccess || hipStreamSynchronize(groups[0].stream) != hipSuccess)
            return die(e0->fail(SWP_EHIP, "reading the leader's control block: %s", hipGetErrorString(hipGetLastError())));
        if (hb.error != ERR_NONE) return die(e0->fail(SWP_ERANGE, "per-node task-count spread exceeds the %d level planes of the block resolver", R6_NP));
        if (hb.pos <= pos) return die(e0->fail(SWP_EHIP, "sharded rounds made no progress at task %u", pos));
        const double recent = (double)(hb.pos - pos) / (double)std::max<uint32_t>(hb.rounds - rounds_seen, 1);
        rounds_seen = hb.rounds;
        pos = hb.pos;
        chunk = (uint32_t)std::min<double>(4096.0, (double)(T - pos) / std::max(1.0, recent) * 1.05 + 4.0);
        if (!env_blk && pos < T) {
            const uint32_t nb = r7_next_block(cur_block, block, recent);
            if (nb != cur_block) {
                cur_block = nb;
                for (uint32_t g = 0; g < G; ++g) ra[g].block = nb;
                for (Group& gr : groups) {   // (behind the stretch's kernels on the device's stream)
                    (void)hipSetDevice(gr.device);
                    if (hipMemcpyAsync(gr.d_args.p, &ra[gr.g0], (size_t)gr.count * sizeof(R6Args), hipMemcpyHostToDevice, gr.stream) != hipSuccess)
                        return die(e0->fail(SWP_EHIP, "device %d: %s", gr.device, hipGetErrorString(hipGetLastError())));
                }
            }
            if (cur_block < block) chunk = std::min<uint32_t>(chunk, 64u);   // (look again before long while the block is small)
        }
    }
    if (csi) {   // the volumes reserved in the last round: to the shards that did not place that task (every device's rounds are done first)
        for (Group& gr : groups) {
            (void)hipSetDevice(gr.device);
            if (hipStreamSynchronize(gr.stream) != hipSuccess) return die(e0->fail(SWP_EHIP, "device %d: %s", gr.device, hipGetErrorString(hipGetLastError())));
        }
        for (Group& gr : groups) {
            (void)hipSetDevice(gr.device);
            const hipError_t x = launch_r7_settle(gr.d_args.as<R6Args>(), gr.count, gr.d_m.as<R7Args>(), gr.g0, gr.stream);
            if (x != hipSuccess) return die(e0->fail(SWP_EHIP, "k_r7_settle on device %d: %s", gr.device, hipGetErrorString(x)));
        }
    }
    for (Group& gr : groups) {
        (void)hipSetDevice(gr.device);
        if (hipStreamSynchronize(gr.stream) != hipSuccess) return die(e0->fail(SWP_EHIP, "device %d: %s", gr.device, hipGetErrorString(hipGetLastError())));
    }
    const auto t_rounds1 = std::chrono::steady_clock::now();
    // every shard: wait, check, explain the unplaceable tasks over its own nodes, results back, host mirror
    std::vector<int32_t> local(T);
    std::vector<uint32_t> hist;
    if (out_fail_hist) hist.resize((size_t)T * SWP_NFILTERS);
    for (uint32_t g = 0; g < G; ++g) {
        swp_engine* e = engines[g];
        swp_batch* b = batches[g];
        (void)hipSetDevice(e->device);
        Ctl ctl{};
        Blk6 sb{};
        if (hipMemcpyAsync(&ctl, b->d_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipMemcpyAsync(&sb, b->d_blk6.p, sizeof sb, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
            return die(e0->fail(SWP_EHIP, "shard %u: %s", g, hipGetErrorString(hipGetLastError())));
        if (sb.error != ERR_NONE) return die(e0->fail(SWP_ERANGE, "shard %u: per-node task-count spread exceeds the %d level planes", g, R6_NP));
        if (sb.pos != T) return die(e0->fail(SWP_EHIP, "shard %u stopped at task %u of %u", g, sb.pos, T));
        if (ctl.ninf) {
            int rc = run_explain(e, b, ctl.ninf);
            if (rc) return die(rc);
        }
        if (hipMemcpyAsync(local.data(), b->d_out.p, (size_t)T * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess) return die(e0->fail(SWP_EHIP, "results of shard %u", g));
        if (out_fail_hist && ctl.ninf && hipMemcpyAsync(hist.data(), b->d_hist.p, (size_t)T * 8 * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess)
            return die(e0->fail(SWP_EHIP, "histograms of shard %u", g));
        if (hipStreamSynchronize(e->stream) != hipSuccess) return die(e0->fail(SWP_EHIP, "shard %u: %s", g, hipGetErrorString(hipGetLastError())));
        if (csi && !(flags & SWP_SHARD_NO_FOLD)) {   // (with SWP_SHARD_NO_FOLD the caller fetches them: the shard set does)
            if (int rcv = download_volumes(e, b, true)) return die(rcv);
        }
        uint64_t placed = 0;
        for (uint32_t i = 0; i < T; ++i) {
            const int32_t nloc = local[i];
            if (nloc < 0) continue;
            if ((uint32_t)nloc >= e->nodes.size() || !e->nodes[nloc].present || out_shard[i] >= 0) return die(e0->fail(SWP_EHIP, "shard %u returned an invalid placement for task %u", g, i));
            out_shard[i] = (int32_t)g;
            out_node[i] = nloc;
            const swp_task_desc& d = b->desc(i);
            if (!(flags & SWP_SHARD_NO_FOLD)) host_apply_placement(e, (uint32_t)nloc, d.service, d.cpu, d.mem, d.port_set, !(d.flags & 0x2u), true, d.generic_set);
            ++placed;
        }
        if (out_fail_hist && ctl.ninf)
            for (size_t q = 0; q < hist.size(); ++q) out_fail_hist[q] += hist[q];
        e->stats.batches++;
        e->stats.tasks += T;
        e->stats.placed += placed;
        e->stats.pair_evals += (uint64_t)T * e->n_present;
        e->stats.last_resolver = 7;   // node-range shards, rounds on the device
        e->stats.resolve_launches += (uint32_t)rounds;
    }
    if (dbg_bits & 16) {
        Blk6 hh{};
        (void)hipSetDevice(e0->device);
        (void)hipMemcpy(&hh, batches[0]->d_blk6.p, sizeof hh, hipMemcpyDeviceToHost);
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[swp] sharded rounds over %u engines: %u rounds of %u (%.1f decided each) | cut by an exhausted list %u, an exception-list task %u, an uncounted task %u | set-up %.2f ms, rounds %.2f ms, explain + results %.2f ms\n", G,
                hh.rounds, block, (double)T / std::max<uint32_t>(hh.rounds, 1), hh.cut_exhausted, hh.cut_exception, hh.cut_uncounted, ms(t_begin, t_rounds0), ms(t_rounds0, t_rounds1),
                ms(t_rounds1, std::chrono::steady_clock::now()));        const double rr_ = std::max<uint32_t>(hh.rounds, 1);
        fprintf(stderr, "[swp] k_r7_commit (shard 0) shader cycles per round: prologue + fold %.0f, matching (wave 0) %.0f (list loads %.0f, walks %.0f), apply %.0f | %.1f matcher stops at an emptied half-word per round\n",
                hh.cyc[0] * 64.0 / rr_, hh.cyc[1] * 64.0 / rr_, hh.cyc_load * 64.0 / rr_, hh.cyc_walk * 64.0 / rr_, hh.cyc[3] * 64.0 / rr_, hh.reseats / rr_);
        fprintf(stderr, "[swp] ... of the list loads: waiting for a group's lists %.0f, its head records %.0f, seating %.0f (%.1f steps of the seating loop, stops included) per round\n", hh.cyc_g[0] * 64.0 / rr_,
                hh.cyc_g[1] * 64.0 / rr_, hh.cyc_g[2] * 64.0 / rr_, hh.cyc_g[3] / rr_);
    }
    cleanup();
    return SWP_OK;
}

// ---- RCCL, loaded lazily: the four entry points the rank variant needs ---------------------------------------------------------
namespace {
struct RcclId { char internal[SWP_RCCL_ID_BYTES]; };   // ncclUniqueId, passed by value
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // an RCCL the process already holds (PyTorch ships its own) is the one to use: two copies in one process corrupt each other
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.lib) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (r.lib) {
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
            if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) r.lib = nullptr;
        }
    }
    return r.lib ? &r : nullptr;
}
}  // namespace

int swp_rccl_finalize(swp_engine* e) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set lives in one process: it has no RCCL communicator");
    if (!e) return SWP_EINVAL;
    Rccl* r = rccl();
    if (r && e->rccl_comm) {
        (void)hipSetDevice(e->device);
        (void)hipStreamSynchronize(e->stream);
        (void)r->CommDestroy(e->rccl_comm);
    }
    e->rccl_comm = nullptr;
    e->rccl_ranks = 0;
    return SWP_OK;
}

int swp_rccl_available(swp_engine* e) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set lives in one process: it has no RCCL communicator");
    if (!e) return SWP_EINVAL;
    return rccl() ? SWP_OK : e->fail(SWP_EUNSUPPORTED, "librccl.so cannot be loaded (or lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy)");
}

int swp_rccl_unique_id(swp_engine* e, uint8_t* id_out) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set lives in one process: it has no RCCL communicator");
    if (!e || !id_out) return SWP_EINVAL;
    Rccl* r = rccl();
    if (!r) return e->fail(SWP_EUNSUPPORTED, "librccl.so cannot be loaded (or lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy)");
    (void)hipSetDevice(e->device);
    const int rc = r->GetUniqueId(id_out);
    return rc == 0 ? SWP_OK : e->fail(SWP_EHIP, "ncclGetUniqueId: %s", r->GetErrorString ? r->GetErrorString(rc) : "error");
}

int swp_rccl_init(swp_engine* e, const uint8_t* id, uint32_t rank, uint32_t n_ranks) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set lives in one process: it has no RCCL communicator");
    if (!e || !id || n_ranks == 0 || rank >= n_ranks) return SWP_EINVAL;
    if (n_ranks > R7_MAXS) return e->fail(SWP_ERANGE, "%u ranks: the device-side rounds take at most %d", n_ranks, R7_MAXS);
    Rccl* r = rccl();
    if (!r) return e->fail(SWP_EUNSUPPORTED, "librccl.so cannot be loaded (or lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy)");
    (void)hipSetDevice(e->device);
    if (e->rccl_comm) (void)swp_rccl_finalize(e);
    RcclId uid;
    std::memcpy(uid.internal, id, SWP_RCCL_ID_BYTES);
    const int rc = r->CommInitRank(&e->rccl_comm, (int)n_ranks, uid, (int)rank);
    if (rc != 0) return e->fail(SWP_EHIP, "ncclCommInitRank: %s", r->GetErrorString ? r->GetErrorString(rc) : "error");
    e->rccl_rank = rank;
    e->rccl_ranks = n_ranks;
    return SWP_OK;
}

// What the ranks of a sharded run tell each other between stretches of rounds: one tiny ncclAllGather on the engine's stream. Every
// rank sees every rank's word, so they all leave the round loop in the SAME place — a rank that returned on its own would leave its
// peers inside the next collective for ever.
namespace {
struct RankStatus { int32_t code; uint32_t pos, error, rounds; };
int rank_agree(swp_engine* e, Rccl* r, hipStream_t st, DevBuf& d_stat, const RankStatus& mine, std::vector<RankStatus>& all) {
    const uint32_t G = e->rccl_ranks;
    HIPCHECK(e, d_stat.reserve((size_t)(G + 1) * sizeof(RankStatus)));
    RankStatus* dev = d_stat.as<RankStatus>();
    HIPCHECK(e, hipMemcpyAsync(dev + G, &mine, sizeof mine, hipMemcpyHostToDevice, st));
    const int nr = r->AllGather(dev + G, dev, sizeof(RankStatus), /* ncclInt8 */ 0, e->rccl_comm, st);
    if (nr != 0) return e->fail(SWP_EHIP, "ncclAllGather (status): %s", r->GetErrorString ? r->GetErrorString(nr) : "error");
    all.resize(G);
    HIPCHECK(e, hipMemcpyAsync(all.data(), dev, (size_t)G * sizeof(RankStatus), hipMemcpyDeviceToHost, st));
    HIPCHECK(e, hipStreamSynchronize(st));
    return SWP_OK;
}
// the verdict every rank derives from the same gathered words (tests/test_product_cpu.py checks the rule through swp_shard_verdict)
int rank_verdict(const RankStatus* all, uint32_t G, uint32_t* who) {
    for (uint32_t g = 0; g < G; ++g)
        if (all[g].code != SWP_OK) { *who = g; return 1; }        // a rank could not take part
    for (uint32_t g = 0; g < G; ++g)
        if (all[g].error != ERR_NONE) { *who = g; return 2; }     // a rank's kernels reported an error (level range)
    for (uint32_t g = 1; g < G; ++g)
        if (all[g].pos != all[0].pos) { *who = g; return 3; }     // the ranks no longer agree on the batch position: they diverged
    return 0;
}
}  // namespace

int swp_shard_verdict(const uint32_t* words, uint32_t n_ranks, uint32_t* who_out) {
    if (!words || !who_out || n_ranks == 0) return SWP_EINVAL;
    *who_out = 0;
    return rank_verdict(reinterpret_cast<const RankStatus*>(words), n_ranks, who_out);
}

int swp_shard_run_rank(swp_engine* e, swp_batch* b, const uint32_t* shard_nodes, uint32_t flags, int32_t* out_node_local, uint32_t* out_fail_hist) {
    if (e && e->set) return e->fail(SWP_EINVAL, "a shard set is not a rank: swp_shard_run_rank takes the engine of ONE range");
    if (!e || !b || !shard_nodes || (!out_node_local && b->T)) return SWP_EINVAL;
    if (!e->rccl_comm) return e->fail(SWP_EINVAL, "swp_shard_run_rank before swp_rccl_init");
    Rccl* r = rccl();
    const uint32_t G = e->rccl_ranks, me = e->rccl_rank, T = b->T;
    (void)hipSetDevice(e->device);
    hipStream_t st = e->stream;
    const char* env_dbg = getenv("SWP_DBG");
    const uint32_t dbg_bits = env_dbg ? (uint32_t)atoi(env_dbg) : 0u;
    const char* env_blk = getenv("SWP_R6_BLOCK");
    uint32_t block = std::min<uint32_t>(r6_block_max(), std::max<uint32_t>(1u, env_blk ? (uint32_t)atoi(env_blk) : R6_BLOCK_DEFAULT_CAP));   // (as large as the commit kernel's LDS allows: below)
    const size_t lds_budget = 160 * 1024 - 512;
    // every rank must choose the same row mode: task rows whenever any rank might (the choice only depends on the task list, which is shared)
    const char* env_tr = getenv("SWP_R6_TASKROWS");
    const bool task_rows = env_tr ? atoi(env_tr) != 0 : (!b->classes_ok || b->n_dc + b->n_dm > 128);
    {   // (and the same block: it follows from shard_nodes and the task list, which every rank holds)
        uint32_t hw_all = 0;
        for (uint32_t g = 0; g < G; ++g) hw_all += (shard_nodes[g] + 31) / 32;
        while (block > 64 && r7_commit_lds_size(hw_all, block, task_rows ? 0u : b->n_dc + b->n_dm) > lds_budget) block = (block - 1u) / 64u * 64u;
    }
    const uint32_t Wn = n_words_of(e->n_nodes);
    auto bad = [&](int rc) { e->dev_dynamic_dirty = true; (void)hipStreamSynchronize(st); return rc; };
    // ---- everything that can go wrong on THIS rank before the first round is checked here, and the outcome is exchanged: either all
    // ranks start the rounds or none does (a lone early return would leave the others inside ncclAllGather)
    R6Args ra{};
    R7Args ma{};
    DevBuf d_all, d_m, d_args, d_stat;
    const bool csi = !b->csi_set.empty();
    const size_t send = r7_send_bytes(block);   // what a rank contributes to a round's exchange: its proposals + the tail (swp_resolve7.hpp)
    size_t lds_commit = 0;
    Blk6 hb{};
    int pre = SWP_OK;
    if (e->n_nodes != b->n_nodes_prepared) pre = e->fail(SWP_EINVAL, "the nodeSet grew since swp_batch_prepare");
    else if (shard_nodes[me] != e->n_nodes) pre = e->fail(SWP_EINVAL, "rank %u holds %u node slots, shard_nodes says %u", me, e->n_nodes, shard_nodes[me]);
    else if (e->n_nodes == 0) pre = e->fail(SWP_EINVAL, "rank %u owns no node", me);
    else if (r6_propose_lds_size(Wn) > lds_budget) pre = e->fail(SWP_ERANGE, "%u nodes exceed the block resolver's LDS", e->n_nodes);
    if (out_fail_hist) std::memset(out_fail_hist, 0, (size_t)T * SWP_NFILTERS * 4);
    for (uint32_t i = 0; i < T; ++i) out_node_local[i] = -1;
    auto setup = [&]() -> int {
        int rc = batch_begin(e, b);
        if (rc) return rc;
        if ((rc = r6_args_for(e, b, block, task_rows, dbg_bits, &ra))) return rc;
        ra.tmpl = nullptr;   // (as in swp_shard_run)
        R7Tail* tail = reinterpret_cast<R7Tail*>(ra.prop + block);   // behind this rank's proposals: it travels with them
        if (csi) ra.trail_out = tail->slot;
        HIPCHECK(e, hipMemsetAsync(tail, 0, sizeof(R7Tail), st));
        hb.end = T;
        HIPCHECK(e, hipMemcpyAsync(b->d_blk6.p, &hb, sizeof hb, hipMemcpyHostToDevice, st));
        hipError_t x = launch_r6_build(ra, st);
        if (x != hipSuccess) return e->fail(SWP_EHIP, "k_r6 build launch: %s", hipGetErrorString(x));
        HIPCHECK(e, d_all.reserve((size_t)G * send));
        HIPCHECK(e, d_m.reserve(sizeof(R7Args)));
        HIPCHECK(e, d_args.reserve(sizeof(R6Args)));
        HIPCHECK(e, hipMemcpyAsync(d_args.p, &ra, sizeof ra, hipMemcpyHostToDevice, st));
        ma.n_shards = G;
        ma.block = block;
        ma.dbg = dbg_bits;
        ma.use_trailers = csi ? 1u : 0u;
        ma.check_dead = 1u;
        uint32_t first = 0, hw = 0;
        for (uint32_t g = 0; g < G; ++g) {
            ma.hw_base[g] = hw;
            ma.first_node[g] = first;
            first += shard_nodes[g];
            hw += (shard_nodes[g] + 31) / 32;
            ma.prop[g] = reinterpret_cast<const R6Prop*>(static_cast<const char*>(d_all.p) + (size_t)g * send);   // the layout ncclAllGather leaves
            ma.tail[g] = reinterpret_cast<const R7Tail*>(ma.prop[g] + block);
        }
        ma.hw_base[G] = ma.hw_total = hw;
        lds_commit = r7_commit_lds_size(hw, block, task_rows ? 0u : b->n_dc + b->n_dm);
        if (hw > 0xFFFFu) return e->fail(SWP_ERANGE, "%u nodes over all ranks: the commit kernel's half-word indices are 16 bits (2^21 nodes)", first);
        if (lds_commit > lds_budget) return e->fail(SWP_ERANGE, "%u nodes over all ranks with blocks of %u tasks exceed the commit kernel's LDS (SWP_R6_BLOCK)", first, block);
        HIPCHECK(e, hipMemcpyAsync(d_m.p, &ma, sizeof ma, hipMemcpyHostToDevice, st));
        // the build kernel's verdict (the level range of THIS rank's nodes) is part of what is exchanged
        HIPCHECK(e, hipMemcpyAsync(&hb, b->d_blk6.p, sizeof hb, hipMemcpyDeviceToHost, st));
        HIPCHECK(e, hipStreamSynchronize(st));
        return SWP_OK;
    };
    if (pre == SWP_OK && T) pre = setup();
    std::vector<RankStatus> all;
    uint32_t who = 0;
    int rc = rank_agree(e, r, st, d_stat, RankStatus{pre, 0u, pre == SWP_OK ? hb.error : 0u, 0u}, all);
    if (rc) return bad(rc);   // (the collective itself failed: nothing sensible is left to agree on)
    switch (rank_verdict(all.data(), G, &who)) {
    case 1: return bad(pre != SWP_OK ? pre : e->fail(SWP_EINVAL, "rank %u could not start the sharded run (code %d): no rank runs it", who, all[who].code));
    case 2: return bad(e->fail(SWP_ERANGE, "rank %u: per-node task-count spread exceeds the %d level planes of the block resolver", who, R6_NP));
    default: break;
    }
    if (T == 0) return SWP_OK;
    uint32_t pos = 0, chunk = std::min<uint32_t>(16u, (T + 255u) / 256u + 1u);
    uint64_t rounds = 0;
    // Inside the rounds NOTHING returns on its own: a rank whose launch / copy failed remembers the first error, marks its tail's dead
    // word, and keeps issuing the chunk's collectives (its peers are enqueued in them) — their commit kernels see the word and stand
    // still; the error travels in the status exchange behind the chunk and every rank leaves there.
    int local_rc = SWP_OK;
    auto note = [&](int code) { if (local_rc == SWP_OK) local_rc = code; };
    R7Tail* my_tail = reinterpret_cast<R7Tail*>(ra.prop + block);
    uint32_t cur_block = block, rounds_seen = 0;
    while (pos < T) {   // (every rank computes the same positions from the same gathered words, hence the same chunks: the collectives line up)
        for (uint32_t q = 0; q < chunk; ++q) {
            if (local_rc == SWP_OK) {
                const hipError_t x = launch_r7_propose(d_args.as<R6Args>(), 1, cur_block, Wn, task_rows, csi, st, e->device);
                if (x != hipSuccess) note(e->fail(SWP_EHIP, "propose: %s", hipGetErrorString(x)));
            }
            if (local_rc != SWP_OK) (void)hipMemsetAsync(&my_tail->dead, 1, sizeof(uint32_t), st);
            const int nr = r->AllGather(ra.prop, d_all.p, send, /* ncclInt8 */ 0, e->rccl_comm, st);
            if (nr != 0) note(e->fail(SWP_EHIP, "ncclAllGather: %s", r->GetErrorString ? r->GetErrorString(nr) : "error"));
            if (local_rc == SWP_OK) {
                const hipError_t x = launch_r7_commit(d_args.as<R6Args>(), 1, d_m.as<R7Args>(), lds_commit, csi, me, st, e->device);
                if (x != hipSuccess) note(e->fail(SWP_EHIP, "commit: %s", hipGetErrorString(x)));
            }
        }
        rounds += chunk;
        if (local_rc == SWP_OK && (hipMemcpyAsync(&hb, b->d_blk6.p, sizeof hb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
            note(e->fail(SWP_EHIP, "reading the control block: %s", hipGetErrorString(hipGetLastError())));
        if ((rc = rank_agree(e, r, st, d_stat, RankStatus{local_rc, hb.pos, hb.error, hb.rounds}, all))) return bad(rc);   // (the collective itself failed: nothing is left to agree on)
        switch (rank_verdict(all.data(), G, &who)) {
        case 1: return bad(local_rc != SWP_OK ? local_rc : e->fail(SWP_EHIP, "rank %u failed inside the sharded rounds (code %d): every rank stops here", who, all[who].code));
        case 2: return bad(e->fail(SWP_ERANGE, "rank %u: per-node task-count spread exceeds the %d level planes of the block resolver", who, R6_NP));
        case 3: return bad(e->fail(SWP_EHIP, "the ranks diverged: rank %u stands at task %u, rank 0 at %u", who, all[who].pos, all[0].pos));
        default: break;
        }
        if (hb.pos <= pos) return bad(e->fail(SWP_EHIP, "sharded rounds made no progress at task %u", pos));   // (the same on every rank: the positions agree)
        // (positions and round counts are the agreed ones — rank 0's words, which every rank's equal: the same pace, chunk and block everywhere)
        const double recent = (double)(all[0].pos - pos) / (double)std::max<uint32_t>(all[0].rounds - rounds_seen, 1);
        rounds_seen = all[0].rounds;
        pos = all[0].pos;
        chunk = (uint32_t)std::min<double>(4096.0, (double)(T - pos) / std::max(1.0, recent) * 1.05 + 4.0);
        if (!env_blk && pos < T) {
            const uint32_t nb = r7_next_block(cur_block, block, recent);
            if (nb != cur_block) {
                cur_block = nb;
                ra.block = nb;
                if (hipMemcpyAsync(d_args.p, &ra, sizeof ra, hipMemcpyHostToDevice, st) != hipSuccess) note(e->fail(SWP_EHIP, "argument record: %s", hipGetErrorString(hipGetLastError())));
            }
            if (cur_block < block) chunk = std::min<uint32_t>(chunk, 64u);
        }
    }
    if (csi) {   // the volumes reserved in the last round: one more exchange (every rank issues it: the batch ended for all of them together), then every rank takes them
        const int nr = r->AllGather(ra.prop, d_all.p, send, /* ncclInt8 */ 0, e->rccl_comm, st);
        if (nr != 0) return bad(e->fail(SWP_EHIP, "ncclAllGather (last trailers): %s", r->GetErrorString ? r->GetErrorString(nr) : "error"));
        const hipError_t x = launch_r7_settle(d_args.as<R6Args>(), 1, d_m.as<R7Args>(), me, st);
        if (x != hipSuccess) return bad(e->fail(SWP_EHIP, "k_r7_settle: %s", hipGetErrorString(x)));
    }
    Ctl ctl{};
    HIPCHECK(e, hipMemcpyAsync(&ctl, b->d_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, st));
    HIPCHECK(e, hipStreamSynchronize(st));
    if (ctl.ninf && (rc = run_explain(e, b, ctl.ninf))) return bad(rc);
    HIPCHECK(e, hipMemcpyAsync(out_node_local, b->d_out.p, (size_t)T * 4, hipMemcpyDeviceToHost, st));
    if (out_fail_hist && ctl.ninf) HIPCHECK(e, hipMemcpyAsync(out_fail_hist, b->d_hist.p, (size_t)T * 8 * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(e, hipStreamSynchronize(st));
    if (csi) {   // the attachments of this rank's tasks with cluster mounts; the volumes' usage as the batch left it (every rank holds the whole table)
        if (int rcv = download_volumes(e, b, !(flags & SWP_SHARD_NO_FOLD))) return bad(rcv);
        if (flags & SWP_SHARD_NO_FOLD) e->vol_dyn_dirty = true;
    }
    uint64_t placed = 0;
    for (uint32_t i = 0; i < T; ++i) {
        const int32_t n = out_node_local[i];
        if (n < 0) continue;
        if ((uint32_t)n >= e->nodes.size() || !e->nodes[n].present) return bad(e->fail(SWP_EHIP, "device returned an invalid node index %d for task %u", n, i));
        const swp_task_desc& d = b->desc(i);
        if (!(flags & SWP_SHARD_NO_FOLD)) host_apply_placement(e, (uint32_t)n, d.service, d.cpu, d.mem, d.port_set, !(d.flags & 0x2u), true, d.generic_set);
        ++placed;
    }
    e->stats.batches++;
    e->stats.tasks += T;
    e->stats.placed += placed;
    e->stats.pair_evals += (uint64_t)T * e->n_present;
    e->stats.last_resolver = 7;
    e->stats.resolve_launches += (uint32_t)rounds;
    return SWP_OK;
}

int swp_schedule_batch(swp_engine* e, const swp_task_desc* tasks, uint32_t n_tasks, int32_t* out_node, uint32_t* out_fail_hist) {
    swp_batch* b = nullptr;
    int rc = swp_batch_prepare(e, tasks, n_tasks, &b);
    if (rc) return rc;
    rc = swp_batch_run(e, b);
    if (!rc) rc = swp_batch_fetch(e, b, out_node, out_fail_hist);
    swp_batch_free(e, b);
    return rc;
}

int swp_state_save(swp_engine* e) {
    if (e && e->set) return ss::state_save(e);
    if (!e) return SWP_EINVAL;
    (void)hipSetDevice(e->device);
    int rc = flush_nodes(e);
    if (rc) return rc;
    size_t cap = e->ncap;
    HIPCHECK(e, e->d_save_cpu.reserve(cap * 8));
    HIPCHECK(e, e->d_save_mem.reserve(cap * 8));
    HIPCHECK(e, e->d_save_total.reserve(cap * 4));
    HIPCHECK(e, hipMemcpyAsync(e->d_save_cpu.p, e->d_cpu.p, cap * 8, hipMemcpyDeviceToDevice, e->stream));
    HIPCHECK(e, hipMemcpyAsync(e->d_save_mem.p, e->d_mem.p, cap * 8, hipMemcpyDeviceToDevice, e->stream));
    HIPCHECK(e, hipMemcpyAsync(e->d_save_total.p, e->d_total.p, cap * 4, hipMemcpyDeviceToDevice, e->stream));
    if (e->dev_gkinds > 1) {
        HIPCHECK(e, e->d_save_gcnt.reserve((size_t)e->dev_gkinds * cap * 4));
        HIPCHECK(e, hipMemcpyAsync(e->d_save_gcnt.p, e->d_gcnt.p, (size_t)e->dev_gkinds * cap * 4, hipMemcpyDeviceToDevice, e->stream));
    }
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    e->saved.nodes = e->nodes;
    e->saved.svc_nodes_ = e->svc_nodes;
    e->saved.port_nodes_ = e->port_nodes;
    e->saved.valid = true;
    e->host_dirty_since_save = false;
    return SWP_OK;
}

int swp_state_restore(swp_engine* e) {
    if (e && e->set) return ss::state_restore(e);
    if (!e) return SWP_EINVAL;
    if (!e->saved.valid) return e->fail(SWP_EINVAL, "swp_state_restore without swp_state_save");
    (void)hipSetDevice(e->device);
    if (e->saved.nodes.size() != e->nodes.size() || e->dev_static_dirty) return e->fail(SWP_EINVAL, "node set changed since swp_state_save");
    size_t cap = e->ncap;
    HIPCHECK(e, hipMemcpyAsync(e->d_cpu.p, e->d_save_cpu.p, cap * 8, hipMemcpyDeviceToDevice, e->stream));
    HIPCHECK(e, hipMemcpyAsync(e->d_mem.p, e->d_save_mem.p, cap * 8, hipMemcpyDeviceToDevice, e->stream));
    HIPCHECK(e, hipMemcpyAsync(e->d_total.p, e->d_save_total.p, cap * 4, hipMemcpyDeviceToDevice, e->stream));
    if (e->dev_gkinds > 1) HIPCHECK(e, hipMemcpyAsync(e->d_gcnt.p, e->d_save_gcnt.p, (size_t)e->dev_gkinds * cap * 4, hipMemcpyDeviceToDevice, e->stream));
    if (e->host_dirty_since_save) {   // the host mirror only moves in swp_batch_fetch / swp_commit / node mutators
        HIPCHECK(e, hipStreamSynchronize(e->stream));
        e->nodes = e->saved.nodes;
        e->svc_nodes = e->saved.svc_nodes_;
        e->port_nodes = e->saved.port_nodes_;
        e->host_dirty_since_save = false;
        e->dev_flags_dirty = true;   // (flag words may have moved since the save: the mirror has the saved ones again, the device follows)
    }
    e->dev_dynamic_dirty = false;
    e->dirty_rows.clear();           // (rows changed since the save: both sides hold the saved ones again)
    return SWP_OK;
}

int swp_commit(swp_engine* e, const swp_placement* p, uint32_t n, int add_or_remove) {
    if (e && e->set) return ss::commit(e, p, n, add_or_remove);
    if (!e || (!p && n)) return SWP_EINVAL;
    if (n == 0) return SWP_OK;
    (void)hipSetDevice(e->device);
    for (uint32_t i = 0; i < n; ++i) {
        if (p[i].node >= e->nodes.size() || !e->nodes[p[i].node].present) return SWP_ENOTFOUND;
        if (p[i].port_set >= e->port_sets.size()) return SWP_EINVAL;
    }
    HostSpan sp("commit: flush_nodes");
    int rc = flush_nodes(e);   // device rows must be current before the residual kernel touches them
    if (rc) return rc;
    sp.next("commit: node mirror");
    std::vector<DevPlacement> dp(n);
    std::vector<BulkItem> bulk;
    bulk.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        dp[i].node = p[i].node;
        dp[i].counted = p[i].counted;
        dp[i].cpu = p[i].cpu;
        dp[i].mem = p[i].mem;
        if (p[i].counted && !p[i].port_set) bulk.push_back(BulkItem{p[i].node, p[i].service, p[i].cpu, p[i].mem});
        else host_apply_placement(e, p[i].node, p[i].service, p[i].cpu, p[i].mem, p[i].port_set, p[i].counted != 0, add_or_remove != 0);
    }
    {
        BulkScratch scr;
        host_apply_bulk(e, bulk, add_or_remove != 0, scr);
    }
    sp.next("commit: upload + k_commit + wait");
    DevBuf d;
    HIPCHECK(e, d.reserve((size_t)n * sizeof(DevPlacement)));
    HIPCHECK(e, hipMemcpyAsync(d.p, dp.data(), (size_t)n * sizeof(DevPlacement), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_commit, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, d.as<DevPlacement>(), add_or_remove, e->d_cpu.as<long long>(),
                       e->d_mem.as<long long>(), e->d_total.as<uint32_t>());
    HIPCHECK(e, hipGetLastError());
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return SWP_OK;
}

int swp_check_node(swp_engine* e, const swp_task_desc* task, uint32_t node, int32_t* first_fail) {
    if (e && e->set) return ss::check_node(e, task, node, first_fail);
    if (!e || !task || !first_fail) return SWP_EINVAL;
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    (void)hipSetDevice(e->device);
    swp_batch* b = nullptr;
    int rc = swp_batch_prepare(e, task, 1, &b);
    if (rc) return rc;
    std::unique_ptr<swp_batch, void (*)(swp_batch*)> guard(b, [](swp_batch* x) { delete x; });
    if ((rc = run_classes(e, b))) return rc;
    CheckArgs ca{};
    ca.node = node;
    ca.n_words = n_words_of(e->n_nodes);
    ca.rt = b->rt[0];
    ca.valid = e->d_valid.as<u64>();
    ca.ready = e->d_ready.as<u64>();
    ca.con = b->d_con.as<u64>();
    ca.plat = b->d_plat.as<u64>();
    ca.plug = b->d_plug.as<u64>();
    ca.cpu = e->d_cpu.as<long long>();
    ca.mem = e->d_mem.as<long long>();
    const HostNode& h = e->nodes[node];
    ca.port_busy = 0;
    if (task->port_set)
        for (const swp_port& p : e->port_sets[task->port_set])
            if (h.ports.count(port_key(p.protocol, p.port))) ca.port_busy = 1;
    auto it = h.svc.find(task->service);
    ca.svc_count = it == h.svc.end() ? 0 : it->second;
    if (task->generic_set) {
        for (const swp_generic& g : e->gen_sets[task->generic_set]) {
            ca.gkind[ca.n_gen] = g.kind;
            ca.gval[ca.n_gen++] = (int32_t)g.value;
        }
        ca.gstride = e->ncap;
        ca.gcnt = e->d_gcnt.as<int32_t>();
    }
    DevBuf out;
    HIPCHECK(e, out.reserve(4));
    ca.out = out.as<int32_t>();
    hipLaunchKernelGGL(k_check_pair, dim3(1), dim3(64), 0, e->stream, ca);
    HIPCHECK(e, hipGetLastError());
    int32_t ff = 0;
    HIPCHECK(e, hipMemcpyAsync(&ff, out.p, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    *first_fail = ff;
    if (ff < 0 && (task->flags >> SWP_TASK_MOUNTS_SHIFT)) {   // VolumesFilter, the pipeline's last entry: any mount with a volume on the node
        uint32_t att[SWP_MAX_MOUNTS], na = 0;
        if (!e->has_volumes() || e->volumes.empty()) { *first_fail = 7; return SWP_OK; }
        DevBuf d;
        HIPCHECK(e, d.reserve((SWP_MAX_MOUNTS + 3) * 4));
        VolChooseArgs va{};
        va.vol = vol_view(e);
        va.set = task->flags >> SWP_TASK_MOUNTS_SHIFT;
        va.node = node;
        va.out = d.as<u32>();
        hipError_t r = launch_vol_choose(va, e->stream);
        if (r != hipSuccess) return e->fail(SWP_EHIP, "k_vol_choose launch: %s", hipGetErrorString(r));
        uint32_t h[SWP_MAX_MOUNTS + 3];
        HIPCHECK(e, hipMemcpyAsync(h, d.p, sizeof h, hipMemcpyDeviceToHost, e->stream));
        HIPCHECK(e, hipStreamSynchronize(e->stream));
        (void)att;
        (void)na;
        if (!h[SWP_MAX_MOUNTS + 2]) *first_fail = 7;
    }
    return SWP_OK;
}

int swp_enforce(swp_engine* e, const swp_enforce_node* nodes, uint32_t n_nodes, const swp_enforce_task* tasks, uint32_t n_tasks,
                uint8_t* out_reject) {
    if (e && e->set) return ss::enforce(e, nodes, n_nodes, tasks, n_tasks, out_reject);
    if (!e || (!nodes && n_nodes) || (!tasks && n_tasks) || (!out_reject && n_tasks)) return SWP_EINVAL;
    if (n_tasks == 0 || n_nodes == 0) {
        if (n_tasks) std::memset(out_reject, 0, n_tasks);
        return SWP_OK;
    }
    (void)hipSetDevice(e->device);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        if (nodes[i].node >= e->nodes.size() || !e->nodes[nodes[i].node].present) return e->fail(SWP_ENOTFOUND, "enforce: node %u is not in the nodeSet mirror", nodes[i].node);
        if ((uint64_t)nodes[i].first_task + nodes[i].n_tasks > n_tasks) return e->fail(SWP_EINVAL, "enforce: node %u lists tasks beyond the task array", i);
    }
    // the constraint sets in play become classes exactly as for a scheduling batch (one pseudo task per enforce task)
    uint32_t svc = 0;
    {
        static const char kDummy[] = "\0swp-enforce";
        int rc = swp_intern(e, SWP_SPACE_SERVICE, kDummy, sizeof kDummy - 1, &svc);
        if (rc) return rc;
    }
    std::vector<swp_task_desc> descs(n_tasks);
    std::memset(descs.data(), 0, descs.size() * sizeof(swp_task_desc));
    for (uint32_t i = 0; i < n_tasks; ++i) {
        descs[i].service = svc;
        descs[i].constraint_set = tasks[i].constraint_set;
    }
    int rc = flush_nodes(e);
    if (rc) return rc;
    swp_batch b;
    if ((rc = build_batch(e, descs.data(), n_tasks, &b, nullptr))) return rc;
    if ((rc = flush_nodes(e))) return rc;
    if ((rc = upload_batch(e, &b))) return rc;
    if ((rc = run_classes(e, &b))) return rc;
    const uint32_t Wn = n_words_of(e->n_nodes);
    std::vector<EnfNode> en(n_nodes);
    std::vector<EnfTask> et(n_tasks);
    for (uint32_t i = 0; i < n_nodes; ++i) en[i] = EnfNode{nodes[i].node, nodes[i].first_task, nodes[i].n_tasks, 0u, nodes[i].cpu, nodes[i].mem};
    for (uint32_t i = 0; i < n_tasks; ++i)
        et[i] = EnfTask{tasks[i].cpu, tasks[i].mem, b.rt[i].cls_con, tasks[i].flags & SWP_ENF_RESERVATIONS, tasks[i].desired_state, tasks[i].state};
    DevBuf d_en, d_et, d_out;
    if ((rc = upload(e, d_en, en))) return rc;
    if ((rc = upload(e, d_et, et))) return rc;
    HIPCHECK(e, d_out.reserve(n_tasks));
    hipStream_t st = e->stream;
    HIPCHECK(e, hipMemsetAsync(d_out.p, 0, n_tasks, st));
    hipLaunchKernelGGL(k_enforce, dim3((n_nodes + 255) / 256), dim3(256), 0, st, n_nodes, Wn, d_en.as<EnfNode>(), d_et.as<EnfTask>(),
                       b.d_con.as<u64>(), d_out.as<unsigned char>());
    HIPCHECK(e, hipGetLastError());
    HIPCHECK(e, hipMemcpyAsync(out_reject, d_out.p, n_tasks, hipMemcpyDeviceToHost, st));
    HIPCHECK(e, hipStreamSynchronize(st));
    return SWP_OK;
}

int swp_node_matches(swp_engine* e, const uint32_t* constraint_sets, uint32_t n_sets, uint64_t* out_bitmaps, uint32_t n_words) {
    if (e && e->set) return ss::node_matches(e, constraint_sets, n_sets, out_bitmaps, n_words);
    if (!e || (!constraint_sets && n_sets) || (!out_bitmaps && n_sets)) return SWP_EINVAL;
    if (n_sets == 0) return SWP_OK;
    (void)hipSetDevice(e->device);
    int rc = flush_nodes(e);
    if (rc) return rc;
    const uint32_t Wn = n_words_of(e->n_nodes);
    if (n_words != Wn) return e->fail(SWP_EINVAL, "node_matches: caller passes %u words per row, the nodeSet has %u", n_words, Wn);
    if (e->n_nodes == 0) return SWP_OK;
    uint32_t svc = 0;
    {
        static const char kDummy[] = "\0swp-enforce";
        if ((rc = swp_intern(e, SWP_SPACE_SERVICE, kDummy, sizeof kDummy - 1, &svc))) return rc;
    }
    std::vector<swp_task_desc> descs(n_sets);
    std::memset(descs.data(), 0, descs.size() * sizeof(swp_task_desc));
    for (uint32_t i = 0; i < n_sets; ++i) {
        descs[i].service = svc;
        descs[i].constraint_set = constraint_sets[i];
    }
    swp_batch b;
    if ((rc = build_batch(e, descs.data(), n_sets, &b, nullptr))) return rc;
    if ((rc = flush_nodes(e))) return rc;
    if ((rc = upload_batch(e, &b))) return rc;
    if ((rc = run_classes(e, &b))) return rc;
    hipStream_t st = e->stream;
    for (uint32_t i = 0; i < n_sets; ++i) {
        const uint32_t cls = b.rt[i].cls_con;
        const void* src = cls ? (const void*)(b.d_con.as<u64>() + (size_t)cls * Wn) : (const void*)e->d_valid.p;   // no constraints: every present node
        HIPCHECK(e, hipMemcpyAsync(out_bitmaps + (size_t)i * Wn, src, (size_t)Wn * 8, hipMemcpyDeviceToHost, st));
    }
    HIPCHECK(e, hipStreamSynchronize(st));
    return SWP_OK;
}

int swp_stats(swp_engine* e, swp_stats_t* out) {
    if (e && e->set) return ss::stats(e, out);
    if (!e || !out) return SWP_EINVAL;
    e->stats.n_nodes = e->n_present;
    e->stats.n_words = n_words_of(e->n_nodes);
    *out = e->stats;
    return SWP_OK;
}

int swp_shardset_create(const swp_config* cfg, const int32_t* devices, uint32_t n_shards, uint32_t nodes_per_shard, swp_engine** out) {
    return ss::create(cfg, devices, n_shards, nodes_per_shard, out);
}

}  // extern "C"

#include "swp_shardset.hpp"
