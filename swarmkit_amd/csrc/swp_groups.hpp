// swp_groups.hpp — task groups (SpecVersion != nil): scheduleTaskGroup with k = len(group) (scheduler.go:694-748), nodeSet.tree
// with a bounded max-heap per leaf (nodeset.go:50-124; container/heap mechanics step for step: nodeheap.go:3-31 and the Go stdlib
// heap, decision_tree.go:24-52), scheduleNTasksOnSubtree (:772-825), the fill loop scheduleNTasksOnNodes (:844-924),
// NodeInfo.addTask (nodeinfo.go:108-154) and Pipeline.Process / Explain (pipeline.go:56-103) for a whole tick's groups in ONE launch.
//
// Why it looks the way it does. Which of several equal-key nodes a full heap keeps, and the order heap-sort pops equal keys in, are
// artefacts of container/heap's array mechanics; equal keys are the rule (spread placement keeps every node's task count within
// one or two of the others), so the only way to give the reference's answer is to replay the heap operations in node order. That
// replay — push / replace-root / pop-all / fill — is a strictly serial chain of dependent LDS round trips (one per heap level) and it
// is ALL that one lane has to do here. Everything else is wide and is kept off that chain:
//
//   wave 0 ("the machine")   per group: scans the nodes 64 at a time (first failing filter + nodeLess key per node come precomputed),
//        ballots the nodes that can enter their leaf's heap (heap not full, or key < root as of the chunk's start: roots only drop,
//        so this is a superset), lane 0 replays exactly those; then the tree walk and the fill loops on the heap slots' state,
//        then (all 64 lanes) the write-back of the touched nodes and the re-evaluation of just those nodes for the next group.
//   waves 1..15 ("helpers")  run one group AHEAD of the machine: scatter the next service's (node, svcCount, failures) list into dense
//        columns and add it up along the spread tree (decisionTree.tasks, nodeset.go:88-90,103-105), evaluate Pipeline.Process's
//        first failing filter and the packed nodeLess key for every node against the node rows as they are; count the Explain
//        histogram of a group that could not be placed completely. Commands travel through a small LDS ring; a helper bumps one
//        counter per finished command, which doubles as the barrier between commands.
//   The machine's picks change at most k nodes: those (and only those) are evaluated again for the next group after the write-back
//   ("patch"), so what the helpers computed early is exact when it is used.
//
// No size limits: a group's working set (heap slots = sum over leaves of min(k, nodes of the leaf); tree-node arrays; the walk's
// frame stack) is carved from an arena that lives in LDS when it fits (G2_ARENA_LDS) and in global memory (L2) otherwise — the same
// source instantiated twice, so that the LDS instance keeps ds_* instructions. A tree node may have any number of children (the
// noRoom set is a flag per child), the recursion any depth (frames in the arena), the fill phase logs nothing (running counters).
//
// Written against swp_wave.hpp only: tests/emu/emu_groups.cpp runs this source on CPU fibers against a sequential model.
#pragma once
#include <stddef.h>

#include "swp_types.hpp"
#include "swp_volumes.hpp"

namespace swpdev {

#define G2_THREADS 1024
#define G2_RING 16
#define G2_FF_PASS 255u
#define G2_FF_ABSENT 254u        // no such node (slot not present in the nodeSet)
#ifndef G2_ARENA_LDS
#define G2_ARENA_LDS (120 * 1024)   // (the emulation harness also builds with a tiny one: most groups then run the global-memory instance)
#endif
#define G2_MAXGEN 8
#define G2_NONE 0xFFFFFFFFu
#define ERR_GROUP_HANG 3         // a wave waited for another one beyond any plausible time (protocol bug): the launch ends instead of hanging
#define G2_SPIN_LIMIT (1u << 26)

enum { G2_OP_SCATTER = 1, G2_OP_EVAL = 2, G2_OP_EXPLAIN = 3, G2_OP_UNSCATTER = 4, G2_OP_QUIT = 5 };

// One entry of a group's candidate list (Groups2Args.ccand): a node of the group's static class list with its nodeLess key and spread
// leaf — all the admission scan reads (one 16-byte load per candidate, 64 candidates per "chunk" instead of the handful a node word holds)
struct alignas(16) G2Cand {
    u64 key;     // KEY_NONE: the node dropped out since the list was built (a hole)
    u32 node;
    u32 leaf;
};
static_assert(sizeof(G2Cand) == 16, "G2Cand layout");

struct GroupRec2 {   // one per group, 152 B
    i64 cpu, mem;
    u64 maxrep;
    u32 flags;          // RT_*
    u32 k;              // group size
    u32 svc;            // batch-local service
    u32 out_off;        // first output index
    u32 pset;
    u32 cls_con, cls_plat, cls_plug;
    u32 tree;           // spread set of the call (0-based; a set without levels is a tree of one node)
    u32 n_slots;        // heap slots the group needs: sum over the tree's leaves of min(k, nodes of the leaf)
    u32 n_gen;          // generic reservations (filter.go:86-91), at most G2_MAXGEN
    u32 dep_prev;       // the previous group of the call is of the same service: nothing of this group is prepared ahead of its write-back
    u32 gkind[G2_MAXGEN];
    int32_t gval[G2_MAXGEN];
    u32 mset;           // mount set of the group's tasks (swp_volumes.hpp), 0: no cluster mounts
    u32 att_off;        // first row of the group's tasks in Groups2Args.att
    u32 scls;           // static class of the group: its (cls_plug, cls_con, cls_plat) among the call's distinct ones (Groups2Args.slist)
    u32 pad1;
};
static_assert(sizeof(GroupRec2) == 152, "GroupRec2 layout");

struct Groups2Args {
    u32 n_nodes, n_words, n_groups, gstride;
    u32 max_ntn;             // tree nodes of the largest spread tree of the call
    u32 max_depth;           // its depth in levels (a tree of one node: 0)
    u32 dbg, pad0;
    const GroupRec2* g;
    const u64* valid;
    const u64* ready;
    const u64* con;
    const u64* plat;
    const u64* plug;
    i64* cpu;
    i64* mem;
    u32* total;
    int32_t* gcnt;           // [kind][gstride]
    u64* portmap;
    const u32* pset_off;
    const u32* pset_ids;
    u32* list_node;
    u32* list_svc;
    u32* list_fail;
    const u32* list_off;     // [n_svc+1]
    u32* list_cnt;           // [n_svc] entries in use (compact at the front of the service's range)
    // tree topology per spread set: tnodes of tree t are [tree_off[t], tree_off[t+1]); indices inside are relative; children follow their parent
    const u32* tree_off;
    const u32* tn_parent;    // G2_NONE for the root
    const u32* tn_first;     // first child or G2_NONE
    const u32* tn_next;      // next sibling or G2_NONE
    const u32* tn_nchild;
    const u32* tn_nodes;     // nodes whose leaf this tnode is
    const u32* leaf_of_node; // [n_trees][n_nodes]
    // scratch, all double-buffered by group parity
    unsigned char* ffbuf;    // [2][n_nodes] first failing filter of Pipeline.Process, G2_FF_PASS, G2_FF_ABSENT
    u64* keybuf;             // [2][n_nodes] nodeLess key at tree() time
    // Static class lists (round 6). ReadyFilter, PluginFilter, ConstraintFilter and PlatformFilter do not change while a tick runs, and
    // groups share them (cfg3: 63 distinct (plugin, constraint, platform) classes among 1 000 groups): k_g2_static lists, once per call
    // and per class, the nodes that pass all four IN NODE ORDER — a chip-wide pass, one wave per class. A group is then evaluated over
    // its class's list only (a tenth of the nodes on cfg3), and that evaluation IS the group's compact candidate list: entry i =
    // {nodeLess key, node slist[i], spread leaf}, KEY_NONE for a node that fails Resource / HostPort / MaxReplicas / Volumes (a hole).
    u32 n_scls, pad2;
    const u32* scls_def;     // [n_scls][3] cls_plug, cls_con, cls_plat
    u32* slist;              // [n_scls][n_nodes] the class's nodes in node order
    u32* scnt;               // [n_scls] how many
    u64* sbits;              // [n_scls][n_words] the same as a bitmap (the patch after a write-back asks ONE word whether a node is listed)
    G2Cand* ccand;           // [2][n_nodes] the group's candidates: what the admission scan reads, 64 to a chunk
    u32* cpos;               // [2][n_nodes] a listed node's place (the patch after a write-back finds its entry there)
    u64* cmin;               // [2][n_words] per chunk: the lowest key among its candidates (KEY_NONE: none)
    u32* svc_dense;          // [2][n_nodes] the group's service: ActiveTasksCountByService per node (zero outside the list)
    u32* fail_dense;         // [2][n_nodes] recent failures (only values >= maxFailures are listed)
    u32* lpos_dense;         // [2][n_nodes] list entry of the node + 1, 0: not listed
    i64* tsumbuf;            // [2][max_ntn] decisionTree.tasks
    u64* xroot;              // [max_ntn] Explain: heap root keys and ...
    int32_t* xadm;           // [max_ntn] ... heap lengths as tree() left them
    unsigned char* arena;    // global arena: the largest g2_arena_bytes over the call's groups that do not fit G2_ARENA_LDS
    int32_t* out_node;
    u32* hist;               // [n_groups][8]
    Ctl* ctl;
    // CSI volumes (groups with mset != 0): the VolumesFilter is evaluated from the volumes as they are — at tree() time and whenever the
    // fill loop re-checks a node — and every placement chooses and reserves its volumes (scheduler.go:857-874)
    VolView vol;
    u32* att;                // [tasks of the groups with mounts][VOL_MAX_MOUNTS]
};

struct G2Frame {   // one invocation of scheduleNTasksOnSubtree (scheduler.go:772-825)
    u32 tn;
    int n, scheduled, assign;
    i64 usable, desired, rem;
    u32 child;
    int n_noroom;
    u32 converging;
    int phase;
    u32 pad[2];
};
static_assert(sizeof(G2Frame) == 64, "G2Frame layout");

inline __host__ __device__ size_t g2_arena_bytes(u32 S, u32 ntn, u32 ngen, u32 depth, u32 k) {
    const size_t fw = (size_t)(S + 63) / 64 + 1, kt = k < S ? k : S;
    return 32 * (size_t)S + 64 * ((size_t)depth + 2) + 8 * (2 * (size_t)ntn + fw) + 4 * ((size_t)S * ngen + 2 * kt + 6 * (size_t)ntn) + 64;
}
// mailbox, staging of one chunk of candidates, a few scalars
#define G2_LDS_FIXED 11264     // (the flat mode's scratch and tables: g2_flat_*, g2_pre_table)
#define G2_FLAT_OFF 8192
#define G2_FLAT_MAXK 128u       // heap positions the flat mode's bit mask covers
#define G2_PRE_OFF 6144         // [G2_FLAT_MAXK][2] u64: the positions that come before position p in a post-order walk of the heap's tree (g2_pre_table)
#define G2_LEFT_OFF 9216        // [G2_FLAT_MAXK][2] u64: ... those of them that are not descendants of p
#define G2_VISW 4              // chunks of 64 candidates whose records are staged through LDS together
inline __host__ __device__ size_t g2_lds_bytes() { return (size_t)G2_LDS_FIXED + G2_ARENA_LDS; }

#ifdef SWP_G2_KERNELS
// The machine's section timers (SWP_DBG=16) are compiled in only with -DSWP_G2_PROF (make prof: swarmkit_amd/lib/libswp_prof.so): their
// thirty 64-bit counters live in scalar registers, and a kernel whose scalar registers are long spilled pays for each of them with a
// lane of a vector register — 25 VGPRs went to scratch memory in the product before they were taken out (tools/check_kernels.py).
#ifdef SWP_G2_PROF
#define G2_PROF_ON(a) (((a).dbg & 16u) != 0)
#else
#define G2_PROF_ON(a) false
#endif
#ifndef G2_STAT
#define G2_STAT(i, v) ((void)0)   // (tests/emu/emu_groups.cpp counts which admission path a run took)
#endif
// nodeLess (scheduler.go:708-735) as ONE integer compare: key = (failures if >= 5 else 0, svcCount, total) packed 8 | 24 | 32 bits
// (both sides below 5 failures skip the failure compare; a side at >= 5 loses against any side below).
WV_DEV u64 g2_key(u32 fail, u32 svc, u32 total) {
    const u32 fc = fail >= MAX_FAILURES ? fail : 0u;
    return ((u64)fc << 56) | ((u64)svc << 32) | total;
}
WV_DEV bool g2_key_ok(u32 fail, u32 svc) { return fail < 256u && svc < (1u << 24); }
#define G2_KEY_STEP ((1ull << 32) + 1ull)   // one more task of the service on the node: svcCount + 1, ActiveTasksCount + 1

// A heap position holds ONE 16-byte record — key, node, index into the touched list — so a sift step is one wide load per element
// (both children of a position are adjacent: 32 bytes), and the node's residuals ride in a second 16-byte record at the SAME
// position once the leaf is sorted: nothing the fill loop touches is behind an indirection.
struct alignas(16) G2Ent {
    u64 key;
    u32 node;
    u32 tix;     // G2_NONE until the node got its first task of the group, then its index in tnode / tcount
};
struct alignas(16) G2Res {
    i64 cpu, mem;
};
static_assert(sizeof(G2Ent) == 16 && sizeof(G2Res) == 16, "heap record layout");

struct G2Arena {
    G2Ent* HE;          // [slots] heap position -> entry
    G2Res* PS;          // [slots] residual cpu / memory of the entry's node (loaded once the leaf is sorted)
    i64* tsum;          // decisionTree.tasks
    u64* rootkey;       // per leaf: heap root key as tree() left it (Explain)
    u64* failed;        // fill loop: positions that failed Process (failedConstraints, scheduler.go:846)
    G2Frame* st;
    int32_t* gen;       // [slot][n_gen] generic counts of the group's kinds, by position like PS
    u32* tnode;         // nodes that got a task ...
    u32* tcount;        // ... and how many
    u32* h_off;         // first heap position of a leaf
    int32_t* h_len;     // nodeMaxHeap.length while tree() runs
    int32_t* h_cnt;     // len(nodeMaxHeap.nodes)
    int32_t* h_adm;     // length as tree() left it
    u32* noroom;        // per tree node: member of its parent's noRoom set (scheduler.go:787,810-813)
    u32* h_vis;         // the leaf was handed to scheduleNTasksOnNodes before (orderedNodes filters from the second call on)
};
WV_DEV void g2_carve(G2Arena& A, unsigned char* p, u32 S, u32 ntn, u32 ngen, u32 depth, u32 k) {
    const size_t fw = (size_t)(S + 63) / 64 + 1, kt = k < S ? k : S;
    A.HE = reinterpret_cast<G2Ent*>(p); p += 16 * (size_t)S;
    A.PS = reinterpret_cast<G2Res*>(p); p += 16 * (size_t)S;
    A.st = reinterpret_cast<G2Frame*>(p); p += 64 * ((size_t)depth + 2);
    A.tsum = reinterpret_cast<i64*>(p); p += 8 * (size_t)ntn;
    A.rootkey = reinterpret_cast<u64*>(p); p += 8 * (size_t)ntn;
    A.failed = reinterpret_cast<u64*>(p); p += 8 * fw;
    A.gen = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)S * ngen;
    A.tnode = reinterpret_cast<u32*>(p); p += 4 * kt;
    A.tcount = reinterpret_cast<u32*>(p); p += 4 * kt;
    A.h_off = reinterpret_cast<u32*>(p); p += 4 * (size_t)ntn;
    A.h_len = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)ntn;
    A.h_cnt = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)ntn;
    A.h_adm = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)ntn;
    A.noroom = reinterpret_cast<u32*>(p); p += 4 * (size_t)ntn;
    A.h_vis = reinterpret_cast<u32*>(p);
}

// container/heap (go stdlib) over nodeMaxHeap: Less(i,j) = lessFunc(nodes[j], nodes[i]) (nodeheap.go:17-20). up / down move ONE
// element along a path and swap it with what it meets: the element rides in registers and every step copies the other one into the
// hole — the same comparisons and the same final arrangement as the swap sequence, one memory round trip per level.
// up(j0) for the element `e` that is (conceptually) at j0; returns its final position (it is stored there).
WV_DEV int g2_up_val(const G2Arena& A, u32 base, int j0, G2Ent e) {
    int j = j0;
    while (j > 0) {                  // j == 0: i == j, break (Go's (j-1)/2 truncates to 0)
        const int i = (j - 1) / 2;
        const G2Ent pe = A.HE[base + i];
        if (!(pe.key < e.key)) break;   // !Less(j, i)
        A.HE[base + j] = pe;
        j = i;
    }
    A.HE[base + j] = e;
    return j;
}
// Records travel BY VALUE and are picked field by field: a record that is selected as a whole, or handed out through a reference,
// ends up in scratch memory (a global-memory round trip inside the serial chain).
WV_DEV G2Ent g2_pick(bool c, G2Ent x, G2Ent y) {
    G2Ent r;
    r.key = c ? x.key : y.key;
    r.node = c ? x.node : y.node;
    r.tix = c ? x.tix : y.tix;
    return r;
}
struct G2Down {
    int pos;     // where the element came to rest
    G2Ent top;   // the record that ended up AT i0 (what a caller that tracks the root in registers wants to know)
};
// down(i0, n) for the element `e` that is (conceptually) at i0 (it is stored where it comes to rest)
WV_DEV G2Down g2_down_val(const G2Arena& A, u32 base, int i0, int n, G2Ent e) {
    int i = i0;
    G2Ent top = e;
    for (;;) {
        const int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        const int j2 = j1 + 1;
        const bool two = j2 < n;
        const G2Ent c1 = A.HE[base + j1];
        const G2Ent c2 = A.HE[base + (two ? j2 : j1)];
        const bool right = two && c1.key < c2.key;   // Less(j2, j1)
        const G2Ent cj = g2_pick(right, c2, c1);
        if (!(e.key < cj.key)) break;                // !Less(j, i)
        A.HE[base + i] = cj;
        top = g2_pick(i == i0, cj, top);
        i = right ? j2 : j1;
    }
    A.HE[base + i] = e;
    G2Down r;
    r.pos = i;
    r.top = top;
    return r;
}
// the same for a leaf that is handed out again (orderedNodes' second call on): the residual records move with the entries
WV_DEV void g2_move_full(const G2Arena& A, u32 ngen, u32 dst, u32 src) {
    A.HE[dst] = A.HE[src];
    A.PS[dst] = A.PS[src];
    for (u32 q = 0; q < ngen; ++q) A.gen[(size_t)dst * ngen + q] = A.gen[(size_t)src * ngen + q];
}
WV_DEV bool g2_down_full(const G2Arena& A, u32 ngen, u32 base, int i0, int n) {   // swap-based: only the rare second hand-out pays it
    int i = i0;
    for (;;) {
        const int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int j = j1;
        const int j2 = j1 + 1;
        if (j2 < n && A.HE[base + j1].key < A.HE[base + j2].key) j = j2;
        if (!(A.HE[base + i].key < A.HE[base + j].key)) break;
        const G2Ent te = A.HE[base + i]; A.HE[base + i] = A.HE[base + j]; A.HE[base + j] = te;
        const G2Res tr = A.PS[base + i]; A.PS[base + i] = A.PS[base + j]; A.PS[base + j] = tr;
        for (u32 q = 0; q < ngen; ++q) {
            const int32_t tg = A.gen[(size_t)(base + i) * ngen + q];
            A.gen[(size_t)(base + i) * ngen + q] = A.gen[(size_t)(base + j) * ngen + q];
            A.gen[(size_t)(base + j) * ngen + q] = tg;
        }
        i = j;
    }
    return i > i0;
}
// heap-sort of one leaf in place: heap.Pop until empty (decision_tree.go:46-49; Pop = Swap(0, n-1); down(0, n-1); length--).
// The root rides in registers: after a sift it is either the element that was sifted (it did not move) or the child that took its place.
WV_DEV void g2_pop_all(const G2Arena& A, u32 base, int len) {
    if (len < 2) return;
    G2Ent root = A.HE[base];
    G2Ent last = A.HE[base + len - 1];
    for (int n = len - 1; n >= 1; --n) {
        // the element the NEXT pop takes from the end is requested now; only this pop's sifted element can land on it
        const G2Ent nlast = A.HE[base + (n >= 2 ? n - 1 : 0)];
        A.HE[base + n] = root;
        int fin = 0;
        if (n >= 3) {   // equal keys are the rule: the element from the end usually stays at the root (one compare, no loop)
            const G2Ent c1 = A.HE[base + 1], c2 = A.HE[base + 2];
            const bool right = c1.key < c2.key;
            const G2Ent cj = g2_pick(right, c2, c1);
            if (!(last.key < cj.key)) {
                A.HE[base] = last;
                root = last;
            } else {
                A.HE[base] = cj;
                root = cj;
                fin = g2_down_val(A, base, right ? 2 : 1, n, last).pos;
            }
        } else {
            const G2Down d = g2_down_val(A, base, 0, n, last);
            fin = d.pos;
            root = d.top;
        }
        last = g2_pick(fin == n - 1, last, nlast);
    }
}

// (a group's record is copied to registers field by field; its two small arrays are only ever read through the record in
// memory — `Gm` — because an array indexed by a loop counter would drag the whole copy into scratch memory)

// Pipeline.Process on node n for group G (pipeline.go:56-68: the FIRST failing filter in checklist order: Ready, Resource, Plugin,
// Constraint, Platform, HostPort, MaxReplicas, Volumes — scheduler.go:60-66) and the node's nodeLess key. Returns the first failing
// filter (G2_FF_PASS, G2_FF_ABSENT); `listed`: the node is on the group's static class list (present and passing Ready, Plugin,
// Constraint, Platform). LISTED: the caller took n FROM that list, so those five are not looked at again.
// Every input is REQUESTED before any of them is looked at — one memory round trip, not one per filter of the chain (class 0 = "no such
// filter" has a row like any other: it is loaded and not looked at): g2_load_in asks, g2_decide answers. OVR: the node's residuals and
// task count are not loaded — the caller fills them in (the write-back has the new ones in registers: g2_group's fused pass).
struct G2In {
    u64 vw, rw, pw, cw, tw;
    u32 sv, fl, tot;
    i64 c, m;
};
// CLS: the five static words are not asked for one by one — only the class bitmap's word, in `vw` (g2_decide_patch: a node that is
// not listed needs no verdict of its own: nothing reads its record before the Explain command evaluates it itself).
template <bool LISTED, bool OVR, bool CLS = false>
WV_DEV void g2_load_in(const Groups2Args& a, const GroupRec2& G, u32 b, u32 n, G2In& in) {
    const u32 N = a.n_nodes, Wn = a.n_words, w = n >> 6;
    in.vw = LISTED ? ~0ull : CLS ? a.sbits[(size_t)G.scls * Wn + w] : a.valid[w];
    in.rw = (LISTED || CLS) ? ~0ull : a.ready[w];
    in.pw = (LISTED || CLS) ? ~0ull : a.plug[(size_t)G.cls_plug * Wn + w];
    in.cw = (LISTED || CLS) ? ~0ull : a.con[(size_t)G.cls_con * Wn + w];
    in.tw = (LISTED || CLS) ? ~0ull : a.plat[(size_t)G.cls_plat * Wn + w];
    in.sv = a.svc_dense[(size_t)b * N + n];
    in.fl = a.fail_dense[(size_t)b * N + n];
    in.c = OVR ? 0 : a.cpu[n];
    in.m = OVR ? 0 : a.mem[n];
    in.tot = OVR ? 0u : a.total[n];
}
template <bool LISTED>
WV_DEV u32 g2_decide(const Groups2Args& a, const GroupRec2& G, const GroupRec2* Gm, u32 n, const G2In& in, u64& key, bool& listed) {
    const u32 Wn = a.n_words, w = n >> 6;
    const u64 bit = 1ull << (n & 63);
    key = 0;
    listed = LISTED;
    if (!LISTED && !(in.vw & bit)) return G2_FF_ABSENT;
    const bool ready = LISTED || (in.rw & bit) != 0;
    u32 sff = G2_FF_PASS;   // the first failing one of Plugin / Constraint / Platform
    if (!LISTED) {
        if (G.cls_plug && !(in.pw & bit)) sff = 2;
        else if (G.cls_con && !(in.cw & bit)) sff = 3;
        else if (G.cls_plat && !(in.tw & bit)) sff = 4;
        listed = ready && sff == G2_FF_PASS;
    }
    bool res = true;   // ResourceFilter.Check, filter.go:76-95
    if (G.flags & RT_RES) {
        res = G.cpu <= in.c && G.mem <= in.m;
        for (u32 q = 0; q < G.n_gen; ++q)
            if (a.gcnt[(size_t)Gm->gkind[q] * a.gstride + n] < Gm->gval[q]) res = false;   // HasEnough, validate.go:24-52
    }
    u32 ff = G2_FF_PASS;
    if (!ready) ff = 0;
    else if (!res) ff = 1;
    else if (sff != G2_FF_PASS) ff = sff;
    else {
        bool busy = false;
        if (G.flags & RT_PORTS)
            for (u32 q = a.pset_off[G.pset]; q < a.pset_off[G.pset + 1]; ++q)
                if (a.portmap[(size_t)a.pset_ids[q] * Wn + w] & bit) busy = true;
        if (busy) ff = 5;
        else if ((G.flags & RT_MAXREP) && !((u64)in.sv < G.maxrep)) ff = 6;
        else if (G.mset && !((vol_filter_word(a.vol, G.mset, w) >> (n & 63u)) & 1ull)) ff = 7;   // VolumesFilter, the pipeline's last entry
    }
    if (!g2_key_ok(in.fl, in.sv)) a.ctl->error = ERR_GROUP_RANGE;
    key = g2_key(in.fl, in.sv, in.tot);
    return ff;
}
template <bool LISTED>
WV_DEV u32 g2_process(const Groups2Args& a, const GroupRec2& G, const GroupRec2* Gm, u32 b, u32 n, u64& key, bool& listed) {
    G2In in;
    g2_load_in<LISTED, false>(a, G, b, n, in);
    return g2_decide<LISTED>(a, G, Gm, n, in, key, listed);
}
// A node that got a task since the next group was evaluated is evaluated again: its Explain record, and — if it is on that group's
// static class list — its candidate entry (keys only grow, nodes only drop out: the chunk minima stay lower bounds).
WV_DEV void g2_patch_node(const Groups2Args& a, const GroupRec2& G, const GroupRec2* Gm, u32 b, u32 n) {
    const u32 N = a.n_nodes;
    const u32 cp = a.cpos[(size_t)b * N + n];   // (meaningless unless listed; requested with the rest)
    u64 key;
    bool listed;
    G2In in;
    g2_load_in<false, false, true>(a, G, b, n, in);
    listed = ((in.vw >> (n & 63u)) & 1ull) != 0;
    if (!listed) return;   // (its Explain record is not read: the command evaluates unlisted nodes itself)
    const u32 ff = g2_decide<true>(a, G, Gm, n, in, key, listed);
    a.ffbuf[(size_t)b * N + n] = (unsigned char)ff;
    a.keybuf[(size_t)b * N + n] = key;
    a.ccand[(size_t)b * N + cp].key = ff == G2_FF_PASS ? key : KEY_NONE;
}
// The static class lists (Groups2Args.slist): ONE WAVE per class (k_g2_static: a grid of n_scls workgroups of 64), lanes over the node words.
WV_DEV void g2_static_list(const Groups2Args& a, u32 c) {
    const u32 lane = wv::lane(), N = a.n_nodes, Wn = a.n_words;
    const u32 cpl = a.scls_def[3u * c], cco = a.scls_def[3u * c + 1u], cpa = a.scls_def[3u * c + 2u];
    u32* out = a.slist + (size_t)c * N;
    u32 run = 0;
    for (u32 w0 = 0; w0 < Wn; w0 += 64u) {
        const u32 w = w0 + lane;
        u64 m = 0;
        if (w < Wn) {
            m = a.valid[w] & a.ready[w];
            if (cpl) m &= a.plug[(size_t)cpl * Wn + w];
            if (cco) m &= a.con[(size_t)cco * Wn + w];
            if (cpa) m &= a.plat[(size_t)cpa * Wn + w];
            if (w == Wn - 1u && (N & 63u)) m &= (1ull << (N & 63u)) - 1ull;
        }
        if (w < Wn) a.sbits[(size_t)c * Wn + w] = m;
        const u32 cnt = (u32)wv::popc64(m);
        const u32 incl = wv::scan_incl_u32(cnt);
        u32 pos = run + incl - cnt;
        while (m) {
            out[pos++] = w * 64u + (u32)wv::ffs64(m);
            m &= m - 1ull;
        }
        run += wv::readlane(incl, 63);
    }
    if (lane == 0) a.scnt[c] = run;
}
WV_DEV u32 g2_fls64(u64 v) {   // index of the highest set bit (v != 0)
    const u32 hi = (u32)(v >> 32);
    return hi ? 63u - (u32)wv::clz32(hi) : 31u - (u32)wv::clz32((u32)v);
}
WV_DEV u64 g2_wave_min64(u64 v) {
    const u32 hi = (u32)(v >> 32), lo = (u32)v;
    const u32 mh = wv::min_u32(hi);
    const u32 ml = wv::min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((u64)mh << 32) | ml;
}

struct G2Mail {   // LDS
    u32 posted;            // commands posted so far (wave 0 publishes)
    u32 done;              // helper waves x commands finished
    u32 quit;              // somebody gave up waiting (ERR_GROUP_HANG)
    u32 pad;
    u32 op[G2_RING], grp[G2_RING];
    // Explain of the current group
    u32 cntx[8];
    u32 x_lastp, x_k;
    // what lane 0 of the machine hands to its other lanes
    u32 sh[16];
};
enum { SH_LEFT = 2, SH_NTOUCH = 3, SH_ERR = 4, SH_LASTP = 5, SH_C1 = 6, SH_C5 = 7, SH_C6 = 8, SH_FPASS = 9, SH_C7 = 10 };

struct G2Stage {   // LDS: the candidates of one 64-node word, compacted in node order: {key, node, leaf} as one 16-byte record each
    G2Ent ent[64];
    u32 scan[64];  // scratch of the offset scan over the tree's leaves
};

WV_DEV bool g2_wait_ge(const u32* p, u32 want, G2Mail* mb) {
    u32 spins = 0;
    while (wv::lds_poll32(p) < want) {
        if (wv::lds_poll32(&mb->quit)) return false;
        if (++spins > G2_SPIN_LIMIT) {
            wv::lds_publish32(&mb->quit, 1u);
            return false;
        }
        wv::spin_pause();
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// helpers: waves 1 .. nh
// ---------------------------------------------------------------------------------------------------------------------------
WV_DEV void g2_helper(const Groups2Args& a, G2Mail* mb, u32 hid, u32 nh) {
    const u32 lane = wv::lane(), N = a.n_nodes, Wn = a.n_words;
    for (u32 my = 0;; ++my) {
        if (!g2_wait_ge(&mb->posted, my + 1, mb)) return;
        if (!g2_wait_ge(&mb->done, my * nh, mb)) return;   // every helper is through the command before: its writes are this one's inputs
        const u32 op = mb->op[my % G2_RING], gi = mb->grp[my % G2_RING];
        if (op == G2_OP_QUIT) return;
        const GroupRec2* Gm = a.g + gi;
        const GroupRec2 G = *Gm;
        const u32 b = gi & 1u;
        if (op == G2_OP_SCATTER || op == G2_OP_UNSCATTER) {
            const u32 e0 = a.list_off[G.svc], e1 = e0 + a.list_cnt[G.svc];
            const u32 tbase = a.tree_off[G.tree];
            const u32* leaf_of = a.leaf_of_node + (size_t)G.tree * N;
            for (u32 e = e0 + hid * 64u + lane; e < e1; e += nh * 64u) {
                const u32 n = a.list_node[e];
                if (n == LIST_EMPTY) continue;
                if (op == G2_OP_UNSCATTER) {
                    a.svc_dense[(size_t)b * N + n] = 0;
                    a.fail_dense[(size_t)b * N + n] = 0;
                    a.lpos_dense[(size_t)b * N + n] = 0;
                    continue;
                }
                const u32 sv = a.list_svc[e];
                a.svc_dense[(size_t)b * N + n] = sv;
                a.fail_dense[(size_t)b * N + n] = a.list_fail[e];
                a.lpos_dense[(size_t)b * N + n] = e + 1u;
                // tree(): the node's service count is added at its leaf and at every level above it, whether or not the node is
                // feasible (nodeset.go:88-90,103-105)
                if (sv && ((a.valid[n >> 6] >> (n & 63)) & 1ull))
                    for (u32 t = leaf_of[n]; t != G2_NONE; t = a.tn_parent[tbase + t]) wv::g_add64(&a.tsumbuf[(size_t)b * a.max_ntn + t], (i64)sv);
            }
        } else if (op == G2_OP_EVAL) {
            // The group's candidates: its static class's nodes, a chunk of 64 per wave and turn, + the chunk's lowest key: what lets the
            // machine skip 64 candidates at a time. (Keys only grow and nodes only drop out while the tick runs, so a minimum taken
            // early stays a lower bound: the patch after a write-back need not touch it.)
            const u32 M = a.scnt[G.scls];
            const u32* sl = a.slist + (size_t)G.scls * N;
            const u32* leaf_of = a.leaf_of_node + (size_t)G.tree * N;
            for (u32 c = hid; c * 64u < M; c += nh) {
                const u32 i = c * 64u + lane;
                u64 ck = KEY_NONE;
                if (i < M) {
                    const u32 n = sl[i];
                    u64 key;
                    bool listed;
                    const u32 ff = g2_process<true>(a, G, Gm, b, n, key, listed);
                    a.ffbuf[(size_t)b * N + n] = (unsigned char)ff;
                    a.keybuf[(size_t)b * N + n] = key;
                    if (ff == G2_FF_PASS) ck = key;
                    G2Cand ce;
                    ce.key = ck; ce.node = n; ce.leaf = leaf_of[n];
                    a.ccand[(size_t)b * N + i] = ce;
                    a.cpos[(size_t)b * N + n] = i;
                }
                const u64 mk = g2_wave_min64(ck);
                if (lane == 0) a.cmin[(size_t)b * Wn + c] = mk;
            }
        } else if (op == G2_OP_EXPLAIN) {
            // Every passing Process zeroes the counters (pipeline.go:64-66), so only the calls AFTER the last passing one count. Inside
            // tree() the heaps stop changing after that call: a later node was "called" (nodeset.go:108-116) iff its leaf's heap was not
            // full or the node is less than the final root.
            const u32 lastp = mb->x_lastp, k = mb->x_k;
            const u32* leaf_of = a.leaf_of_node + (size_t)G.tree * N;
            for (u32 w = hid; w < Wn; w += nh) {
                const u32 n = w * 64u + lane;
                // (a node on the group's static class list has its record from EVAL / the patch; any other node is evaluated here — its
                // rows have not moved since tree(): a group only ever touches listed nodes, and this command is through before the NEXT
                // group's write-back)
                u32 f = 0xFFu;
                if (n < N && n >= lastp) {
                    u64 key;
                    bool listed;
                    u32 ffn = g2_process<false>(a, G, Gm, b, n, key, listed);
                    if (listed) { ffn = a.ffbuf[(size_t)b * N + n]; key = a.keybuf[(size_t)b * N + n]; }
                    if (ffn < 8u) {
                        const u32 lf = leaf_of[n];
                        if (a.xadm[lf] < (int32_t)k || key < a.xroot[lf]) f = ffn;
                    }
                }
                for (u32 q = 0; q < 8; ++q) {
                    const u64 bm = wv::ballot(f == q);
                    if (bm && lane == 0) wv::lds_add32(&mb->cntx[q], (u32)wv::popc64(bm));
                }
            }
        }
        wv::lockstep();   // every lane of the wave is through the command before lane 0 says so (the device runs them together: this is for the CPU fibers)
        if (lane == 0) wv::lds_add_release32(&mb->done, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// FLAT MODE of a full heap with ONE leaf whose keys take two values (the rule while a tick runs: spread placement keeps every
// node's task count within one of the others' — "heavy" = the root's key, "light" = the other one).
// A candidate with the light key replaces the root and sifts down (heap.Fix(0): down(), go stdlib): it moves to the larger child
// while that child is greater — with two values: to the LEFT child if it is heavy, else to the right one if that is heavy, else it
// stays. So (1) the path runs through heavy positions only, every element on it moves up one position, the root's leaves the heap,
// the new element comes to rest at the path's end; light elements never move. (2) The heavy positions form a subtree that contains
// the root (a max-heap: a heavy child has a heavy parent), the path's end is the node a post-order walk of that subtree (left,
// right, node) visits first, and removing it leaves a subtree of the same kind: the r-th replacement comes to rest at the r-th node of
// the heavy subtree in POST-ORDER. (3) While a heavy element is left the root is heavy, so every light candidate is admitted; once
// all |H| are gone the root is light and light candidates are not less than it any more.
// Hence, as long as only light candidates come by: they are COUNTED and remembered, nothing is sifted; when |H| of them have come
// the heap is what it was with candidate r at post-order position r of H — one parallel scatter. If the stream ends (or a candidate
// with a third key shows up) before all heavy elements are gone, the heavy elements that are left HAVE moved: then the remembered
// candidates are replayed one after the other by the ordinary down() (g2_flat_flush), and the ordinary code carries on.
struct G2Flat {
    bool on;          // wave-uniform, every lane's copy
    u64 hi, lo;       // the two keys (lo == KEY_NONE: every element is heavy so far, the first candidate below hi names it)
    u64 m0, m1;       // heavy positions 0..63, 64..127
    u32 nh, n;        // |H|; candidates remembered so far (n <= nh)
};
// Position q comes before position p in a post-order walk (left, right, node) of the binary tree of heap positions: q is a descendant
// of p, or q's subtree hangs to the left of the way from the root to p. (1-based indices: the ancestor of x at depth d is x >> (depth(x) - d).)
WV_DEV bool g2_post_before(u32 q, u32 p) {
    const u32 x = q + 1u, y = p + 1u;
    const u32 dx = 31u - (u32)wv::clz32(x), dy = 31u - (u32)wv::clz32(y);
    if (dx >= dy) return x != y && (x >> (dx - dy)) <= y;   // (x's ancestor at p's depth is p itself, or lies to its left)
    return x < (y >> (dy - dx));                             // (p's ancestor at q's depth: q itself comes AFTER its descendant p)
}
// pre[p] = {positions 0..63, positions 64..127} that come before p: the table the flat mode's scatter counts in (built once per launch)
WV_DEV bool g2_is_desc(u32 q, u32 p) {   // q lies in the subtree below p
    const u32 x = q + 1u, y = p + 1u;
    const u32 dx = 31u - (u32)wv::clz32(x), dy = 31u - (u32)wv::clz32(y);
    return dx > dy && (x >> (dx - dy)) == y;
}
WV_DEV void g2_pre_table(u64* pre, u64* left) {
    const u32 lane = wv::lane();
    for (u32 p = lane; p < G2_FLAT_MAXK; p += 64u) {
        u64 lo = 0, hi = 0, llo = 0, lhi = 0;
        for (u32 q = 0; q < 64u; ++q) {
            if (g2_post_before(q, p)) { lo |= 1ull << q; if (!g2_is_desc(q, p)) llo |= 1ull << q; }
            if (g2_post_before(q + 64u, p)) { hi |= 1ull << q; if (!g2_is_desc(q + 64u, p)) lhi |= 1ull << q; }
        }
        pre[2u * p] = lo;
        pre[2u * p + 1u] = hi;
        left[2u * p] = llo;
        left[2u * p + 1u] = lhi;
    }
    wv::wave_sync();
}
// The heap's keys as the flat mode sees them: hi = the root's key, the heavy positions, lo = the LARGEST key below hi (KEY_NONE: every
// element is heavy). Candidates with the key lo are counted (the argument above only needs that no light ELEMENT exceeds a light
// CANDIDATE: the sift then still stops where the heavy children end, and lighter elements further down are never looked at).
WV_DEV void g2_flat_init(const G2Arena& A, G2Flat& F, u32 k) {
    const u32 lane = wv::lane();
    const u64 k0 = lane < k ? A.HE[lane].key : 0ull, k1 = lane + 64u < k ? A.HE[lane + 64u].key : 0ull;
    F.hi = wv::readlane64(k0, 0);
    F.m0 = wv::ballot(lane < k && k0 == F.hi);
    F.m1 = wv::ballot(lane + 64u < k && k1 == F.hi);
    // (the largest key below hi = the least of the complements)
    const u64 o0 = (lane < k && k0 != F.hi) ? ~k0 : KEY_NONE, o1 = (lane + 64u < k && k1 != F.hi) ? ~k1 : KEY_NONE;
    const u64 mn = g2_wave_min64(o0 < o1 ? o0 : o1);
    F.lo = mn == KEY_NONE ? KEY_NONE : ~mn;
    F.nh = (u32)wv::popc64(F.m0) + (u32)wv::popc64(F.m1);
    F.n = 0;
    F.on = true;
    G2_STAT(0, 1);
}
// The remembered candidates enter the heap. cand[r] = node of the r-th one; pre / left: g2_pre_table.
// After r replacements candidate j < r sits at the j-th heavy position in post-order (light elements never move), and a heavy element
// that started at position p has moved up once per replacement from the first one that came to rest inside p's subtree on — the way
// from the root to a landing place passes every ancestor of it, and the post-order successor of a landing place inside a subtree lies
// inside the parent's subtree: once an element is on the move, every further replacement moves it, until it leaves at the root. The
// first landing inside p's subtree is number b(p) = the heavy positions in front of p that are not its descendants. So the element
// from p is min(r - b(p), ...) steps up, or gone after depth(p) + 1 moves: every lane moves its own element, no replay.
WV_DEV void g2_flat_flush(const G2Arena& A, G2Flat& F, u32 k, const u32* cand, const u64* pre, const u64* left) {
    const u32 lane = wv::lane();
    if (F.n == F.nh) G2_STAT(1, F.n);
    else G2_STAT(2, F.n);
    if (F.n != 0) {
        const u32 r = F.n;
        G2Ent old[2];
        u32 rp[2], mv[2];
        bool heavy[2];
        WV_UNROLL
        for (u32 h = 0; h < 2u; ++h) {
            const u32 p = lane + 64u * h;
            heavy[h] = p < k && (((h ? F.m1 : F.m0) >> lane) & 1ull) != 0;
            rp[h] = 0; mv[h] = 0;
            old[h] = A.HE[heavy[h] ? p : 0u];
            if (heavy[h]) {
                rp[h] = (u32)wv::popc64(F.m0 & pre[2u * p]) + (u32)wv::popc64(F.m1 & pre[2u * p + 1u]);
                const u32 bp = (u32)wv::popc64(F.m0 & left[2u * p]) + (u32)wv::popc64(F.m1 & left[2u * p + 1u]);
                mv[h] = r > bp ? r - bp : 0u;
            }
        }
        wv::lockstep();   // every lane holds its old elements before any position is rewritten
        WV_UNROLL
        for (u32 h = 0; h < 2u; ++h) {
            const u32 p = lane + 64u * h;
            if (!heavy[h]) continue;
            if (rp[h] < r) {
                G2Ent he;
                he.key = F.lo; he.node = cand[rp[h]]; he.tix = G2_NONE;
                A.HE[p] = he;
            }
            const u32 dp = 31u - (u32)wv::clz32(p + 1u);
            if (mv[h] != 0 && mv[h] <= dp) {
                G2Ent he;   // (field by field: a record handed on as a whole lives in scratch memory)
                he.key = old[h].key; he.node = old[h].node; he.tix = old[h].tix;
                A.HE[((p + 1u) >> mv[h]) - 1u] = he;
            }
        }
    }
    wv::wave_sync();
    F.on = false;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the machine: wave 0
// ---------------------------------------------------------------------------------------------------------------------------
struct G2Post {   // wave 0's view of the ring
    u32 n;        // commands posted
};
WV_DEV void g2_post(G2Mail* mb, G2Post& P, u32 op, u32 gi) {
    if (wv::lane() == 0) {
        mb->op[P.n % G2_RING] = op;
        mb->grp[P.n % G2_RING] = gi;
    }
    wv::wave_sync();
    P.n += 1;
    if (wv::lane() == 0) wv::lds_publish32(&mb->posted, P.n);
}

// A group's Explain command is not waited for where it is posted: the helpers count while the machine writes the group back and admits
// the next one. It is taken before the NEXT group's Explain section — which may overwrite xroot / xadm / cntx, and which comes in front
// of that group's write-back (the command evaluates unlisted nodes from their rows: only ITS OWN group's write-back leaves those alone).
struct G2Pend {
    u32 on, gi, cmd;
};
WV_DEV bool g2_take_explain(const Groups2Args& a, G2Mail* mb, G2Pend& X, u32 nh) {
    if (!X.on) return true;
    if (!g2_wait_ge(&mb->done, X.cmd * nh, mb)) return false;
    const u32 lane = wv::lane();
    if (lane < 8) a.hist[(size_t)X.gi * 8 + lane] += mb->cntx[lane];
    wv::wave_sync();
    X.on = 0;
    return true;
}

// One group, state in the arena A (LDS instance: L == true). Returns false when the launch must end (error / hang).
// gi + 1 < n_groups: `eval_next` is the ring position behind the next group's EVAL command if it was posted ahead (0: it was not,
// dep_prev); on return it always is.
// GEN: the group reserves generic resources (the instance without them carries none of that code)
template <bool L, bool GEN>
WV_DEV bool g2_group(const Groups2Args& a, G2Mail* mb, G2Stage* sg, unsigned char* arena_base, const GroupRec2& G, u32 gi, u64* gt, u64* gx, G2Post& P, u32 nh,
                     u32& eval_next, G2Pend& X) {
    const u32 lane = wv::lane(), N = a.n_nodes, Wn = a.n_words, b = gi & 1u, k = G.k;
    const u32 tbase = a.tree_off[G.tree], ntn = a.tree_off[G.tree + 1] - tbase;
    const bool single = ntn == 1;
    const GroupRec2* Gm = a.g + gi;
    const u32 NG = GEN ? G.n_gen : 0u;   // generic kinds the group reserves
    G2Arena A;
    g2_carve(A, arena_base, G.n_slots, ntn, G.n_gen, a.max_depth, k);
    u64 tk = G2_PROF_ON(a) ? wv::clock64() : 0ull;
#define G2_TICK(q) do { if (G2_PROF_ON(a)) { const u64 n_ = wv::clock64(); gt[q] += n_ - tk; tk = n_; } } while (0)
    // (the admission's own partition: every mark charges the time since the last one to a bucket — [0] waiting for a batch's records,
    // [1] whole chunks appended while a heap of one key fills, [2] flat mode's counting, [3] its flushes, [4] staging, [5] the per-chunk
    // pass, [6] lane 0's replay and the pipelined replacements, [7] the scan over the chunk minima)
    u64 tx = 0;
#define G2_X(q) do { if (G2_PROF_ON(a)) { const u64 n_ = wv::clock64(); gx[q] += n_ - tx; tx = n_; } } while (0)

    // ---------- per-group reset: tree-node arrays, leaf heap offsets (a leaf's heap holds at most min(k, its nodes)) ----------
    if (single) {
        // a tree of one node: its heap takes all the group's slots (the host counted min(k, nodes of the leaf)); decisionTree.tasks is
        // only read by the walk over a tree with levels — nothing to fetch from global memory here
        if (lane == 0) {
            A.tsum[0] = 0;
            A.h_len[0] = 0; A.h_cnt[0] = 0; A.h_adm[0] = 0; A.noroom[0] = 0; A.h_vis[0] = 0;
            A.h_off[0] = 0;
            mb->sh[SH_ERR] = 0; mb->sh[SH_LASTP] = 0;
        }
        const u32 kt = k < G.n_slots ? k : G.n_slots;
        for (u32 i = lane; i < kt; i += 64) A.tcount[i] = 0;
        wv::wave_sync();
    } else {
        u32 run = 0;
        for (u32 i0 = 0; i0 < ntn; i0 += 64) {
            const u32 i = i0 + lane;
            u32 cap = 0;
            if (i < ntn) {
                A.tsum[i] = a.tsumbuf[(size_t)b * a.max_ntn + i];
                A.h_len[i] = 0; A.h_cnt[i] = 0; A.h_adm[i] = 0; A.noroom[i] = 0; A.h_vis[i] = 0;
                if (a.tn_nchild[tbase + i] == 0) cap = min(k, a.tn_nodes[tbase + i]);
            }
            const u32 incl = wv::scan_incl_u32(cap);   // exclusive prefix over the wave
            if (i < ntn) A.h_off[i] = run + incl - cap;
            run += wv::readlane(incl, 63);
        }
        if (run != G.n_slots) {   // the host's slot count and the tree disagree: refuse rather than overrun the arena
            if (lane == 0) a.ctl->error = ERR_GROUP_RANGE;
            return false;
        }
        {
            const u32 kt = k < G.n_slots ? k : G.n_slots;
            for (u32 i = lane; i < kt; i += 64) A.tcount[i] = 0;
        }
        if (lane == 0) { mb->sh[SH_ERR] = 0; mb->sh[SH_LASTP] = 0; }
        wv::wave_sync();
    }
    G2_TICK(1);

    // ---------- tree(): heap admission in node order (nodeset.go:107-120) ----------
    {
        u32 len0 = 0;          // single leaf: its heap length and root key ride in registers (every lane's copy, for the pre-filter)
        u64 root0 = 0;
        u32 lastp = 0;         // node + 1 of the last Process that returned true inside tree() (every lane's copy)
        u32 c_lf = G2_NONE, hbase = 0;   // lane 0 only: the leaf of the last replayed entry, its heap's base / length / root key
        int hlen = 0;
        u64 hroot = 0;
        // One leaf whose heap is full: root replacements in flight, one per lane (see g2_pipe_tick). p_since: ticks since the last one
        // started; p_slots: how many were started (the next one takes lane p_slots % 64).
        bool p_act = false;
        u64 p_key = 0;
        u32 p_node = 0, p_hole = 0, p_since = 2, p_slots = 0;
        bool u_valid = true;   // one leaf, still filling: every element so far has the same key u_key (nothing ever moved) — then a batch of words
        u64 u_key = KEY_NONE;  // whose candidates all have that key too is ONE append (below)
        G2Flat F{false, 0, 0, 0, 0, 0, 0};   // the flat mode (above): tried once, when the one heap has just become full
        u32 flat_fail = 0;   // flat sessions that a foreign candidate ended before they had admitted anything: after two the group gives up on them
        u32* f_cand = reinterpret_cast<u32*>(reinterpret_cast<unsigned char*>(mb) + G2_FLAT_OFF);   // [G2_FLAT_MAXK] the remembered candidates' nodes
        const u64* f_pre = reinterpret_cast<const u64*>(reinterpret_cast<unsigned char*>(mb) + G2_PRE_OFF);   // the post-order tables (k_groups2 builds them)
        const u64* f_left = reinterpret_cast<const u64*>(reinterpret_cast<unsigned char*>(mb) + G2_LEFT_OFF);
        // The scan runs over the group's candidate list (the helpers' EVAL command over the static class list): chunk c = entries
        // [64 c, 64 c + 64) in node order, `minb[c]` a lower bound of the chunk's keys. A lane's candidate is a record {key, node, leaf};
        // a hole (a listed node that fails a dynamic filter, or dropped out since) carries KEY_NONE.
        const u32 M = a.scnt[G.scls], Cn = (M + 63u) / 64u;
        const u64* minb = a.cmin + (size_t)b * Wn;
        const G2Cand* ccb = a.ccand + (size_t)b * N;
        u64* vis = reinterpret_cast<u64*>(reinterpret_cast<unsigned char*>(mb) + 2048);                       // [G2_VISW][64] candidate keys
        u32* visn = reinterpret_cast<u32*>(reinterpret_cast<unsigned char*>(mb) + 2048 + G2_VISW * 64 * 8);   // [G2_VISW][64] nodes
        u32* visl = visn + G2_VISW * 64;                                                                       // [G2_VISW][64] leaves
        if (G2_PROF_ON(a)) tx = wv::clock64();
        for (u32 w0 = 0; w0 < Cn; w0 += 64) {
            // 64 chunks at a time: which of them can hold a candidate at all? (a superset: the root only drops from here on)
            const u64 mk = w0 + lane < Cn ? minb[w0 + lane] : KEY_NONE;
            bool visit = mk != KEY_NONE;
            if (visit && single) visit = len0 < k || mk < root0;
            u64 vm = wv::ballot(visit);
            G2_X(7);
            while (vm) {
              // (the root may have dropped since the 64 chunks were looked at: the ones it rules out by now are not even loaded)
              if (single) vm &= wv::ballot(len0 < k || mk < root0);
              if (vm == 0) break;
              // The candidate keys of the next G2_VISW words to visit go through LDS: their loads are all in flight together, so the
              // replay below never waits for global memory.
              u64 sm = 0;
              u32 vq0 = 0;   // words at the front of the batch that were taken whole
              {
                u64 r[G2_VISW];
                u32 rn[G2_VISW], rl[G2_VISW];
                const u64 tw_ = G2_PROF_ON(a) ? wv::clock64() : 0ull;
                WV_UNROLL
                for (int q = 0; q < G2_VISW; ++q) {
                    r[q] = KEY_NONE;
                    rn[q] = 0;
                    rl[q] = 0;
                    if (vm) {
                        const u32 c = (u32)wv::ffs64(vm);
                        vm &= vm - 1ull;
                        sm |= 1ull << c;
                        const u32 ci_ = (w0 + c) * 64u + lane;
                        if (ci_ < M) {
                            const G2Cand ce = ccb[ci_];
                            r[q] = ce.key;
                            rn[q] = ce.node;
                            rl[q] = ce.leaf;
                        }
                    }
                }
                if (G2_PROF_ON(a)) { wv::wait_vm(); G2_X(0); gx[8] += wv::clock64() - tw_; gx[9] += 1; }
                // Flat mode (above) with nothing but the two keys in the whole batch: the light candidates of all its words are counted
                // and remembered straight from the registers, word after word in node order — no staging, no per-word pass.
                bool batch_done = false;
                // The heap is still filling and every element has the same key (heap.Push moves nothing: the new element is never above
                // its parent): words whose candidates all have that key, and fit, are appended whole, straight from the registers.
                if (single && len0 < k && u_valid) {
                    u64 uk = len0 == 0 ? KEY_NONE : u_key;
                    if (uk == KEY_NONE) {   // an empty heap: its first element names the key
                        WV_UNROLL
                        for (int q = 0; q < G2_VISW; ++q) {
                            const u64 bq = wv::ballot(r[q] != KEY_NONE);
                            if (uk == KEY_NONE && bq) uk = wv::readlane64(r[q], (u32)wv::ffs64(bq));
                        }
                    }
                    if (uk != KEY_NONE) {
                        u64 t = sm;
                        WV_UNROLL
                        for (int q = 0; q < G2_VISW; ++q) {
                            if (t == 0) break;
                            if (wv::ballot(r[q] != KEY_NONE && r[q] != uk)) break;   // a candidate with another key: the per-chunk pass takes over here
                            const bool have = r[q] != KEY_NONE;
                            const u64 bq = wv::ballot(have);
                            const u32 cnt = (u32)wv::popc64(bq);
                            if (len0 + cnt > k) break;   // this chunk fills the heap beyond its size: the per-chunk pass takes over here
                            t &= t - 1ull;
                            ++vq0;
                            if (have) {
                                G2Ent he;
                                he.key = uk; he.node = rn[q]; he.tix = G2_NONE;
                                A.HE[len0 + wv::mbcnt(bq)] = he;
                            }
                            if (bq) {
                                const u32 hib = (u32)(bq >> 32), lob = (u32)bq;
                                const u32 top = hib ? 63u - (u32)wv::clz32(hib) : 31u - (u32)wv::clz32(lob);
                                lastp = wv::readlane(rn[q], top) + 1u;
                                if (len0 == 0) root0 = uk;
                                len0 += cnt;
                                if (G2_PROF_ON(a)) { gt[11] += cnt; gt[14] += cnt; }
                            }
                            if (G2_PROF_ON(a)) gt[10] += 1;
                            G2_STAT(6, 1);
                        }
                        wv::wave_sync();
                        u_key = uk;
                        sm = t;
                        if (sm == 0) batch_done = true;
                    }
                    G2_X(1);
                }
                if (!batch_done && single && !F.on && len0 >= k && k <= G2_FLAT_MAXK && flat_fail < 2u && p_slots == 0u) g2_flat_init(A, F, k);
                if (!batch_done && single && F.on && F.lo != KEY_NONE) {
                    u64 t = sm;
                    WV_UNROLL
                    for (int q = 0; q < G2_VISW; ++q) {
                        if (q < (int)vq0) continue;
                        if (t == 0) break;
                        if (wv::ballot(r[q] != KEY_NONE && r[q] != F.lo && r[q] < F.hi)) break;   // a third key: the per-chunk pass takes over here
                        t &= t - 1ull;
                        ++vq0;
                        const bool isl = r[q] == F.lo;
                        const u32 idx = F.n + wv::mbcnt(wv::ballot(isl));
                        const bool acc = isl && idx < F.nh;   // (admitted while a heavy element is left: the root is heavy until then)
                        if (acc) f_cand[idx] = rn[q];
                        const u64 ab = wv::ballot(acc);
                        if (ab) {
                            const u32 hib = (u32)(ab >> 32), lob = (u32)ab;
                            const u32 top = hib ? 63u - (u32)wv::clz32(hib) : 31u - (u32)wv::clz32(lob);
                            lastp = wv::readlane(rn[q], top) + 1u;   // the last Process that returned true inside tree()
                            F.n += (u32)wv::popc64(ab);
                            if (G2_PROF_ON(a)) gt[12] += (u32)wv::popc64(ab);
                        }
                        if (G2_PROF_ON(a)) gt[10] += 1;
                        G2_STAT(5, 1);
                    }
                    wv::wave_sync();
                    root0 = F.n < F.nh ? F.hi : F.lo;
                    G2_X(2);
                    if (F.n == F.nh) {   // all heavy elements are gone: the remembered candidates enter the heap
                        g2_flat_flush(A, F, k, f_cand, f_pre, f_left);
                        root0 = A.HE[0].key;
                        G2_X(3);
                    }
                    sm = t;
                    if (sm == 0) batch_done = true;
                }
                if (batch_done) {
                    if (G2_PROF_ON(a)) gt[15] += wv::clock64() - tw_;
                    continue;
                }
                WV_UNROLL
                for (int q = 0; q < G2_VISW; ++q) {
                    vis[q * 64 + (int)lane] = r[q];
                    visn[q * 64 + (int)lane] = rn[q];
                    if (!single) visl[q * 64 + (int)lane] = rl[q];
                }
                if (G2_PROF_ON(a)) gt[15] += wv::clock64() - tw_;   // (mostly: waiting for the loads)
                G2_X(4);
              }
              u32 vq = vq0;
              while (sm) {
                G2_X(5);
                const u32 c = (u32)wv::ffs64(sm);
                sm &= sm - 1ull;
                const u64 key = vis[vq * 64u + lane];   // (a lane reads back what it wrote itself)
                const u32 n = visn[vq * 64u + lane];
                u32 leaf = single ? 0u : visl[vq * 64u + lane];
                ++vq;
                const u64 mkc = single ? wv::readlane64(mk, c) : 0ull;
                if (single && !(len0 < k || mkc < root0)) continue;   // the root dropped below the word's minimum since the batch was looked at
                if (single && F.on && F.lo != KEY_NONE && !wv::ballot(key != KEY_NONE && key != F.lo && key < F.hi)) {
                    // a flat session is open and the chunk holds no foreign candidate: its light candidates are counted as the batch path
                    // counts them (no compaction, no staging of the chunk's candidates)
                    const bool isl = key == F.lo;
                    const u32 idx = F.n + wv::mbcnt(wv::ballot(isl));
                    const bool acc = isl && idx < F.nh;   // (admitted while a heavy element is left: the root is heavy until then)
                    if (acc) f_cand[idx] = n;
                    const u64 ab = wv::ballot(acc);
                    if (ab) {
                        const u32 hib = (u32)(ab >> 32), lob = (u32)ab;
                        const u32 top = hib ? 63u - (u32)wv::clz32(hib) : 31u - (u32)wv::clz32(lob);
                        lastp = wv::readlane(n, top) + 1u;   // the last Process that returned true inside tree()
                        F.n += (u32)wv::popc64(ab);
                        if (G2_PROF_ON(a)) gt[12] += (u32)wv::popc64(ab);
                    }
                    if (G2_PROF_ON(a)) gt[10] += 1;
                    G2_STAT(5, 1);
                    wv::wave_sync();
                    root0 = F.n < F.nh ? F.hi : F.lo;
                    if (F.n == F.nh) {   // all heavy elements are gone: the remembered candidates enter the heap
                        G2_X(2);
                        g2_flat_flush(A, F, k, f_cand, f_pre, f_left);
                        root0 = A.HE[0].key;
                        G2_X(3);
                    }
                    continue;
                }
                bool cand = key != KEY_NONE;
                if (cand) {
                    if (single) cand = len0 < k || key < root0;
                    else {
                        const int len = A.h_len[leaf];
                        cand = len < (int)k || key < A.HE[A.h_off[leaf]].key;
                    }
                }
                const u64 bal = wv::ballot(cand);
                if (bal == 0) continue;
                const u32 ne = (u32)wv::popc64(bal), myj = wv::mbcnt(bal);
                if (cand) {
                    G2Ent se;
                    se.key = key; se.node = n; se.tix = leaf;
                    sg->ent[myj] = se;
                }
                wv::wave_sync();
                if (G2_PROF_ON(a)) { gt[10] += 1; gt[11] += ne; }
                // heap.Push moves nothing when the new element is not above its parent — with equal keys all around, the usual case.
                // All lanes check that for their candidate at once (the parent is in the heap already or an earlier candidate of this
                // word); if it holds for every push of the word, the pushes are one parallel append.
                u32 first = 0;
                if (single && len0 < k) {
                    // Candidates [first, ne) are pushed while the heap has room. Every lane checks its own candidate against its parent
                    // (in the heap already, or an earlier candidate of this stretch): the pushes in FRONT of the first one that moves
                    // move nothing — one parallel append. The one that moves is then sifted up BY THE WAVE: lane d holds the ancestor at
                    // depth d of the new position; the ancestors that are less than the new element are the lowest ones on the path (a
                    // max-heap: keys do not grow downwards), each of them goes one step down, the new element takes the place of the
                    // topmost — the same comparisons and the same final array as container/heap's up(), one LDS round trip instead of
                    // one per level. Then the parallel check again, from the next candidate on.
                    for (;;) {
                        const u32 np = min(ne - first, k - len0);
                        if (np == 0) break;
                        bool moves = false;
                        const bool mine = cand && myj >= first && myj < first + np;
                        if (mine) {
                            const u32 pos = len0 + (myj - first);
                            if (pos > 0) {
                                const u32 pp = (pos - 1u) >> 1;
                                const u64 pk = pp < len0 ? A.HE[pp].key : sg->ent[first + (pp - len0)].key;
                                moves = pk < key;   // Less(j, i)
                            }
                        }
                        const u64 mvb = wv::ballot(moves);
                        const u32 npp = mvb ? wv::readlane(myj, (u32)wv::ffs64(mvb)) - first : np;
                        if (npp != 0) {
                            if (mine && myj < first + npp) {
                                G2Ent he;
                                he.key = key; he.node = n; he.tix = G2_NONE;
                                A.HE[len0 + (myj - first)] = he;
                            }
                            if (len0 == 0) { root0 = sg->ent[first].key; u_key = root0; }
                            if (wv::ballot(mine && myj < first + npp && key != u_key)) u_valid = false;
                            lastp = sg->ent[first + npp - 1u].node + 1u;
                            len0 += npp;
                            first += npp;
                            if (G2_PROF_ON(a)) gt[14] += npp;
                            wv::wave_sync();
                        }
                        if (npp == np) break;
                        // candidate `first` goes to position len0 and moves up
                        const G2Ent e = sg->ent[first];
                        const u32 j1 = len0 + 1u, dj = 31u - (u32)wv::clz32(j1);   // (1-based index; its ancestor at depth d is j1 >> (dj - d))
                        const bool anc = lane < dj;
                        const G2Ent ae = A.HE[anc ? (j1 >> (dj - lane)) - 1u : 0u];
                        const bool less = anc && ae.key < e.key;
                        const u32 mv = (u32)wv::popc64(wv::ballot(less));
                        wv::lockstep();   // every lane has read its ancestor before any lane overwrites one
                        if (less) A.HE[(j1 >> (dj - lane - 1u)) - 1u] = ae;
                        if (lane == 0) {
                            G2Ent he;
                            he.key = e.key; he.node = e.node; he.tix = G2_NONE;
                            A.HE[(j1 >> mv) - 1u] = he;
                        }
                        if (mv == dj) root0 = e.key;
                        u_valid = false;
                        lastp = e.node + 1u;
                        len0 += 1u;
                        first += 1u;
                        if (G2_PROF_ON(a)) gt[12] += 1;
                        G2_STAT(7, 1);
                        wv::wave_sync();
                    }
                }
                if (first == ne) continue;
                // ---- the heap is full, one leaf, k <= 128: FLAT sessions over the chunk's remaining candidates (see G2Flat). A session ends when
                // all heavy elements are replaced (nothing else of the chunk can enter then: the root is light), or in front of a FOREIGN
                // candidate — admitted, with a key other than the light one: the session's candidates enter the heap (g2_flat_flush: closed
                // form, whether or not heavy elements are left), lane 0 replaces the root by that ONE candidate the ordinary way, and the
                // next session starts from the heap as it is then.
                while (first < ne && single && len0 >= k && k <= G2_FLAT_MAXK && flat_fail < 2u && p_slots == 0u) {
                    if (!F.on) g2_flat_init(A, F, k);
                    // the chunk's remaining candidates in lane order (= node order: the compaction keeps it)
                    const bool mineC = cand && myj >= first;
                    if (F.lo == KEY_NONE) {   // every element is heavy: the first candidate below the root names the light key
                        const u64 b0 = wv::ballot(mineC && key < F.hi);
                        if (b0) F.lo = wv::readlane64(key, (u32)wv::ffs64(b0));
                    }
                    const bool is_lo = mineC && F.lo != KEY_NONE && key == F.lo;
                    const u32 before = wv::mbcnt(wv::ballot(is_lo));          // light candidates in front of this lane
                    const bool root_heavy = F.n + before < F.nh;              // ... all admitted while a heavy element was left
                    const bool foreign = mineC && key != F.lo && key < (root_heavy ? F.hi : F.lo);   // admitted, with another key
                    const u64 fb = wv::ballot(foreign);
                    const u32 fl = fb ? (u32)wv::ffs64(fb) : 64u;
                    const bool acc = is_lo && root_heavy && lane < fl;
                    const u64 ab = wv::ballot(acc);
                    if (acc) f_cand[F.n + before] = n;
                    if (ab) {
                        u32 hib = (u32)(ab >> 32), lob = (u32)ab;
                        const u32 top = hib ? 63u - (u32)wv::clz32(hib) : 31u - (u32)wv::clz32(lob);   // the last admitted lane
                        lastp = wv::readlane(n, top) + 1u;       // the last Process that returned true inside tree()
                        F.n += (u32)wv::popc64(ab);
                        if (G2_PROF_ON(a)) gt[12] += (u32)wv::popc64(ab);
                    }
                    wv::wave_sync();
                    root0 = F.n < F.nh ? F.hi : F.lo;
                    if (fb == 0 && F.n < F.nh) { first = ne; break; }   // the chunk is through, the session goes on
                    if (fb) G2_STAT(3, 1);
                    if (fb && F.n == 0) ++flat_fail;
                    G2_X(5);
                    g2_flat_flush(A, F, k, f_cand, f_pre, f_left);
                    G2_X(3);
                    if (fb == 0) { root0 = A.HE[0].key; first = ne; break; }   // every heavy element is replaced: the root is light, nothing else of the chunk enters
                    // the foreign candidate: heap.Fix(0) with it at the root (it is less than the root: that is what made it foreign)
                    first = (u32)wv::popc64(bal & ((1ull << fl) - 1ull));   // (fl < 64 here)
                    u64 nroot = 0;
                    if (lane == 0) {
                        const G2Ent fe = sg->ent[first];
                        G2Ent he;
                        he.key = fe.key; he.node = fe.node; he.tix = G2_NONE;
                        nroot = g2_down_val(A, 0u, 0, (int)k, he).top.key;
                    }
                    root0 = ((u64)wv::readfirstlane((u32)(nroot >> 32)) << 32) | wv::readfirstlane((u32)nroot);
                    lastp = sg->ent[first].node + 1u;
                    first += 1u;
                    if (G2_PROF_ON(a)) gt[12] += 1;
                    G2_STAT(4, 1);
                    wv::wave_sync();
                    G2_X(6);
                }
                if (first == ne) continue;
                if (single && len0 >= k) {
                    // ---- the heap is full: every remaining candidate either replaces the root (heap.Fix(0), a sift from the top) or is not
                    // less than it. A sift only ever touches deeper levels as it goes, so the next replacement may start two steps behind
                    // the previous one: the operation that started 2j steps ago works on level 2j while the new one works on level 0
                    // (it reads level 1, which the one before it finished with a step ago). Lanes hold the operations in flight; every
                    // tick all of them take one step. Same comparisons, same writes, same final array as one after the other.
                    const u64 ta_ = G2_PROF_ON(a) ? wv::clock64() : 0ull;
                    G2_X(5);
                    const G2Ent mine = sg->ent[lane < ne ? lane : 0u];   // lane j looks after candidate j of the word
                    u32 ci = first;
                    while (ci < ne) {
                        // one step of every operation in flight
                        {
                            const u32 j1 = 2u * p_hole + 1u;
                            const bool has = p_act && j1 < k;
                            const G2Ent c1 = A.HE[has ? j1 : 0u];
                            const G2Ent c2 = A.HE[has && j1 + 1u < k ? j1 + 1u : 0u];
                            const bool right = j1 + 1u < k && c1.key < c2.key;   // Less(j2, j1)
                            const G2Ent cj = g2_pick(right, c2, c1);
                            const bool moves = has && p_key < cj.key;             // Less(j, i)
                            G2Ent pe;
                            pe.key = p_key; pe.node = p_node; pe.tix = G2_NONE;
                            if (p_act) A.HE[p_hole] = g2_pick(moves, cj, pe);
                            p_hole = moves ? (right ? j1 + 1u : j1) : p_hole;
                            p_act = moves;
                        }
                        ++p_since;
                        wv::lockstep();   // what this tick wrote is what the next tick reads
                        if (p_since >= 2u) {   // the next candidate that is less than the root starts; those in front of it never enter
                            const u64 root = A.HE[0].key;
                            const u64 hb = wv::ballot(lane >= ci && lane < ne && mine.key < root);
                            root0 = root;
                            if (hb == 0) ci = ne;
                            else {
                                const u32 f = (u32)wv::ffs64(hb);
                                const u64 ek = wv::readlane64(mine.key, f);
                                const u32 en = wv::readlane(mine.node, f);
                                if (lane == (p_slots & 63u)) { p_act = true; p_key = ek; p_node = en; p_hole = 0; }
                                ++p_slots;
                                p_since = 0;
                                lastp = en + 1u;       // the last Process that returned true inside tree()
                                ci = f + 1u;
                                G2_STAT(4, 1);
                                if (G2_PROF_ON(a)) gt[12] += 1;
                            }
                        }
                    }
                    if (G2_PROF_ON(a)) gt[13] += wv::clock64() - ta_;
                    G2_X(6);
                    continue;
                }
                u_valid = false;
                G2_X(5);   // (lane 0 replays pushes that move something: the heap is no longer known to hold one key)
                if (lane == 0) {
                    const u64 ta_ = G2_PROF_ON(a) ? wv::clock64() : 0ull;
                    if (single) { c_lf = 0; hbase = 0; hlen = (int)len0; hroot = root0; }
                    G2Ent cur = sg->ent[first];
                    for (u32 i = first; i < ne; ++i) {
                        const G2Ent e = cur;
                        if (i + 1 < ne) cur = sg->ent[i + 1];   // in flight while this one is replayed
                        const u32 lf = e.tix;
                        if (lf != c_lf) {
                            if (c_lf != G2_NONE) A.h_len[c_lf] = hlen;
                            c_lf = lf;
                            hbase = A.h_off[lf];
                            hlen = A.h_len[lf];
                            hroot = hlen ? A.HE[hbase].key : 0ull;
                        }
                        G2Ent he;
                        he.key = e.key; he.node = e.node; he.tix = G2_NONE;
                        // Equal keys are the rule, so both operations usually end at their first compare: those paths are straight-line.
                        if (hlen < (int)k) {            // heap.Push
                            if (hlen == 0) {
                                A.HE[hbase] = he;
                                hroot = he.key;
                            } else {
                                const int pp = (hlen - 1) >> 1;
                                const G2Ent pe = A.HE[hbase + pp];
                                if (!(pe.key < he.key)) A.HE[hbase + hlen] = he;
                                else {
                                    A.HE[hbase + hlen] = pe;
                                    if (g2_up_val(A, hbase, pp, he) == 0) hroot = he.key;
                                }
                            }
                            ++hlen;
                        } else if (he.key < hroot) {    // replaces the root + heap.Fix(0)
                            if (hlen >= 3) {
                                const G2Ent c1 = A.HE[hbase + 1], c2 = A.HE[hbase + 2];
                                const bool right = c1.key < c2.key;
                                const G2Ent cj = g2_pick(right, c2, c1);
                                if (!(he.key < cj.key)) {
                                    A.HE[hbase] = he;
                                    hroot = he.key;
                                } else {
                                    A.HE[hbase] = cj;
                                    hroot = cj.key;
                                    g2_down_val(A, hbase, right ? 2 : 1, hlen, he);
                                }
                            } else hroot = g2_down_val(A, hbase, 0, hlen, he).top.key;
                        } else continue;
                        lastp = e.node + 1;            // the last Process that returned true inside tree()
                        if (G2_PROF_ON(a)) gt[12] += 1;
                    }
                    if (G2_PROF_ON(a)) gt[13] += wv::clock64() - ta_;
                    if (!single && c_lf != G2_NONE) A.h_len[c_lf] = hlen;   // the other lanes' pre-filter reads the lengths of all leaves
                }
                G2_X(6);
                lastp = wv::readfirstlane(lastp);
                if (single) {   // ... and lane 0's registers when there is one leaf
                    len0 = wv::readfirstlane((u32)hlen);
                    root0 = ((u64)wv::readfirstlane((u32)(hroot >> 32)) << 32) | wv::readfirstlane((u32)hroot);
                } else wv::wave_sync();
              }
              G2_X(5);
            }
        }
        G2_X(7);
        if (F.on) { g2_flat_flush(A, F, k, f_cand, f_pre, f_left); G2_X(3); }   // the stream ended in flat mode
        while (wv::ballot(p_act)) {   // the replacements still in flight run to their ends
            const u32 j1 = 2u * p_hole + 1u;
            const bool has = p_act && j1 < k;
            const G2Ent c1 = A.HE[has ? j1 : 0u];
            const G2Ent c2 = A.HE[has && j1 + 1u < k ? j1 + 1u : 0u];
            const bool right = j1 + 1u < k && c1.key < c2.key;
            const G2Ent cj = g2_pick(right, c2, c1);
            const bool moves = has && p_key < cj.key;
            G2Ent pe;
            pe.key = p_key; pe.node = p_node; pe.tix = G2_NONE;
            if (p_act) A.HE[p_hole] = g2_pick(moves, cj, pe);
            p_hole = moves ? (right ? j1 + 1u : j1) : p_hole;
            p_act = moves;
            wv::lockstep();
        }
        if (lane == 0) {
            if (single) A.h_len[0] = (int)len0;
            else if (c_lf != G2_NONE) A.h_len[c_lf] = hlen;
        }
        if (lane == 0) mb->sh[SH_LASTP] = lastp;
        wv::wave_sync();
    }
    G2_TICK(2);
    // Every leaf is heap-sorted NOW, a lane per leaf (the reference pops a leaf's heap when scheduleNTasksOnSubtree first reaches it,
    // decision_tree.go:46-49; no Process call is involved, so doing it for all leaves at once — and for leaves the walk never
    // reaches — is unobservable). Before that: heap roots and lengths as tree() left them, for the Explain pass. After it: the residuals
    // of every node that sits in a heap, by position.
    {
        for (u32 i = lane; i < ntn; i += 64) {
            const int len = A.h_len[i];
            const u32 base = A.h_off[i];
            A.rootkey[i] = len ? A.HE[base].key : 0ull;
            A.h_adm[i] = len;
            A.h_cnt[i] = len;
            if (single) continue;
            g2_pop_all(A, base, len);
            for (int q = 0; q < len; ++q) {
                const u32 pos = base + (u32)q, n = A.HE[pos].node;
                G2Res r;
                r.cpu = a.cpu[n]; r.mem = a.mem[n];
                A.PS[pos] = r;
                for (u32 z = 0; z < NG; ++z) A.gen[(size_t)pos * NG + z] = a.gcnt[(size_t)Gm->gkind[z] * a.gstride + n];
            }
        }
        wv::wave_sync();
        if (G2_PROF_ON(a)) tx = wv::clock64();
        if (single) {
            // One leaf. When every key in the heap is the same, no pop moves anything but the two elements it swaps, and the heap-sort
            // comes out as a rotation by one — new[j] = old[(j + 1) % len] — which all lanes do together; otherwise lane 0 pops.
            const u32 len = (u32)A.h_cnt[0];
            const u64 k0 = len ? A.HE[0].key : 0ull;
            bool differs = false;
            for (u32 pos = lane; pos < len; pos += 64) differs |= A.HE[pos].key != k0;
            bool same_ = wv::ballot(differs) == 0;
            G2_X(10);
            u32 rl_ = len;   // the heap's front [0, rl_) still to be sorted when the keys in it are all the same
            if (!same_ && len <= G2_FLAT_MAXK) {
                // TWO keys (the root's: "heavy", and one other): heap-sort by the WAVE. While a heavy element is left the root is heavy. A pop
                // takes the last element x: if x is heavy it stays at the root (no child is greater); if it is light it sifts down — to the
                // left child if that is heavy, else to the right one if that is heavy, else it stays: it comes to rest at the heavy position
                // a post-order walk visits first, and the heavy elements on its way move up one place (the flat mode's argument, G2Flat).
                // Either way one heavy position goes: the path is known from the MASK of heavy positions alone, and the moves of one pop are
                // one LDS round trip for the lanes along the path. When the front that is left holds one key, the rotation finishes it.
                const u64 kk0 = lane < len ? A.HE[lane].key : k0, kk1 = lane + 64u < len ? A.HE[lane + 64u].key : k0;
                const u64 o0 = kk0 != k0 ? kk0 : KEY_NONE, o1 = kk1 != k0 ? kk1 : KEY_NONE;
                const u64 lo_ = g2_wave_min64(o0 < o1 ? o0 : o1);
                if (wv::ballot((kk0 != k0 && kk0 != lo_) || (kk1 != k0 && kk1 != lo_)) == 0) {
                    const u64* pre_ = reinterpret_cast<const u64*>(reinterpret_cast<unsigned char*>(mb) + G2_PRE_OFF);
                    const u64* left_ = reinterpret_cast<const u64*>(reinterpret_cast<unsigned char*>(mb) + G2_LEFT_OFF);
                    u64 s0 = wv::ballot(lane < len && kk0 == k0), s1 = wv::ballot(lane + 64u < len && kk1 == k0);
                    u32 m = len - 1u;
                    for (;;) {
                        const u32 hcnt = (u32)wv::popc64(s0) + (u32)wv::popc64(s1);
                        if (hcnt == 0u || hcnt == m + 1u) break;   // one key in [0, m]
                        const bool last_heavy = ((m < 64u ? s0 >> m : s1 >> (m - 64u)) & 1ull) != 0;
                        if (last_heavy) {   // x stays at the root, the root goes to m: positions 0 and m change places
                            const bool act = lane == 0u || lane == 63u;
                            const G2Ent e = A.HE[lane == 0u ? m : 0u];
                            wv::lockstep();
                            if (act) {
                                G2Ent he;
                                he.key = e.key; he.node = e.node; he.tix = e.tix;
                                A.HE[lane == 0u ? 0u : m] = he;
                            }
                            wv::wave_sync();
                            if (m < 64u) s0 &= ~(1ull << m);
                            else s1 &= ~(1ull << (m - 64u));
                            m -= 1u;
                            G2_STAT(8, 1);
                            continue;
                        }
                        // A RUN of light pops: the positions m, m - 1, ... down to the highest heavy one are light, so the next t pops (t <= the
                        // heavy elements left) each take a light x — t root replacements in a row with the "candidates" HE[m], HE[m - 1], ...:
                        // g2_flat_flush's closed form (x_j comes to rest at the j-th heavy position in post-order; the element from heavy
                        // position p moves up once per pop from pop b(p) on), and the element that leaves at the root in pop j — the one
                        // with b(p) + depth(p) == j — is what the pop puts at position m - j.
                        const u32 hb = s1 ? 64u + g2_fls64(s1) : g2_fls64(s0);   // the highest heavy position (below m: m is light)
                        const u32 t = min(m - hb, hcnt);
                        G2Ent old[2], xin[2];
                        u32 rp[2], mv[2], bp[2];
                        bool hv[2];
                        WV_UNROLL
                        for (u32 h = 0; h < 2u; ++h) {
                            const u32 p = lane + 64u * h;
                            hv[h] = (((h ? s1 : s0) >> lane) & 1ull) != 0;
                            rp[h] = 0; mv[h] = 0; bp[h] = 0;
                            if (hv[h]) {
                                rp[h] = (u32)wv::popc64(s0 & pre_[2u * p]) + (u32)wv::popc64(s1 & pre_[2u * p + 1u]);
                                bp[h] = (u32)wv::popc64(s0 & left_[2u * p]) + (u32)wv::popc64(s1 & left_[2u * p + 1u]);
                                mv[h] = t > bp[h] ? t - bp[h] : 0u;
                            }
                            old[h] = A.HE[hv[h] ? p : 0u];
                            xin[h] = A.HE[hv[h] && rp[h] < t ? m - rp[h] : 0u];
                        }
                        wv::lockstep();   // every lane holds what it needs before any position is rewritten
                        WV_UNROLL
                        for (u32 h = 0; h < 2u; ++h) {
                            const u32 p = lane + 64u * h;
                            if (!hv[h]) continue;
                            if (rp[h] < t) {
                                G2Ent he;
                                he.key = xin[h].key; he.node = xin[h].node; he.tix = xin[h].tix;
                                A.HE[p] = he;
                            }
                            if (mv[h] != 0) {
                                const u32 dp = 31u - (u32)wv::clz32(p + 1u);
                                G2Ent he;
                                he.key = old[h].key; he.node = old[h].node; he.tix = old[h].tix;
                                A.HE[mv[h] <= dp ? ((p + 1u) >> mv[h]) - 1u : m - (bp[h] + dp)] = he;
                            }
                        }
                        wv::wave_sync();
                        s0 = wv::ballot(hv[0] && rp[0] >= t);
                        s1 = wv::ballot(hv[1] && rp[1] >= t);
                        m -= t;
                        G2_STAT(8, t);
                    }
                    rl_ = m + 1u;
                    same_ = true;
                }
            }
            if (same_) {
                if (rl_ >= 2) {
                    const G2Ent head = A.HE[0];
                    for (u32 j0 = 0; j0 < rl_; j0 += 64) {
                        const u32 j = j0 + lane;
                        const G2Ent nx = A.HE[j + 1 < rl_ ? j + 1 : 0u];
                        const G2Ent v = g2_pick(j + 1 < rl_, nx, head);   // (field by field: a record selected as a whole lives in scratch memory)
                        wv::lockstep();   // every lane has read its element before any lane overwrites one
                        if (j < rl_) A.HE[j] = v;
                    }
                }
            } else {
                G2_STAT(9, 1);
                if (lane == 0) g2_pop_all(A, 0u, (int)len);
            }
            wv::wave_sync();
            G2_X(11);
        }
        if (single) {   // one leaf: all lanes load its positions
            const u32 len = (u32)A.h_cnt[0];
            for (u32 pos = lane; pos < len; pos += 64) {
                const u32 n = A.HE[pos].node;
                G2Res r;
                r.cpu = a.cpu[n]; r.mem = a.mem[n];
                A.PS[pos] = r;
                for (u32 z = 0; z < NG; ++z) A.gen[(size_t)pos * NG + z] = a.gcnt[(size_t)Gm->gkind[z] * a.gstride + n];
            }
            wv::wave_sync();
            G2_X(12);
        }
    }
    G2_TICK(3);

    // ---------- the usual fill, done by all lanes at once ----------
    // One leaf, at least as many nodes in it as tasks, every task counts on its node, and every sorted node's key + one task exceeds
    // its successor's: scheduleNTasksOnNodes (scheduler.go:844-924) then gives task j to node j — after each placement the next node is
    // the lesser one (:899-903), it passes Process (it did at admission and nothing touched it since), and the k-th placement returns
    // (:893-895). All lanes verify the key condition for their positions; if it holds, they place their tasks themselves.
    bool filled = false;
    if (single && !(G.flags & RT_UNCOUNTED) && k <= (u32)A.h_cnt[0] && !G.mset) {   // (a task with cluster mounts changes what the NEXT node's Process sees)
        bool off = false;
        for (u32 j = lane; j < k; j += 64) {
            const u64 kj = A.HE[j].key + G2_KEY_STEP;
            if (((kj >> 32) & 0xFFFFFFull) == 0) off = true;               // svcCount would leave its 24 bits: the serial path reports it
            if (j + 1 < k && !(A.HE[j + 1].key < kj)) off = true;          // the fill loop would stay on node j
        }
        if (wv::ballot(off) == 0) {
            for (u32 j = lane; j < k; j += 64) {
                G2Ent e = A.HE[j];
                G2Res r = A.PS[j];
                a.out_node[G.out_off + j] = (int32_t)e.node;
                r.cpu -= G.cpu;
                r.mem -= G.mem;
                A.PS[j] = r;
                for (u32 q = 0; q < NG; ++q) A.gen[(size_t)j * NG + q] -= Gm->gval[q];
                e.key += G2_KEY_STEP;
                e.tix = j;
                A.HE[j] = e;
                A.tnode[j] = e.node;
                A.tcount[j] = 1;
            }
            if (lane == 0) {
                mb->sh[SH_LEFT] = 0; mb->sh[SH_NTOUCH] = k; mb->sh[SH_ERR] = 0;
                mb->sh[SH_C1] = 0; mb->sh[SH_C5] = 0; mb->sh[SH_C6] = 0; mb->sh[SH_C7] = 0; mb->sh[SH_FPASS] = k >= 2 ? 1u : 0u;
            }
            filled = true;
        }
    }

    // ---------- tree walk + fill loops: lane 0, on the arena only ----------
    if (!filled && lane == 0) {
        u32 next_task = 0, ntouch = 0;
        u32 c1 = 0, c5 = 0, c6 = 0, c7 = 0, fpass = 0;   // Explain counters of the fill phase (only Resource / HostPort / MaxReplicas / Volumes can fail there)
        bool bad_key = false;
        const bool has_ports = (G.flags & RT_PORTS) != 0, has_res = (G.flags & RT_RES) != 0, has_maxrep = (G.flags & RT_MAXREP) != 0;
        const bool counted = !(G.flags & RT_UNCOUNTED);
        const u32 ngen = NG;
        // Pipeline.Process on the entry at a heap position (its records are passed in: the fill loop has them in registers): the
        // static filters passed at admission and cannot change
        auto process = [&](u32 pos, G2Ent e, G2Res r) -> bool {
            u32 ff = G2_FF_PASS;
            if (has_res) {
                if (!(G.cpu <= r.cpu && G.mem <= r.mem)) ff = 1;
                else
                    for (u32 q = 0; q < ngen; ++q)
                        if (A.gen[(size_t)pos * ngen + q] < Gm->gval[q]) ff = 1;
            }
            if (ff == G2_FF_PASS) {
                if (has_ports && e.tix != G2_NONE) ff = 5;
                else if (has_maxrep && !((u64)((u32)(e.key >> 32) & 0xFFFFFFu) < G.maxrep)) ff = 6;
            }
            if (ff == G2_FF_PASS && G.mset && !((vol_filter_word(a.vol, G.mset, e.node >> 6) >> (e.node & 63u)) & 1ull)) ff = 7;   // the volumes as the group's placements left them
            if (ff == G2_FF_PASS) { c1 = c5 = c6 = c7 = 0; fpass = 1; }
            else if (ff == 1) ++c1;
            else if (ff == 5) ++c5;
            else if (ff == 6) ++c6;
            else ++c7;
            return ff == G2_FF_PASS;
        };
        // scheduleNTasksOnNodes, scheduler.go:844-924, on the leaf's positions [base, base+cnt). The current entry and the one behind
        // it ride in registers; the one behind is loaded while the current one is worked on (it cannot change meanwhile: a placement
        // only touches the current entry).
        auto fill = [&](int want, u32 base, int cnt) -> int {
            int scheduled = 0, iter = 0, ix = 0;
            u32 nfailed = 0;
            for (u32 q = base >> 6; q <= (base + (u32)cnt - 1u) >> 6; ++q) A.failed[q] = 0;
            G2Ent ce = A.HE[base];
            G2Res cr = A.PS[base];
            int nx = cnt > 1 ? 1 : 0;
            G2Ent ne = A.HE[base + (u32)nx];
            G2Res nr = A.PS[base + (u32)nx];
            const u64 kstep = counted ? G2_KEY_STEP : 0ull;
            const bool one = cnt == 1;
            while (next_task < k) {
                const u32 pos = base + (u32)ix;
                a.out_node[G.out_off + next_task] = (int32_t)ce.node;
                if (G.mset) {   // chooseTaskVolumes + reserveTaskVolumes on the node (scheduler.go:857-874); a mount without a volume: no attachments
                    u32 att[VOL_MAX_MOUNTS];
                    const u32 na = vol_choose(a.vol, G.mset, ce.node, att, nullptr);
                    if (na) vol_reserve(a.vol, G.mset, ce.node, att, na);
                    for (u32 q = 0; q < VOL_MAX_MOUNTS; ++q) a.att[((size_t)G.att_off + next_task) * VOL_MAX_MOUNTS + q] = att[q];
                }
                ++next_task;
                cr.cpu -= G.cpu;   // NodeInfo.addTask (nodeinfo.go:108-154)
                cr.mem -= G.mem;
                A.PS[pos] = cr;
                for (u32 q = 0; q < ngen; ++q) A.gen[(size_t)pos * ngen + q] -= Gm->gval[q];   // Claim, resource_management.go:11-39 (counts)
                // the node's entry in the touched list (tcount starts all zero): no branch on "first task here"
                const bool fresh = ce.tix == G2_NONE;
                const u32 t = fresh ? ntouch : ce.tix;
                ntouch += fresh ? 1u : 0u;
                ce.tix = t;
                A.tnode[t] = ce.node;
                wv::lds_add32(&A.tcount[t], 1u);
                ce.key += kstep;
                bad_key |= counted && ((ce.key >> 32) & 0xFFFFFFull) == 0;   // svcCount left its 24 bits
                A.HE[pos] = ce;
                ++scheduled;
                if (scheduled == want) return scheduled;
                if (one) { ne = ce; nr = cr; }   // the entry behind the only entry is that entry
                // first pass: on to the next node once it is the lesser; later passes: round robin
                if (iter + 1 >= cnt || ne.key < ce.key) {
                    ++iter;
                    ix = nx; ce = ne; cr = nr;
                    nx = ix + 1 == cnt ? 0 : ix + 1;
                    ne = A.HE[base + (u32)nx]; nr = A.PS[base + (u32)nx];
                }
                const int orig = iter;
                for (;;) {
                    const u32 bi = base + (u32)ix;
                    const bool bad = nfailed != 0 && ((A.failed[bi >> 6] >> (bi & 63)) & 1ull);
                    if (!bad && process(bi, ce, cr)) break;
                    if (!bad) { A.failed[bi >> 6] |= 1ull << (bi & 63); ++nfailed; }
                    ++iter;
                    ix = nx; ce = ne; cr = nr;
                    nx = ix + 1 == cnt ? 0 : ix + 1;
                    ne = A.HE[base + (u32)nx]; nr = A.PS[base + (u32)nx];
                    if (iter - orig == cnt) return scheduled;
                }
            }
            return scheduled;
        };
        // decisionTree.orderedNodes, decision_tree.go:24-52. The first hand-out of a leaf finds it sorted already (above); from the
        // second one on the nodes that no longer pass are dropped, the rest is heapified and popped again.
        auto ordered = [&](u32 lf) -> int {
            int cnt = A.h_cnt[lf];
            if (!A.h_vis[lf]) {
                A.h_vis[lf] = 1;
                return cnt;
            }
            if (cnt == 0) return 0;
            const u32 base = A.h_off[lf];
            for (int i = 0; i < cnt;) {
                const G2Ent e = A.HE[base + i];
                const G2Res r = A.PS[base + i];
                if (process(base + (u32)i, e, r)) ++i;
                else {
                    --cnt;
                    if (i != cnt) g2_move_full(A, ngen, base + (u32)i, base + (u32)cnt);   // nodes[i] = nodes[last]; nodes = nodes[:last]
                }
            }
            for (int i = cnt / 2 - 1; i >= 0; --i) g2_down_full(A, ngen, base, i, cnt);   // heap.Init
            for (int n = cnt - 1; n >= 1; --n) {                                          // heap.Pop until empty
                const G2Ent te = A.HE[base]; A.HE[base] = A.HE[base + n]; A.HE[base + n] = te;
                const G2Res tr = A.PS[base]; A.PS[base] = A.PS[base + n]; A.PS[base + n] = tr;
                for (u32 q = 0; q < ngen; ++q) {
                    const int32_t tg = A.gen[(size_t)base * ngen + q];
                    A.gen[(size_t)base * ngen + q] = A.gen[(size_t)(base + n) * ngen + q];
                    A.gen[(size_t)(base + n) * ngen + q] = tg;
                }
                g2_down_full(A, ngen, base, 0, n);
            }
            A.h_cnt[lf] = cnt;
            return cnt;
        };
        // scheduleNTasksOnSubtree, scheduler.go:772-825, as an explicit stack machine (frames in the arena: any depth)
        int sp = 0, ret = 0;
        {
            G2Frame& f0 = A.st[0];
            f0.tn = 0; f0.n = (int)k; f0.scheduled = 0; f0.assign = 0; f0.usable = 0; f0.desired = 0; f0.rem = 0; f0.child = G2_NONE;
            f0.n_noroom = 0; f0.converging = 1; f0.phase = 0;
        }
        while (sp >= 0) {
            G2Frame& f = A.st[sp];
            const u32 nch = a.tn_nchild[tbase + f.tn];
            if (f.phase == 0) {
                if (nch == 0) {   // leaf
                    const u64 t0_ = G2_PROF_ON(a) ? wv::clock64() : 0ull;
                    const int cnt = ordered(f.tn);
                    const u64 t1_ = G2_PROF_ON(a) ? wv::clock64() : 0ull;
                    ret = cnt == 0 ? 0 : fill(f.n, A.h_off[f.tn], cnt);
                    if (G2_PROF_ON(a)) { gt[8] += t1_ - t0_; gt[9] += wv::clock64() - t1_; }
                    --sp;
                    continue;
                }
                f.scheduled = 0;
                f.usable = A.tsum[f.tn];
                f.n_noroom = 0;   // var noRoom map[*decisionTree]struct{} — fresh per invocation
                for (u32 c = a.tn_first[tbase + f.tn]; c != G2_NONE; c = a.tn_next[tbase + c]) A.noroom[c] = 0;
                f.converging = 1;
                f.phase = 1;
            }
            if (f.phase == 3) {   // a child call returned `ret`
                if (ret < f.assign) {
                    A.noroom[f.child] = 1;
                    f.n_noroom++;
                    f.usable -= A.tsum[f.child];
                } else if (f.rem > 0) f.rem--;
                f.scheduled += ret;
                f.child = a.tn_next[tbase + f.child];
                f.phase = 2;
            }
            if (f.phase == 1) {   // while condition + per-round quantities
                const int room = (int)nch - f.n_noroom;
                if (!(f.scheduled != f.n && room != 0 && f.converging)) {
                    ret = f.scheduled;
                    --sp;
                    continue;
                }
                const i64 tot = f.usable + f.n - f.scheduled;
                f.desired = tot / room;
                f.rem = tot % room;
                f.converging = 0;
                f.child = a.tn_first[tbase + f.tn];
                f.phase = 2;
            }
            // phase 2: `for _, subtree := range tree.next` in creation order
            bool called = false;
            while (f.child != G2_NONE) {
                if (!A.noroom[f.child]) {
                    const i64 sub = A.tsum[f.child];
                    if (sub < f.desired || (sub == f.desired && f.rem > 0)) {
                        f.converging = 1;
                        f.assign = (int)(f.desired - sub) + (f.rem > 0 ? 1 : 0);
                        f.phase = 3;
                        G2Frame& c = A.st[sp + 1];
                        c.tn = f.child; c.n = f.assign; c.scheduled = 0; c.assign = 0; c.usable = 0; c.desired = 0; c.rem = 0; c.child = G2_NONE;
                        c.n_noroom = 0; c.converging = 1; c.phase = 0;
                        ++sp;
                        called = true;
                        break;
                    }
                }
                f.child = a.tn_next[tbase + f.child];
            }
            if (!called) f.phase = 1;
        }
        for (u32 i = next_task; i < k; ++i) a.out_node[G.out_off + i] = -1;
        mb->sh[SH_LEFT] = k - next_task;
        mb->sh[SH_NTOUCH] = ntouch;
        mb->sh[SH_ERR] = bad_key ? 1u : 0u;
        mb->sh[SH_C1] = c1; mb->sh[SH_C5] = c5; mb->sh[SH_C6] = c6; mb->sh[SH_C7] = c7; mb->sh[SH_FPASS] = fpass;
    }
    wv::wave_sync();
    if (mb->sh[SH_ERR]) {
        if (lane == 0) a.ctl->error = ERR_GROUP_RANGE;
        return false;
    }
    G2_TICK(4);

    // ---------- Explain counters for a group with leftovers (pipeline.go:56-68 call sequence: tree()'s calls, then the fill phase's) ----------
    if (!g2_take_explain(a, mb, X, nh)) return false;   // (the group before)
    if (mb->sh[SH_LEFT] > 0) {
        const u32 fpass = mb->sh[SH_FPASS];
        if (lane < 8) {   // the fill phase's share now, tree()'s share when the helpers have counted it (g2_take_explain)
            u32 v = 0;
            if (lane == 1) v = mb->sh[SH_C1];
            if (lane == 5) v = mb->sh[SH_C5];
            if (lane == 6) v = mb->sh[SH_C6];
            if (lane == 7) v = mb->sh[SH_C7];
            a.hist[(size_t)gi * 8 + lane] = v;
        }
        if (!fpass) {   // no Process passed after tree(): the failing calls inside tree() behind its last passing one count, too
            for (u32 i = lane; i < ntn; i += 64) { a.xroot[i] = A.rootkey[i]; a.xadm[i] = A.h_adm[i]; }
            if (lane < 8) mb->cntx[lane] = 0;
            if (lane == 0) { mb->x_lastp = mb->sh[SH_LASTP]; mb->x_k = k; }
            wv::wait_vm();
            g2_post(mb, P, G2_OP_EXPLAIN, gi);
            X.on = 1; X.gi = gi; X.cmd = P.n;
        }
    }
    G2_TICK(5);

    // ---------- write-back of the nodes that got a task: node rows, host ports, generic counts, the service's (node, count) list ----------
    const u32 nt = mb->sh[SH_NTOUCH];
    // The usual case — the next group was evaluated ahead, this one takes neither host ports nor generic resources — does the write-back
    // and the next group's patch in ONE pass over the touched nodes: the node's row, its place in the service's list and everything the
    // next group's Process reads are requested together (one memory round trip), the patch is evaluated on the new residuals from
    // registers, then everything is stored.
    const bool fuse = gi + 1 < a.n_groups && eval_next != 0 && !(G.flags & RT_PORTS) && NG == 0;
    if (fuse) {
        const GroupRec2* Gnm = a.g + gi + 1;
        const GroupRec2 Gn = *Gnm;
        const u32 bn = (gi + 1) & 1u;
        const bool counted = !(G.flags & RT_UNCOUNTED);
        const u32 lbase = a.list_off[G.svc];
        u32 lcnt = a.list_cnt[G.svc];
        if (!g2_wait_ge(&mb->done, eval_next * nh, mb)) return false;
        for (u32 i0 = 0; i0 < nt; i0 += 64) {
            // (two nodes a lane and turn — the inputs of both in flight together — was tried: 32 vector registers went to scratch memory
            // elsewhere in the kernel and the tick got no faster)
            const u32 i = i0 + lane;
            const bool act = i < nt;
            const u32 n = act ? A.tnode[i] : 0u, pl = act ? A.tcount[i] : 0u;
            // (after the all-lanes fill task j sits at heap position j: the node's new residuals and task count are in the arena already —
            // three gathers less)
            i64 oc = 0, om = 0;
            u32 ot = 0;
            if (!filled) { oc = a.cpu[n]; om = a.mem[n]; ot = a.total[n]; }
            const G2Res fr = A.PS[filled && act ? i : 0u];
            const u64 fkey = A.HE[filled && act ? i : 0u].key;
            const u32 e1 = a.lpos_dense[(size_t)b * N + n], osv = a.svc_dense[(size_t)b * N + n];
            const u32 cp = a.cpos[(size_t)bn * N + n];
            G2In in;
            g2_load_in<false, true, true>(a, Gn, bn, n, in);
            const i64 nc = filled ? fr.cpu : oc - (i64)pl * G.cpu, nm = filled ? fr.mem : om - (i64)pl * G.mem;   // NodeInfo.addTask's arithmetic for the node's `pl` new tasks (nodeinfo.go:128-153)
            const u32 ntot = filled ? (u32)fkey : ot + (counted ? pl : 0u);
            in.c = nc; in.m = nm; in.tot = ntot;
            u64 key;
            bool as_listed;
            const bool listed = ((in.vw >> (n & 63u)) & 1ull) != 0;   // (on the next group's static class list: only then is there a record to renew)
            const u32 ff = g2_decide<true>(a, Gn, Gnm, n, in, key, as_listed);
            const bool app = act && counted && e1 == 0u;
            if (act) {
                a.cpu[n] = nc;
                a.mem[n] = nm;
                if (counted) {
                    a.total[n] = ntot;
                    if (e1) a.list_svc[e1 - 1u] = osv + pl;   // (the dense column IS the list entry's count since the scatter)
                }
                if (listed) {
                    a.ffbuf[(size_t)bn * N + n] = (unsigned char)ff;
                    a.keybuf[(size_t)bn * N + n] = key;
                    a.ccand[(size_t)bn * N + cp].key = ff == G2_FF_PASS ? key : KEY_NONE;
                }
            }
            const u64 bal = wv::ballot(app);
            if (app) {
                const u32 e = lbase + lcnt + wv::mbcnt(bal);
                a.list_node[e] = n; a.list_svc[e] = pl; a.list_fail[e] = 0;
            }
            lcnt += (u32)wv::popc64(bal);
        }
        if (lane == 0) a.list_cnt[G.svc] = lcnt;
        wv::wait_vm();
        g2_post(mb, P, G2_OP_UNSCATTER, gi);
        G2_TICK(6);
        wv::lockstep();
        G2_TICK(7);
        return true;
    }
    {
        const bool counted = !(G.flags & RT_UNCOUNTED);
        const u32 lbase = a.list_off[G.svc];
        u32 lcnt = a.list_cnt[G.svc];
        for (u32 i0 = 0; i0 < nt; i0 += 64) {
            const u32 i = i0 + lane;
            bool app = false;
            u32 n = 0, pl = 0;
            if (i < nt) {
                n = A.tnode[i];
                pl = A.tcount[i];
                a.cpu[n] -= (i64)pl * G.cpu;   // NodeInfo.addTask's arithmetic for the node's `pl` new tasks (nodeinfo.go:128-153)
                a.mem[n] -= (i64)pl * G.mem;
                for (u32 q = 0; q < NG; ++q) a.gcnt[(size_t)Gm->gkind[q] * a.gstride + n] -= (int32_t)pl * Gm->gval[q];
                if (counted) {
                    a.total[n] += pl;
                    const u32 e1 = a.lpos_dense[(size_t)b * N + n];
                    if (e1) a.list_svc[e1 - 1u] += pl;
                    else app = true;
                }
                if (G.flags & RT_PORTS)
                    for (u32 z = a.pset_off[G.pset]; z < a.pset_off[G.pset + 1]; ++z) wv::g_or64(&a.portmap[(size_t)a.pset_ids[z] * Wn + (n >> 6)], 1ull << (n & 63));
            }
            const u64 bal = wv::ballot(app);
            if (app) {
                const u32 e = lbase + lcnt + wv::mbcnt(bal);
                a.list_node[e] = n; a.list_svc[e] = pl; a.list_fail[e] = 0;
            }
            lcnt += (u32)wv::popc64(bal);
        }
        if (lane == 0) a.list_cnt[G.svc] = lcnt;
        wv::wait_vm();
    }
    g2_post(mb, P, G2_OP_UNSCATTER, gi);
    G2_TICK(6);

    // ---------- the next group: what the helpers evaluated ahead is evaluated again for exactly the nodes this group touched ----------
    if (gi + 1 < a.n_groups) {
        const GroupRec2 Gn = a.g[gi + 1];
        const u32 bn = (gi + 1) & 1u;
        if (eval_next == 0) {   // nothing was prepared ahead (same service): prepare it now, against the rows as they are
            const u32 ntn_n = a.tree_off[Gn.tree + 1] - a.tree_off[Gn.tree];
            for (u32 i = lane; i < ntn_n; i += 64) a.tsumbuf[(size_t)bn * a.max_ntn + i] = 0;
            wv::wait_vm();
            g2_post(mb, P, G2_OP_SCATTER, gi + 1);
            g2_post(mb, P, G2_OP_EVAL, gi + 1);
            eval_next = P.n;
        } else {
            if (!g2_wait_ge(&mb->done, eval_next * nh, mb)) return false;
            for (u32 i = lane; i < nt; i += 64) g2_patch_node(a, Gn, a.g + gi + 1, bn, A.tnode[i]);
            wv::wait_vm();
        }
    }
    wv::lockstep();   // every lane has read what it needs from the arena before lane 0 goes on to reset it for the next group
    G2_TICK(7);
    return true;
#undef G2_TICK
#undef G2_X
}

// the static class lists of a call: a grid of n_scls workgroups of ONE wave each, in front of k_groups2 on the same stream
WV_KERNEL(64) void k_g2_static(Groups2Args a) {
    if (wv::block() < a.n_scls) g2_static_list(a, wv::block());
}

WV_KERNEL(G2_THREADS) void k_groups2(Groups2Args a) {
    unsigned char* l = reinterpret_cast<unsigned char*>(wv::lds());
    G2Mail* mb = reinterpret_cast<G2Mail*>(l);
    G2Stage* sg = reinterpret_cast<G2Stage*>(l + 512);
    static_assert(sizeof(G2Mail) <= 512 && 512 + sizeof(G2Stage) <= 2048 && 2048 + G2_VISW * 64 * 16 <= G2_PRE_OFF && G2_PRE_OFF + 16 * G2_FLAT_MAXK <= G2_FLAT_OFF && G2_FLAT_OFF + 8 * G2_FLAT_MAXK <= G2_LEFT_OFF && G2_LEFT_OFF + 16 * G2_FLAT_MAXK <= G2_LDS_FIXED, "fixed LDS layout");
    const u32 wave = wv::wave(), lane = wv::lane(), nh = wv::nthreads() / 64u - 1u;
    if (wv::tid() == 0) { mb->posted = 0; mb->done = 0; mb->quit = 0; }
    wv::barrier();
    if (a.ctl->error != ERR_NONE || a.n_groups == 0) return;
    if (wave != 0) {
        g2_helper(a, mb, wave - 1u, nh);
        return;
    }
    g2_pre_table(reinterpret_cast<u64*>(l + G2_PRE_OFF), reinterpret_cast<u64*>(l + G2_LEFT_OFF));
    wv::setprio<3>();   // the machine's instructions go first on its SIMD: the helper waves it shares it with only fill the gaps
    u64 gt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 gx[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 tk = G2_PROF_ON(a) ? wv::clock64() : 0ull;
    G2Post P{0};
    {   // the first group is prepared with nothing to overlap
        const GroupRec2 G0 = a.g[0];
        const u32 ntn0 = a.tree_off[G0.tree + 1] - a.tree_off[G0.tree];
        for (u32 i = lane; i < ntn0; i += 64) a.tsumbuf[i] = 0;
        wv::wait_vm();
        g2_post(mb, P, G2_OP_SCATTER, 0);
        g2_post(mb, P, G2_OP_EVAL, 0);
    }
    u32 eval_cur = P.n;
    bool ok = true;
    G2Pend X{0, 0, 0};
    for (u32 gi = 0; gi < a.n_groups && ok; ++gi) {
        if (!g2_wait_ge(&mb->done, eval_cur * nh, mb)) { ok = false; break; }
        if (wv::g_fresh32(&a.ctl->error) != ERR_NONE) { ok = false; break; }
        if (G2_PROF_ON(a)) { const u64 n_ = wv::clock64(); gt[0] += n_ - tk; tk = n_; }
        const GroupRec2 G = a.g[gi];
        u32 eval_next = 0;
        if (gi + 1 < a.n_groups) {
            const GroupRec2 Gn = a.g[gi + 1];
            if (!Gn.dep_prev) {   // the helpers run one group ahead
                const u32 ntn_n = a.tree_off[Gn.tree + 1] - a.tree_off[Gn.tree], bn = (gi + 1) & 1u;
                for (u32 i = lane; i < ntn_n; i += 64) a.tsumbuf[(size_t)bn * a.max_ntn + i] = 0;
                wv::wait_vm();
                g2_post(mb, P, G2_OP_SCATTER, gi + 1);
                g2_post(mb, P, G2_OP_EVAL, gi + 1);
                eval_next = P.n;
            }
        }
        const u32 ntn = a.tree_off[G.tree + 1] - a.tree_off[G.tree];
        const bool in_lds = g2_arena_bytes(G.n_slots, ntn, G.n_gen, a.max_depth, G.k) <= G2_ARENA_LDS;
        if (in_lds && !G.n_gen) ok = g2_group<true, false>(a, mb, sg, l + G2_LDS_FIXED, G, gi, gt, gx, P, nh, eval_next, X);
        else if (in_lds) ok = g2_group<true, true>(a, mb, sg, l + G2_LDS_FIXED, G, gi, gt, gx, P, nh, eval_next, X);
        else ok = g2_group<false, true>(a, mb, sg, a.arena, G, gi, gt, gx, P, nh, eval_next, X);
        eval_cur = eval_next;
        if (G2_PROF_ON(a)) tk = wv::clock64();
    }
    if (ok) ok = g2_take_explain(a, mb, X, nh);   // the last group's
    if (!ok && lane == 0) {
        if (wv::lds_poll32(&mb->quit) && wv::g_fresh32(&a.ctl->error) == ERR_NONE) a.ctl->error = ERR_GROUP_HANG;
        wv::lds_publish32(&mb->quit, 1u);   // helpers leave their wait loops
    }
    g2_post(mb, P, G2_OP_QUIT, 0);
    if (lane == 0 && G2_PROF_ON(a)) {
        for (int q = 0; q < 8; ++q) a.ctl->cyc[q] = gt[q];
        a.ctl->m_cyc[0] = gt[8];   // inside the walk: orderedNodes ...
        a.ctl->m_cyc[1] = gt[9];   // ... and the fill loops
        for (int q = 0; q < 6; ++q) a.ctl->l_cyc[q] = gt[10 + q];
        for (int q = 0; q < 8; ++q) a.ctl->wave_cyc[q] = gx[q];
        a.ctl->wave_cyc[13] = gx[8];
        a.ctl->wave_cyc[14] = gx[9];
        for (int q = 0; q < 4; ++q) a.ctl->wave_cyc[8 + q] = gx[10 + q];   // admission: words visited, candidates staged, heap operations, cycles inside lane 0's replay
    }
}

#endif   // SWP_G2_KERNELS
}  // namespace swpdev
