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

struct GroupRec2 {   // one per group, 144 B
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
    u32 pad[2];
};
static_assert(sizeof(GroupRec2) == 144, "GroupRec2 layout");

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
    return 8 * (3 * (size_t)S + 2 * (size_t)ntn + fw) + 64 * ((size_t)depth + 2) + 4 * (4 * (size_t)S + kt + (size_t)S * ngen + 5 * (size_t)ntn) + 64;
}
// mailbox, staging of one chunk of candidates, a few scalars
#define G2_LDS_FIXED 4096
inline __host__ __device__ size_t g2_lds_bytes() { return (size_t)G2_LDS_FIXED + G2_ARENA_LDS; }

#ifdef SWP_G2_KERNELS
// nodeLess (scheduler.go:708-735) as ONE integer compare: key = (failures if >= 5 else 0, svcCount, total) packed 8 | 24 | 32 bits
// (both sides below 5 failures skip the failure compare; a side at >= 5 loses against any side below).
WV_DEV u64 g2_key(u32 fail, u32 svc, u32 total) {
    const u32 fc = fail >= MAX_FAILURES ? fail : 0u;
    return ((u64)fc << 56) | ((u64)svc << 32) | total;
}
WV_DEV bool g2_key_ok(u32 fail, u32 svc) { return fail < 256u && svc < (1u << 24); }
#define G2_KEY_STEP ((1ull << 32) + 1ull)   // one more task of the service on the node: svcCount + 1, ActiveTasksCount + 1

struct G2Arena {
    u64* HK;            // heap position -> key
    i64* s_cpu;         // slot state (a slot's state never moves; heap positions hold (key, slot))
    i64* s_mem;
    i64* tsum;          // decisionTree.tasks
    u64* rootkey;       // per leaf: heap root key as tree() left it (Explain)
    u64* failed;        // fill loop: positions that failed Process (failedConstraints, scheduler.go:846)
    G2Frame* st;
    u32* HP;            // heap position -> slot
    u32* s_node;
    u32* s_placed;
    u32* live;          // slots ever filled
    u32* touched;       // slots that got a task
    int32_t* s_gen;     // [slot][n_gen] generic counts of the group's kinds
    u32* h_off;         // first heap position of a leaf
    int32_t* h_len;     // nodeMaxHeap.length
    int32_t* h_cnt;     // len(nodeMaxHeap.nodes)
    int32_t* h_adm;     // length as tree() left it
    u32* noroom;        // per tree node: member of its parent's noRoom set (scheduler.go:787,810-813)
};
WV_DEV void g2_carve(G2Arena& A, unsigned char* p, u32 S, u32 ntn, u32 ngen, u32 depth, u32 k) {
    const size_t fw = (size_t)(S + 63) / 64 + 1, kt = k < S ? k : S;
    A.HK = reinterpret_cast<u64*>(p); p += 8 * (size_t)S;
    A.s_cpu = reinterpret_cast<i64*>(p); p += 8 * (size_t)S;
    A.s_mem = reinterpret_cast<i64*>(p); p += 8 * (size_t)S;
    A.tsum = reinterpret_cast<i64*>(p); p += 8 * (size_t)ntn;
    A.rootkey = reinterpret_cast<u64*>(p); p += 8 * (size_t)ntn;
    A.failed = reinterpret_cast<u64*>(p); p += 8 * fw;
    A.st = reinterpret_cast<G2Frame*>(p); p += 64 * ((size_t)depth + 2);
    A.HP = reinterpret_cast<u32*>(p); p += 4 * (size_t)S;
    A.s_node = reinterpret_cast<u32*>(p); p += 4 * (size_t)S;
    A.s_placed = reinterpret_cast<u32*>(p); p += 4 * (size_t)S;
    A.live = reinterpret_cast<u32*>(p); p += 4 * (size_t)S;
    A.touched = reinterpret_cast<u32*>(p); p += 4 * kt;
    A.s_gen = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)S * ngen;
    A.h_off = reinterpret_cast<u32*>(p); p += 4 * (size_t)ntn;
    A.h_len = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)ntn;
    A.h_cnt = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)ntn;
    A.h_adm = reinterpret_cast<int32_t*>(p); p += 4 * (size_t)ntn;
    A.noroom = reinterpret_cast<u32*>(p);
}

// container/heap (go stdlib) over nodeMaxHeap: Less(i,j) = lessFunc(nodes[j], nodes[i]) (nodeheap.go:17-20). up / down move ONE
// element along a path and swap it with what it meets: the element rides in registers and every step copies the other one into the
// hole — the same comparisons and the same final arrangement as the swap sequence, one memory round trip per level (both children
// are requested together).
WV_DEV void g2_up(const G2Arena& A, u32 base, int j0) {
    int j = j0;
    const u64 kv = A.HK[base + j];
    const u32 pv = A.HP[base + j];
    for (;;) {
        const int i = (j - 1) / 2;   // j == 0: i == 0 (Go's integer division truncates, too)
        if (i == j) break;
        const u64 ki = A.HK[base + i];
        const u32 pi = A.HP[base + i];
        if (!(ki < kv)) break;       // !Less(j, i)
        A.HK[base + j] = ki;
        A.HP[base + j] = pi;
        j = i;
    }
    if (j != j0) {
        A.HK[base + j] = kv;
        A.HP[base + j] = pv;
    }
}
// down(i0, n) for the element (kv, pv) that is (conceptually) at i0; the caller has NOT necessarily stored it there. Returns the
// element's final position; it is stored there unless `lazy_same` and it did not move.
WV_DEV int g2_down_val(const G2Arena& A, u32 base, int i0, int n, u64 kv, u32 pv) {
    int i = i0;
    for (;;) {
        const int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        const int j2 = j1 + 1;
        const bool two = j2 < n;
        const u64 k1 = A.HK[base + j1], k2 = two ? A.HK[base + j2] : 0ull;
        const u32 p1 = A.HP[base + j1], p2 = two ? A.HP[base + j2] : 0u;
        const bool right = two && k1 < k2;   // Less(j2, j1)
        const u64 kj = right ? k2 : k1;
        if (!(kv < kj)) break;               // !Less(j, i)
        A.HK[base + i] = kj;
        A.HP[base + i] = right ? p2 : p1;
        i = right ? j2 : j1;
    }
    A.HK[base + i] = kv;
    A.HP[base + i] = pv;
    return i;
}
WV_DEV bool g2_down(const G2Arena& A, u32 base, int i0, int n) {
    return g2_down_val(A, base, i0, n, A.HK[base + i0], A.HP[base + i0]) > i0;
}

// (a group's record is copied to registers field by field; its two small arrays are only ever read through the record in
// memory — `Gm` — because an array indexed by a loop counter would drag the whole copy into scratch memory)
WV_DEV bool g2_res_ok(const Groups2Args& a, const GroupRec2& G, const GroupRec2* Gm, u32 n) {   // ResourceFilter.Check, filter.go:76-95
    if (!(G.cpu <= a.cpu[n] && G.mem <= a.mem[n])) return false;
    for (u32 q = 0; q < G.n_gen; ++q)
        if (a.gcnt[(size_t)Gm->gkind[q] * a.gstride + n] < Gm->gval[q]) return false;   // HasEnough, validate.go:24-52
    return true;
}

// Pipeline.Process on node n for group G (pipeline.go:56-68: the FIRST failing filter in checklist order) and the node's key
WV_DEV void g2_eval_node(const Groups2Args& a, const GroupRec2& G, const GroupRec2* Gm, u32 b, u32 n) {
    const u32 N = a.n_nodes, Wn = a.n_words, w = n >> 6;
    const u64 bit = 1ull << (n & 63);
    u32 ff = G2_FF_PASS;
    u64 key = 0;
    if (!(a.valid[w] & bit)) ff = G2_FF_ABSENT;
    else {
        const u32 sv = a.svc_dense[(size_t)b * N + n], fl = a.fail_dense[(size_t)b * N + n];
        if (!(a.ready[w] & bit)) ff = 0;
        else if ((G.flags & RT_RES) && !g2_res_ok(a, G, Gm, n)) ff = 1;
        else if (G.cls_plug && !(a.plug[(size_t)G.cls_plug * Wn + w] & bit)) ff = 2;
        else if (G.cls_con && !(a.con[(size_t)G.cls_con * Wn + w] & bit)) ff = 3;
        else if (G.cls_plat && !(a.plat[(size_t)G.cls_plat * Wn + w] & bit)) ff = 4;
        else {
            bool busy = false;
            if (G.flags & RT_PORTS)
                for (u32 q = a.pset_off[G.pset]; q < a.pset_off[G.pset + 1]; ++q)
                    if (a.portmap[(size_t)a.pset_ids[q] * Wn + w] & bit) busy = true;
            if (busy) ff = 5;
            else if ((G.flags & RT_MAXREP) && !((u64)sv < G.maxrep)) ff = 6;
        }
        if (!g2_key_ok(fl, sv)) a.ctl->error = ERR_GROUP_RANGE;
        key = g2_key(fl, sv, a.total[n]);
    }
    a.ffbuf[(size_t)b * N + n] = (unsigned char)ff;
    a.keybuf[(size_t)b * N + n] = key;
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
enum { SH_LEN0 = 0, SH_NLIVE = 1, SH_LEFT = 2, SH_NTOUCH = 3, SH_ERR = 4, SH_LASTP = 5, SH_C1 = 6, SH_C5 = 7, SH_C6 = 8, SH_FPASS = 9, SH_ROOT_LO = 10, SH_ROOT_HI = 11 };

struct G2Stage {   // LDS: the candidates of one 64-node chunk, compacted in node order
    u64 key[64];
    u32 node[64];
    u32 leaf[64];
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
            for (u32 w = hid; w < Wn; w += nh) {
                const u32 n = w * 64u + lane;
                if (n < N) g2_eval_node(a, G, Gm, b, n);
            }
        } else if (op == G2_OP_EXPLAIN) {
            // Every passing Process zeroes the counters (pipeline.go:64-66), so only the calls AFTER the last passing one count. Inside
            // tree() the heaps stop changing after that call: a later node was "called" (nodeset.go:108-116) iff its leaf's heap was not
            // full or the node is less than the final root.
            const u32 lastp = mb->x_lastp, k = mb->x_k;
            const u32* leaf_of = a.leaf_of_node + (size_t)G.tree * N;
            for (u32 w = hid; w < Wn; w += nh) {
                const u32 n = w * 64u + lane;
                u32 f = 0xFFu;
                if (n < N && n >= lastp) {
                    const u32 ffn = a.ffbuf[(size_t)b * N + n];
                    if (ffn < 8u) {
                        const u32 lf = leaf_of[n];
                        if (a.xadm[lf] < (int32_t)k || a.keybuf[(size_t)b * N + n] < a.xroot[lf]) f = ffn;
                    }
                }
                for (u32 q = 0; q < 7; ++q) {
                    const u64 bm = wv::ballot(f == q);
                    if (bm && lane == 0) wv::lds_add32(&mb->cntx[q], (u32)wv::popc64(bm));
                }
            }
        }
        if (lane == 0) wv::lds_add_release32(&mb->done, 1u);
    }
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

// One group, state in the arena A (LDS instance: L == true). Returns false when the launch must end (error / hang).
// gi + 1 < n_groups: `eval_next` is the ring position behind the next group's EVAL command if it was posted ahead (0: it was not,
// dep_prev); on return it always is.
template <bool L>
WV_DEV bool g2_group(const Groups2Args& a, G2Mail* mb, G2Stage* sg, unsigned char* arena_base, const GroupRec2& G, u32 gi, u64* gt, G2Post& P, u32 nh,
                     u32& eval_next) {
    const u32 lane = wv::lane(), N = a.n_nodes, Wn = a.n_words, b = gi & 1u, k = G.k;
    const u32 tbase = a.tree_off[G.tree], ntn = a.tree_off[G.tree + 1] - tbase;
    const u32* leaf_of = a.leaf_of_node + (size_t)G.tree * N;
    const unsigned char* ffb = a.ffbuf + (size_t)b * N;
    const u64* keyb = a.keybuf + (size_t)b * N;
    const bool single = ntn == 1;
    const GroupRec2* Gm = a.g + gi;
    G2Arena A;
    g2_carve(A, arena_base, G.n_slots, ntn, G.n_gen, a.max_depth, k);
    u64 tk = (a.dbg & 16u) ? wv::clock64() : 0ull;
#define G2_TICK(q) do { if (a.dbg & 16u) { const u64 n_ = wv::clock64(); gt[q] += n_ - tk; tk = n_; } } while (0)

    // ---------- per-group reset: tree-node arrays, leaf heap offsets (a leaf's heap holds at most min(k, its nodes)) ----------
    {
        u32 run = 0;
        for (u32 i0 = 0; i0 < ntn; i0 += 64) {
            const u32 i = i0 + lane;
            u32 cap = 0;
            if (i < ntn) {
                A.tsum[i] = a.tsumbuf[(size_t)b * a.max_ntn + i];
                A.h_len[i] = 0; A.h_cnt[i] = 0; A.h_adm[i] = 0; A.noroom[i] = 0;
                if (a.tn_nchild[tbase + i] == 0) cap = min(k, a.tn_nodes[tbase + i]);
            }
            // exclusive prefix over the wave (ntn is small: a shuffle-free ballot-per-bit scan would not pay; lanes add through LDS)
            sg->node[lane] = cap;
            wv::wave_sync();
            u32 before = 0;
            for (u32 q = 0; q < lane; ++q) before += sg->node[q];
            u32 tot = 0;
            if (lane == 63) tot = before + cap;
            tot = wv::readlane(tot, 63);
            if (i < ntn) A.h_off[i] = run + before;
            run += tot;
            wv::wave_sync();
        }
        if (run != G.n_slots) {   // the host's slot count and the tree disagree: refuse rather than overrun the arena
            if (lane == 0) a.ctl->error = ERR_GROUP_RANGE;
            return false;
        }
        if (lane == 0) {
            mb->sh[SH_LEN0] = 0; mb->sh[SH_NLIVE] = 0; mb->sh[SH_ERR] = 0; mb->sh[SH_LASTP] = 0;
            mb->sh[SH_ROOT_LO] = 0; mb->sh[SH_ROOT_HI] = 0;
        }
        wv::wave_sync();
    }
    G2_TICK(1);

    // ---------- tree(): heap admission in node order (nodeset.go:107-120) ----------
    {
        u32 len0 = 0;          // single leaf: its heap length and root key ride in registers
        u64 root0 = 0;
        u32 nlive = 0, lastp = 0;   // lane 0 only
        u32 ffn = lane < N ? (u32)ffb[lane] : G2_FF_ABSENT;
        u64 keyn = lane < N ? keyb[lane] : 0ull;
        for (u32 n0 = 0; n0 < N; n0 += 64) {
            const u32 n = n0 + lane;
            const u32 ff = ffn;
            const u64 key = keyn;
            if (n0 + 64 < N) {   // the next chunk's loads are in flight while this one is replayed
                const u32 nn = n + 64;
                ffn = nn < N ? (u32)ffb[nn] : G2_FF_ABSENT;
                keyn = nn < N ? keyb[nn] : 0ull;
            }
            bool cand = ff == G2_FF_PASS;
            u32 leaf = 0;
            if (cand) {
                if (single) cand = len0 < k || key < root0;
                else {
                    leaf = leaf_of[n];
                    const int len = A.h_len[leaf];
                    cand = len < (int)k || key < A.HK[A.h_off[leaf]];
                }
            }
            const u64 bal = wv::ballot(cand);
            if (bal == 0) continue;
            if (cand) {
                const u32 pos = wv::mbcnt(bal);
                sg->key[pos] = key; sg->node[pos] = n; sg->leaf[pos] = leaf;
            }
            wv::wave_sync();
            if (lane == 0) {
                const u32 ne = (u32)wv::popc64(bal);
                // the leaf of the last entry, its heap's base / length / root key ride in registers: a run of entries of one leaf
                // (every group without spread preferences) pays one round trip per entry that does not enter the heap
                u32 c_lf = G2_NONE, base = 0;
                int len = 0;
                u64 root = 0;
                for (u32 i = 0; i < ne; ++i) {
                    const u32 lf = sg->leaf[i];
                    const u64 ek = sg->key[i];
                    if (lf != c_lf) {
                        if (c_lf != G2_NONE) A.h_len[c_lf] = len;
                        c_lf = lf;
                        base = A.h_off[lf];
                        len = A.h_len[lf];
                        root = len ? A.HK[base] : 0ull;
                    }
                    if (len < (int)k) {            // heap.Push: a fresh slot
                        const u32 sl = base + (u32)len;
                        A.HK[sl] = ek; A.HP[sl] = sl;
                        A.s_node[sl] = sg->node[i];
                        A.live[nlive++] = sl;
                        g2_up(A, base, len);
                        ++len;
                        root = A.HK[base];
                    } else if (ek < root) {        // replaces the root (the evicted node's slot is reused) + heap.Fix(0)
                        const u32 sl = A.HP[base];
                        A.s_node[sl] = sg->node[i];
                        g2_down_val(A, base, 0, len, ek, sl);
                        root = A.HK[base];
                    } else continue;
                    lastp = sg->node[i] + 1;       // the last Process that returned true inside tree()
                }
                if (c_lf != G2_NONE) A.h_len[c_lf] = len;
                if (single) { mb->sh[SH_LEN0] = (u32)len; mb->sh[SH_ROOT_LO] = (u32)root; mb->sh[SH_ROOT_HI] = (u32)(root >> 32); }
            }
            wv::wave_sync();
            if (single) { len0 = mb->sh[SH_LEN0]; root0 = ((u64)mb->sh[SH_ROOT_HI] << 32) | mb->sh[SH_ROOT_LO]; }
        }
        if (lane == 0) { mb->sh[SH_NLIVE] = nlive; mb->sh[SH_LASTP] = lastp; }
        wv::wave_sync();
    }
    G2_TICK(2);
    // heap roots and lengths as tree() left them (the Explain pass needs them after the walk has spent the heaps); the state of
    // every slot that holds a node, loaded by all lanes at once
    {
        for (u32 i = lane; i < ntn; i += 64) {
            const int len = A.h_len[i];
            A.rootkey[i] = len ? A.HK[A.h_off[i]] : 0ull;
            A.h_adm[i] = len;
            A.h_cnt[i] = len;
        }
        const u32 nlive = mb->sh[SH_NLIVE];
        for (u32 i = lane; i < nlive; i += 64) {
            const u32 sl = A.live[i], n = A.s_node[sl];
            A.s_cpu[sl] = a.cpu[n];
            A.s_mem[sl] = a.mem[n];
            A.s_placed[sl] = 0;
            for (u32 q = 0; q < G.n_gen; ++q) A.s_gen[(size_t)sl * G.n_gen + q] = a.gcnt[(size_t)Gm->gkind[q] * a.gstride + n];
        }
        wv::wave_sync();
    }
    G2_TICK(3);

    // ---------- tree walk + fill loops: lane 0, on the arena only ----------
    if (lane == 0) {
        u32 next_task = 0, ntouch = 0;
        u32 c1 = 0, c5 = 0, c6 = 0, fpass = 0;   // Explain counters of the fill phase (only Resource / HostPort / MaxReplicas can fail there)
        bool bad_key = false;
        const bool has_ports = (G.flags & RT_PORTS) != 0, has_res = (G.flags & RT_RES) != 0, has_maxrep = (G.flags & RT_MAXREP) != 0;
        const bool counted = !(G.flags & RT_UNCOUNTED);
        const u32 ngen = G.n_gen;
        // Pipeline.Process on a heap position: the static filters passed at admission and cannot change
        auto process = [&](u32 pos) -> bool {
            const u32 sl = A.HP[pos];
            u32 ff = G2_FF_PASS;
            if (has_res) {
                if (!(G.cpu <= A.s_cpu[sl] && G.mem <= A.s_mem[sl])) ff = 1;
                else
                    for (u32 q = 0; q < ngen; ++q)
                        if (A.s_gen[(size_t)sl * ngen + q] < Gm->gval[q]) ff = 1;
            }
            if (ff == G2_FF_PASS) {
                if (has_ports && A.s_placed[sl] > 0) ff = 5;
                else if (has_maxrep && !((u64)((u32)(A.HK[pos] >> 32) & 0xFFFFFFu) < G.maxrep)) ff = 6;
            }
            if (ff == G2_FF_PASS) { c1 = c5 = c6 = 0; fpass = 1; }
            else if (ff == 1) ++c1;
            else if (ff == 5) ++c5;
            else ++c6;
            return ff == G2_FF_PASS;
        };
        // scheduleNTasksOnNodes, scheduler.go:844-924, on the leaf's positions [base, base+cnt)
        auto fill = [&](int want, u32 base, int cnt) -> int {
            int scheduled = 0, iter = 0, ix = 0;
            for (u32 q = base >> 6; q <= (base + (u32)cnt - 1u) >> 6; ++q) A.failed[q] = 0;
            while (next_task < k) {
                const u32 pos = base + (u32)ix;
                const u32 sl = A.HP[pos];
                a.out_node[G.out_off + next_task] = (int32_t)A.s_node[sl];
                ++next_task;
                A.s_cpu[sl] -= G.cpu;   // NodeInfo.addTask (nodeinfo.go:108-154)
                A.s_mem[sl] -= G.mem;
                for (u32 q = 0; q < ngen; ++q) A.s_gen[(size_t)sl * ngen + q] -= Gm->gval[q];   // Claim, resource_management.go:11-39 (counts)
                const u32 pl = A.s_placed[sl];
                A.s_placed[sl] = pl + 1;
                if (pl == 0) A.touched[ntouch++] = sl;
                u64 kcur = A.HK[pos];
                if (counted) {
                    kcur += G2_KEY_STEP;
                    if (((kcur >> 32) & 0xFFFFFFull) == 0) bad_key = true;   // svcCount left its 24 bits
                    A.HK[pos] = kcur;
                }
                ++scheduled;
                if (scheduled == want) return scheduled;
                const int nx = ix + 1 == cnt ? 0 : ix + 1;
                if (iter + 1 < cnt) {
                    if (A.HK[base + (u32)nx] < kcur) { ++iter; ix = nx; }   // first pass: on to the next node once it is the lesser
                } else { ++iter; ix = nx; }                                 // later passes: round robin
                const int orig = iter;
                for (;;) {
                    const u32 bi = base + (u32)ix;
                    const bool bad = (A.failed[bi >> 6] >> (bi & 63)) & 1ull;
                    if (!bad && process(bi)) break;
                    A.failed[bi >> 6] |= 1ull << (bi & 63);
                    ++iter;
                    ix = ix + 1 == cnt ? 0 : ix + 1;
                    if (iter - orig == cnt) return scheduled;
                }
            }
            return scheduled;
        };
        // decisionTree.orderedNodes, decision_tree.go:24-52
        auto ordered = [&](u32 lf) -> int {
            const u32 base = A.h_off[lf];
            int len = A.h_len[lf], cnt = A.h_cnt[lf];
            if (len != cnt) {
                for (int i = 0; i < cnt;) {
                    if (process(base + (u32)i)) ++i;
                    else {
                        --cnt;
                        if (i != cnt) {   // nodes[i] = nodes[last]; nodes = nodes[:last] (the dropped node's slot stays on the touched list)
                            A.HK[base + i] = A.HK[base + cnt];
                            A.HP[base + i] = A.HP[base + cnt];
                        }
                    }
                }
                len = cnt;
                for (int i = cnt / 2 - 1; i >= 0; --i) g2_down(A, base, i, cnt);   // heap.Init
            }
            while (len > 0) {   // heap.Pop: Swap(0, n-1); down(0, n-1); length--
                const int nn = len - 1;
                const u64 kr = A.HK[base]; const u32 pr = A.HP[base];
                const u64 kl = A.HK[base + nn]; const u32 plast = A.HP[base + nn];
                A.HK[base + nn] = kr; A.HP[base + nn] = pr;
                if (nn > 0) g2_down_val(A, base, 0, nn, kl, plast);
                len = nn;
            }
            A.h_cnt[lf] = cnt;
            A.h_len[lf] = 0;
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
                    const int cnt = ordered(f.tn);
                    ret = cnt == 0 ? 0 : fill(f.n, A.h_off[f.tn], cnt);
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
        mb->sh[SH_C1] = c1; mb->sh[SH_C5] = c5; mb->sh[SH_C6] = c6; mb->sh[SH_FPASS] = fpass;
    }
    wv::wave_sync();
    if (mb->sh[SH_ERR]) {
        if (lane == 0) a.ctl->error = ERR_GROUP_RANGE;
        return false;
    }
    G2_TICK(4);

    // ---------- Explain counters for a group with leftovers (pipeline.go:56-68 call sequence: tree()'s calls, then the fill phase's) ----------
    if (mb->sh[SH_LEFT] > 0) {
        const u32 fpass = mb->sh[SH_FPASS];
        if (!fpass) {   // no Process passed after tree(): the failing calls inside tree() behind its last passing one count, too
            for (u32 i = lane; i < ntn; i += 64) { a.xroot[i] = A.rootkey[i]; a.xadm[i] = A.h_adm[i]; }
            if (lane < 8) mb->cntx[lane] = 0;
            if (lane == 0) { mb->x_lastp = mb->sh[SH_LASTP]; mb->x_k = k; }
            g2_post(mb, P, G2_OP_EXPLAIN, gi);
            if (!g2_wait_ge(&mb->done, P.n * nh, mb)) return false;
        }
        if (lane < 8) {
            u32 v = fpass ? 0u : mb->cntx[lane];
            if (lane == 1) v += mb->sh[SH_C1];
            if (lane == 5) v += mb->sh[SH_C5];
            if (lane == 6) v += mb->sh[SH_C6];
            a.hist[(size_t)gi * 8 + lane] = v;
        }
    }
    G2_TICK(5);

    // ---------- write-back of the nodes that got a task: node rows, host ports, generic counts, the service's (node, count) list ----------
    const u32 nt = mb->sh[SH_NTOUCH];
    {
        const bool counted = !(G.flags & RT_UNCOUNTED);
        const u32 lbase = a.list_off[G.svc];
        u32 lcnt = a.list_cnt[G.svc];
        for (u32 i0 = 0; i0 < nt; i0 += 64) {
            const u32 i = i0 + lane;
            bool app = false;
            u32 n = 0, pl = 0;
            if (i < nt) {
                const u32 sl = A.touched[i];
                n = A.s_node[sl];
                pl = A.s_placed[sl];
                a.cpu[n] = A.s_cpu[sl];
                a.mem[n] = A.s_mem[sl];
                for (u32 q = 0; q < G.n_gen; ++q) a.gcnt[(size_t)Gm->gkind[q] * a.gstride + n] = A.s_gen[(size_t)sl * G.n_gen + q];
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
            for (u32 i = lane; i < nt; i += 64) g2_eval_node(a, Gn, a.g + gi + 1, bn, A.s_node[A.touched[i]]);
            wv::wait_vm();
        }
    }
    wv::lockstep();   // every lane has read what it needs from the arena before lane 0 goes on to reset it for the next group
    G2_TICK(7);
    return true;
#undef G2_TICK
}

WV_KERNEL(G2_THREADS) void k_groups2(Groups2Args a) {
    unsigned char* l = reinterpret_cast<unsigned char*>(wv::lds());
    G2Mail* mb = reinterpret_cast<G2Mail*>(l);
    G2Stage* sg = reinterpret_cast<G2Stage*>(l + 512);
    static_assert(sizeof(G2Mail) <= 512 && 512 + sizeof(G2Stage) <= G2_LDS_FIXED, "fixed LDS layout");
    const u32 wave = wv::wave(), lane = wv::lane(), nh = wv::nthreads() / 64u - 1u;
    if (wv::tid() == 0) { mb->posted = 0; mb->done = 0; mb->quit = 0; }
    wv::barrier();
    if (a.ctl->error != ERR_NONE || a.n_groups == 0) return;
    if (wave != 0) {
        g2_helper(a, mb, wave - 1u, nh);
        return;
    }
    u64 gt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 tk = (a.dbg & 16u) ? wv::clock64() : 0ull;
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
    for (u32 gi = 0; gi < a.n_groups && ok; ++gi) {
        if (!g2_wait_ge(&mb->done, eval_cur * nh, mb)) { ok = false; break; }
        if (wv::g_fresh32(&a.ctl->error) != ERR_NONE) { ok = false; break; }
        if (a.dbg & 16u) { const u64 n_ = wv::clock64(); gt[0] += n_ - tk; tk = n_; }
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
        if (g2_arena_bytes(G.n_slots, ntn, G.n_gen, a.max_depth, G.k) <= G2_ARENA_LDS) ok = g2_group<true>(a, mb, sg, l + G2_LDS_FIXED, G, gi, gt, P, nh, eval_next);
        else ok = g2_group<false>(a, mb, sg, a.arena, G, gi, gt, P, nh, eval_next);
        eval_cur = eval_next;
        if (a.dbg & 16u) tk = wv::clock64();
    }
    if (!ok && lane == 0) {
        if (wv::lds_poll32(&mb->quit) && wv::g_fresh32(&a.ctl->error) == ERR_NONE) a.ctl->error = ERR_GROUP_HANG;
        wv::lds_publish32(&mb->quit, 1u);   // helpers leave their wait loops
    }
    g2_post(mb, P, G2_OP_QUIT, 0);
    if (lane == 0 && (a.dbg & 16u))
        for (int q = 0; q < 8; ++q) a.ctl->cyc[q] = gt[q];
}

#endif   // SWP_G2_KERNELS
}  // namespace swpdev
