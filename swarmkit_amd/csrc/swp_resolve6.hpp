// swp_resolve6.hpp — the BLOCK resolver: the sequential argmin + commit pass of the tick (nodeSet.tree with a heap of one,
// nodeset.go:50-124; nodeLess, scheduler.go:708-735; NodeInfo.addTask, nodeinfo.go:108-154). The whole chip builds the candidate
// lists, one wave matches them. The engine's default from 16 384 tasks on and for every node set beyond k_resolve5's LDS.
//
// State in global memory (L2-resident), all of it bitmaps over the node words, kept exact by every commit:
//   planes[b]   bit b of (ActiveTasksCount − base) per node, base = the lowest count among the valid nodes at build time
//   rr[c]       ResourceFilter (filter.go:77-84) as set membership: nodes whose residual cpu (rows 0..n_dc−1) / memory (rows n_dc..)
//               is >= the row's threshold; thresholds = the batch's distinct reservations. In TASK-ROWS mode (a batch with more
//               than 128 of them) the rows are built per task of the coming block instead (k_r6_taskrows): no limit.
//   rg[r]       generic reservations (filter.go:86-91): nodes whose count of the row's kind is >= the row's value
//   sc, X, portmap   static class rows, per-service exception bitmaps, host ports: the rows every resolver uses
//
// One ROUND decides a block of up to `block` tasks, two launches:
//   k_r6_propose   one workgroup of eight wavefronts per task, lanes over the node WORDS: m = sc & ~X & RC & RM & RG & ~ports (the
//                  task's plain candidates, 64 nodes per operation); the level planes of a lane's words are requested together and
//                  the candidates are narrowed from the top plane down IN REGISTERS (m & ~plane ≠ ∅ ? keep that : the bit is set in
//                  the word's minimum level); one reduction gives the task's minimum level, the words whose own minimum equals it
//                  hold exactly its candidates. The first 2 * R6_CAND non-empty half-words go into the task's proposal, with the
//                  best node of the service's exception list by the full key when there is no plain candidate.
//   k_r6_commit    one workgroup of 1 024. Every thread stages its task's list into LDS (entry-major). Wave 0 walks the block in
//                  task order, 64 tasks at a time: a lane seats the first two half-words of its list that still have a candidate
//                  (cursor: an entry is looked at once) and wv::match_seq64 gives every task in turn the first listed node nobody
//                  before it took, struck from every later list. The block is CUT — committed up to there, proposed again from
//                  there — in front of a task whose listed nodes are all taken, around a task that must use its exception list
//                  (its order moves with every placement of the service) and behind an uncounted task (its node did NOT move
//                  up). Waves 1..15 move the cursors of "their" group over dead entries while they wait, then apply the group's
//                  picks: NodeInfo.addTask on the node rows + the bitmaps above + commit log.
//   k_r6_compact   (only while the host sees the symptom: a matcher stop every few tasks, rounds cut after a fraction of their block)
//                  one workgroup, in front of the round's propose: numbers the ready nodes on the level the block's first task aims
//                  at in node order. Tasks whose minimum level is that one list half-words of those POSITIONS — 32 candidates each
//                  where plain half-words hold three of the few emptied nodes — the others plain half-words behind them
//                  (R6Args.compact, the _c instances of the two kernels).
// The exactness argument is the list rule: inside a batch levels only grow and feasibility only shrinks, and a list
// holds ALL plain candidates of the task's minimum level in node order up to its last half-word.
//
// Written against swp_wave.hpp only, so tests/emu runs the same source on CPU fibers (tests/test_emu_resolve6.py).
#pragma once
#include <stddef.h>

#include "swp_shard.hpp"
#include "swp_types.hpp"
#include "swp_volumes.hpp"

namespace swpdev {

#define R6_NP 16                 // level planes: 65 535 levels above the lowest valid node
#ifndef R6_CAND
#define R6_CAND 16               // a proposal lists 2 * R6_CAND non-empty 32-node half-words: a block is cut where a task finds all its listed nodes taken
#endif
#define R6_SEAT 2                 // list entries a seating step of the matcher looks at (4 measured no better: fewer steps, each longer)
#define R6_BMAX 1024             // largest block
#define R6_COMMIT_THREADS 1024      // == R6_BMAX: one accepted pick per thread in the apply phase
#define R6_NONE 0xFFFFFFFFu

struct Blk6 {   // control block, global memory
    u32 pos, end;          // next undecided task, end of the stretch
    u32 base, maxrel;      // lowest task count among the valid nodes at build time; highest level above it so far
    u32 error, rounds, cut_exhausted, cut_exception, cut_uncounted;
    u32 cyc[4];            // R6Args.dbg & 16: shader cycles / 64 of k_r6_commit's sections (prologue, matching, wait for it, apply)
    u32 reseats, cyc_load, cyc_walk;   // how often the matcher stopped at an emptied half-word (always counted); dbg: cycles / 64 of a group's list load and of its walk
    u32 cyc_g[4];          // ... of the list load: waiting for the group's lists, the head records, seating; seating steps
    u32 csize, clevel;     // compact index of this round (k_r6_compact): its nodes (0: none, every list in plain half-words), their level above the base
    u32 crounds;           // rounds that had one
    u32 dbg_cut[3];        // dbg: of the cuts at an exhausted list, those whose list was full (more candidates on the level), in compact positions, one entry long
    u32 scan_skipped;      // k_scanb: tasks answered "no node" without a look (an identical task found none earlier in the stretch)
    u32 scan_batches;      // k_scanb: barriers it took for the tasks it did look at
};
static_assert(sizeof(Blk6) == 112, "Blk6 layout");

struct R6Prop {   // one task's proposal: the shard protocol's record (include/swp.h swp_proposal) with more candidates, as 32-node half-words
    u32 level, n_cand;       // minimum level among the plain candidates (R6_NONE: there is none); half-words listed | bit 31: there are more
    unsigned short hw[2 * R6_CAND];   // half-word index (node = 32 * hw + bit), ascending: the first 2 * R6_CAND NON-EMPTY half-words of the level — 16 bits
                                      // (round 6: node sets end at 2^21 nodes): a record is 224 bytes instead of 288, to write, to stage, to fold and to all-gather
    u32 hb[2 * R6_CAND];     // the level's survivors inside each
    u64 exc_hi, exc_lo;      // best node of the service's exception list by nodeLess' key, KEY_NONE: none
    u32 exc_entry, flags;    // flags bit 0: the task does not count on its node; bit 1: it has cluster mounts
};
static_assert(sizeof(R6Prop) == 8 + 12 * R6_CAND + 24, "R6Prop layout");
static_assert(2 * R6_CAND <= 64, "one lane per list entry");

struct R6Args {
    u32 n_nodes, n_words, xs, block;
    u32 n_dc, n_dm;
    u32 dbg;                 // timing experiments only (env SWP_DBG); 0 in production
    u32 task_rows;           // != 0: ResourceFilter rows per TASK of the block, rebuilt every round (k_r6_taskrows), instead of per demand class
    const u64* valid;        // [n_words]
    const u64* sc;           // [n_sc][n_words]
    u64* X;                  // [n_svc][xs]
    const RTask* rt;
    i64* cpu;
    i64* mem;
    u32* total;
    u32* list_node;
    u32* list_svc;
    u32* list_fail;
    const u32* list_off;
    u64* portmap;
    const u32* pset_off;
    const u32* pset_ids;
    int32_t* out_node;
    u32* log_node;
    u32* log_task;
    int32_t* log_prev;
    int32_t* last;
    u32* inf_task;
    u32* inf_pos;
    Ctl* ctl;
    u64* planes;             // [R6_NP][n_words]
    u64* rr;                 // [n_dc + n_dm][n_words]
    const i64* thr;          // [n_dc + n_dm] the distinct cpu reservations (ascending), then the memory ones
    Blk6* blk;
    R6Prop* prop;            // [block]
    u64* trows;              // [block][n_words] task_rows mode: {cpu residual >= the task's NanoCPUs && memory residual >= its MemoryBytes}
    // generic reservations (filter.go:86-91, validate.go:24-52): n_rg more demand-class rows, rg[r] = {count[rg_kind[r]] >= rg_val[r]},
    // sorted by (kind, value); a task's set (tg[task] -> gs_off / gs_row) names its rows. n_rg == 0: none of the pointers is read.
    u32 n_rg, gstride;       // rows; node stride of gcnt
    int32_t* gcnt;           // [kind][gstride] the node's count of the kind (0: the kind is absent)
    u64* rg;                 // [n_rg][n_words]
    const u32* tg;           // [tasks of the batch] batch-local generic set, 0 = none
    const u32* gs_off;       // [sets + 1]
    const u32* gs_row;
    const u32* rg_kind;      // [n_rg]
    const int32_t* rg_val;   // [n_rg]
    const u32* rg_k0;        // [n_rg] first row of the row's kind ...
    const u32* rg_k1;        // ... and one past its last
    // [tasks of the batch] the first task of the batch with this task's descriptor: equal values = identical tasks, whose plain candidates
    // of a round are the same set. nullptr: lists start at the level's first candidate (the shard drivers: a range sees only its own part).
    const u32* tmpl;
    // CSI volumes (swp_volumes.hpp). csi_of == nullptr: the batch has no task with cluster mounts. Such a task's candidates depend on the
    // volumes every earlier one took — on ANY node — so a block decides at most one of them: the second one is where the block is cut.
    const u32* csi_of;       // [tasks of the batch] index among the batch's tasks with mounts, R6_NONE: it has none
    const u32* csi_set;      // [those tasks] mount set
    u64* vrows;              // [those tasks][n_words] VolumesFilter.Check (filter.go:424-432) as the volumes stood at the start of the task's last round
    u32* att;                // [those tasks][VOL_MAX_MOUNTS] the volumes chosen for its mounts on its node (chooseTaskVolumes), VOL_NONE: none
    VolView vol;
    // The compact index (k_r6_compact, one launch in front of a round's propose). Re-placements after a drain all aim at the few emptied
    // nodes: their level holds, say, 1 000 of 10 000 nodes, three to a half-word, so a list of 32 half-words names 100 of them and the
    // matcher's two seats hold six. The ready nodes ON the level the block's first task aims at, numbered in node order, are the round's
    // compact positions; a task whose minimum level is that one lists half-words of POSITIONS (dense: 32 candidates each), every other task plain half-words behind
    // them (+ 2 * ceil(csize / 64)). One level per list, so a list lives in one of the two address ranges; a node has one address per round,
    // so strikes meet; numbering in node order keeps every list in node order. The applying threads translate a picked address back.
    u32 compact;             // != 0: the rounds launch k_r6_compact and the kernels honour Blk6.csize; 2: k_r6_commit_c builds the NEXT round's index itself
                             // at its end (launch_r6_rounds: one k_r6_compact in front of a chunk's first round only)
    const u64* cbase;        // [n_words] the nodes an index is drawn from: READY && valid — every static class row is a subset, so every task's
                             // candidates are (a drained node keeps its place in `valid` and its level, which is usually the lowest)
    u64* cmask;              // [n_words] the nodes of the compact index
    u32* crank;              // [n_words] compact position of the word's first such node
    u32* cidx;               // [r6_compact_cap(n_words)] node of a compact position
    // node-range shards (swp_resolve7.hpp): where this shard publishes the volumes of a task with cluster mounts it placed ([2] slots, by
    // round parity, right behind its proposals — they travel with them); nullptr: no volumes in the batch
    struct R7Trail* trail_out;   // = the slots of its R7Tail
};
struct R7Args;   // swp_resolve7.hpp: what a shard's commit kernel knows of the other shards
#define R6_STAGES_IN_FLIGHT 2u    // waves that stage their lists at the same time on a single engine (measured, cfg3 / cfg4 1M x 100k: all at once 11.8 / 126.1 ms,
                                  // 1: 11.6 / 122.7, 2: 11.5 / 122.9, 4: 11.65 / 124.5, 6: 11.7 / 125.6 — the matcher's own group no longer queues behind the others' loads)
#define R7_FOLDS_IN_FLIGHT 4u   // waves that fold at the same time (the R7 staging below); SWP_DBG bits 8-11 override it for A/B runs (tools/gpu_r5_foldwin.sh)
#define R7M_EXC 0x100u   // H_meta of a folded record: list length | the task has an exception-list candidate on some shard | it does not count on its node | it has cluster mounts
#define R7M_UNC 0x200u
#define R7M_CSI 0x400u

#define R6_UNROLL 4              // 64-word chunks a propose wave has in flight together ...
#define R6_SMALL_WORDS 512u      // ... on node sets beyond this many words (32 768 nodes); up to there a wave has ONE chunk: a quarter of the registers, so that
                                 // the workgroups of a large block are all resident at once
#define R6_PW 8                  // waves per task in the propose kernel
inline __host__ __device__ u32 r6_unroll(u32 n_words) { return n_words <= R6_SMALL_WORDS ? 1u : (u32)R6_UNROLL; }
inline __host__ __device__ u32 r6_chunks(u32 n_words) { const u32 g = r6_unroll(n_words) * R6_PW; return (((n_words + 63u) >> 6) + g - 1u) / g * g; }
inline __host__ __device__ size_t r6_propose_lds(u32 n_words) { return (size_t)2 * r6_chunks(n_words) * 64 * 8 + 128; }
// TK row, thresholds, the picks of the block, a few scalars, the block's lists (entry-major)
// (per task: pick node / index / aux, the cursor, 2 * R6_CAND candidate masks and as many half-word indices of 16 bits: node sets of up to 2^21 nodes)
// (the TK row: the node words, and in front of them the words of a compact index — at most a quarter of the nodes)
#define R6_COMPACT_MAX_WORDS 16384u   // node words up to which a compact index may be built (half-word indices are 16 bits in the commit kernel's LDS); the
                                      // engine builds one only where its quarter more of TK row fits next to the block's lists as they are
inline __host__ __device__ u32 r6_compact_cap(u32 n_words) { return 16u * n_words; }
inline __host__ __device__ u32 r6_tk_words(u32 n_words) { return n_words + n_words / 4u + 2u; }   // (with a compact index; n_words otherwise)
inline __host__ __device__ size_t r6_commit_lds(u32 n_words, u32 block, u32 n_rr, bool compact = false) {
    return (size_t)((compact ? r6_tk_words(n_words) : n_words) + n_rr) * 8 + (size_t)block * (16 + 2 * R6_CAND * 6) + 128;
}

#ifdef SWP_R6_KERNELS   // the kernels: swp_resolve6.hip and the emulation harness only (the engine TU shares the argument records)
// the node-range shards' side of the commit kernel (defined in swp_resolve7.hpp; only its R7 instances use them)
WV_DEV u32 r7_tk_words(const R7Args* m);
WV_DEV bool r7_any_dead(const R7Args* m);
WV_DEV void r7_fold_into(const R7Args* m, u32 i, bool have, u32 block, unsigned short* L_hw, u32* L_hb, u32* H_level, u32* H_meta, u32* sh);
WV_DEV u32 r7_addr(const R7Args* m, u32 shard, u32 node);
WV_DEV u32 r7_local(const R7Args* m, u32 addr, u32 my, bool* here);
WV_DEV u32 r7_take_trailers(const R6Args& a, const R7Args* m, u32 my, u32 round);
WV_DEV void r7_leave_trailer(const R6Args& a, u32 my, u32 round, u32 set, u32 node, const u32* att, u32 n);

WV_DEV u64 r6_wave_min64(u64 v) {
    const u32 hi = (u32)(v >> 32), lo = (u32)v;
    const u32 mh = wv::min_u32(hi);
    const u32 ml = wv::min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((u64)mh << 32) | ml;
}

// ---- build: base / highest level (one workgroup of 1024), then planes and demand-class rows (one wave per node word) ----
WV_KERNEL(1024) void k_r6_minmax(R6Args a) {
    u32* red = reinterpret_cast<u32*>(wv::lds());
    u32 lo = 0xFFFFFFFFu, hi = 0;
    for (u32 n = wv::tid(); n < a.n_nodes; n += 1024)
        if ((a.valid[n >> 6] >> (n & 63)) & 1) {
            const u32 t = a.total[n];
            lo = min(lo, t);
            hi = max(hi, t);
        }
    lo = wv::min_u32(lo);
    hi = ~wv::min_u32(~hi);
    if (wv::lane() == 0) {
        red[wv::wave()] = lo;
        red[16 + wv::wave()] = hi;
    }
    wv::barrier();
    if (wv::tid() == 0) {
        for (u32 i = 1; i < 16; ++i) {
            lo = min(lo, red[i]);
            hi = max(hi, red[16 + i]);
        }
        if (lo == 0xFFFFFFFFu) lo = hi = 0;   // no valid node
        a.blk->base = lo;
        a.blk->maxrel = hi - lo;
        a.blk->error = (hi - lo) >> R6_NP ? (u32)ERR_LEVEL_RANGE : (u32)ERR_NONE;
    }
}

WV_KERNEL(256) void k_r6_rows(R6Args a) {
    const u32 w = wv::block() * 4 + wv::wave(), lane = wv::lane();
    if (w >= a.n_words) return;
    const u32 base = wv::uload(&a.blk->base);   // written by the launch before this one
    const u32 n = w * 64 + lane;
    const bool in = n < a.n_nodes;
    const bool v = in && ((wv::uload(a.valid + w) >> lane) & 1);
    const u32 lvl = v ? a.total[n] - base : 0u;
    for (u32 b = 0; b < R6_NP; ++b) {
        const u64 word = wv::ballot(v && ((lvl >> b) & 1u));
        if (lane == 0) a.planes[(size_t)b * a.n_words + w] = word;
    }
    const i64 qc = in ? a.cpu[n] : 0, qm = in ? a.mem[n] : 0;
    for (u32 c = 0; c < a.n_dc + a.n_dm; ++c) {
        const u64 word = wv::ballot(in && (c < a.n_dc ? qc : qm) >= wv::uload(a.thr + c));
        if (lane == 0) a.rr[(size_t)c * a.n_words + w] = word;
    }
    for (u32 r = 0; r < a.n_rg; ++r) {
        const u64 word = wv::ballot(in && a.gcnt[(size_t)wv::uload(a.rg_kind + r) * a.gstride + n] >= wv::uload(a.rg_val + r));
        if (lane == 0) a.rg[(size_t)r * a.n_words + w] = word;
    }
}

// ---- task rows: ResourceFilter.Check (filter.go:77-84) of every task of the coming block against the residuals as they are ------
// A batch whose tasks carry hundreds of distinct reservations (every service its own NanoCPUs / MemoryBytes) would make every commit
// cross hundreds of demand-class thresholds. For such a batch the rows are not kept per class and patched by the commits but built
// per task at the start of every round, from the exact residuals: one wave per node word (a ballot IS a row word), the block's
// reservations staged in LDS. No limit on the number of distinct reservations.
// `g_first`, `g_step`: the 64-task groups of the block this workgroup works for (k_r6_taskrows: one group per workgroup row of the
// grid, so that the launch fills the chip; the shard driver's launch uses its second grid dimension for the shards and walks all groups).
WV_DEV void r6_taskrows(const R6Args& a, u32 g_first, u32 g_step) {
    const u32 pos = wv::uload(&a.blk->pos), end = wv::uload(&a.blk->end);
    if (pos >= end || wv::uload(&a.blk->error) != ERR_NONE) return;
    const u32 cnt = min(a.block, end - pos);
    if (g_first * 64u >= cnt) return;
    i64* res = reinterpret_cast<i64*>(wv::lds());   // [block][2], a task without the filter: (INT64_MIN, INT64_MIN): every node passes
    for (u32 g0 = g_first * 64u; g0 < cnt; g0 += g_step * 64u)
        for (u32 j = g0 + wv::tid(); j < min(g0 + 64u, cnt); j += 256) {
            const RTask* rt = a.rt + pos + j;
            const bool on = (rt->flags & RT_RES) != 0;
            res[2 * j] = on ? rt->cpu : (i64)0x8000000000000000ull;
            res[2 * j + 1] = on ? rt->mem : (i64)0x8000000000000000ull;
        }
    wv::barrier();
    const u32 w = wv::block() * 4 + wv::wave(), lane = wv::lane();
    if (w >= a.n_words) return;
    const u32 n = w * 64 + lane;
    const bool in = n < a.n_nodes;
    const i64 qc = in ? a.cpu[n] : 0, qm = in ? a.mem[n] : 0;
    const u64 inmask = wv::ballot(in);
    // lanes = 64 tasks of the block at a time, the word's 64 residual pairs walked on the scalar side: ten instructions per (node, 64
    // tasks) and one store per lane and group, instead of a ballot and a one-lane store per task
    for (u32 g0 = g_first * 64u; g0 < cnt; g0 += g_step * 64u) {
        const u32 j = g0 + lane;
        const bool have = j < cnt;
        const i64 rc = have ? res[2 * j] : 0, rm = have ? res[2 * j + 1] : 0;
        u32 lo = 0, hi = 0;
        for (u32 i = 0; i < 32; ++i) {
            const i64 c0 = (i64)wv::readlane64((u64)qc, i), m0 = (i64)wv::readlane64((u64)qm, i);
            const i64 c1 = (i64)wv::readlane64((u64)qc, i + 32), m1 = (i64)wv::readlane64((u64)qm, i + 32);
            if (rc <= c0 && rm <= m0) lo |= 1u << i;
            if (rc <= c1 && rm <= m1) hi |= 1u << i;
        }
        if (have) a.trows[(size_t)j * a.n_words + w] = (((u64)hi << 32) | lo) & inmask;
    }
}

WV_KERNEL(256) void k_r6_taskrows(R6Args a) { r6_taskrows(a, wv::block_y(), (a.block + 63u) / 64u); }   // grid (words / 4, groups of the block)

// ---- volume rows: VolumesFilter.Check of the block's tasks with cluster mounts, from the volumes as they are (grid: words / 256, block) ----
WV_KERNEL(256) void k_r6_volrows(R6Args a) {
    const u32 pos = wv::uload(&a.blk->pos), end = wv::uload(&a.blk->end);
    const u32 t = pos + wv::block_y();
    if (t >= end || wv::uload(&a.blk->error) != ERR_NONE) return;
    const u32 ck = wv::uload(a.csi_of + t);
    if (ck == R6_NONE) return;
    const u32 w = wv::block() * 256 + wv::tid();
    if (w < a.n_words) a.vrows[(size_t)ck * a.n_words + w] = vol_filter_word(a.vol, wv::uload(a.csi_set + ck), w);
}

// ---- compact index: the ready nodes on the level the block's first task aims at, numbered in node order (one workgroup of 1 024) ----
// The level is the minimum among the plain candidates of the task at Blk6.pos (the rows the propose kernel combines, narrowed over the
// level planes the same way): after a drain that is the level of the emptied nodes, which the tasks behind it aim at as well — not the
// lowest level among the ready nodes, where nodes that nobody can use (no room, the wrong platform) sit for ever. The index holds ALL
// ready nodes on that level (their running count is a word's first position, crank; the nodes themselves go to cidx). More than a
// quarter of the node set on it, or no plain candidate: no index this round.
// (FUSED: the same pass at the END of k_r6_commit_c, for the round that follows — R6Args.compact == 2: the launch, its ramp and its drain
// are a fifth of a churn round. The rows it reads were changed by this workgroup's atomics a moment ago, so every load of a row that
// moves goes past the vector L1 (g_fresh*), and Blk6.maxrel — which the kernel's scalar loads may have cached with the control block —
// is read the same way. `t`: the first task of the coming round; `red`: 32 words of LDS nobody else uses any more.)
template <bool FUSED> WV_DEV void r6_compact_body(const R6Args& a, u32 t, u32* red) {
    const u32 tid = wv::tid(), lane = wv::lane(), wave = wv::wave();
    const u32 Wn = a.n_words;
    auto ld = [](const u64* q) { return FUSED ? wv::g_fresh64(q) : *q; };
    const u32 maxrel_v = FUSED ? wv::g_fresh32(&a.blk->maxrel) : 0u;   // (requested with the task's record, not in front of it)
    const RTask* rt = a.rt + t;
    const u32 flags = wv::uload(&rt->flags), svc = wv::uload(&rt->svc), scid = wv::uload(&rt->sc), pset = wv::uload(&rt->pset);
    const int nb = 32 - wv::clz32(FUSED ? wv::readfirstlane(maxrel_v) : wv::uload(&a.blk->maxrel));
    const u64* scrow = a.sc + (size_t)scid * Wn;
    const u64* xrow = a.X + (size_t)svc * a.xs;
    const bool res = (flags & RT_RES) != 0;
    const u64* rc = a.task_rows ? a.trows : a.rr + (size_t)((flags >> RT_DC_SHIFT) & RT_DCLS_MASK) * Wn;   // (task-rows mode: the block's first row)
    const u64* rm = a.task_rows ? rc : a.rr + (size_t)(a.n_dc + ((flags >> RT_DM_SHIFT) & RT_DCLS_MASK)) * Wn;
    u32 p0 = 0, p1 = 0, g0 = 0, g1 = 0;
    if (flags & RT_PORTS) {
        p0 = wv::uload(a.pset_off + pset);
        p1 = wv::uload(a.pset_off + pset + 1);
    }
    if (a.n_rg) {
        const u32 gset = wv::uload(a.tg + t);
        g0 = wv::uload(a.gs_off + gset);
        g1 = wv::uload(a.gs_off + gset + 1);
    }
    const u32 ck = a.csi_of ? wv::uload(a.csi_of + t) : R6_NONE;
    const u64* vrow = ck != R6_NONE ? a.vrows + (size_t)ck * Wn : nullptr;
    u32 best = R6_NONE;
    for (u32 w = tid; w < Wn; w += 1024) {
        u64 m = scrow[w] & ~ld(xrow + w);
        if (res) m &= ld(rc + w) & ld(rm + w);
        for (u32 g = g0; g < g1; ++g) m &= ld(a.rg + (size_t)wv::uload(a.gs_row + g) * Wn + w);
        if (vrow) m &= vrow[w];
        for (u32 p = p0; p < p1; ++p) m &= ~ld(a.portmap + (size_t)wv::uload(a.pset_ids + p) * Wn + w);
        u32 rel = 0;
        for (int b = nb - 1; b >= 0; --b) {
            const u64 c = m & ~ld(a.planes + (size_t)b * Wn + w);
            if (c) m = c;
            else if (m) rel |= 1u << b;
        }
        best = min(best, m ? rel : R6_NONE);
    }
    best = wv::min_u32(best);
    if (lane == 0) red[wave] = best;
    wv::barrier();
    u32 gl = R6_NONE;
    for (u32 v = 0; v < 16; ++v) gl = min(gl, red[v]);
    u32 running = 0, at0 = 0;
    u64 c0 = 0;   // this thread's first word: its nodes of the level and their first position
    for (u32 w0 = 0; w0 < Wn; w0 += 1024) {
        const u32 w = w0 + tid;
        u64 c = (w < Wn && gl != R6_NONE) ? a.cbase[w] : 0ull;   // the ready nodes whose level IS gl
        for (int b = 0; b < nb && c; ++b) {
            const u64 pl = ld(a.planes + (size_t)b * Wn + w);
            c &= ((gl >> b) & 1u) ? pl : ~pl;
        }
        const u32 pc = (u32)wv::popc64(c);
        u32 ex = 0, tot = 0;   // the counts (<= 64: seven bits) summed over the lower lanes plane by plane
        WV_UNROLL
        for (u32 bit = 0; bit < 7; ++bit) {
            const u64 bm = wv::ballot(((pc >> bit) & 1u) != 0);
            ex += wv::mbcnt(bm) << bit;
            tot += (u32)wv::popc64(bm) << bit;
        }
        wv::barrier();   // (the counts of the chunk before are read)
        if (lane == 0) red[16 + wave] = tot;
        wv::barrier();
        u32 off = 0, all = 0;
        for (u32 v = 0; v < 16; ++v) {
            const u32 x = red[16 + v];
            off += v < wave ? x : 0u;
            all += x;
        }
        if (w < Wn) {
            a.cmask[w] = c;   // (the words beyond a thread's first are read back below by the thread that wrote them)
            a.crank[w] = running + off + ex;
        }
        if (w0 == 0) { c0 = c; at0 = off + ex; }
        running += all;
    }
    const bool on = running != 0 && running <= r6_compact_cap(Wn);
    if (on)
        for (u32 w = tid; w < Wn; w += 1024) {
            u64 c = w == tid ? c0 : FUSED ? wv::g_fresh64(a.cmask + w) : a.cmask[w];   // (a thread's first word: from its registers, no round trip)
            u32 at = w == tid ? at0 : FUSED ? wv::g_fresh32(a.crank + w) : a.crank[w];
            while (c) {
                a.cidx[at++] = w * 64u + (u32)wv::ffs64(c);
                c &= c - 1ull;
            }
        }
    if (tid == 0) {
        a.blk->csize = on ? running : 0u;
        a.blk->clevel = gl;
    }
}
WV_KERNEL(1024) void k_r6_compact(R6Args a) {
    const u32 t = wv::uload(&a.blk->pos);
    if (t >= wv::uload(&a.blk->end) || wv::uload(&a.blk->error) != ERR_NONE) return;
    r6_compact_body<false>(a, t, reinterpret_cast<u32*>(wv::lds()));   // [16] a wave's minimum, [16] a wave's count
}

// ---- propose: one workgroup of R6_PW waves per task of the block -----------------------------------------------------------
// The waves split the task's node words (wave v owns the chunks of 64 words k = v, v + R6_PW, ...): the passes are latency-bound,
// so more waves per task is what shortens them. A pass ends with one barrier (has any wave a candidate left?).
template <int UN, bool CPT = false> WV_DEV void r6_propose_t(const R6Args& a) {
    const u32 lane = wv::lane(), wave = wv::wave();
    const u32 t = wv::uload(&a.blk->pos) + wv::block();
    if (t >= wv::uload(&a.blk->end) || wv::uload(&a.blk->error) != ERR_NONE) return;   // (an error stops the rounds until the host has seen it)
    const RTask* rt = a.rt + t;
    const i64 rcpu = wv::uload(&rt->cpu), rmem = wv::uload(&rt->mem);
    const u32 flags = wv::uload(&rt->flags), svc = wv::uload(&rt->svc), scid = wv::uload(&rt->sc), pset = wv::uload(&rt->pset);
    const u64 maxrep = wv::uload(&rt->maxrep);
    const u32 Wn = a.n_words, KC = r6_chunks(Wn);   // a multiple of UN * R6_PW; the chunks beyond the row hold no candidates
    u64* A = wv::lds();
    u64* Bf = A + (size_t)KC * 64;
    u32* flag = reinterpret_cast<u32*>(Bf + (size_t)KC * 64);   // [R6_NP + 2] "some wave still has a candidate", one word per pass
    const u64* scrow = a.sc + (size_t)scid * Wn;
    const u64* xrow = a.X + (size_t)svc * a.xs;
    const bool res = (flags & RT_RES) != 0;
    // the two demand-class rows of the task — or, in task-rows mode, its own row of this round (twice)
    const u64* rc = a.task_rows ? a.trows + (size_t)wv::block() * Wn : a.rr + (size_t)((flags >> RT_DC_SHIFT) & RT_DCLS_MASK) * Wn;
    const u64* rm = a.task_rows ? rc : a.rr + (size_t)(a.n_dc + ((flags >> RT_DM_SHIFT) & RT_DCLS_MASK)) * Wn;
    u32 p0 = 0, p1 = 0;
    if (flags & RT_PORTS) {
        p0 = wv::uload(a.pset_off + pset);
        p1 = wv::uload(a.pset_off + pset + 1);
    }
    u32 g0 = 0, g1 = 0;   // the task's generic rows
    if (a.n_rg) {
        const u32 gset = wv::uload(a.tg + t);
        g0 = wv::uload(a.gs_off + gset);
        g1 = wv::uload(a.gs_off + gset + 1);
    }
    const u32 ck = a.csi_of ? wv::uload(a.csi_of + t) : R6_NONE;   // a task with cluster mounts: its VolumesFilter row of this round
    const u64* vrow = ck != R6_NONE ? a.vrows + (size_t)ck * a.n_words : nullptr;
    // The task's plain candidates AND the minimum level among them in one pass: lane l of wave v owns words {l + 64 k}, k = v (mod
    // R6_PW); per word the candidate set is narrowed over the level planes from the top in registers (m & ~plane ≠ ∅ ? keep that : the
    // bit is set in the word's minimum), the planes of UN words requested together; the minimum over the words is one reduction.
    // How many tasks of the block in front of this one are IDENTICAL to it (same descriptor): each of them takes — strikes — the first
    // candidate nobody took before it, out of the same set in the same order, so when this task's turn comes the first `twins`
    // candidates of the level are gone whatever the other tasks did. Its list starts behind them (below): a block of one service's
    // tasks then walks 32 half-words PER TASK into the level instead of sharing one window of 32. The ids are requested here and
    // counted behind the pass.
    u32 tw_id[(R6_BMAX + 64 * R6_PW - 1) / (64 * R6_PW)], tw_mine = 0;
    const u32 tw_pos = wv::uload(&a.blk->pos);
    if (a.tmpl) {
        tw_mine = wv::uload(a.tmpl + t);
        WV_UNROLL
        for (u32 q = 0; q < (R6_BMAX + 64 * R6_PW - 1) / (64 * R6_PW); ++q) {
            const u32 u = tw_pos + (q * R6_PW + wave) * 64 + lane;
            tw_id[q] = u < t ? a.tmpl[u] : R6_NONE;
        }
    }
    const u32 maxrel = wv::uload(&a.blk->maxrel);
    const int nb = 32 - wv::clz32(maxrel);          // planes in use (0: every valid node sits on the base level)
    u32* R = reinterpret_cast<u32*>(Bf);            // [KC * 64] the word's own minimum level above the base, R6_NONE: no candidate in it
    // the round's compact index (R6Args.compact): VW words of positions, the plain half-words are numbered behind them
    const u32 csize = CPT ? wv::uload(&a.blk->csize) : 0u, VW = (csize + 63u) >> 6;
    u64* V = reinterpret_cast<u64*>(R + (size_t)KC * 64);   // [VW <= KC * 16 + 1] the task's candidates by compact position (the upper half of Bf)
    for (u32 i = wave * 64 + lane; i < VW; i += 64 * R6_PW) V[i] = 0;
    u32 best = R6_NONE;
    for (u32 k0 = wave; k0 < KC; k0 += UN * R6_PW) {
        u64 m[UN], f[UN];
        WV_UNROLL
        for (int u = 0; u < UN; ++u) {
            const u32 w = (k0 + u * R6_PW) * 64 + lane;
            const bool in = w < Wn;
            m[u] = in ? scrow[w] : 0ull;
            f[u] = in ? xrow[w] : 0ull;
            if (res && in) f[u] |= ~(rc[w] & rm[w]);
            for (u32 g = g0; g < g1; ++g)
                if (in) f[u] |= ~a.rg[(size_t)wv::uload(a.gs_row + g) * Wn + w];
            if (vrow && in) f[u] |= ~vrow[w];
        }
        u32 rel[UN];
        WV_UNROLL
        for (int u = 0; u < UN; ++u) {
            const u32 w = (k0 + u * R6_PW) * 64 + lane;
            m[u] &= ~f[u];
            for (u32 p = p0; p < p1; ++p)
                if (m[u]) m[u] &= ~a.portmap[(size_t)wv::uload(a.pset_ids + p) * Wn + w];
            rel[u] = 0;
        }
        for (int hi = nb; hi > 0; hi -= 8) {   // eight planes a batch (a batch is all there is below 256 levels)
            const int lo = hi > 8 ? hi - 8 : 0;
            u64 q[UN][8];
            WV_UNROLL
            for (int u = 0; u < UN; ++u) {
                const u32 w = (k0 + u * R6_PW) * 64 + lane;
                // (a word without candidates needs no planes — but on a small node set waiting for m costs more than the loads: there the top
                // batch is requested together with the rows)
                const bool want = (hi == nb && Wn <= 512u) ? w < Wn : m[u] != 0;   // m != 0 only inside the row
                WV_UNROLL
                for (int b = 0; b < 8; ++b) q[u][b] = (lo + b < hi && want) ? a.planes[(size_t)(lo + b) * Wn + w] : 0ull;
            }
            WV_UNROLL
            for (int u = 0; u < UN; ++u) {
                WV_UNROLL
                for (int b = 7; b >= 0; --b) {
                    if (lo + b >= hi) continue;
                    const u64 c = m[u] & ~q[u][b];
                    if (c) m[u] = c;
                    else if (m[u]) rel[u] |= 1u << (lo + b);
                }
            }
        }
        WV_UNROLL
        for (int u = 0; u < UN; ++u) {
            const u32 idx = (k0 + u * R6_PW) * 64 + lane;
            A[idx] = m[u];
            const u32 r = m[u] ? rel[u] : R6_NONE;
            R[idx] = r;
            best = min(best, r);
        }
    }
    best = wv::min_u32(best);
    u32 tw_cnt = 0;
    if (a.tmpl) {
        WV_UNROLL
        for (u32 q = 0; q < (R6_BMAX + 64 * R6_PW - 1) / (64 * R6_PW); ++q) tw_cnt += (u32)wv::popc64(wv::ballot(tw_id[q] == tw_mine && tw_pos + (q * R6_PW + wave) * 64 + lane < t));
    }
    if (lane == 0) {
        flag[wave] = best;
        flag[R6_PW + wave] = tw_cnt;
    }
    wv::barrier();
    u32 gmin = R6_NONE, twins = 0;
    for (u32 v = 0; v < R6_PW; ++v) {
        gmin = min(gmin, flag[v]);
        twins += flag[R6_PW + v];
    }
    const u32 level = gmin == R6_NONE ? R6_NONE : wv::uload(&a.blk->base) + gmin;
    // the task's minimum level is the compact index's: its candidates are among the index's nodes; every wave moves those of its share of
    // the words to their positions
    const bool cmode = CPT && csize != 0 && gmin == wv::uload(&a.blk->clevel);
    if (CPT && cmode) {
        u32* V32 = reinterpret_cast<u32*>(V);
        for (u32 idx = wave * 64 + lane; idx < Wn; idx += 64 * R6_PW) {
            u64 m = R[idx] == gmin ? A[idx] : 0ull;
            if (!m) continue;
            const u64 cm = a.cmask[idx];
            const u32 r0 = a.crank[idx];
            while (m) {
                const u64 low = m & (0ull - m);
                const u32 vp = r0 + (u32)wv::popc64(cm & (low - 1ull));
                wv::lds_or32(V32 + (vp >> 5), 1u << (vp & 31u));
                m ^= low;
            }
        }
        wv::barrier();
    }
    if (wave != 0) return;   // (every wave is past the last barrier) wave 0 lists the candidates and writes the proposal
    const u32 nw = cmode ? VW : Wn, hw_off = cmode ? 0u : 2u * VW;
    R6Prop* out = a.prop + wv::block();
    // its first non-empty half-words, in node order (what the matcher walks: a word that is half empty does not cost a list entry):
    // 64 words a step, a lane's place in the list = the non-empty half-words in front of it (two ballots)
    u32 cnt = 0, more = 0;
    u32 skip = twins, last_hw = 0, last_hb = 0;   // candidates still to pass over; the last non-empty half-word passed over
    if (level != R6_NONE)
        for (u32 k = 0; k < (nw + 63u) / 64u && !more; ++k) {   // (the chunks beyond the row hold nothing)
            const u64 m = cmode ? (k * 64 + lane < VW ? V[k * 64 + lane] : 0ull) : (R[k * 64 + lane] == gmin ? A[k * 64 + lane] : 0ull);   // the words whose own minimum is the task's
            u32 lo = (u32)m, hi = (u32)(m >> 32);
            u64 b_lo = wv::ballot(lo != 0), b_hi = wv::ballot(hi != 0);
            if (!(b_lo | b_hi)) continue;
            if (skip) {
                // candidates in front of a lane's word: the popcounts (<= 64: seven bits) summed over the lower lanes plane by plane
                const u32 pl = (u32)wv::popc64((u64)lo), pc = pl + (u32)wv::popc64((u64)hi);
                u32 ex = 0, tot = 0;
                WV_UNROLL
                for (u32 bit = 0; bit < 7; ++bit) {
                    const u64 bm = wv::ballot(((pc >> bit) & 1u) != 0);
                    ex += wv::mbcnt(bm) << bit;
                    tot += (u32)wv::popc64(bm) << bit;
                }
                // a half-word whose candidates ALL lie among the first `skip` is passed over (the others keep their struck bits: the
                // matcher meets them in the TK row)
                const bool drop_lo = ex + pl <= skip, drop_hi = ex + pc <= skip;
                const u64 gone = wv::ballot(pc != 0 && drop_hi);   // words passed over entirely ...
                const u64 half = wv::ballot(lo != 0 && drop_lo && !(pc != 0 && drop_hi));   // ... and the one word whose low half alone is
                if (gone | half) {   // remember the last half-word passed over: a level with no more than `skip` candidates lists that one
                    const u32 lg = gone ? 63u - (u32)__builtin_clzll(gone) : 0u, lh = half ? 63u - (u32)__builtin_clzll(half) : 0u;
                    if (half && (!gone || lh > lg)) {
                        last_hw = hw_off + 2 * (k * 64 + lh);
                        last_hb = wv::readlane(lo, lh);
                    } else {
                        const u32 hi_l = wv::readlane(hi, lg), lo_l = wv::readlane(lo, lg);
                        last_hw = hw_off + 2 * (k * 64 + lg) + (hi_l ? 1u : 0u);
                        last_hb = hi_l ? hi_l : lo_l;
                    }
                }
                if (drop_lo) lo = 0;
                if (drop_hi) hi = 0;
                skip = tot <= skip ? skip - tot : 0u;
                b_lo = wv::ballot(lo != 0);
                b_hi = wv::ballot(hi != 0);
                if (!(b_lo | b_hi)) continue;
            }
            const u32 at_lo = cnt + wv::mbcnt(b_lo) + wv::mbcnt(b_hi), at_hi = at_lo + (lo != 0 ? 1u : 0u);
            if (lo && at_lo < 2 * R6_CAND) {
                out->hw[at_lo] = (unsigned short)(hw_off + 2 * (k * 64 + lane));
                out->hb[at_lo] = lo;
            }
            if (hi && at_hi < 2 * R6_CAND) {
                out->hw[at_hi] = (unsigned short)(hw_off + 2 * (k * 64 + lane) + 1);
                out->hb[at_hi] = hi;
            }
            const u32 total = cnt + (u32)wv::popc64(b_lo) + (u32)wv::popc64(b_hi);
            if (total > 2 * R6_CAND) more = 1;
            cnt = min(total, (u32)(2 * R6_CAND));
        }
    // no plain candidate: the service's exception list by the full key (scheduler.go:708-735), lanes stride over the entries
    u64 bhi = KEY_NONE, blo = KEY_NONE;
    u32 be = 0;
    const u32 e0 = wv::uload(a.list_off + svc), e1 = level == R6_NONE ? wv::uload(a.list_off + svc + 1) : e0;
    for (u32 e = e0 + lane; e < e1; e += 64) {
        const u32 n = a.list_node[e];
        if (n == LIST_EMPTY) continue;
        const u32 w = n >> 6;
        const u64 bit = 1ull << (n & 63);
        if (!(scrow[w] & bit)) continue;
        if (res && !(rcpu <= a.cpu[n] && rmem <= a.mem[n])) continue;
        bool lacks = false;   // HasEnough per reservation, validate.go:24-52
        for (u32 g = g0; g < g1; ++g) {
            const u32 r = a.gs_row[g];
            if (a.gcnt[(size_t)a.rg_kind[r] * a.gstride + n] < a.rg_val[r]) lacks = true;
        }
        if (lacks) continue;
        if (vrow && !(vrow[w] & bit)) continue;
        bool used = false;
        for (u32 p = p0; p < p1; ++p)
            if (a.portmap[(size_t)a.pset_ids[p] * Wn + w] & bit) used = true;
        if (used) continue;
        const u32 sv = a.list_svc[e], fl = a.list_fail[e];
        if ((flags & RT_MAXREP) && !((u64)sv < maxrep)) continue;   // filter.go:373-375
        const u32 fcl = fl >= MAX_FAILURES ? fl - (MAX_FAILURES - 1) : 0u;
        const u64 hi = ((u64)fcl << 32) | sv, lo = ((u64)a.total[n] << 32) | n;
        if (hi < bhi || (hi == bhi && lo < blo)) {
            bhi = hi;
            blo = lo;
            be = e;
        }
    }
    const u64 ghi = r6_wave_min64(bhi);
    const u64 glo = r6_wave_min64(bhi == ghi ? blo : KEY_NONE);
    const u64 who = wv::ballot(bhi == ghi && blo == glo && ghi != KEY_NONE);
    const u32 gentry = who ? wv::readlane(be, (u32)wv::ffs64(who)) : 0u;
    if (level != R6_NONE && cnt == 0) {   // every candidate of the level lies among the twins' share: the level's last half-word, all taken by then (a cut)
        if (lane == 0) {
            out->hw[0] = (unsigned short)last_hw;
            out->hb[0] = last_hb;
        }
        cnt = 1;
    }
    if (lane >= cnt && lane < 2 * R6_CAND) {
        out->hw[lane] = 0;
        out->hb[lane] = 0;
    }
    if (lane == 0) {
        out->level = level;
        out->n_cand = cnt | (more ? 0x80000000u : 0u);
        out->exc_hi = ghi;
        out->exc_lo = ghi == KEY_NONE ? KEY_NONE : glo;
        out->exc_entry = gentry;
        out->flags = ((flags & RT_UNCOUNTED) ? 1u : 0u) | (ck != R6_NONE ? 2u : 0u);
    }
}

// (UN follows the node set: r6_unroll(n_words); the shard drivers' ranges may differ in size, so they pass it per range)
WV_DEV void r6_propose(const R6Args& a) {
    if (a.n_words <= R6_SMALL_WORDS) r6_propose_t<1>(a);
    else r6_propose_t<R6_UNROLL>(a);
}
WV_KERNEL(64 * R6_PW) void k_r6_propose(R6Args a) { r6_propose(a); }
WV_KERNEL(64 * R6_PW) void k_r6_propose_small(R6Args a) { r6_propose_t<1>(a); }   // n_words <= R6_SMALL_WORDS only: the registers of ONE chunk
// ... the instances that honour a round's compact index (launched behind k_r6_compact only)
WV_KERNEL(64 * R6_PW) void k_r6_propose_c(R6Args a) {
    if (a.n_words <= R6_SMALL_WORDS) r6_propose_t<1, true>(a);
    else r6_propose_t<R6_UNROLL, true>(a);
}
WV_KERNEL(64 * R6_PW) void k_r6_propose_small_c(R6Args a) { r6_propose_t<1, true>(a); }

// index of the first of n ascending thresholds that is greater than q (n: none)
WV_DEV u32 r6_first_above(const i64* thr, u32 n, i64 q) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (thr[mid] > q) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// ---- commit: match the block in task order (wave 0), then apply the accepted picks (all threads) ---------------------------
// (CSI: the instance for batches with cluster mounts — choosing volumes indexes a small array dynamically, which costs the kernel a scratch
// segment; the ordinary instances have none)
// (R7: the instance of the node-range shards, swp_resolve7.hpp — the same kernel on every shard: a thread FOLDS the proposals all shards
// made for its task into one list in global node order while it stages it, wave 0 matches the block exactly as below — every shard
// computes the same picks —, and a pick is applied by the shard that owns its node. m7 / my: the other shards' proposals, this shard's number.)
template <bool CPT, bool CSI, bool R7 = false> WV_DEV void r6_commit_t(const R6Args& a, const R7Args* m7 = nullptr, u32 my = 0) {
    const u32 tid = wv::tid(), lane = wv::lane();
    const u32 pos = a.blk->pos, end = a.blk->end;
    if (pos >= end || a.blk->error != ERR_NONE) return;   // a level beyond the planes: the proposals of this round were not written
    if constexpr (R7) {
        if (r7_any_dead(m7)) return;
    }
    const u32 n = min(a.block, end - pos), Wn = a.n_words, n_rr = a.n_dc + a.n_dm;
    // (with a compact index the positions' words come first, the node words behind them: addresses as the lists carry them)
    // (R7: the TK row covers the padded half-word space of ALL shards)
    const u32 VW = CPT ? (a.blk->csize + 63u) >> 6 : 0u;
    u32 tkw = CPT ? r6_tk_words(Wn) : Wn;
    if constexpr (R7) tkw = r7_tk_words(m7);
    const u32 rnd = a.blk->rounds;   // (R7: the same on every shard — the parity of the trailer slots)
    u64* tk = wv::lds();                                         // [tkw] addresses taken by this block so far
    i64* thr = reinterpret_cast<i64*>(tk + tkw);                 // [n_rr] the demand-class thresholds
    u32* pk_node = reinterpret_cast<u32*>(thr + n_rr);           // [block] node, R6_NONE = no suitable node
    u32* pk_idx = pk_node + a.block;                             // [block] commit index / index among the unplaceable tasks
    u32* pk_aux = pk_idx + a.block;                              // [block] exception-list entry (LIST_EMPTY: plain node) / commits before an unplaceable task
    u32* sh = pk_aux + a.block;                                  // [0] accepted, [1] ncommit, [2] ninf, [3] tasks decided so far, [4] the matching is over
    u32* staged = sh + 16;                                       // [16] group g's lists are in LDS
    u32* L_hb = sh + 32;                                         // [2 * R6_CAND][block] the block's lists, entry-major: lane i of the matcher reads
    u32* L_cur = L_hb + (size_t)2 * R6_CAND * a.block;           // ... entry k of task i at [k * block + i] (no bank conflicts); [block] a lower bound of every task's cursor: the entries in front of it are dead
    unsigned short* L_hw = reinterpret_cast<unsigned short*>(L_cur + a.block);   // [2 * R6_CAND][block] the half-word indices, 16 bits each (the engine refuses node sets beyond 2^21 nodes)
    u32* H_level = reinterpret_cast<u32*>(L_hw + (size_t)2 * R6_CAND * a.block);   // R7 only: [block] the folded record's level ...
    u32* H_meta = H_level + a.block;                                                // ... and its list length | R7M_* flags
    for (u32 j = tid; j < a.block; j += R6_COMMIT_THREADS) L_cur[j] = 0;
    for (u32 w = tid; w < (R7 ? tkw : Wn + VW); w += R6_COMMIT_THREADS) tk[w] = 0;
    for (u32 c = tid; c < n_rr; c += R6_COMMIT_THREADS) thr[c] = a.thr[c];
    if (tid < 16) staged[tid] = 0;
    // Wave v >= 1 applies the picks of the block's group v - 1 (tasks 64 (v - 1) ...) as soon as wave 0 has matched that group, while
    // it matches the next ones; the task records are requested now. The groups no wave is left for (a block of more than 960 tasks)
    // are applied by wave 0 behind its matching.
    const u32 wave_ = wv::wave();
    const u32 mine = wave_ == 0 ? 15u * 64u + lane : (wave_ - 1u) * 64u + lane;   // the block-local task this thread applies
    RTask r{};
    if (mine < n) r = a.rt[pos + mine];
    if (tid == 0) { sh[3] = 0; sh[4] = 0; }
    if constexpr (R7 && CSI) {
        if (tid == 0) sh[12] = r7_take_trailers(a, m7, my, rnd);   // the volumes the other shards reserved in the round before: into this shard's table
    }
    const bool prof = (a.dbg & 16u) != 0;
    const u64 t0 = prof ? wv::clock64() : 0;
    wv::barrier();
    // every thread stages the lists of its task (wave g: the block's group g): the matcher walks them with a cursor (an entry is looked
    // at once) instead of holding all of them in registers. No barrier: a group's flag is published when its lists are in LDS.
    if constexpr (R7) {
        // A few folds at a time, in group order: a fold is three dependent batches of uncoalesced loads, and the batches of eleven waves
        // queue up in the CU's one address path — the matcher's own group (wave 0 stages it) then waits behind all the others': 27.8 k
        // cycles before the matching starts. Alone a fold takes 11.9 k (latency bound), but then the chain is slower than the matcher
        // (a group is matched in ≈ 11 k): measured on cfg4 200k x 40k over 4 / 8 shards, window 1: 33.6 / 45.1 ms, 2: 29.8 / 38.1,
        // 4: 30.1 / 37.2, all at once: 30.9 / 38.4. (Between GPUs a batch takes longer — peer reads — so rather more in flight than fewer.)
        const u32 fw = ((a.dbg >> 8) & 15u) ? ((a.dbg >> 8) & 15u) : R7_FOLDS_IN_FLIGHT;
        if (wave_ >= fw)
            while (wv::lds_poll32(staged + (wave_ - fw)) == 0) wv::spin_pause();
        if ((tid & ~63u) < n) r7_fold_into(m7, tid < n ? tid : 0u, tid < n, a.block, L_hw, L_hb, H_level, H_meta, sh);   // (whole waves: it ballots)
    } else {
        // (in group order, a few at a time — as the folds above: every wave's 64 uncoalesced loads per lane share the CU's one address path)
        const u32 sw = ((a.dbg >> 8) & 15u) ? ((a.dbg >> 8) & 15u) : R6_STAGES_IN_FLIGHT;
        if (wave_ >= sw)
            while (wv::lds_poll32(staged + (wave_ - sw)) == 0) wv::spin_pause();
        if (tid < n) {
            const R6Prop* q = a.prop + tid;
            const u32* qh = reinterpret_cast<const u32*>(q->hw);   // (two indices a dword)
            for (int k = 0; k < R6_CAND; ++k) {
                const u32 v = qh[k];
                L_hw[(size_t)(2 * k) * a.block + tid] = (unsigned short)(v & 0xFFFFu);
                L_hw[(size_t)(2 * k + 1) * a.block + tid] = (unsigned short)(v >> 16);
            }
            for (int k = 0; k < 2 * R6_CAND; ++k) L_hb[(size_t)k * a.block + tid] = q->hb[k];
        }
    }
    wv::lockstep();
    if (lane == 0) wv::lds_publish32(staged + wave_, 1u);
    const u64 t1 = prof ? wv::clock64() : 0;
    u32 reseats = 0, seat_steps = 0;
    u64 cy_load = 0, cy_walk = 0, cy_g0 = 0, cy_g1 = 0, cy_g2 = 0;
    if (wave_ == 0) {
        u32 nc = a.ctl->ncommit, ni = a.ctl->ninf, acc = 0, why = 0;
        u32* tk32 = reinterpret_cast<u32*>(tk);   // the same row as 32-node half-words
        bool stop = false, csi_seen = false;   // csi_seen: a task with cluster mounts is decided in this block already
        // the scalar part of a task's record (level, list length, exception-list candidate); the next group's is in flight while a group is matched
        struct Head { u32 level, n_cand; u64 exc_hi, exc_lo; u32 exc_entry, flags; };
        auto head_of = [&](u32 j) {
            const R6Prop* q = a.prop + j;
            return Head{q->level, q->n_cand, q->exc_hi, q->exc_lo, q->exc_entry, q->flags};
        };
        // (R7: the folded heads are in LDS once the group is staged — read behind the wait below; only the block's first task may use its
        // exception-list candidate, whose record its staging thread left in sh[8..10])
        auto head7 = [&](u32 j) {
            const u32 mt = H_meta[j];
            return Head{H_level[j], mt & 0xFFu, (mt & R7M_EXC) ? 0ull : KEY_NONE, 0ull, 0u, ((mt & R7M_UNC) ? 1u : 0u) | ((mt & R7M_CSI) ? 2u : 0u)};
        };
        Head nxt{}, nx2{};   // the heads of the next group and of the one behind it (a group that is one run of identical tasks is through
                              // before a load issued at its start has answered: two groups ahead since round 6)
        if constexpr (!R7) {
            nxt = head_of(lane < n ? lane : 0);
            nx2 = head_of(64u + lane < n ? 64u + lane : 0);
        }
        if (R7 && CSI && sh[12]) csi_seen = true;   // a reservation arrived from another shard only now: the proposals of tasks with mounts were made without it
        for (u32 g0 = 0; g0 < n && !stop; g0 += 64) {
            const u32 i = g0 + lane, glim = min(64u, n - g0);
            const bool have = i < n;
            const u32 tmid = (!R7 && a.tmpl != nullptr && have) ? a.tmpl[pos + i] : 0u;   // the task's descriptor id (a RUN of identical tasks: below); requested ahead of the wait
            const u64 tg0 = prof ? wv::clock64() : 0;
            while (wv::lds_poll32(staged + (g0 >> 6)) == 0) wv::spin_pause();   // (long there, but for the first groups of a block)
            const u64 tga = prof ? wv::clock64() : 0;
            Head rec = nxt;
            if constexpr (R7) rec = head7(have ? i : 0u);
            const Head* p = &rec;
            if constexpr (!R7) {
                nxt = nx2;
                if (g0 + 128 < n) nx2 = head_of(i + 128 < n ? i + 128 : 0);
            }
            const u32 level = have ? p->level : 0u;
            const u32 nent = (have && level != R6_NONE) ? (p->n_cand & 0x7FFFFFFFu) : 0u;
            const bool plain = nent != 0;
            const bool exc = have && level == R6_NONE && p->exc_hi != KEY_NONE;
            const bool inf = have && level == R6_NONE && !exc;
            // Every lane carries two half-words of its list — the current one (bits, w) and the next one with a candidate left
            // (bits2, w2), which it steps to inside the walk — and a cursor behind them. An entry is looked at ONCE, against the TK
            // row as it is then (the picks of the earlier groups, and of this group's tasks in front of a stop): what it finds empty
            // stays empty, what it seats is struck by the walk from then on.
            u32 bits = 0, w = 0, bits2 = 0, w2 = 0, cur = have ? L_cur[i] : 0u;   // (the group's applying wave has skipped the dead entries so far)
            // (straight-line: every lane reads R6_SEAT entries — clamped to its list, the unused entries of a list are zero — and the TK
            // words behind them, the seats are filled by selects: a step is two LDS round trips and no divergent branch. An entry is
            // consumed while a seat is free; the first one that finds none stays for the next visit, and so do those behind it)
            const u32 li = have ? i : 0u;
            auto seat = [&](bool want, u32 opt_steps) {   // lanes with `want` fill their free seats from their cursor on
                for (u32 step = 0;; ++step) {   // (after opt_steps steps: only as long as a lane has NO seat)
                    const bool go = want && bits2 == 0 && cur < nent;
                    if (!(step < opt_steps ? wv::ballot(go) : wv::ballot(go && bits == 0))) break;
                    ++seat_steps;
                    u32 h[R6_SEAT], b[R6_SEAT], t[R6_SEAT];
                    WV_UNROLL
                    for (int q = 0; q < R6_SEAT; ++q) {
                        const u32 c = min(cur + (u32)q, 2u * R6_CAND - 1u);
                        h[q] = L_hw[c * a.block + li];
                        b[q] = L_hb[c * a.block + li];
                    }
                    WV_UNROLL
                    for (int q = 0; q < R6_SEAT; ++q) t[q] = tk32[h[q]];
                    bool open = go;
                    u32 adv = 0;
                    WV_UNROLL
                    for (int q = 0; q < R6_SEAT; ++q) {
                        open = open && cur + (u32)q < nent && bits2 == 0;
                        const u32 c = open ? b[q] & ~t[q] : 0u;
                        const bool first = bits == 0;
                        bits2 = (c && !first) ? c : bits2;
                        w2 = (c && !first) ? h[q] : w2;
                        w = (c && first) ? h[q] : w;
                        bits = (c && first) ? c : bits;
                        adv += open ? 1u : 0u;
                    }
                    cur += adv;
                }
            };
            // ---- A group that is ONE RUN of identical tasks (service-major batches: a service's tasks follow each other) needs no walk.
            // Identical tasks have the same candidates in the same order, and each takes the first one nobody took: task j of the run
            // takes the (j + 1)-th candidate that is still free when the group starts — of lane 0's list, which starts in front of all
            // the others' (a twin's list starts a twin's share further on: k_r6_propose; the candidates in front of lane 0's are taken
            // by then). Lane h looks at entry h of that list behind its cursor and counts what the TK row leaves of it; a prefix sum
            // places the half-words on the lanes' ranks (a marker at every half-word's first rank, summed up: the half-word a rank
            // falls into); every lane picks its bit. If the list holds fewer free candidates than the group has tasks, the ordinary
            // seating and walk take over (they cut the block where the list is exhausted).
            // (A group usually holds the end of one service's run and the start of the next: up to eight runs — or stretches of a run that
            // one list's free candidates cover — in front of the group's first task that is not plain-and-counted are taken this way, one after the other — a run's picks are in the TK row before the
            // next run counts; the lanes behind them are seated and walked as ever, from there.)
            u32 fdone = 0;   // lanes [0, fdone) have their picks from the run path
            u32 fr_w = 0, fr_b = 0;
            if constexpr (!R7) {
                if (a.tmpl != nullptr && g0 + 64u <= a.block) {   // (the scratch below: 64 pick slots from g0 on)
                    const bool simple = have && plain && !(p->flags & 3u);
                    const u64 ns = ~wv::ballot(simple);
                    const u32 nsimple = min(glim, ns ? (u32)wv::ffs64(ns) : 64u);
                    for (u32 it = 0; it < 8u && fdone + 8u <= nsimple; ++it) {
                        const u32 ra = fdone, t_ra = wv::readlane(tmid, ra);
                        const u64 df = wv::ballot(lane >= ra && lane < nsimple && tmid != t_ra);
                        u32 rb = df ? (u32)wv::ffs64(df) : nsimple, rl = rb - ra;
                        if (rl < 8u) break;   // (a short run: the walk is as good)
                        const u32 cur0 = wv::readlane(cur, ra), nent0 = wv::readlane(nent, ra);
                        const u32 k = cur0 + lane;
                        const bool hv = lane < 32u && k < nent0;
                        const u32 hwv = hv ? L_hw[(size_t)k * a.block + g0 + ra] : 0u;
                        const u32 hbv = hv ? L_hb[(size_t)k * a.block + g0 + ra] : 0u;
                        const u32 av = hv ? hbv & ~tk32[hwv] : 0u;
                        const u32 cnt = (u32)wv::popc64((u64)av);
                        const u32 incl = wv::scan_incl_u32(cnt), excl = incl - cnt;
                        const u32 free_ = wv::readlane(incl, 63);
                        if (free_ < rl) {   // lane ra's list holds fewer free candidates than the run has tasks: it serves as many, the next turn
                            if (free_ < 8u) break;   // goes on with the list of the first lane behind them (it starts further into the level)
                            rl = free_;
                            rb = ra + rl;
                        }
                        const u32 ord = wv::mbcnt(wv::ballot(cnt != 0));
                        pk_node[g0 + lane] = 0;   // (scratch: this group's pick slots are written further down, read by its applying wave behind the publish)
                        wv::wave_sync();
                        if (cnt != 0) {
                            if (ra + excl < 64u) pk_node[g0 + ra + excl] = 1;   // a marker at the lane whose rank is the half-word's first
                            pk_idx[g0 + ord] = hwv;
                            pk_idx[g0 + 32u + ord] = av;
                            pk_aux[g0 + ord] = excl;
                        }
                        wv::wave_sync();
                        const u32 upto = wv::scan_incl_u32(pk_node[g0 + lane]);   // half-words that begin at or in front of this lane's rank
                        const bool inr = lane >= ra && lane < rb;
                        const u32 oo = inr ? upto - 1u : 0u;
                        const u32 mav = pk_idx[g0 + 32u + oo], mhw = pk_idx[g0 + oo], rk = inr ? (lane - ra) - pk_aux[g0 + oo] : 0u;
                        u32 bsel = mav;
                        for (u32 t = 0; t < 32u; ++t) {
                            const bool more_ = inr && t < rk;
                            if (!wv::ballot(more_)) break;
                            if (more_) bsel &= bsel - 1u;
                        }
                        if (inr) {
                            fr_w = mhw;
                            fr_b = bsel & (0u - bsel);
                            wv::lds_or32(tk32 + fr_w, fr_b);   // (the next run, the seating and the later groups meet it there)
                        }
                        wv::wave_sync();
                        fdone = rb;
                    }
                }
            }
            const u64 tgb = prof ? wv::clock64() : 0;
            if (fdone < glim) seat(plain && lane >= fdone, 1u);   // (a second optional step costs more than the stops it saves)
            const u64 tgc = prof ? wv::clock64() : 0;
            if (prof) { cy_g0 += tga - tg0; cy_g1 += tgb - tga; cy_g2 += tgc - tgb; }
            const u64 lanes = glim == 64 ? ~0ull : (1ull << glim) - 1ull;
            const u64 m_plain = wv::ballot(plain), m_inf = wv::ballot(inf), m_exc = wv::ballot(exc), m_unc = wv::ballot(plain && (p->flags & 1u));
            // the group ends in front of a task that must use its exception list (its order moves with every placement of the service:
            // only a block's first task may) and behind an uncounted task (its node stays on its level: the later lists are stale about it)
            u32 cut = glim;
            bool last = false;   // the block ends with this group even if the group is walked to its end
            if (m_exc) { cut = (u32)wv::ffs64(m_exc); why = 2; }
            if (m_unc && (u32)wv::ffs64(m_unc) < cut) { cut = (u32)wv::ffs64(m_unc) + 1; why = 3; last = true; }
            {   // tasks with cluster mounts: the block's first one is decided, the block is cut in front of the next (their lists know nothing of each other's volumes)
                const u64 all_csi = wv::ballot(have && (p->flags & 2u));
                u64 m_csi = all_csi;
                if (m_csi && !csi_seen) m_csi &= m_csi - 1ull;   // (the first one of the block stays)
                if (m_csi && (u32)wv::ffs64(m_csi) < cut) { cut = (u32)wv::ffs64(m_csi); why = 2; last = false; }
                if (all_csi & (cut >= 64 ? ~0ull : (1ull << cut) - 1ull)) csi_seen = true;
            }
            u32 m_pick = R6_NONE;
            if (g0 == 0 && (m_exc & 1ull)) {   // the block's first task, from its exception list; the block ends behind it
                if (lane == 0) {
                    pk_node[0] = (u32)p->exc_lo + 64u * VW;   // (an address, like the matcher's picks)
                    pk_idx[0] = nc;
                    pk_aux[0] = p->exc_entry;
                    if constexpr (R7) {
                        pk_node[0] = r7_addr(m7, sh[8], sh[9]);
                        pk_aux[0] = sh[10];
                    }
                }
                ++nc;
                acc = 1;
                why = 2;
                break;
            }
            // the unrolled walk (wv::match_seq64) passes every lane in order: it serves the plain tasks in front of the cut, the others
            // carry a dummy, and the lane at the cut (if the cut is inside the group) stops it with empty bits
            const bool served = plain && lane < cut;
            if (!served) { bits = lane == cut ? 0u : 1u; bits2 = 0; w = WV_DUMMY_W | lane; }
            u32 pickb = 0, from = fdone;
            u32 flushed = fdone;   // picks of lanes < flushed are in the TK row
            const u64 tg1 = prof ? wv::clock64() : 0;
            if (lane < fdone) { w = fr_w; pickb = fr_b; }   // (the runs' picks are known: the walk starts behind them)
            if (fdone < cut) for (;;) {
                const u32 at = wv::match_seq64(bits, w, bits2, w2, pickb, lane, from);
                if (at >= cut) break;
                ++reseats;
                // task `at` ran out of both its half-words — and so, usually, did others that sat on them: this group's picks so far go to
                // the TK row (a pick = the lowest bit the lane had at its turn, in the half-word it still sits on), and every lane still
                // to be served fills its free seats from its cursor on
                if (served && lane >= flushed && lane < at) wv::lds_or32(tk32 + w, pickb & (0u - pickb));
                flushed = at;
                wv::lockstep();   // one wave's LDS operations execute in order: the reads below see the atomics above
                {
                    seat(served && lane >= at, 1u);
                    if (wv::readlane(bits, at) == 0) {   // every listed node is taken: propose again against the new state
                        if (prof && lane == at) {
                            if (p->n_cand >> 31) a.blk->dbg_cut[0] += 1;
                            if (CPT && L_hw[li] < 2u * VW) a.blk->dbg_cut[1] += 1;
                            if ((p->n_cand & 0x7FFFFFFFu) == 1) a.blk->dbg_cut[2] += 1;
                        }
                        cut = at;
                        why = 1;
                        break;
                    }
                }
                from = at;
            }
            if (prof) { const u64 tg2 = wv::clock64(); cy_load += tg1 - tg0; cy_walk += tg2 - tg1; }
            if (served && lane < cut) m_pick = (w << 5) + (u32)wv::ffs64((u64)pickb);
            if (served && lane >= flushed && lane < cut) wv::lds_or32(tk32 + w, pickb & (0u - pickb));   // for the later groups
            const u64 below = cut == 64 ? ~0ull : (1ull << cut) - 1ull;
            const u64 mc = m_plain & below & lanes, mi = m_inf & below & lanes;
            if (lane < cut && have) {
                if (plain) {
                    pk_node[i] = m_pick;
                    pk_idx[i] = nc + wv::mbcnt(mc);
                    pk_aux[i] = LIST_EMPTY;
                } else {   // no suitable node: final whatever the earlier tasks of the block did (feasibility only shrinks)
                    pk_node[i] = R6_NONE;
                    pk_idx[i] = ni + wv::mbcnt(mi);
                    pk_aux[i] = nc + wv::mbcnt(mc);
                }
            }
            nc += (u32)wv::popc64(mc);
            ni += (u32)wv::popc64(mi);
            acc = g0 + cut;
            if (cut < glim || (last && why == 3)) stop = true;
            else why = 0;
            wv::wave_sync();
            if (lane == 0) wv::lds_publish32(sh + 3, acc);   // the picks of tasks < acc are in the pk arrays: their wave applies them now
        }
        if (lane == 0) {
            sh[0] = acc;
            sh[1] = nc;
            sh[2] = ni;
            a.blk->pos = pos + acc;   // (every thread read the old position before the barrier above)
            a.ctl->ncommit = nc;
            a.ctl->ninf = ni;
            wv::lds_publish32(sh + 3, acc);   // (the first task from its exception list leaves the loop before the publish above)
            wv::lds_publish32(sh + 4, 1u);
            a.blk->rounds += 1;
            if (CPT && VW) a.blk->crounds += 1;   // a round with a compact index
            if (why == 1) a.blk->cut_exhausted += 1;
            if (why == 2) a.blk->cut_exception += 1;
            if (why == 3) a.blk->cut_uncounted += 1;
            a.blk->reseats += reseats;   // (the host's sign of sparse lists: run_blocks switches the compact index on by it)
            if (prof) {
                a.blk->cyc[0] += (u32)((t1 - t0) >> 6);
                a.blk->cyc[1] += (u32)((wv::clock64() - t1) >> 6);
                a.blk->cyc_load += (u32)(cy_load >> 6);
                a.blk->cyc_walk += (u32)(cy_walk >> 6);
                a.blk->cyc_g[0] += (u32)(cy_g0 >> 6);
                a.blk->cyc_g[1] += (u32)(cy_g1 >> 6);
                a.blk->cyc_g[2] += (u32)(cy_g2 >> 6);
                a.blk->cyc_g[3] += seat_steps;
            }
        }
    }
    const u64 t2 = prof ? wv::clock64() : 0;
    const u32 base = a.blk->base;
    // a wave waits (polling LDS, asleep in between) until its group is matched or the matching is over, and applies what was accepted of it
    if (wave_ != 0) {
        const u32 g0 = (wave_ - 1u) * 64u, need = min(g0 + 64u, n);
        if (g0 > 0 && g0 < n)
            while (wv::lds_poll32(staged + (g0 >> 6)) == 0) wv::spin_pause();
        if (g0 > 0 && mine < n) {
            // until the matcher reaches this wave's group: move every task's cursor over the entries that are dead by now (taken nodes
            // stay taken: what is dead against an older TK row is dead), so that the matcher seats the group in a step or two
            u32 lv = 0, nent = 0;
            if constexpr (R7) { lv = H_level[mine]; nent = lv != R6_NONE ? (H_meta[mine] & 0xFFu) : 0u; }
            else { lv = a.prop[mine].level; nent = lv != R6_NONE ? (a.prop[mine].n_cand & 0x7FFFFFFFu) : 0u; }
            const u32* tk32 = reinterpret_cast<const u32*>(tk);
            u32 cur = 0;
            while (wv::lds_poll32(sh + 3) + 64u < g0 && wv::lds_poll32(sh + 4) == 0) {   // (stops a group early: the last value must be in LDS when it is read)
                for (int q = 0; q < 4 && cur < nent; ++q) {
                    if (L_hb[(size_t)cur * a.block + mine] & ~tk32[L_hw[(size_t)cur * a.block + mine]]) break;
                    ++cur;
                }
                L_cur[mine] = cur;
                wv::spin_pause();
            }
        }
        if (g0 < n)
            while (wv::lds_poll32(sh + 3) < need && wv::lds_poll32(sh + 4) == 0) wv::spin_pause();
    }
    const u32 acc_now = wv::readfirstlane(wv::lds_poll32(sh + 3));
    if (mine < acc_now) {
        const u32 t = pos + mine, addr = pk_node[mine];
        u32 nd = addr;
        if (CPT) nd = addr == R6_NONE ? R6_NONE : addr < 64u * VW ? a.cidx[addr] : addr - 64u * VW;   // a compact position, or a node behind them
        bool here = true;   // R7: the pick lies in this shard's range
        if constexpr (R7) {
            if (addr != R6_NONE) nd = r7_local(m7, addr, my, &here);
        }
        if (nd == R6_NONE) {   // (every shard records the unplaceable tasks: each explains them over its own nodes)
            a.inf_task[pk_idx[mine]] = t;
            a.inf_pos[pk_idx[mine]] = pk_aux[mine];
        } else if (here) {
            const u32 w = nd >> 6, ci = pk_idx[mine], entry = pk_aux[mine];
            const u64 bit = 1ull << (nd & 63);
            // the node row, requested in one go. No two picks of one block share a node: plain read-modify-write of the row; bitmap
            // words are shared between nodes: atomics
            const i64 qc = a.cpu[nd] - r.cpu, qm = a.mem[nd] - r.mem;
            const u32 old = a.total[nd];
            const int32_t prev = a.last[nd];
            // the node leaves the demand-class rows whose threshold lies in (new residual, old residual]: the rows above the old
            // residual never held it. Thresholds ascend; a batch may have thousands of them, so the upper end is found by bisection.
            if (r.cpu) {
                a.cpu[nd] = qc;
                for (int c = (int)r6_first_above(thr, a.n_dc, qc + r.cpu) - 1; c >= 0 && thr[c] > qc; --c) wv::g_andn64(a.rr + (size_t)c * Wn + w, bit);
            }
            if (r.mem) {
                a.mem[nd] = qm;
                for (int c = (int)r6_first_above(thr + a.n_dc, a.n_dm, qm + r.mem) - 1; c >= 0 && thr[a.n_dc + c] > qm; --c)
                    wv::g_andn64(a.rr + (size_t)(a.n_dc + c) * Wn + w, bit);
            }
            if (r.flags & RT_PORTS)
                for (u32 q = a.pset_off[r.pset]; q < a.pset_off[r.pset + 1]; ++q) wv::g_or64(a.portmap + (size_t)a.pset_ids[q] * Wn + w, bit);
            if (a.n_rg) {   // Claim (resource_management.go:11-72): the count drops by the request; the node leaves the kind's rows it no longer meets
                const u32 gset = a.tg[t];
                for (u32 g = a.gs_off[gset]; g < a.gs_off[gset + 1]; ++g) {
                    const u32 row = a.gs_row[g];
                    int32_t* cp = a.gcnt + (size_t)a.rg_kind[row] * a.gstride + nd;
                    const int32_t c = *cp - a.rg_val[row];
                    *cp = c;
                    for (u32 r2 = a.rg_k0[row]; r2 < a.rg_k1[row]; ++r2)
                        if (a.rg_val[r2] > c) wv::g_andn64(a.rg + (size_t)r2 * Wn + w, bit);
                }
            }
            if (!(r.flags & RT_UNCOUNTED)) {
                a.total[nd] = old + 1;
                const u32 rl = old - base, nl = rl + 1, xm = rl ^ nl;   // the bits a +1 flips: a run of ones from bit 0
                for (u32 b = 0; b < R6_NP && ((xm >> b) & 1u); ++b) wv::g_xor64(a.planes + (size_t)b * Wn + w, bit);
                wv::g_max32(&a.blk->maxrel, nl);
                if (nl >> R6_NP) a.blk->error = ERR_LEVEL_RANGE;
                if (entry == LIST_EMPTY) {
                    wv::g_or64(a.X + (size_t)r.svc * a.xs + w, bit);
                    a.list_node[r.slot] = nd;
                    a.list_svc[r.slot] = 1;
                    a.list_fail[r.slot] = 0;
                } else
                    a.list_svc[entry] += 1;
            }
            a.log_node[ci] = nd;
            a.log_task[ci] = t;
            a.log_prev[ci] = prev;
            a.last[nd] = (int32_t)ci;
            a.out_node[t] = (int32_t)nd;
            if (CSI && a.csi_of) {   // chooseTaskVolumes + reserveTaskVolumes on the node (scheduler.go:857-874); a mount without a volume: assigned without attachments
                const u32 ck = a.csi_of[t];
                if (ck != R6_NONE) {
                    u32 att[VOL_MAX_MOUNTS];
                    const u32 set = a.csi_set[ck], na = vol_choose(a.vol, set, nd, att, nullptr);
                    if (na) vol_reserve(a.vol, set, nd, att, na);
                    for (u32 q = 0; q < VOL_MAX_MOUNTS; ++q) a.att[(size_t)ck * VOL_MAX_MOUNTS + q] = att[q];
                    if constexpr (R7) {
                        if (na) r7_leave_trailer(a, my, rnd, set, nd, att, na);   // the other shards reserve the same volumes at the start of the next round
                    }
                }
            }
        }
    }
    if (prof && wave_ == 1 && lane == 0) a.blk->cyc[3] += (u32)((wv::clock64() - t2) >> 6);   // the first group's applying wave: waiting for it + applying
    if constexpr (CPT && !R7) {
        if (a.compact == 2u) {   // the next round's compact index, here instead of in a launch of its own (r6_compact_body)
            wv::wait_vm();       // this wave's atomics on the rows have arrived ...
            wv::barrier();       // ... and every other wave's; nobody reads the TK row any more (the body's 32 words lie on it)
            const u32 np = pos + sh[0];
            const bool go = np < end && wv::readfirstlane(wv::g_fresh32(&a.blk->error)) == ERR_NONE;
            wv::barrier();       // (sh[0] is read before the body's first barrier lets anybody write LDS)
            if (go) r6_compact_body<true>(a, np, reinterpret_cast<u32*>(wv::lds()));
        }
    }
}
WV_KERNEL(R6_COMMIT_THREADS) void k_r6_commit(R6Args a) { r6_commit_t<false, false>(a); }
WV_KERNEL(R6_COMMIT_THREADS) void k_r6_commit_c(R6Args a) { r6_commit_t<true, false>(a); }   // ... with a compact index (launched behind k_r6_compact only)
WV_KERNEL(R6_COMMIT_THREADS) void k_r6_commit_v(R6Args a) { r6_commit_t<false, true>(a); }   // ... for a batch with cluster mounts (R6Args.csi_of)

#endif   // SWP_R6_KERNELS

}  // namespace swpdev
