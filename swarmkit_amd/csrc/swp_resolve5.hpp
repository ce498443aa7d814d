// swp_resolve5.hpp — k_resolve5, the ROUND resolver: the sequential argmin + commit pass of the tick
// (nodeSet.tree with a heap of one, nodeset.go:50-124; nodeLess, scheduler.go:708-735; NodeInfo.addTask,
// nodeinfo.go:108-154) decided R5_B tasks at a time by ONE workgroup of 16 wavefronts.
//
// Why rounds. A lone wavefront retires one instruction per ≈ 4 ns; a resolver that decides task after task on one wave (round
// 1: ≈ 85 instructions per task) IS the whole batch time. The work per task that is really sequential is tiny — "take the first
// candidate nobody before you took" — while finding the candidates (157 words of F & ~X against the node levels at 10k nodes,
// F = static class & demand-class rows) is wide and independent of the other tasks of the round as long as every node is
// taken at most once per round. So:
//
//   lister waves 1..14  build, for each task of the NEXT round, a candidate list: the first R5_Q non-empty words of
//                       F & ~X restricted to the task's minimum level, in node order. ResourceFilter.Check (filter.go:77-84)
//                       is membership in two demand-class rows (the batch's distinct reservations, kept exact in LDS by
//                       every commit). One wave per task, 4 tasks per
//                       wave and round; everything is word-parallel over the lanes {lane + 64k}. Levels come from a
//                       ring of R5_J per-level node masks in LDS (the levels where the picks happen) and, for anything
//                       below or above the ring, from the bit-planes of all levels.
//   wave 0 (matcher)    walks the CURRENT round's tasks in order on the scalar unit: task i takes the lowest candidate
//                       bit of its current word and the bit is struck from every lane that sits on the same word.
//                       Then it applies the round to the LDS state (planes, ring, residuals) and leaves a hand-over record.
//   wave 15 (committer) applies the memory side effects of the PREVIOUS round from that record while the next round is
//                       matched: residual update of the node row, task count, X row, exception list, commit log and its
//                       per-node chain, placement. Nothing on the matcher's path ever waits for memory.
//
// Exactness (the list rule). Inside a batch a node's level only grows and feasibility only shrinks. A task's list holds
// ALL its feasible plain nodes of its minimum level L in node order up to the last listed word; a node taken since the
// list's snapshot (by the previous round, whose picks are in the TK rows, or by an earlier task of this round) moves to
// level L+1 > L, so "first listed bit nobody took" is exactly the sequential argmin(level, index) — as long as the list
// is not exhausted. When it is, the round is CUT there: the tasks before it are committed and the next round starts at
// that task with fresh lists (its own list cannot be exhausted at position 0). A task that must look at its service's
// exception list (no plain candidate, but F & X ≠ 0), with host ports or
// uncounted: cut, and that one task runs the generic workgroup path (r5_generic: one task decided by all 1024 threads from
// this kernel's state). A task with F == 0 and no exception candidate ("no suitable node") passes through the round as a no-op.
//
// Pipeline: while the matcher works on round r, the listers build round r+1 against the state after round r-1; the
// picks of round r are removed from those lists when the matcher loads them (TK row of the previous round).
//
// Written against swp_wave.hpp only (no raw intrinsics) so that tests/emu can run the same source on CPU fibers.
#pragma once
#include "swp_types.hpp"

namespace swpdev {

#define R5_Q 4                    // candidate words per task
#define R5_LW 14                  // lister waves (waves 1..14); wave 0 matches, wave 15 commits to memory
#define R5_TPW 4                  // task slots per lister wave and round
#define R5_B (R5_LW * R5_TPW)     // tasks per round
#define R5_CW 15                  // the committer wave
#define R5_NBMAX 8                // level planes (255 levels above the lowest valid node)
#define R5_J 4                    // levels in the mask ring
#define R5_THREADS 1024
#define R5_KMAX 4                 // node words per lane: n_words <= 256 (16 384 nodes)
#define R5_QLIM (1 << 30)         // residuals in resource units must stay below this (host checks)
#define R5_RRMAX 64               // demand-class rows (distinct cpu reservations + distinct memory reservations)
#define R5_TREC 256               // task-record ring in LDS (entries; power of two, > 3 rounds + slack)
#define R5_TREC_U32 8             // flags, svc, sc, kc, km, slot, -, -
#define R5_AHEAD (3 * R5_B + 8)   // records staged this far beyond the current round's first task

enum { R5_NONE = 0, R5_FAST = 1, R5_INFEASIBLE = 2, R5_COMPLEX = 3 };
enum { R5_CUT_NOT = 0, R5_CUT_RELIST = 1, R5_CUT_GENERIC = 2 };
enum {   // u32 scalars in LDS
    R5S_NCOMMIT = 0, R5S_NINF, R5S_BASE, R5S_NB, R5S_LB, R5S_REBUILD, R5S_OK, R5S_QUIET, R5S_ADVANCE,
    R5S_CUT0, R5S_CUT1,           // cut position of the round, by round parity
    R5S_WHY0, R5S_WHY1,           // why it was cut
    R5S_HJ0, R5S_HJ1,             // hand-over: window-local first task of the round, by round parity
    R5S_HV0, R5S_HV1,             // hand-over record is pending
    R5S_BELANY, R5S_RETRIES, R5S_SLOW, R5S_GENERIC, R5S_REBASES, R5S_ROUNDS, R5S_FULL, R5S_CUT_CLASS, R5S_CUT_EMPTY, R5S_RINGADV, R5S_DEEP, R5S_NREL, R5S_SHIFTS,
    R5S_COUNT = 32
};
#define R5_HDR_U32 8                            // list header: class, entries, level, service, kc, km, list slot, -
#define R5_LIST_U32 (R5_HDR_U32 + R5_Q * 4)     // + entries {word index, -, bits lo, bits hi}; the matcher walks them as 32-node half-words
                                                // (a stride of 26 dwords, free of bank conflicts for the matcher's lane-per-list reads, measured 2 % slower)
#define R5_HAND_U32 8                           // hand-over record: kind, node, commit / inf index, commits before, slot, service, kc, km
enum { R5H_NONE = 0, R5H_COMMIT = 1, R5H_INF = 2 };

struct R5Lds {
    u64* planes;     // [R5_NBMAX][rs]  bit b of (level - base) per node
    u64* lv;         // [R5_J][rs]      nodes at level lb .. lb+J-1 (slot = level % J)
    u64* below;      // [rs]            nodes below the ring
    u64* tk;         // [2][rs]         picks of the previous / the current round
    u64* rr;         // [n_rr][rs]      nodes whose residual cpu (rows 0..n_dc-1) / memory (rows n_dc..) is >= the row's threshold
    int32_t* thr;    // [R5_RRMAX]      the thresholds, resource units
    u32* trec;       // [R5_TREC][R5_TREC_U32]  the listers' fields of the upcoming tasks, staged by the committer wave (ring by task index)
    u64* red;        // [64]            block reductions
    u32* lists;      // [3][R5_B][R5_LIST_U32]  current round / next round / spare (target of a carry-over)
    u32* ring;       // [2][R5_B][2]    (service, node) of the last two rounds' commits, by round parity
    u32* hand;       // [2][R5_B][R5_HAND_U32]  hand-over to the committer, by round parity
    u32* relist;     // [R5_B]          slots of the round whose lists came out empty when a cut round's lists were carried over
    u32* sh;         // [R5S_COUNT]
    int32_t* q;      // [n_nodes][2]    residual cpu / mem in resource units
    u32 rs;          // row stride in words
};

// rows are padded to whole 64-word windows: lane l of a wave owns words {l + 64 k}, k < K, all of them inside the row (no bounds checks)
inline __host__ __device__ u32 r5_row_stride(u32 n_words) { return (n_words + 63u) & ~63u; }
inline __host__ __device__ size_t r5_lds_bytes(u32 n_nodes, u32 n_words, u32 n_rr) {
    const size_t rs = r5_row_stride(n_words);
    return (size_t)(R5_NBMAX + R5_J + 1 + 2 + n_rr) * rs * 8 + 64 * 8 + (size_t)3 * R5_B * R5_LIST_U32 * 4 + (size_t)2 * R5_B * 2 * 4 +
           (size_t)2 * R5_B * R5_HAND_U32 * 4 + (size_t)R5_B * 4 + R5S_COUNT * 4 + R5_RRMAX * 4 + (size_t)R5_TREC * R5_TREC_U32 * 4 + (size_t)n_nodes * 8;
}
WV_DEV R5Lds r5_layout(u64* lds, u32 n_words, u32 n_rr) {
    R5Lds L;
    L.rs = r5_row_stride(n_words);
    L.planes = lds;
    L.lv = L.planes + (size_t)R5_NBMAX * L.rs;
    L.below = L.lv + (size_t)R5_J * L.rs;
    L.tk = L.below + L.rs;
    L.rr = L.tk + 2 * L.rs;
    L.red = L.rr + (size_t)n_rr * L.rs;
    L.lists = reinterpret_cast<u32*>(L.red + 64);
    L.ring = L.lists + 3 * R5_B * R5_LIST_U32;
    L.hand = L.ring + 2 * R5_B * 2;
    L.relist = L.hand + 2 * R5_B * R5_HAND_U32;
    L.sh = L.relist + R5_B;
    L.thr = reinterpret_cast<int32_t*>(L.sh + R5S_COUNT);
    L.trec = reinterpret_cast<u32*>(L.thr + R5_RRMAX);
    L.q = reinterpret_cast<int32_t*>(L.trec + R5_TREC * R5_TREC_U32);
    return L;
}

// ---- block reductions (all 1024 threads; one barrier each) ---------------------------------------------------------
WV_DEV u32 r5_block_min32(u32 v, const R5Lds& L, u32& par) {
    u32 m = wv::min_u32(v);
    u32* r = reinterpret_cast<u32*>(L.red) + par * 32;
    if (wv::lane() == 0) r[wv::wave()] = m;
    wv::barrier();
    u32 g = r[0];
    for (u32 i = 1; i < R5_THREADS / 64; ++i) g = min(g, r[i]);
    par ^= 1;
    return g;
}
WV_DEV u64 r5_block_min64(u64 v, const R5Lds& L, u32& par) {
    u32 hi = (u32)(v >> 32), lo = (u32)v;
    u32 mh = wv::min_u32(hi);
    u32 ml = wv::min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    u64* r = L.red + 32 + par * 16;
    if (wv::lane() == 0) r[wv::wave()] = ((u64)mh << 32) | ml;
    wv::barrier();
    u64 g = r[0];
    for (u32 i = 1; i < R5_THREADS / 64; ++i) g = r[i] < g ? r[i] : g;
    par ^= 1;
    return g;
}

// ---- the level-mask ring from the planes: LV[l % J] = {level == l} for l in [lb, lb+J), BELOW = {level < lb} --------
// One thread per node word (rs <= 256 < 1024). Three barriers; ends with one.
WV_DEV void r5_ring_build(const R5Lds& L, u32 lb) {
    const u32 w = wv::tid(), NB = L.sh[R5S_NB];
    if (w < L.rs) {
        u64 eq[R5_J], lt = 0, e0 = ~0ull;
        for (int j = 0; j < R5_J; ++j) eq[j] = ((lb + j) >> NB) ? 0ull : ~0ull;   // a level the planes cannot express has no node
        for (int b = (int)NB - 1; b >= 0; --b) {
            const u64 p = L.planes[(size_t)b * L.rs + w];
            for (int j = 0; j < R5_J; ++j) eq[j] &= (((lb + j) >> b) & 1u) ? p : ~p;
            if ((lb >> b) & 1u) { lt |= e0 & ~p; e0 &= p; } else e0 &= ~p;
        }
        for (int j = 0; j < R5_J; ++j) L.lv[(size_t)((lb + j) % R5_J) * L.rs + w] = eq[j];
        L.below[w] = lt;
    }
    const bool some_below = w < L.rs && L.below[w] != 0;
    wv::barrier();   // every thread has read the flags that sent it here before they are reset
    if (wv::tid() == 0) L.sh[R5S_BELANY] = 0;
    wv::barrier();
    if (some_below) L.sh[R5S_BELANY] = 1;   // many writers, one value
    if (wv::tid() == 0) {
        L.sh[R5S_LB] = lb;
        L.sh[R5S_QUIET] = 0;
        L.sh[R5S_ADVANCE] = 0;
    }
    wv::barrier();
}

// ---- level planes from total[] (window start, level overflow). One wave per node word: a ballot IS a plane word. ----
// Returns false (uniformly) when the level span of the valid nodes does not fit R5_NBMAX planes. Ends with a barrier.
WV_DEV bool r5_build_planes(const ResolveArgs& a, const R5Lds& L, u32& par) {
    const u32 lane = wv::lane(), wave = wv::wave();
    u32 lo = 0xFFFFFFFFu, hi = 0;
    for (u32 w = wave; w < a.n_words; w += R5_THREADS / 64) {
        const u64 vm = wv::uload(a.valid + w);
        const u32 n = w * 64 + lane;
        if ((vm >> lane) & 1) {
            u32 t = wv::g_fresh32(a.total + n);
            lo = min(lo, t);
            hi = max(hi, t);
        }
    }
    lo = r5_block_min32(lo, L, par);
    hi = ~r5_block_min32(~hi, L, par);
    if (lo == 0xFFFFFFFFu) { lo = 0; hi = 0; }   // no valid node
    const u32 span = hi - lo;
    if (span >= (1u << R5_NBMAX) - 1u) return false;
    u32 need = 1;
    while ((1u << need) <= span) ++need;
    const u32 NB = min((u32)R5_NBMAX, need + 1);   // one spare bit: room to grow before the next rebuild
    for (u32 w = wave; w < L.rs; w += R5_THREADS / 64) {
        u32 lvl = 0;
        bool v = false;
        if (w < a.n_words) {
            const u64 vm = wv::uload(a.valid + w);
            v = (vm >> lane) & 1;
            if (v) lvl = wv::g_fresh32(a.total + w * 64 + lane) - lo;
        }
        for (u32 b = 0; b < R5_NBMAX; ++b) {
            u64 word = wv::ballot(v && ((lvl >> b) & 1u));
            if (lane == 0) L.planes[(size_t)b * L.rs + w] = word;
        }
    }
    if (wv::tid() == 0) {
        L.sh[R5S_BASE] = lo;
        L.sh[R5S_NB] = NB;
        L.sh[R5S_REBUILD] = 0;
    }
    wv::barrier();
    r5_ring_build(L, 0);
    return true;
}

// level of one node above the base, read back from the planes
WV_DEV u32 r5_level_of(const R5Lds& L, u32 NB, u32 w, u64 bit) {
    u32 lvl = 0;
    for (u32 b = 0; b < NB; ++b)
        if (L.planes[(size_t)b * L.rs + w] & bit) lvl |= 1u << b;
    return lvl;
}

// one node moves from level rl to rl + 1 in the LDS level structures; true when rl + 1 does not fit the planes
WV_DEV bool r5_bump_level(const R5Lds& L, u32 NB, u32 lb, u32 w, u64 bit, u32 rl) {
    const u32 nl = rl + 1, xm = rl ^ nl;   // the bits a +1 flips are a run of ones from bit 0: mostly just bit 0
    for (u32 b = 0; b < NB && ((xm >> b) & 1u); ++b) wv::lds_xor64(L.planes + (size_t)b * L.rs + w, bit);
    if (rl < lb) {
        if (nl >= lb) wv::lds_andn64(L.below + w, bit);
    } else if (rl < lb + R5_J)
        wv::lds_andn64(L.lv + (size_t)(rl % R5_J) * L.rs + w, bit);
    if (nl >= lb && nl < lb + R5_J) wv::lds_or64(L.lv + (size_t)(nl % R5_J) * L.rs + w, bit);
    return (nl >> NB) != 0;
}

// ---- the demand-class rows -------------------------------------------------------------------------------------------
// ResourceFilter.Check (filter.go:77-84) as set membership: node n passes a task with reservations (kc, km) <=> n is in
// RC[class of kc] & RM[class of km], RC[c] = {q_cpu >= thr[c]}, RM[c] = {q_mem >= thr[n_dc + c]}. Residuals only shrink inside a
// batch, so a commit can only take bits away. One wave per node word: a ballot IS a row word. Needs q and thr in LDS.
WV_DEV void r5_rr_build(const ResolveArgs& a, const R5Lds& L) {
    const u32 lane = wv::lane(), n_rr = a.n_dc + a.n_dm;
    for (u32 w = wv::wave(); w < L.rs; w += R5_THREADS / 64) {
        const u32 n = w * 64 + lane;
        const bool in = n < a.n_nodes;
        const int32_t qc = in ? L.q[2 * n] : 0, qm = in ? L.q[2 * n + 1] : 0;
        for (u32 c = 0; c < n_rr; ++c) {
            const u64 word = wv::ballot(in && (c < a.n_dc ? qc : qm) >= L.thr[c]);
            if (lane == 0) L.rr[(size_t)c * L.rs + w] = word;
        }
    }
}
// after a commit left node (w, bit) with residuals (qc, qm): the node leaves every row whose threshold it no longer meets.
// Thresholds ascend within each group, so the largest one decides whether there is anything to do (the common case: not).
WV_DEV void r5_rr_update(const R5Lds& L, u32 n_dc, u32 n_dm, u32 w, u64 bit, int32_t qc, int32_t qm) {
    if (n_dc && qc < L.thr[n_dc - 1])
        for (u32 c = 0; c < n_dc; ++c)
            if (qc < L.thr[c]) wv::lds_andn64(L.rr + (size_t)c * L.rs + w, bit);
    if (n_dm && qm < L.thr[n_dc + n_dm - 1])
        for (u32 c = 0; c < n_dm; ++c)
            if (qm < L.thr[n_dc + c]) wv::lds_andn64(L.rr + (size_t)(n_dc + c) * L.rs + w, bit);
}

// ---- task records: the six fields the listers need, from the batch's task array into the LDS ring -------------------
// The task array is read once, front to back: every record is an L2 miss. The committer wave, idle most of a round, fetches
// the records of the rounds to come; the listers then start from LDS instead of waiting for scalar loads. One lane per task.
WV_DEV void r5_stage_records(const ResolveArgs& a, const R5Lds& L, u32 from, u32 to) {
    for (u32 jj = from + wv::lane(); jj < to; jj += 64) {
        const RTask* rt = a.rt + a.j0 + jj;
        u32* o = L.trec + (size_t)(jj & (R5_TREC - 1)) * R5_TREC_U32;
        o[0] = rt->flags;
        o[1] = rt->svc;
        o[2] = rt->sc;
        o[3] = rt->kc;
        o[4] = rt->km;
        o[5] = rt->slot;
    }
}

// ---- lister: candidate lists of the round that starts at window-local task jbase, into list buffer `buf` ------------
// The static-class and X rows are requested one task ahead. A task's feasible set is sc[static class] & RC[cpu class] & RM[memory
// class], exact at the list's snapshot (the demand-class rows are kept exact by every commit).
// slots == nullptr: the wave lists the tasks of slots lw, lw + R5_LW, ... of the round that starts at jbase. Otherwise `slots`
// (LDS, n_slots entries) names the slots to list — the relist pass after a cut round — and the wave takes entries lw, lw + R5_LW, ...
template <int K>
WV_DEV void r5_list(const ResolveArgs& a, const R5Lds& L, u32 jbase, u32 buf, u32 lw, u64* lt, const u32* slots = nullptr, u32 n_slots = 0) {
    const u32 lane = wv::lane();
    u32 sl_[R5_TPW];   // slot of the wave's t-th task, 0xFFFFFFFF = none
    for (u32 t = 0; t < R5_TPW; ++t) {
        const u32 idx = lw + R5_LW * t;
        sl_[t] = slots == nullptr ? idx : (idx < n_slots ? wv::readfirstlane(slots[idx]) : 0xFFFFFFFFu);
    }
    u64 lmark = lt ? wv::clock64() : 0;
#define R5_LT(slot) do { if (lt) { const u64 n_ = wv::clock64(); lt[slot] += n_ - lmark; lmark = n_; } } while (0)
    const u32 NB = L.sh[R5S_NB], lb = L.sh[R5S_LB];
    const bool bel_any = L.sh[R5S_BELANY] != 0;
    u32 fl_[R5_TPW], sv_[R5_TPW], sc_[R5_TPW];
    for (u32 t = 0; t < R5_TPW; ++t) {
        const u32 jj = sl_[t] == 0xFFFFFFFFu ? 0xFFFFFFFFu : jbase + sl_[t];
        fl_[t] = 0;
        sv_[t] = 0;
        sc_[t] = 0;
        if (jj < a.count) {   // uniform
            const u32* rec = L.trec + (size_t)(jj & (R5_TREC - 1)) * R5_TREC_U32;   // broadcast reads, made scalar for the row addresses
            fl_[t] = wv::readfirstlane(rec[0]);
            sv_[t] = wv::readfirstlane(rec[1]);
            sc_[t] = wv::readfirstlane(rec[2]);
        }
    }
    u64 Fn[K], Xn[K];
    for (int k = 0; k < K; ++k) {
        const u32 w = lane + 64 * k;
        const u32 j0_ = sl_[0] == 0xFFFFFFFFu ? 0xFFFFFFFFu : jbase + sl_[0];
        const bool in = j0_ < a.count && w < a.n_words;
        Fn[k] = in ? a.sc[(size_t)sc_[0] * a.n_words + w] : 0ull;
        Xn[k] = in ? wv::g_fresh64(a.X + (size_t)sv_[0] * a.xs + w) : 0ull;
    }
    // the lowest ring level and BELOW of this wave's words stay in registers for the round
    u64 LV0[K], BEL[K];
    for (int k = 0; k < K; ++k) {
        const u32 w = lane + 64 * k;
        LV0[k] = L.lv[(size_t)(lb % R5_J) * L.rs + w];
        BEL[k] = L.below[w];
    }
    // (service, node) of the last two rounds' commits: lanes 0..B-1 hold one round, lanes ... the other needs a second pair
    const u32 r0_svc = lane < R5_B ? L.ring[2 * lane] : 0xFFFFFFFFu, r0_node = lane < R5_B ? L.ring[2 * lane + 1] : 0u;
    const u32 r1_svc = lane < R5_B ? L.ring[2 * (R5_B + lane)] : 0xFFFFFFFFu, r1_node = lane < R5_B ? L.ring[2 * (R5_B + lane) + 1] : 0u;
    R5_LT(0);   // prologue: task records, first row requests, ring masks

    for (u32 t = 0; t < R5_TPW; ++t) {
        const u32 s = sl_[t];   // task slot of the round == lane of the matcher
        const bool none = s == 0xFFFFFFFFu;   // (relist pass: fewer slots than waves x tasks)
        const u32 jj = none ? 0xFFFFFFFFu : jbase + s;
        u32* out = L.lists + ((size_t)buf * R5_B + (none ? 0u : s)) * R5_LIST_U32;
        u64 F[K], X[K];
        for (int k = 0; k < K; ++k) {
            F[k] = Fn[k];
            X[k] = Xn[k];
        }
        if (t + 1 < R5_TPW) {
            const u32 jn = sl_[t + 1] == 0xFFFFFFFFu ? 0xFFFFFFFFu : jbase + sl_[t + 1];
            for (int k = 0; k < K; ++k) {
                const u32 w = lane + 64 * k;
                const bool in = jn < a.count && w < a.n_words;
                Fn[k] = in ? a.sc[(size_t)sc_[t + 1] * a.n_words + w] : 0ull;
                Xn[k] = in ? wv::g_fresh64(a.X + (size_t)sv_[t + 1] * a.xs + w) : 0ull;
            }
        }
        if (none) continue;   // uniform
        if (jj >= a.count) {   // uniform
            if (lane == 0) { out[0] = R5_NONE; out[1] = 0; }
            continue;
        }
        const u32* rec = L.trec + (size_t)(jj & (R5_TREC - 1)) * R5_TREC_U32;
        const u32 flags = fl_[t], svc = sv_[t];
        if (flags & (RT_PORTS | RT_UNCOUNTED)) {
            if (lane == 0) { out[0] = R5_COMPLEX; out[1] = 0; }
            continue;
        }
        if (flags & RT_RES) {   // uniform
            const u64* rc = L.rr + (size_t)((flags >> RT_DC_SHIFT) & RT_DCLS_MASK) * L.rs;
            const u64* rm = L.rr + (size_t)(a.n_dc + ((flags >> RT_DM_SHIFT) & RT_DCLS_MASK)) * L.rs;
            for (int k = 0; k < K; ++k) F[k] &= rc[lane + 64 * k] & rm[lane + 64 * k];
        }
        u64 mk[K];
        // commits of the last two rounds may still be on their way to X in memory: patch them in from the ring
        u64 match = wv::ballot(r0_svc == svc || r1_svc == svc);
        R5_LT(1);   // rows have arrived
        while (match) {   // seldom any: one ring entry at a time, on the scalar side
            const u32 l = (u32)wv::ffs64(match);
            match &= match - 1;
            const u32 s0 = wv::readlane(r0_svc, l), n0 = wv::readlane(r0_node, l), s1 = wv::readlane(r1_svc, l), n1 = wv::readlane(r1_node, l);
            for (int k = 0; k < K; ++k) {
                if (s0 == svc && (n0 >> 6) == lane + 64 * k) X[k] |= 1ull << (n0 & 63);
                if (s1 == svc && (n1 >> 6) == lane + 64 * k) X[k] |= 1ull << (n1 & 63);
            }
        }
        u64 any_mk = 0;
        for (int k = 0; k < K; ++k) {
            mk[k] = F[k] & ~X[k];
            any_mk |= mk[k];
        }
        if (!wv::ballot(any_mk != 0)) {
            // no plain candidate: "no suitable node" unless a node of the service's exception list is feasible
            u64 any_fx = 0;
            for (int k = 0; k < K; ++k) any_fx |= F[k] & X[k];
            const u32 cls = wv::ballot(any_fx != 0) ? R5_COMPLEX : R5_INFEASIBLE;
            if (lane == 0) { out[0] = cls; out[1] = 0; }
            continue;
        }
        u64 c[K];
        u32 lvl = 0xFFFFFFFFu;
        bool below = false;
        if (bel_any) {   // uniform; the ring seldom has anything below it
            u64 in_bel = 0;
            for (int k = 0; k < K; ++k) in_bel |= mk[k] & BEL[k];
            below = wv::ballot(in_bel != 0) != 0;
        }
        if (!below) {
            u64 in_0 = 0;
            for (int k = 0; k < K; ++k) {
                c[k] = mk[k] & LV0[k];
                in_0 |= c[k];
            }
            if (wv::ballot(in_0 != 0)) lvl = lb;
            for (u32 jl = 1; jl < R5_J && lvl == 0xFFFFFFFFu; ++jl) {
                u64 any = 0;
                for (int k = 0; k < K; ++k) {
                    c[k] = mk[k] & L.lv[(size_t)((lb + jl) % R5_J) * L.rs + lane + 64 * k];
                    any |= c[k];
                }
                if (wv::ballot(any != 0)) lvl = lb + jl;
            }
        }
        if (lvl == 0xFFFFFFFFu) {
            // below or above the ring: bit-sliced minimum per word over the planes, then the wave minimum of the levels
            u32 lv[K], lmin = 0xFFFFFFFFu;
            for (int k = 0; k < K; ++k) {
                const u32 w = lane + 64 * k;
                u64 m = mk[k];
                lv[k] = 0xFFFFFFFFu;
                if (m) {
                    u32 l = 0;
                    for (int b = (int)NB - 1; b >= 0; --b) {
                        const u64 tt = m & ~L.planes[(size_t)b * L.rs + w];
                        if (tt) m = tt;
                        else l |= 1u << b;
                    }
                    lv[k] = l;
                    lmin = min(lmin, l);
                }
                c[k] = m;
            }
            lvl = wv::min_u32(lmin);
            for (int k = 0; k < K; ++k)
                if (lv[k] != lvl) c[k] = 0;
            if (lane == 0) L.sh[R5S_DEEP] += 1;   // statistics only (several waves may race: the count is approximate)
        }
        R5_LT(2);   // level search
        // the first R5_Q non-empty words in node order: the lane that holds the r-th one writes entry r itself
        // (entry = {word index, -, bits lo, bits hi}; the matcher reads it as two 32-node half-words)
        u32 cnt = 0;
        for (int k = 0; k < K; ++k) {
            const u64 bal = wv::ballot(c[k] != 0);
            if (c[k] != 0) {
                const u32 r = cnt + wv::mbcnt(bal);
                if (r < R5_Q) {
                    out[R5_HDR_U32 + 4 * r] = lane + 64 * k;
                    *reinterpret_cast<u64*>(out + R5_HDR_U32 + 4 * r + 2) = c[k];
                }
            }
            cnt += (u32)wv::popc64(bal);
        }
        cnt = min(cnt, (u32)R5_Q);
        R5_LT(3);   // entries
        const int32_t kc = (int32_t)rec[3], km = (int32_t)rec[4];
        if (lane == 0) {
            out[0] = R5_FAST;
            out[1] = 2 * cnt;
            out[2] = lvl;
            out[3] = svc;
            out[4] = (u32)kc;
            out[5] = (u32)km;
            out[6] = rec[5];
        }
        R5_LT(4);   // header
    }
#undef R5_LT
}

// ---- committer: the memory side effects of one finished round, from its hand-over record ---------------------------
// Nothing is waited for inside a round: the per-node chain link (the exchange's result) is stored by the NEXT call, and
// the X / list updates of a round become visible to the listers through the two-round ring until they have landed.
// drain = true: wait for everything and store the links now (cut, plane rebuild, end of the window).
struct R5Pend { u32 ci; int32_t prev; };
WV_DEV void r5_commit_memory(const ResolveArgs& a, const R5Lds& L, u32 hp, R5Pend& pend, bool drain) {
    const u32 lane = wv::lane();
    const u32 pending = L.sh[R5S_HV0 + hp], hj = L.sh[R5S_HJ0 + hp];
    wv::wave_sync();   // every lane has read the flag before lane 0 clears it below
    wv::wait_vm();     // the previous call's operations (issued a round ago) have landed, its exchange results are here
    if (pend.ci != 0xFFFFFFFFu) a.log_prev[pend.ci] = pend.prev;
    pend.ci = 0xFFFFFFFFu;
    if (pending) {   // uniform
        if (lane < R5_B) {
            const u32* h = L.hand + ((size_t)hp * R5_B + lane) * R5_HAND_U32;
            const u32 kind = h[0], gj = a.j0 + hj + lane;
            if (kind == R5H_COMMIT) {
                const u32 n = h[1], ci = h[2], slot = h[4], svc = h[5], w = n >> 6;
                const u64 bit = 1ull << (n & 63);
                const i64 rcpu = (i64)h[6] * a.unit_cpu, rmem = (i64)h[7] * a.unit_mem;
                if (rcpu) wv::g_add64(a.cpu + n, -rcpu);
                if (rmem) wv::g_add64(a.mem + n, -rmem);
                wv::g_add32(a.total + n, 1u);
                wv::g_or64(a.X + (size_t)svc * a.xs + w, bit);
                a.list_node[slot] = n;
                a.list_svc[slot] = 1;
                a.list_fail[slot] = 0;
                a.log_node[ci] = n;
                a.log_task[ci] = gj;
                pend.prev = (int32_t)wv::g_exch32(reinterpret_cast<u32*>(a.last + n), ci);
                pend.ci = ci;
                a.out_node[gj] = (int32_t)n;
            } else if (kind == R5H_INF) {
                a.inf_task[h[2]] = gj;
                a.inf_pos[h[2]] = h[3];
            }
        }
        wv::wave_sync();
        if (lane == 0) L.sh[R5S_HV0 + hp] = 0;
    }
    if (drain) {
        wv::wait_vm();
        if (pend.ci != 0xFFFFFFFFu) a.log_prev[pend.ci] = pend.prev;
        pend.ci = 0xFFFFFFFFu;
        wv::wait_vm();
    }
}

// ---- one task through the generic workgroup path: k_resolve's algorithm on this kernel's state ----------------------
// (plain nodes by bit-sliced minimum with re-check, then the service's exception list; scheduler.go:708-735.)
// All 1024 threads; thread t owns node word t. Ends with a barrier; counters and flags are in L.sh.
struct R5Rt { i64 cpu, mem; u32 flags, svc, slot, pset, sc; u64 maxrep; int32_t kc, km; };
WV_DEV R5Rt r5_load_rt(const RTask* rt) {
    R5Rt r;
    r.cpu = wv::uload(&rt->cpu);
    r.mem = wv::uload(&rt->mem);
    r.flags = wv::uload(&rt->flags);
    r.svc = wv::uload(&rt->svc);
    r.slot = wv::uload(&rt->slot);
    r.pset = wv::uload(&rt->pset);
    r.sc = wv::uload(&rt->sc);
    r.maxrep = wv::uload(&rt->maxrep);
    r.kc = (int32_t)wv::uload(&rt->kc);
    r.km = (int32_t)wv::uload(&rt->km);
    return r;
}

// residual update of ONE placement by ONE thread (NodeInfo.addTask, nodeinfo.go:108-154); e = exception-list entry or LIST_EMPTY
WV_DEV void r5_commit_one(const ResolveArgs& a, const R5Lds& L, const R5Rt& r, u32 gj, u32 n, u32 e) {
    const u32 NB = L.sh[R5S_NB], w = n >> 6;
    const u64 bit = 1ull << (n & 63);
    const u32 ci = L.sh[R5S_NCOMMIT];
    if (r.cpu) wv::g_add64(a.cpu + n, -r.cpu);
    if (r.mem) wv::g_add64(a.mem + n, -r.mem);
    L.q[2 * n] -= r.kc;
    L.q[2 * n + 1] -= r.km;
    if (r.kc | r.km) r5_rr_update(L, a.n_dc, a.n_dm, w, bit, L.q[2 * n], L.q[2 * n + 1]);
    if (r.flags & RT_PORTS)
        for (u32 p = a.pset_off[r.pset]; p < a.pset_off[r.pset + 1]; ++p) wv::g_or64(a.portmap + (size_t)a.pset_ids[p] * a.n_words + w, bit);
    if (!(r.flags & RT_UNCOUNTED)) {
        wv::g_add32(a.total + n, 1u);
        if (r5_bump_level(L, NB, L.sh[R5S_LB], w, bit, r5_level_of(L, NB, w, bit))) L.sh[R5S_REBUILD] = 1;
        if (e == LIST_EMPTY) {
            wv::g_or64(a.X + (size_t)r.svc * a.xs + w, bit);
            a.list_node[r.slot] = n;
            a.list_svc[r.slot] = 1;
            a.list_fail[r.slot] = 0;
        } else {
            wv::g_store32_fresh(a.list_svc + e, wv::g_fresh32(a.list_svc + e) + 1);
        }
    }
    a.log_node[ci] = n;
    a.log_task[ci] = gj;
    a.log_prev[ci] = (int32_t)wv::g_exch32(reinterpret_cast<u32*>(a.last + n), ci);
    a.out_node[gj] = (int32_t)n;
    L.sh[R5S_NCOMMIT] = ci + 1;
    wv::wait_vm();   // the listers read X / the lists through L2 right after the next barrier
}

WV_DEV void r5_generic(const ResolveArgs& a, const R5Lds& L, u32 jj, u32& par) {
    const u32 tid = wv::tid();
    const u32 gj = a.j0 + jj;
    const R5Rt r = r5_load_rt(a.rt + gj);
    const u32 NB = L.sh[R5S_NB];
    const bool mine = tid < a.n_words;
    u64 f = 0;
    if (mine) {
        // the task's feasible set against the state as it is now: static class & demand-class rows & ~used host ports
        f = a.sc[(size_t)r.sc * a.n_words + tid];
        if (r.flags & RT_RES)
            f &= L.rr[(size_t)((r.flags >> RT_DC_SHIFT) & RT_DCLS_MASK) * L.rs + tid] & L.rr[(size_t)(a.n_dc + ((r.flags >> RT_DM_SHIFT) & RT_DCLS_MASK)) * L.rs + tid];
        if (r.flags & RT_PORTS)
            for (u32 p = a.pset_off[r.pset]; p < a.pset_off[r.pset + 1]; ++p) f &= ~wv::g_fresh64(a.portmap + (size_t)a.pset_ids[p] * a.n_words + tid);
    }
    u64 mk = mine ? f & ~wv::g_fresh64(a.X + (size_t)r.svc * a.xs + tid) : 0ull;
    const u32 idx_bits = 14, idx_mask = (1u << idx_bits) - 1u;   // n_nodes <= 16 384, levels < 256
    bool placed = false;
    for (;;) {
        u32 best = 0xFFFFFFFFu;
        if (mk) {
            u64 m = mk;
            u32 l = 0;
            for (int b = (int)NB - 1; b >= 0; --b) {
                const u64 t = m & ~L.planes[(size_t)b * L.rs + tid];
                if (t) m = t;
                else l |= 1u << b;
            }
            best = (l << idx_bits) | (tid * 64 + (u32)wv::ffs64(m));
        }
        const u32 g = r5_block_min32(best, L, par);
        if (g == 0xFFFFFFFFu) break;
        const u32 n = g & idx_mask, w = n >> 6;
        const u64 bit = 1ull << (n & 63);
        if (w == tid) {
            bool ok = true;
            if (r.flags & RT_RES) ok = L.q[2 * n] >= r.kc && L.q[2 * n + 1] >= r.km;
            if (ok && (r.flags & RT_PORTS))
                for (u32 p = a.pset_off[r.pset]; p < a.pset_off[r.pset + 1]; ++p)
                    if (wv::g_fresh64(a.portmap + (size_t)a.pset_ids[p] * a.n_words + w) & bit) ok = false;
            if (ok) r5_commit_one(a, L, r, gj, n, LIST_EMPTY);
            else {
                mk &= ~bit;
                L.sh[R5S_RETRIES] += 1;
            }
            L.sh[R5S_OK] = ok ? 1u : 0u;
        }
        wv::barrier();
        if (L.sh[R5S_OK]) { placed = true; break; }
        wv::barrier();   // everyone has read the flag before the next winner rewrites it
    }
    if (!placed) {
        // the service's exception list: nodes where it already runs or that failed it ≥ 5 times recently
        const u32 e0 = wv::uload(a.list_off + r.svc), e1 = wv::uload(a.list_off + r.svc + 1);
        u64 bhi = KEY_NONE, blo = KEY_NONE;
        u32 be = 0;
        for (u32 e = e0 + tid; e < e1; e += R5_THREADS) {
            const u32 n = wv::g_fresh32(a.list_node + e);
            if (n == LIST_EMPTY) continue;
            const u32 w = n >> 6;
            const u64 bit = 1ull << (n & 63);
            if (!(a.sc[(size_t)r.sc * a.n_words + w] & bit)) continue;
            if ((r.flags & RT_RES) && !(L.q[2 * n] >= r.kc && L.q[2 * n + 1] >= r.km)) continue;
            if (r.flags & RT_PORTS) {
                bool used = false;
                for (u32 p = a.pset_off[r.pset]; p < a.pset_off[r.pset + 1]; ++p)
                    if (wv::g_fresh64(a.portmap + (size_t)a.pset_ids[p] * a.n_words + w) & bit) used = true;
                if (used) continue;
            }
            const u32 sv = wv::g_fresh32(a.list_svc + e), fl = wv::g_fresh32(a.list_fail + e);
            if ((r.flags & RT_MAXREP) && !((u64)sv < r.maxrep)) continue;   // filter.go:373-375
            const u32 fcl = fl >= MAX_FAILURES ? fl - (MAX_FAILURES - 1) : 0u;
            const u32 tot = L.sh[R5S_BASE] + r5_level_of(L, NB, w, bit);
            const u64 hi = ((u64)fcl << 32) | sv, lo = ((u64)tot << 32) | n;
            if (hi < bhi || (hi == bhi && lo < blo)) { bhi = hi; blo = lo; be = e; }
        }
        const u64 ghi = r5_block_min64(bhi, L, par);
        if (ghi != KEY_NONE) {
            const u64 glo = r5_block_min64(bhi == ghi ? blo : KEY_NONE, L, par);
            if (bhi == ghi && blo == glo) {
                r5_commit_one(a, L, r, gj, (u32)glo, be);
                L.sh[R5S_SLOW] += 1;
            }
            placed = true;
        }
    }
    if (!placed && tid == 0) {
        const u32 ni = L.sh[R5S_NINF];
        a.inf_task[ni] = gj;
        a.inf_pos[ni] = L.sh[R5S_NCOMMIT];
        L.sh[R5S_NINF] = ni + 1;
    }
    if (tid == 0) L.sh[R5S_GENERIC] += 1;
    wv::barrier();
}

// ---- the kernel -----------------------------------------------------------------------------------------------------
template <int K>
WV_KERNEL(R5_THREADS) void k_resolve5(ResolveArgs a) {
    const u32 tid = wv::tid(), lane = wv::lane(), wave = wv::wave();
    if (wv::uload(&a.ctl->error) != ERR_NONE) return;   // an earlier window stopped: the host carries on from ctl->resume
    const R5Lds L = r5_layout(wv::lds(), a.n_words, a.n_dc + a.n_dm);
    u32 par = 0;   // parity of the reduction scratch

    for (u32 i = tid; i < 2 * L.rs; i += R5_THREADS) L.tk[i] = 0;
    for (u32 i = tid; i < a.n_nodes * 2; i += R5_THREADS) L.q[i] = a.qres[i];
    for (u32 i = tid; i < 2 * R5_B * 2; i += R5_THREADS) L.ring[i] = (i & 1) ? 0u : 0xFFFFFFFFu;
    if (tid < R5S_COUNT) L.sh[tid] = 0;
    if (tid < a.n_dc + a.n_dm) L.thr[tid] = a.thr[tid];
    if (wave == R5_CW) r5_stage_records(a, L, 0, min((u32)R5_AHEAD, a.count));
    u32 staged = min((u32)R5_AHEAD, a.count);   // committer wave: records of window-local tasks < staged are in the ring
    wv::barrier();
    if (tid == 0) {
        L.sh[R5S_NCOMMIT] = a.ctl->ncommit;
        L.sh[R5S_NINF] = a.ctl->ninf;
    }
    r5_rr_build(a, L);   // the barriers of r5_build_planes order it before the first lister
    bool fatal = !r5_build_planes(a, L, par);   // ends with a barrier
    // matcher wave: the demand-class thresholds it compares every commit with. Up to 8 + 8 of them live in
    // registers (padding = a residual no node can have: never "below"); more than that are re-read through the scalar cache.
    int32_t thr_c[8], thr_m[8], thr_cmax = -R5_QLIM, thr_mmax = -R5_QLIM;
    const bool small_rr = a.n_dc <= 8 && a.n_dm <= 8;
    for (u32 c = 0; c < 8; ++c) {
        thr_c[c] = (c < a.n_dc && small_rr) ? wv::uload(a.thr + c) : -R5_QLIM;
        thr_m[c] = (c < a.n_dm && small_rr) ? wv::uload(a.thr + a.n_dc + c) : -R5_QLIM;
    }
    if (a.n_dc) thr_cmax = wv::uload(a.thr + a.n_dc - 1);
    if (a.n_dm) thr_mmax = wv::uload(a.thr + a.n_dc + a.n_dm - 1);

    u32 j = 0;           // next window-local task
    u32 buf = 0;         // list buffer of the current round
    u32 bnx = 1, bsp = 2;   // ... of the round being prepared; spare (target of a carry-over)
    u32 tkp = 0;         // TK row that holds the previous round's picks
    u32 rpar = 0;        // round parity (cut flag / hand-over / ring slot)
    bool have_lists = false;
    // matcher state that lives from the match to the LDS commit (wave 0 only)
    u32 m_cls = R5_NONE, m_lvl = 0, m_pick = 0xFFFFFFFFu, m_svc = 0, m_slot = 0, m_kc = 0, m_km = 0;
    // committer state (wave 15 only): the chain link whose exchange is still in flight
    R5Pend pend{0xFFFFFFFFu, -1};
    // section timers (a.dbg & 16): shader cycles spent by wave 0 in match / barrier 1 / LDS commit / barrier 2 and by wave 1
    // in list / barrier 1 / - / barrier 2; cyc[4..7] of the control block, pad1 = between the rounds (cuts, refills, generic)
    const bool prof = (a.dbg & 16u) != 0;
    u64 lcy[5] = {0, 0, 0, 0, 0};
    u64 tc[5] = {0, 0, 0, 0, 0}, tm[3] = {0, 0, 0}, nadv = 0, tmark = prof ? wv::clock64() : 0;
#define R5_TICK(slot) do { if (prof) { const u64 n_ = wv::clock64(); tc[slot] += n_ - tmark; tmark = n_; } } while (0)

    while (!fatal && j < a.count) {
        const u32 nb = min((u32)R5_B, a.count - j);
        // ---------------- phase 1: match round [j, j+nb) || list round [j+nb, ...) || memory side of the last round ----------
        // A (re)fill — no lists yet, or the prefetched ones were dropped by a cut — is the same phase with nothing to match:
        // the lists of [j, j+nb) are built against the state as it is (one call site: the lister is the bulk of the code).
        const bool matching = have_lists;
        if (wave >= 1 && wave <= R5_LW) {
            r5_list<K>(a, L, matching ? j + nb : j, matching ? bnx : buf, wave - 1, (prof && wave == 1) ? lcy : nullptr);
        } else if (wave == R5_CW) {
            r5_commit_memory(a, L, rpar ^ 1, pend, false);
            const u32 want = min(j + (u32)R5_AHEAD, a.count);   // read by the listers from the next round on (barriers in between)
            if (staged < want) {
                r5_stage_records(a, L, staged, want);
                staged = want;
            }
        } else if (!matching) {
            for (u32 i = lane; i < 2 * L.rs; i += 64) L.tk[i] = 0;   // no earlier picks to strike
        } else {
            const u32* li = L.lists + ((size_t)buf * R5_B + lane) * R5_LIST_U32;
            const u32* tkprev = reinterpret_cast<const u32*>(L.tk + (size_t)tkp * L.rs);
            u32* tkcur = reinterpret_cast<u32*>(L.tk + (size_t)(tkp ^ 1) * L.rs);
            u32 cnt = 0;
            u32 eb[2 * R5_Q], ew[2 * R5_Q];   // the list as 32-node half-words (registers): candidate bits, half-word index
            for (int k = 0; k < 2 * R5_Q; ++k) eb[k] = ew[k] = 0;
            m_cls = R5_NONE;
            m_pick = 0xFFFFFFFFu;
            if (lane < nb) {
                m_cls = li[0];
                cnt = li[1];
            }
            const bool fast = m_cls == R5_FAST;
            if (fast) {
                m_lvl = li[2];
                m_svc = li[3];
                m_kc = li[4];
                m_km = li[5];
                m_slot = li[6];
                // half-word k of the list = half (k & 1) of its 64-node entry k / 2, minus the previous round's picks
                for (int k = 0; k < 2 * R5_Q; ++k)
                    if ((u32)k < cnt) {
                        ew[k] = 2 * li[R5_HDR_U32 + 4 * (k >> 1)] + (k & 1);
                        eb[k] = li[R5_HDR_U32 + 4 * (k >> 1) + 2 + (k & 1)] & ~tkprev[ew[k]];
                    }
            }
            // current half-word of every lane: the first one of its list that still has a candidate — and the one after it, which a lane
            // moves on to in a single step when its current half-word runs empty (cleaned of this round's picks only then)
            u32 bits = 0, w = 0, bits2 = 0, w2 = 0;
            for (int k = 2 * R5_Q - 1; k >= 0; --k)
                if (eb[k]) { bits2 = bits; w2 = w; bits = eb[k]; w = ew[k]; }
            const u64 fastmask = wv::ballot(fast), infmask = wv::ballot(m_cls == R5_INFEASIBLE);
            u64 tq = prof ? wv::clock64() : 0;
            if (prof) tm[0] += tq - tmark;
            // the round ends in front of the first task that is neither fast nor a no-op
            const u64 lanes = (1ull << nb) - 1ull;   // nb <= R5_B < 64
            const u64 stop = ~(fastmask | infmask) & lanes;
            u32 cut = stop ? (u32)wv::ffs64(stop) : nb;
            u32 why = stop ? (u32)R5_CUT_GENERIC : (u32)R5_CUT_NOT;
            // the unrolled walk (wv::match_seq64) passes every lane in order: the tasks it serves are the fast ones in front of the cut, the
            // no-ops among them carry a dummy, and the lane at the cut (cut <= R5_B < 64: there always is one) stops it with empty bits
            const bool served = ((fastmask >> lane) & 1ull) && lane < cut;
            if (!served) { bits = lane == cut ? 0u : 1u; bits2 = 0; w = WV_DUMMY_W | lane; }
            u32 pickb = 0, from = 0;
            u32 flushed = 0;   // picks of lanes < flushed are in tkcur
            for (;;) {
                const u64 ta_ = prof ? wv::clock64() : 0;
                const u32 at = wv::match_seq64(bits, w, bits2, w2, pickb, lane, from);
                if (prof) { tm[2] += wv::clock64() - ta_; ++nadv; }
                if (at >= cut) break;
                // Task `at` ran out of its current half-word — and so, usually, did others that sat on it. This round's picks so far go
                // to the TK row (a pick = the lowest bit of the bits the lane had at its turn, in the half-word it still sits on), then
                // every such lane steps to its next half-word, cleaned of them: one gather. Only when that one is empty too does the
                // lane clean all its half-words (the gathers are in flight together) and take the first two that still have a candidate.
                // Straight-line code: a lane's earlier half-words stay empty once they are (picks only accumulate), so recomputing from
                // the registers is idempotent. Served lanes in front of `at` are never touched: their w is where their pick was made.
                if (served && lane >= flushed && lane < at) wv::lds_or32(tkcur + w, pickb & (0u - pickb));
                flushed = at;
                wv::lockstep();   // one wave's LDS operations execute in order: the gathers below see the atomics above without a wait
                {
                    if (served && lane >= at && bits == 0) {
                        u32 t[2 * R5_Q];
                        for (int k = 0; k < 2 * R5_Q; ++k) t[k] = eb[k] & ~tkcur[ew[k]];
                        for (int k = 2 * R5_Q - 1; k >= 0; --k)
                            if (t[k]) { bits2 = bits; w2 = w; bits = t[k]; w = ew[k]; }
                    }
                    if (wv::readlane(bits, at) == 0) {   // list exhausted: the next round starts here with a fresh list
                        cut = at;
                        why = R5_CUT_RELIST;
                        break;
                    }
                }
                from = at;
            }
            if (prof) { const u64 n_ = wv::clock64(); tm[1] += n_ - tq; tq = n_; }
            if (served && lane < cut) m_pick = (w << 5) + (u32)wv::ffs64((u64)pickb);
            if (served && lane >= flushed && lane < cut) wv::lds_or32(tkcur + w, pickb & (0u - pickb));
            if (lane == 0) {
                L.sh[R5S_CUT0 + rpar] = cut;
                L.sh[R5S_WHY0 + rpar] = why;
            }
            // the round's outcome per task, for the wave that writes the hand-over record in phase 2 (this one updates the levels)
            if (lane < R5_B) {
                u32* h = L.hand + ((size_t)rpar * R5_B + lane) * R5_HAND_U32;
                h[0] = m_cls;   // R5_FAST / R5_INFEASIBLE / anything else: turned into R5H_* below
                h[1] = m_pick;
                h[4] = m_slot;
                h[5] = m_svc;
                h[6] = m_kc;
                h[7] = m_km;
            }
        }
        R5_TICK(0);
        wv::barrier();
        R5_TICK(1);
        if (!matching) {   // uniform
            have_lists = true;
            continue;
        }
        // ---------------- phase 2: the round's prefix into the LDS state + hand-over record ----------------
        // Three waves share it: the matcher moves the picked nodes one level up (planes, ring masks); waves 1 and 2 — idle, like
        // all listers, until the next phase — read the matcher's outcome back from LDS: wave 1 turns it into the hand-over
        // record and numbers the commits, wave 2 charges the reservations to the residuals and the demand-class rows.
        static_assert((int)R5_FAST == (int)R5H_COMMIT && (int)R5_INFEASIBLE == (int)R5H_INF, "the matcher's class codes double as hand-over kinds");
        if (wave == 0) {
            const u32 cut = L.sh[R5S_CUT0 + rpar];
            const u32 NB = L.sh[R5S_NB], lb = L.sh[R5S_LB];
            const bool com = lane < cut && m_cls == R5_FAST;
            bool over = false;
            if (com) over = r5_bump_level(L, NB, lb, m_pick >> 6, 1ull << (m_pick & 63), m_lvl);
            const u64 mc = wv::ballot(com);
            const bool low_pick = wv::ballot(com && m_lvl <= lb) != 0;
            const bool any_over = wv::ballot(over) != 0;
            if (lane == 0) {
                if (any_over) L.sh[R5S_REBUILD] = 1;
                // ring policy: two rounds in a row without a pick at the ring's lowest level → the ring moves up one level
                if (mc) {
                    if (!low_pick) {
                        const u32 qn = L.sh[R5S_QUIET] + 1;
                        L.sh[R5S_QUIET] = qn;
                        if (qn >= 2) L.sh[R5S_ADVANCE] = 1;
                    } else
                        L.sh[R5S_QUIET] = 0;
                }
                if (prof) {   // statistics of the SWP_DBG=16 report
                    L.sh[R5S_ROUNDS] += 1;
                    if (cut == nb) L.sh[R5S_FULL] += 1;
                    else if (L.sh[R5S_WHY0 + rpar] == R5_CUT_GENERIC) L.sh[R5S_CUT_CLASS] += 1;
                    else L.sh[R5S_CUT_EMPTY] += 1;
                }
            }
        } else if (wave == 1) {
            const u32 cut = L.sh[R5S_CUT0 + rpar];
            u32* h = L.hand + ((size_t)rpar * R5_B + (lane < R5_B ? lane : 0u)) * R5_HAND_U32;
            const u32 kind0 = lane < R5_B ? h[0] : (u32)R5_NONE;
            const bool act = lane < cut;
            const bool com = act && kind0 == R5_FAST, inf = act && kind0 == R5_INFEASIBLE;
            const u64 mc = wv::ballot(com), mi = wv::ballot(inf);
            const u32 nc0 = L.sh[R5S_NCOMMIT], ni0 = L.sh[R5S_NINF];
            const u32 ci = nc0 + wv::mbcnt(mc);
            u32 rsvc = 0xFFFFFFFFu, rnode = 0;
            if (com) {
                rsvc = h[5];
                rnode = h[1];
                h[2] = ci;
            } else if (inf) {
                h[2] = ni0 + wv::mbcnt(mi);
                h[3] = ci;   // commits before this task
            } else if (lane < R5_B)
                h[0] = R5H_NONE;
            if (lane < R5_B) {
                L.ring[2 * (rpar * R5_B + lane)] = rsvc;
                L.ring[2 * (rpar * R5_B + lane) + 1] = rnode;
            }
            wv::lockstep();   // every lane has read the counters before lane 0 moves them on
            if (lane == 0) {
                L.sh[R5S_NCOMMIT] = nc0 + (u32)wv::popc64(mc);
                L.sh[R5S_NINF] = ni0 + (u32)wv::popc64(mi);
                L.sh[R5S_HJ0 + rpar] = j;
                L.sh[R5S_HV0 + rpar] = 1;
            }
        } else if (wave == 2) {
            const u32 cut = L.sh[R5S_CUT0 + rpar];
            const u32* h = L.hand + ((size_t)rpar * R5_B + (lane < R5_B ? lane : 0u)) * R5_HAND_U32;
            // (wave 1 rewrites h[0] of the tasks behind the cut only: the ones read here keep the matcher's code)
            const bool com = lane < cut && h[0] == R5_FAST;
            const u32 n = com ? h[1] : 0u;
            int32_t qc = R5_QLIM, qm = R5_QLIM;
            if (com) {
                qc = L.q[2 * n] - (int32_t)h[6];
                qm = L.q[2 * n + 1] - (int32_t)h[7];
                L.q[2 * n] = qc;
                L.q[2 * n + 1] = qm;
            }
            {
                // demand-class rows: a committed node leaves every row whose threshold its residual no longer meets
                // (ascending thresholds within each group: the largest one tells whether any lane has anything to do)
                const u32 ndc = a.n_dc, ndm = a.n_dm;
                u64* const rw = L.rr + (n >> 6);
                const u64 rbit = 1ull << (n & 63);
                if (small_rr) {   // thresholds in registers: no load on this path
                    if (wv::ballot(qc < thr_cmax))
                        for (u32 c = 0; c < 8; ++c)
                            if (c < ndc && qc < thr_c[c]) wv::lds_andn64(rw + (size_t)c * L.rs, rbit);
                    if (wv::ballot(qm < thr_mmax))
                        for (u32 c = 0; c < 8; ++c)
                            if (c < ndm && qm < thr_m[c]) wv::lds_andn64(rw + (size_t)(ndc + c) * L.rs, rbit);
                } else {
                    if (ndc && wv::ballot(qc < thr_cmax))
                        for (u32 c = 0; c < ndc; ++c)
                            if (qc < wv::uload(a.thr + c)) wv::lds_andn64(rw + (size_t)c * L.rs, rbit);
                    if (ndm && wv::ballot(qm < thr_mmax))
                        for (u32 c = 0; c < ndm; ++c)
                            if (qm < wv::uload(a.thr + ndc + c)) wv::lds_andn64(rw + (size_t)(ndc + c) * L.rs, rbit);
                }
            }
        } else if (wave == R5_CW) {
            // the previous round's picks are history — unless this round was cut by an exhausted list: the carry-over below
            // strikes them from the lists it keeps
            const bool relist_cut = L.sh[R5S_CUT0 + rpar] < nb && L.sh[R5S_WHY0 + rpar] == R5_CUT_RELIST;
            if (!relist_cut)
                for (u32 i = lane; i < L.rs; i += 64) L.tk[(size_t)tkp * L.rs + i] = 0;
        }
        R5_TICK(2);
        wv::barrier();
        R5_TICK(3);
        const u32 cut = L.sh[R5S_CUT0 + rpar], why = L.sh[R5S_WHY0 + rpar];
        const bool rebuild = L.sh[R5S_REBUILD] != 0, advance = L.sh[R5S_ADVANCE] != 0;
        j += cut;
        bool flush = false;
        // A round cut by an exhausted list keeps its pipeline: see the carry-over below
        const bool shift = cut < nb && why == R5_CUT_RELIST && !rebuild;
        if ((cut < nb && !shift) || rebuild) {
            // off the pipeline: the memory side of both outstanding rounds lands now (the generic path, the plane rebuild and
            // the fresh lists read it back)
            if (wave == R5_CW) {
                r5_commit_memory(a, L, rpar ^ 1, pend, false);
                r5_commit_memory(a, L, rpar, pend, true);
            }
            wv::barrier();
            flush = true;
        }
        if (rebuild) {   // a node outgrew the planes: rebuild them around the current minimum
            if (tid == 0) L.sh[R5S_REBASES] += 1;
            if (!r5_build_planes(a, L, par)) { fatal = true; break; }
        } else if (advance) {
            if (tid == 0) L.sh[R5S_RINGADV] += 1;
            r5_ring_build(L, L.sh[R5S_LB] + 1);   // lists carry absolute levels: the pipeline goes on
        }
        if (shift) {
            // ---- carry-over. The round was cut at a task whose list ran out. Nothing is flushed and nothing is listed twice: the
            // next round [j, j+nb2) takes the lists of the cut round's unprocessed tasks (this buffer, slots cut..nb-1: built
            // before round r-1, so the picks of r-1 and r are struck from them now) and, behind them, the first lists of the
            // round that was being prepared (the other buffer: built before r, struck by r's picks). A list struck of the picks
            // made since its snapshot IS a list built after them (the list rule), so the TK rows start empty again. Lists that
            // came out empty — the task that cut the round is one — are built afresh by the lister waves, one task each.
            const u32 keep = nb - cut, nb2 = min((u32)R5_B, a.count - j);
            const u32* tkprev = reinterpret_cast<const u32*>(L.tk + (size_t)tkp * L.rs);
            const u32* tkcur = reinterpret_cast<const u32*>(L.tk + (size_t)(tkp ^ 1) * L.rs);
            // one list word per thread, into the spare buffer (no list is read and written in the same pass)
            for (u32 e = tid; e < nb2 * R5_LIST_U32; e += R5_THREADS) {
                const u32 sl = e / R5_LIST_U32, q = e % R5_LIST_U32;
                const bool old = sl < keep;
                const u32* src = L.lists + ((size_t)(old ? buf : bnx) * R5_B + (old ? cut + sl : sl - keep)) * R5_LIST_U32;
                u32 v = src[q];
                if (q >= R5_HDR_U32 && ((q - R5_HDR_U32) & 3u) >= 2u && src[0] == R5_FAST) {   // a half of an entry's candidate word
                    const u32 r = (q - R5_HDR_U32) >> 2, hw = 2 * src[R5_HDR_U32 + 4 * r] + ((q - R5_HDR_U32) & 1u);
                    if (2 * r < src[1]) {
                        v &= ~tkcur[hw];
                        if (old) v &= ~tkprev[hw];
                    } else
                        v = 0;
                }
                L.lists[((size_t)bsp * R5_B + sl) * R5_LIST_U32 + q] = v;
            }
            wv::barrier();
            for (u32 i = tid; i < 2 * L.rs; i += R5_THREADS) L.tk[i] = 0;
            if (wave == 0) {
                bool empty = false;
                if (lane < nb2) {
                    const u32* li = L.lists + ((size_t)bsp * R5_B + lane) * R5_LIST_U32;
                    if (li[0] == R5_FAST) {
                        u32 any = 0;
                        for (u32 r = 0; r < R5_Q; ++r) any |= li[R5_HDR_U32 + 4 * r + 2] | li[R5_HDR_U32 + 4 * r + 3];
                        empty = any == 0;
                    }
                }
                const u64 em = wv::ballot(empty);
                if (empty) L.relist[wv::mbcnt(em)] = lane;
                if (lane == 0) {
                    L.sh[R5S_NREL] = (u32)wv::popc64(em);
                    L.sh[R5S_SHIFTS] += 1;

                }
            }
            wv::barrier();
            {   // the spare buffer is the current one now
                const u32 t_ = buf;
                buf = bsp;
                bsp = t_;
            }
            if (wave >= 1 && wave <= R5_LW) r5_list<K>(a, L, j, buf, wave - 1, nullptr, L.relist, L.sh[R5S_NREL]);
            wv::barrier();
        }
        if (cut < nb && why == R5_CUT_GENERIC) {
            r5_generic(a, L, j, par);   // ends with a barrier
            j += 1;
            if (L.sh[R5S_REBUILD]) {
                if (tid == 0) L.sh[R5S_REBASES] += 1;
                wv::barrier();
                if (!r5_build_planes(a, L, par)) { fatal = true; break; }
            }
        }
        rpar ^= 1;
        if (flush) {
            have_lists = false;   // the prefetched lists are for the wrong tasks (or the wrong base)
            if (wave == 0 && lane < R5_B) {   // everything has landed
                L.ring[2 * lane] = 0xFFFFFFFFu;
                L.ring[2 * (R5_B + lane)] = 0xFFFFFFFFu;
            }
            tkp = 0;
        } else if (shift) {
            tkp = 0;   // the carried lists are in `buf` already; both TK rows are empty
        } else {
            const u32 t_ = buf;   // the next round's lists become the current ones; the old current buffer is listed into next
            buf = bnx;
            bnx = t_;
            tkp ^= 1;
        }
        R5_TICK(4);
    }
#undef R5_TICK
    wv::barrier();
    if (wave == R5_CW) {   // the last rounds' memory side
        r5_commit_memory(a, L, 0, pend, false);
        r5_commit_memory(a, L, 1, pend, true);
    }
    if (prof && lane == 0) a.ctl->wave_cyc[wave] += tc[0] >> 6;
    if (prof && lane == 0 && wave == 1)
        for (int q = 0; q < 5; ++q) a.ctl->l_cyc[q] += lcy[q] >> 6;
    if (prof && tid == 0)
    {
        for (int q = 0; q < 3; ++q) a.ctl->m_cyc[q] += tm[q] >> 6;
        a.ctl->m_cyc[3] += nadv;
    }
    if (prof && lane == 0 && wave < 2) {
        u64* c = a.ctl->cyc + 4;   // 32-bit halves, units of 64 cycles
        c[2 * wave] += (tc[0] >> 6) | ((tc[1] >> 6) << 32);
        c[2 * wave + 1] += (tc[2] >> 6) | ((tc[3] >> 6) << 32);
        if (wave == 0) a.ctl->pad1 += tc[4] >> 6;
    }
    wv::barrier();
    for (u32 i = tid; i < a.n_nodes * 2; i += R5_THREADS) a.qres[i] = L.q[i];
    if (tid == 0) {
        a.ctl->ncommit = L.sh[R5S_NCOMMIT];
        a.ctl->ninf = L.sh[R5S_NINF];
        a.ctl->verify_retries += L.sh[R5S_RETRIES];
        a.ctl->slow_tasks += L.sh[R5S_SLOW];
        a.ctl->rebases += L.sh[R5S_REBASES];
        a.ctl->generic_tasks += L.sh[R5S_GENERIC];
        a.ctl->cyc[0] += L.sh[R5S_ROUNDS];
        a.ctl->cyc[1] += L.sh[R5S_FULL];
        a.ctl->cyc[2] += L.sh[R5S_CUT_CLASS];
        a.ctl->cyc[3] += L.sh[R5S_CUT_EMPTY];
        a.ctl->spin_waits += L.sh[R5S_DEEP];
        if (fatal) {
            a.ctl->error = ERR_LEVEL_RANGE;
            a.ctl->resume = a.j0 + j;
        }
    }
}

}  // namespace swpdev
