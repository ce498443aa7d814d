// swp_resolve5.hpp — k_resolve5, the ROUND resolver: the sequential argmin + commit pass of the tick
// (nodeSet.tree with a heap of one, nodeset.go:50-124; nodeLess, scheduler.go:708-735; NodeInfo.addTask,
// nodeinfo.go:108-154) decided R5_B tasks at a time by ONE workgroup of 16 wavefronts.
//
// Why rounds. A lone wavefront retires one instruction per ≈ 4 ns; k_resolve3 needs ≈ 85 per task and is the whole batch
// time (0.38 µs per task). The work per task that is really sequential is tiny — "take the first candidate nobody before
// you took" — while finding the candidates (157 words of F & ~X against the level planes at 10k nodes) is wide and
// independent of the other tasks of the round as long as every node is taken at most once per round. So:
//
//   lister waves 1..15  build, for each task of the NEXT round, a candidate list: the first R5_Q non-empty words of
//                       F & ~X restricted to the task's minimum level, in node order, each word validated against the
//                       exact residuals (kept in LDS in the batch's resource units). One wave per task, 4 tasks per
//                       wave and round; everything is word-parallel over the lanes {lane + 64k}.
//   wave 0 (matcher)    walks the CURRENT round's tasks in order on the scalar unit: task i takes the lowest candidate
//                       bit of its current word and the bit is struck from every lane that sits on the same word
//                       (one v_cmp + two v_cndmask per task) — 12 instructions per task instead of 85. Then commits the
//                       round lane-parallel (planes, residuals, X, exception list, commit log: fire-and-forget).
//
// Exactness (the list rule). Inside a batch a node's level only grows and feasibility only shrinks. A task's list holds
// ALL its feasible plain nodes of its minimum level L in node order up to the last listed word; a node taken since the
// list's snapshot (by the previous round, whose picks are in the TK rows, or by an earlier task of this round) moves to
// level L+1 > L, so "first listed bit nobody took" is exactly the sequential argmin(level, index) — as long as the list
// is not exhausted. An exhausted list, a task that must look at its service's exception list (no plain candidate, but
// F & X ≠ 0), host ports, uncounted tasks: the round is CUT there, the tasks before it are committed, and that one task
// runs the generic workgroup path (r5_generic: k_resolve's algorithm on this kernel's state). A task with F == 0 and no
// exception candidate ("no suitable node") passes through the round as a no-op.
//
// Pipeline: while the matcher works on round r, the listers build round r+1 against the state after round r-1; the
// picks of round r are removed from those lists when the matcher loads them (TK row of the previous round).
//
// Written against swp_wave.hpp only (no raw intrinsics) so that tests/emu can run the same source on CPU fibers.
#pragma once
#include "swp_types.hpp"

namespace swpdev {

#define R5_Q 4                    // candidate words per task
#define R5_LW 15                  // lister waves (waves 1..15)
#define R5_TPW 4                  // task slots per lister wave and round
#define R5_B (R5_LW * R5_TPW)     // tasks per round
#define R5_NBMAX 8                // level planes (255 levels above the lowest valid node)
#define R5_THREADS 1024
#define R5_KMAX 4                 // node words per lane: n_words <= 256 (16 384 nodes)
#define R5_QLIM (1 << 30)         // residuals in resource units must stay below this (host checks)

enum { R5_NONE = 0, R5_FAST = 1, R5_INFEASIBLE = 2, R5_COMPLEX = 3 };
enum {   // u32 scalars in LDS
    R5S_NCOMMIT = 0, R5S_NINF, R5S_BASE, R5S_NB, R5S_HOT, R5S_REBUILD, R5S_ERR, R5S_OK, R5S_ENTRY, R5S_PLACED,
    R5S_CUT0, R5S_CUT1,           // cut position of the round, by round parity
    R5S_RETRIES, R5S_SLOW, R5S_GENERIC, R5S_REBASES, R5S_ROUNDS, R5S_FULL, R5S_CUT_CLASS, R5S_CUT_EMPTY,
    R5S_COUNT = 32
};
#define R5_LIST_U32 ((1 + R5_Q) * 4)   // header + entries, 16 B each

struct R5Lds {
    u64* planes;     // [R5_NBMAX][rs]
    u64* tk;         // [2][rs]      picks of the previous / the current round
    u64* scratch;    // [R5_LW][rs]  per lister wave: same-service commits of the last round as a row
    u64* red;        // [64]         block reductions
    u32* lists;      // [2][R5_B][R5_LIST_U32]
    u32* ring;       // [R5_B][2]    (service, node) of the last round's commits
    u32* sh;         // [R5S_COUNT]
    int32_t* q;      // [n_nodes][2] residual cpu / mem in resource units
    u32 rs;          // row stride in words
};

inline __host__ __device__ u32 r5_row_stride(u32 n_words) { return (n_words + 7u) & ~7u; }
inline __host__ __device__ size_t r5_lds_bytes(u32 n_nodes, u32 n_words) {
    const size_t rs = r5_row_stride(n_words);
    return (size_t)(R5_NBMAX + 2 + R5_LW) * rs * 8 + 64 * 8 + (size_t)2 * R5_B * R5_LIST_U32 * 4 + (size_t)R5_B * 2 * 4 + R5S_COUNT * 4 +
           (size_t)n_nodes * 8;
}
WV_DEV R5Lds r5_layout(u64* lds, u32 n_nodes, u32 n_words) {
    R5Lds L;
    L.rs = r5_row_stride(n_words);
    L.planes = lds;
    L.tk = L.planes + (size_t)R5_NBMAX * L.rs;
    L.scratch = L.tk + 2 * L.rs;
    L.red = L.scratch + (size_t)R5_LW * L.rs;
    L.lists = reinterpret_cast<u32*>(L.red + 64);
    L.ring = L.lists + 2 * R5_B * R5_LIST_U32;
    L.sh = L.ring + R5_B * 2;
    L.q = reinterpret_cast<int32_t*>(L.sh + R5S_COUNT);
    (void)n_nodes;
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

// ---- level planes from total[] (window start, level overflow). One wave per node word: a ballot IS a plane word. ----
// Returns false (uniformly) when the level span of the valid nodes does not fit R5_NBMAX planes.
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
        L.sh[R5S_HOT] = 0;
        L.sh[R5S_REBUILD] = 0;
    }
    wv::barrier();
    return true;
}

// level of one node above the base, read back from the planes
WV_DEV u32 r5_level_of(const R5Lds& L, u32 NB, u32 w, u64 bit) {
    u32 lvl = 0;
    for (u32 b = 0; b < NB; ++b)
        if (L.planes[(size_t)b * L.rs + w] & bit) lvl |= 1u << b;
    return lvl;
}

// ---- lister: candidate lists of the round that starts at window-local task jbase, into list buffer `buf` ------------
template <int K>
WV_DEV void r5_list(const ResolveArgs& a, const R5Lds& L, u32 jbase, u32 buf, u32 lw) {
    const u32 lane = wv::lane();
    const u32 NB = L.sh[R5S_NB], hot = L.sh[R5S_HOT];
    // hot-level masks of this wave's words from the planes: LA level == hot, LB level == hot + 1, LO level < hot
    u64 LA[K], LB[K], LO[K];
    {
        const u32 hb = hot + 1;
        const bool hb_ok = (hb >> NB) == 0;
        for (int k = 0; k < K; ++k) {
            const u32 w = lane + 64 * k;
            u64 ea = ~0ull, eb = hb_ok ? ~0ull : 0ull, lt = 0, eq = ~0ull;
            if (w < L.rs) {
                for (int b = (int)NB - 1; b >= 0; --b) {
                    const u64 p = L.planes[(size_t)b * L.rs + w];
                    ea &= ((hot >> b) & 1u) ? p : ~p;
                    eb &= ((hb >> b) & 1u) ? p : ~p;
                    if ((hot >> b) & 1u) { lt |= eq & ~p; eq &= p; } else eq &= ~p;
                }
            } else { ea = 0; eb = 0; }
            LA[k] = ea;
            LB[k] = eb;
            LO[k] = lt;
        }
    }
    const u32 ring_svc = lane < R5_B ? L.ring[2 * lane] : 0xFFFFFFFFu;
    const u32 ring_node = lane < R5_B ? L.ring[2 * lane + 1] : 0u;
    u64* sr = L.scratch + (size_t)lw * L.rs;

    for (u32 t = 0; t < R5_TPW; ++t) {
        const u32 s = lw + R5_LW * t;   // task slot of the round == lane of the matcher
        const u32 jj = jbase + s;
        u32* out = L.lists + ((size_t)buf * R5_B + s) * R5_LIST_U32;
        if (jj >= a.count) {   // uniform
            if (lane == 0) { out[0] = R5_NONE; out[1] = 0; out[2] = 0; }
            continue;
        }
        const RTask* rt = a.rt + a.j0 + jj;
        const u32 flags = wv::uload(&rt->flags), svc = wv::uload(&rt->svc);
        if (flags & (RT_PORTS | RT_UNCOUNTED)) {
            if (lane == 0) { out[0] = R5_COMPLEX; out[1] = 0; out[2] = 0; }
            continue;
        }
        u64 F[K], X[K];
        for (int k = 0; k < K; ++k) {
            const u32 w = lane + 64 * k;
            const bool in = w < a.n_words;
            F[k] = in ? a.F[(size_t)jj * a.n_words + w] : 0ull;
            X[k] = in ? wv::g_fresh64(a.X + (size_t)svc * a.xs + w) : 0ull;
        }
        // commits of the last round may still be on their way to X in memory: patch them in from the ring
        const u64 match = wv::ballot(ring_svc == svc);
        if (match) {
            if (ring_svc == svc) wv::lds_or64(sr + (ring_node >> 6), 1ull << (ring_node & 63));
            wv::wave_sync();
            for (int k = 0; k < K; ++k) {
                const u32 w = lane + 64 * k;
                if (w < L.rs) X[k] |= sr[w];
            }
            wv::wave_sync();
            if (ring_svc == svc) sr[ring_node >> 6] = 0;
        }
        u64 mk[K];
        u64 any_mk = 0, any_fx = 0, in_lo = 0, in_a = 0, in_b = 0;
        for (int k = 0; k < K; ++k) {
            mk[k] = F[k] & ~X[k];
            any_mk |= mk[k];
            any_fx |= F[k] & X[k];
            in_lo |= mk[k] & LO[k];
            in_a |= mk[k] & LA[k];
            in_b |= mk[k] & LB[k];
        }
        if (!wv::ballot(any_mk != 0)) {
            // no plain candidate: "no suitable node" unless a node of the service's exception list is feasible
            const u32 cls = wv::ballot(any_fx != 0) ? R5_COMPLEX : R5_INFEASIBLE;
            if (lane == 0) { out[0] = cls; out[1] = 0; out[2] = 0; }
            continue;
        }
        u64 c[K];
        u32 lvl;
        const bool below = wv::ballot(in_lo != 0) != 0;
        if (!below && wv::ballot(in_a != 0)) {
            lvl = hot;
            for (int k = 0; k < K; ++k) c[k] = mk[k] & LA[k];
        } else if (!below && wv::ballot(in_b != 0)) {
            lvl = hot + 1;
            for (int k = 0; k < K; ++k) c[k] = mk[k] & LB[k];
        } else {
            // generic: bit-sliced minimum per word, then the wave minimum of the levels
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
        }
        // the first R5_Q non-empty words in node order; entry q is parked in lane q
        u32 my_w = 0, cnt = 0;
        u64 my_bits = 0;
        for (int k = 0; k < K; ++k) {
            u64 bal = wv::ballot(c[k] != 0);
            while (bal && cnt < R5_Q) {   // uniform
                const u32 l = (u32)wv::ffs64(bal);
                bal &= bal - 1;
                const u64 word = wv::readlane64(c[k], l);
                if (lane == cnt) { my_w = l + 64 * k; my_bits = word; }
                ++cnt;
            }
        }
        // ResourceFilter against the exact residuals (filter.go:77-84 in resource units): lane b checks node 64w + b
        if (flags & RT_RES) {
            const int32_t kc = (int32_t)wv::uload(&rt->kc), km = (int32_t)wv::uload(&rt->km);
            for (u32 q = 0; q < R5_Q; ++q) {
                if (q < cnt) {   // uniform
                    const u32 sw = wv::readlane(my_w, q);
                    const u64 sb = wv::readlane64(my_bits, q);
                    const u32 n = sw * 64 + lane;
                    bool ok = (sb >> lane) & 1;
                    if (ok) ok = L.q[2 * n] >= kc && L.q[2 * n + 1] >= km;
                    const u64 v = wv::ballot(ok);
                    if (lane == q) my_bits = v;
                }
            }
        }
        const bool some = wv::ballot(lane < cnt && my_bits != 0) != 0;
        if (lane < cnt) {
            out[4 + 4 * lane] = my_w;
            *reinterpret_cast<u64*>(out + 4 + 4 * lane + 2) = my_bits;
        }
        if (lane == 0) {
            out[0] = some ? R5_FAST : R5_COMPLEX;   // every listed node is full by now: let the generic path look further
            out[1] = cnt;
            out[2] = lvl;
        }
    }
}

// ---- one task through the generic workgroup path: k_resolve's algorithm on this kernel's state ----------------------
// (plain nodes by bit-sliced minimum with re-check, then the service's exception list; scheduler.go:708-735.)
// All 1024 threads; thread t owns node word t. Ends with a barrier; counters and flags are in L.sh.
struct R5Rt { i64 cpu, mem; u32 flags, svc, slot, pset; u64 maxrep; int32_t kc, km; };
WV_DEV R5Rt r5_load_rt(const RTask* rt) {
    R5Rt r;
    r.cpu = wv::uload(&rt->cpu);
    r.mem = wv::uload(&rt->mem);
    r.flags = wv::uload(&rt->flags);
    r.svc = wv::uload(&rt->svc);
    r.slot = wv::uload(&rt->slot);
    r.pset = wv::uload(&rt->pset);
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
    if (r.flags & RT_PORTS)
        for (u32 p = a.pset_off[r.pset]; p < a.pset_off[r.pset + 1]; ++p) wv::g_or64(a.portmap + (size_t)a.pset_ids[p] * a.n_words + w, bit);
    if (!(r.flags & RT_UNCOUNTED)) {
        wv::g_add32(a.total + n, 1u);
        const u32 rl = r5_level_of(L, NB, w, bit), nl = rl + 1, xm = rl ^ nl;
        for (u32 b = 0; b < NB; ++b)
            if ((xm >> b) & 1u) wv::lds_xor64(L.planes + (size_t)b * L.rs + w, bit);
        if (nl >> NB) L.sh[R5S_REBUILD] = 1;
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
    if (wv::wave() == 0) wv::wait_vm();   // the matcher's fire-and-forget updates of X / lists / portmap have landed
    wv::barrier();
    const u32 gj = a.j0 + jj;
    const R5Rt r = r5_load_rt(a.rt + gj);
    const u32 NB = L.sh[R5S_NB];
    const bool mine = tid < a.n_words;
    const u64 f = mine ? a.F[(size_t)jj * a.n_words + tid] : 0ull;
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
            if (!(a.F[(size_t)jj * a.n_words + w] & bit)) continue;
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
    const R5Lds L = r5_layout(wv::lds(), a.n_nodes, a.n_words);
    u32 par = 0;   // parity of the reduction scratch

    for (u32 i = tid; i < (2 + R5_LW) * L.rs; i += R5_THREADS) L.tk[i] = 0;   // TK rows and scratch rows are contiguous
    for (u32 i = tid; i < a.n_nodes * 2; i += R5_THREADS) L.q[i] = a.qres[i];
    for (u32 i = tid; i < R5_B * 2; i += R5_THREADS) L.ring[i] = (i & 1) ? 0u : 0xFFFFFFFFu;
    if (tid < R5S_COUNT) L.sh[tid] = 0;
    wv::barrier();
    if (tid == 0) {
        L.sh[R5S_NCOMMIT] = a.ctl->ncommit;
        L.sh[R5S_NINF] = a.ctl->ninf;
    }
    bool fatal = !r5_build_planes(a, L, par);   // ends with a barrier

    u32 j = 0;           // next window-local task
    u32 buf = 0;         // list buffer of the current round
    u32 tkp = 0;         // TK row that holds the previous round's picks
    u32 rpar = 0;        // round parity (cut flag slot)
    bool have_lists = false;
    // matcher state that lives across phases (wave 0 only)
    u32 m_cls = R5_NONE, m_lvl = 0, m_pick = 0xFFFFFFFFu;
    u32 pend_ci = 0xFFFFFFFFu;
    int32_t pend_prev = -1;

    while (!fatal && j < a.count) {
        const u32 nb = min((u32)R5_B, a.count - j);
        if (!have_lists) {
            // (re)fill: lists of [j, j+nb) against the state as it is; no earlier picks to strike
            if (wave != 0) r5_list<K>(a, L, j, buf, wave - 1);
            else
                for (u32 i = lane; i < 2 * L.rs; i += 64) L.tk[i] = 0;
            wv::barrier();
            have_lists = true;
        }
        // ---------------- phase 1: match round [j, j+nb) || list round [j+nb, ...) ----------------
        u32 cut = nb;
        if (wave != 0) {
            r5_list<K>(a, L, j + nb, buf ^ 1, wave - 1);
        } else {
            const u32* li = L.lists + ((size_t)buf * R5_B + lane) * R5_LIST_U32;
            const u64* tkprev = L.tk + (size_t)tkp * L.rs;
            u64* tkcur = L.tk + (size_t)(tkp ^ 1) * L.rs;
            u32 cnt = 0, e = 0, w = 0;
            u64 bits = 0;
            m_cls = R5_NONE;
            m_pick = 0xFFFFFFFFu;
            if (lane < nb) {
                m_cls = li[0];
                cnt = li[1];
                m_lvl = li[2];
            }
            const bool fast = m_cls == R5_FAST;
            if (fast) {
                w = li[4];
                bits = *reinterpret_cast<const u64*>(li + 6) & ~tkprev[w];
            }
            u32 flushed = 0;   // picks of lanes < flushed are in tkcur
            // a lane whose current word ran empty moves to its next listed word (minus everything taken since the snapshot)
#define R5_ADVANCE(upto)                                                                                    \
    for (;;) {                                                                                              \
        const u64 need_ = wv::ballot(fast && bits == 0 && e + 1 < cnt);                                    \
        if (!need_) break;                                                                                  \
        if (flushed < (upto)) {                                                                             \
            if (lane >= flushed && lane < (upto) && m_pick != 0xFFFFFFFFu) wv::lds_or64(tkcur + (m_pick >> 6), 1ull << (m_pick & 63)); \
            flushed = (upto);                                                                               \
            wv::wave_sync();                                                                                \
        }                                                                                                   \
        if (fast && bits == 0 && e + 1 < cnt) {                                                             \
            ++e;                                                                                            \
            w = li[4 + 4 * e];                                                                              \
            bits = *reinterpret_cast<const u64*>(li + 6 + 4 * e) & ~tkprev[w] & ~tkcur[w];                 \
        }                                                                                                   \
    }
            R5_ADVANCE(0u)
            for (u32 i = 0; i < nb; ++i) {
                const u32 s_cls = wv::readlane(m_cls, i);
                if (s_cls == R5_INFEASIBLE) continue;
                if (s_cls != R5_FAST) { cut = i; if (lane == 0) L.sh[R5S_CUT_CLASS] += 1; break; }
                const u32 s_w = wv::readlane(w, i);
                const u64 s_bits = wv::readlane64(bits, i);
                if (s_bits == 0) { cut = i; if (lane == 0) L.sh[R5S_CUT_EMPTY] += 1; break; }   // list exhausted: the generic path looks further
                const u32 b = (u32)wv::ffs64(s_bits);
                m_pick = wv::writelane(m_pick, s_w * 64 + b, i);
                if (w == s_w) bits &= ~(1ull << b);
                R5_ADVANCE(i + 1)
            }
#undef R5_ADVANCE
            if (lane >= flushed && lane < cut && m_pick != 0xFFFFFFFFu) wv::lds_or64(tkcur + (m_pick >> 6), 1ull << (m_pick & 63));
            if (lane == 0) L.sh[R5S_CUT0 + rpar] = cut;
        }
        wv::barrier();
        // ---------------- phase 2: commit the round's prefix ----------------
        if (wave == 0) {
            cut = L.sh[R5S_CUT0 + rpar];
            wv::wait_vm();   // the previous round's fire-and-forget updates have landed (the ring only covers one round)
            if (pend_ci != 0xFFFFFFFFu) a.log_prev[pend_ci] = pend_prev;
            pend_ci = 0xFFFFFFFFu;
            const u32 NB = L.sh[R5S_NB];
            const u32 gj = a.j0 + j + lane;
            const bool act = lane < cut;
            const bool com = act && m_cls == R5_FAST, inf = act && m_cls == R5_INFEASIBLE;
            const u64 mc = wv::ballot(com), mi = wv::ballot(inf);
            const u32 nc0 = L.sh[R5S_NCOMMIT], ni0 = L.sh[R5S_NINF];
            const u32 ci = nc0 + wv::mbcnt(mc);
            u32 rsvc = 0xFFFFFFFFu, rnode = 0;
            bool over = false;
            if (com) {
                const RTask* rt = a.rt + gj;
                const i64 rcpu = rt->cpu, rmem = rt->mem;
                const u32 svc = rt->svc, slot = rt->slot;
                const int32_t kc = (int32_t)rt->kc, km = (int32_t)rt->km;
                const u32 n = m_pick, w = n >> 6;
                const u64 bit = 1ull << (n & 63);
                const u32 nl = m_lvl + 1, xm = m_lvl ^ nl;
                for (u32 b = 0; b < NB; ++b)
                    if ((xm >> b) & 1u) wv::lds_xor64(L.planes + (size_t)b * L.rs + w, bit);
                over = (nl >> NB) != 0;
                L.q[2 * n] -= kc;
                L.q[2 * n + 1] -= km;
                rsvc = svc;
                rnode = n;
                if (rcpu) wv::g_add64(a.cpu + n, -rcpu);
                if (rmem) wv::g_add64(a.mem + n, -rmem);
                wv::g_add32(a.total + n, 1u);
                wv::g_or64(a.X + (size_t)svc * a.xs + w, bit);
                a.list_node[slot] = n;
                a.list_svc[slot] = 1;
                a.list_fail[slot] = 0;
                a.log_node[ci] = n;
                a.log_task[ci] = gj;
                pend_prev = (int32_t)wv::g_exch32(reinterpret_cast<u32*>(a.last + n), ci);
                pend_ci = ci;
                a.out_node[gj] = (int32_t)n;
            }
            if (inf) {
                const u32 ii = ni0 + wv::mbcnt(mi);
                a.inf_task[ii] = gj;
                a.inf_pos[ii] = ci;   // commits before this task
            }
            if (lane < R5_B) {
                L.ring[2 * lane] = rsvc;
                L.ring[2 * lane + 1] = rnode;
            }
            const u32 newhot = wv::min_u32(com ? m_lvl : 0xFFFFFFFFu);
            const bool any_over = wv::ballot(over) != 0;
            for (u32 i = lane; i < L.rs; i += 64) L.tk[(size_t)tkp * L.rs + i] = 0;   // the previous round's picks are history
            if (lane == 0) {
                L.sh[R5S_NCOMMIT] = nc0 + (u32)wv::popc64(mc);
                L.sh[R5S_NINF] = ni0 + (u32)wv::popc64(mi);
                if (newhot != 0xFFFFFFFFu) L.sh[R5S_HOT] = newhot;
                if (any_over) L.sh[R5S_REBUILD] = 1;
                L.sh[R5S_ROUNDS] += 1;
                if (cut == nb) L.sh[R5S_FULL] += 1;
            }
        }
        wv::barrier();
        cut = L.sh[R5S_CUT0 + rpar];
        rpar ^= 1;
        j += cut;
        bool flush = false;
        if (L.sh[R5S_REBUILD]) {   // a node outgrew the planes: rebuild them around the current minimum
            if (tid == 0) L.sh[R5S_REBASES] += 1;
            if (wave == 0) wv::wait_vm();
            wv::barrier();
            if (!r5_build_planes(a, L, par)) { fatal = true; break; }
            flush = true;
        }
        if (cut < nb) {
            r5_generic(a, L, j, par);   // ends with a barrier
            j += 1;
            flush = true;
            if (L.sh[R5S_REBUILD]) {
                if (tid == 0) L.sh[R5S_REBASES] += 1;
                wv::barrier();
                if (!r5_build_planes(a, L, par)) { fatal = true; break; }
            }
        }
        if (flush) {
            have_lists = false;   // the prefetched lists are for the wrong tasks (or the wrong base)
            if (wave == 0 && lane < R5_B) L.ring[2 * lane] = 0xFFFFFFFFu;   // everything has landed (wait_vm above)
            tkp = 0;
        } else {
            buf ^= 1;
            tkp ^= 1;
        }
    }
    if (wave == 0 && pend_ci != 0xFFFFFFFFu) a.log_prev[pend_ci] = pend_prev;
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
        if (fatal) {
            a.ctl->error = ERR_LEVEL_RANGE;
            a.ctl->resume = a.j0 + j;
        }
    }
}

}  // namespace swpdev
