// swp_resolve4.hpp — k_resolve4: the sequential argmin + commit pass with G REPLICATED resolver wavefronts that pick
// speculatively for the G tasks of a round. EXPERIMENTAL, opt-in (SWP_RESOLVER=4); the default resolver is k_resolve3.
// Status at the end of round 1 (one GPU run, the round's last): bit-exact against the oracle on the nine parity cases
// of tests/test_engine_resolvers.py (SWP_TEST_R4=1), but SLOWER than k_resolve3 on the headline workload (77.9 ms vs
// 40.8 ms per 100k x 10k batch): 60 % of that workload's tasks are h+1 picks, which today's rounds do not accept, so a
// round commits 0.4 tasks on average (tools/sim_rounds.py; R4_OPT bit 16 below lets them in). docs/NOTES_r01.md §5 also
// lists what the ISA shows: dependent LDS round trips before the pick, an abort-flag read in every poll iteration, SGPR
// spill traffic in the round loop. Included from swp_device.hpp after k_resolve3.
//
// Why: a lone wave issues one instruction per ≈ 4 ns, k_resolve3 needs ≈ 85 per task; the only way past that is to work
// on several tasks at once although task i+1 must see task i's placement. Here every resolver wave holds the SAME
// state (level planes, hot masks, touched set; the fast-commit set D and the commit ring live in per-wave LDS rows) and
// round r handles tasks j .. j+G-1:
//   1. wave w picks for task j+w against the state after task j-1, exactly as k_resolve3 does (staged mk row, LA0 & ~D,
//      slot-specialised), but keeps the first w+1 set bits of its candidate WORD; the task is SIMPLE iff it is not
//      forced generic, has a hot-level candidate and none of the kept candidates is touched;
//   2. one 2x64-bit record per wave goes through LDS (word, then meta+round number: LDS executes one wave's
//      instructions in order, so a reader that sees the round number also sees the word); `s_barrier` cannot be used,
//      it is workgroup-wide and the loader / committer waves run asynchronously;
//   3. every wave resolves the round with the same scalar code: task v takes the lowest kept candidate that no earlier
//      task of the round took in the same word; the round ends at the first task that is not simple or has none left.
//      Exact: a hot-level candidate sits at the minimum level of the whole cluster, so no node raised inside the round
//      can beat it, and a node taken inside the round leaves LA (and the service's mk) in the sequential order too;
//   4. every wave applies the round to its replica: one ds_or on its D row, one ds_write on its ring, scalars.
// If task j itself is not simple, ALL resolver waves run k_resolve3's full iteration for it redundantly (hot-level pick
// with the touched test, h+1 pick, hot-level advance); only the memory-dependent part of the generic path (freshness
// re-checks, exception list) runs on wave 0 alone, which broadcasts {placed, node, level, via list, list entry}. Memory
// is therefore written by wave 0 (and the committer) only, and read by the others only behind wave 0's flush flag.
// Helper waves as in k_resolve3: waves G, G+1 stage rows one block ahead, wave G+2 applies the side effects of finished
// blocks — it reads them straight from wave 0's commit ring (64 entries ≥ the 3 blocks that can be outstanding).
#pragma once

#define ERR_PROTOCOL 3   // k_resolve4: an LDS hand-shake between the resolver replicas timed out

// Tuning candidates for the next round, each behind one bit of R4_OPT (make -C swarmkit_amd/csrc EXTRA=-DR4_OPT=n; default 0 = the code
// that passed the parity cases on the GPU). NONE of them has run on hardware yet:
//   1  issue the D-row reads before they are needed (today: three dependent LDS round trips inside the pick)
//   2  {flags, svc} of the whole block in registers (lane t = task t), read once per block: the pick no longer starts
//      with a dependent LDS read of the round's task records
//   4  test the abort flag every 256 polls instead of in every poll iteration
//   8  publish which kept candidates are touched instead of giving the round up when ANY of them is: a round then ends
//      only if the candidate a task actually takes is touched (what the sequential order does)
//  16  tasks WITHOUT a hot-level candidate take part in the round with their h+1 candidates (k_resolve3's "B" pick).
//      tools/sim_rounds.py: on the headline workload 60 % of the tasks are B picks, 30 % A picks — the cluster sits on two
//      levels at once — so today's rounds commit 0.4 tasks on average (hence 77.9 ms), with B picks 2.4. Exact because a
//      node taken at level h inside the round was a hot-level node of the snapshot, so it is in no mask that had no
//      hot-level candidate; B picks of the round only compete with each other (same "skip the taken bits" rule), and a
//      B candidate that is in D (raised earlier in the window) is touched and ends the round as before.
//  32  a task with no feasible node at all (k_resolve3's quick exit: mk == 0, no exception-list hint, no commit of its
//      service in the ring) passes through the round as a no-op instead of ending it: on the headline workload 9.6 % of
//      the tasks are such (two zone values that no node carries), and with bits 16 + 32 the model commits 3.8 of 4 tasks
//      per round (6.6 of 8 at G = 8).
#ifndef R4_OPT
#define R4_OPT 0
#endif
#if R4_OPT & 8
#define R4_POS(mh) (((mh) >> 4) & 0x3FFu)   // bits 16.. of the meta word carry the touched flags
#else
#define R4_POS(mh) ((mh) >> 4)
#endif

__device__ __forceinline__ u32 lds_addr(const void* p) { return (u32)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
// two LDS accesses by the same lanes, guaranteed to execute in this order (one wave's DS instructions are in order)
__device__ __forceinline__ void lds_write_pair_ordered(u32 a0, u64 v0, u32 a1, u64 v1) {
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %2, %3" ::"v"(a0), "v"(v0), "v"(a1), "v"(v1) : "memory");
}
__device__ __forceinline__ void lds_read_pair_ordered(u32 a0, u32 a1, u64& x, u64& y) {
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ u32 rfl32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 rl32(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u64 rl64(u64 v, u32 l) { return ((u64)rl32((u32)(v >> 32), l) << 32) | rl32((u32)v, l); }

template <int K, int G>
__global__ __launch_bounds__(64 * (G + 3)) void k_resolve4(ResolveArgs a) {
    static_assert(G >= 2 && G <= 4 && K <= 8, "record layout: k in 4 bits, at most 4 kept candidates");
    extern __shared__ unsigned char r4_lds[];
    const u32 Wn = a.n_words, XS = a.xs, TB = a.tb;
    constexpr u32 RS = K * 64;   // staged row stride in words
    constexpr u32 NT = 64 * (G + 3);
    int32_t* last_lds = reinterpret_cast<int32_t*>(r4_lds);                                    // [n_nodes]
    const size_t off_f = (((size_t)a.n_nodes * 4 + 15) / 16) * 16;
    u64* MK = reinterpret_cast<u64*>(r4_lds + off_f);                                          // [2*TB][RS]  F & ~X
    u64* below_lds = MK + (size_t)(2 * TB) * RS;                                               // [RS] published BELOW
    u64* xfix_lds = below_lds + RS;                                                            // [G][RS] scratch, zero between uses
    u64* drow_lds = xfix_lds + (size_t)G * RS;                                                 // [G][RS] D of every replica
    R2Rec* Tb = reinterpret_cast<R2Rec*>(drow_lds + (size_t)G * RS);                           // [2*TB]
    uint4* ring_lds = reinterpret_cast<uint4*>(Tb + 2 * TB);                                   // [G][64] {svc, node, meta, list entry}
    u64* recW = reinterpret_cast<u64*>(ring_lds + G * 64);                                     // [2][G] kept candidate bits
    u64* recM = recW + 2 * G;                                                                  // [2][G] meta<<32 | round
    u32* delta_lds = reinterpret_cast<u32*>(recM + 2 * G);                                     // [8] generic outcome of wave 0
    u32* flags_lds = delta_lds + 8;                                                            // [32]
    // flags: [0..1] ready[buf] (+1 per loader wave and staged block), [2] done, [3] abort, [4] BELOW epoch, [5+2*buf+loader]
    // loader epochs, [9] issued, [10] completed, [11+buf] commits of the block, [13+buf] first commit index of the block,
    // [15] flushes done by wave 0, [16] generic outcomes published, [17+w] blocks finished by replica w
    const u32 tid = threadIdx.x, lane = tid & 63, wave = rfl32(tid >> 6);
    const u32 nblk = (a.count + TB - 1) / TB;
    if (a.ctl->error != ERR_NONE) return;
    if (tid < 32) flags_lds[tid] = 0;
    if (tid < 8) delta_lds[tid] = 0;
    if (tid < 4 * G) recW[tid] = 0;   // recW and recM are contiguous
    for (u32 n = tid; n < a.n_nodes; n += NT) last_lds[n] = a.last[n];
    for (u32 i = tid; i < 2 * G * RS; i += NT) xfix_lds[i] = 0;   // xfix and D rows are contiguous
    for (u32 i = tid; i < G * 64; i += NT) ring_lds[i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
    __syncthreads();

    // Side effects of one commit (as k_resolve3): residual update of the node row (NodeInfo.addTask, nodeinfo.go:108-154),
    // exception bitmap + list entry, commit log + per-node chain, placement.
    auto apply_commit = [&](u32 n, u32 meta, u32 list_entry, u32 ce, u32 blk) __attribute__((always_inline)) {
        const u32 tt = meta & 0xFFu;
        const R2Rec r = Tb[(blk & 1u) * TB + tt];
        const u32 gj = a.j0 + blk * TB + tt;
        if (r.cpu) atomicAdd(reinterpret_cast<u64*>(a.cpu + n), (u64)(-r.cpu));
        if (r.mem) atomicAdd(reinterpret_cast<u64*>(a.mem + n), (u64)(-r.mem));
        if (meta & 0x100u) {   // counted
            atomicAdd(a.total + n, 1u);
            if (meta & 0x200u) atomicAdd(a.list_svc + list_entry, 1u);   // placed through the exception list
            else {
                atomicOr(&a.X[(size_t)r.svc * XS + (n >> 6)], 1ull << (n & 63));
                a.list_node[r.slot] = n;
                a.list_svc[r.slot] = 1;
                a.list_fail[r.slot] = 0;
            }
        }
        a.log_node[ce] = n;
        a.log_task[ce] = gj;
        a.log_prev[ce] = (int32_t)atomicExch(reinterpret_cast<u32*>(&last_lds[n]), ce);
        a.out_node[gj] = (int32_t)n;
    };

    if (wave == G + 2) {
        // =============================== COMMITTER ===============================
        for (u32 b = 0; b < nblk; ++b) {
            u32 spins = 0;
            while (__hip_atomic_load(&flags_lds[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b + 1) {
                if (__hip_atomic_load(&flags_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 26)) return;   // bounded: never hang the GPU
            }
            const u32 cnt = flags_lds[11 + (b & 1u)], first = flags_lds[13 + (b & 1u)];
            if (lane < cnt) {
                const uint4 e = ring_lds[(first + lane) & 63u];   // wave 0's ring
                apply_commit(e.y, e.z, e.w, first + lane, b);
            }
            __hip_atomic_store(&flags_lds[9], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&flags_lds[10], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }

    if (wave >= G) {
        // =============================== LOADERS (two waves, half a block each; as k_resolve3) ===============================
        const u32 lw = wave - G;
        const u32 half = (TB + 1) / 2;
        constexpr int LB = K <= 4 ? 8 : 4;
        for (u32 b = 0; b < nblk; ++b) {
            const u32 buf = b & 1;
            if (b >= 2) {
                u32 spins = 0;
                while (__hip_atomic_load(&flags_lds[9], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b - 1) {
                    if (__hip_atomic_load(&flags_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 26)) return;
                }
            }
            const u32 t0 = b * TB, nt = min(TB, a.count - t0);
            const u32 tb0 = min(nt, lw * half), tb1 = min(nt, (lw + 1) * half);
            R2Rec rec;
            const bool hasrec = tb0 + lane < tb1;
            if (hasrec) {
                const RTask* r = a.rt + a.j0 + t0 + tb0 + lane;
                rec.cpu = r->cpu;
                rec.mem = r->mem;
                rec.flags = r->flags;
                rec.svc = r->svc;
                rec.slot = r->slot;
                rec.pset = r->pset;
            }
            u64 BL[K];
            u32 ep = 0;
            {
                u32 spins = 0;
                for (;;) {
                    ep = __hip_atomic_load(&flags_lds[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!(ep & 1u)) {
#pragma unroll
                        for (int k = 0; k < K; ++k) BL[k] = __hip_atomic_load(&below_lds[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        if (__hip_atomic_load(&flags_lds[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == ep) break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 24)) return;
                }
            }
            u32 extra = 0;
            for (u32 t = tb0; t < tb1; t += LB) {
                u64 f[LB][K], x[LB][K];
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    const bool ht = t + q < tb1;
                    const u32 svc = ht ? cload(&a.rt[a.j0 + t0 + t + q].svc) : 0u;
                    const u64* fs = a.F + (size_t)(t0 + t + q) * Wn;
                    const u64* xs = a.X + (size_t)svc * XS;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const u32 i = lane + 64 * k;
                        const bool ok = ht && i < Wn;
                        f[q][k] = ok ? fs[i] : 0ull;
                        x[q][k] = ok ? __hip_atomic_load(&xs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    }
                }
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    if (t + q < tb1) {
                        u64* dst = MK + ((size_t)buf * TB + t + q) * RS + lane;
                        u64 sbv = 0, fxv = 0;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const u64 m = f[q][k] & ~x[q][k];
                            dst[64 * k] = m;
                            sbv |= m & BL[k];
                            fxv |= f[q][k] & x[q][k];
                        }
                        const u32 bits = (ballot64(sbv != 0) ? 0x80000000u : 0u) | (ballot64(fxv != 0) ? 0x40000000u : 0u);
                        extra = (tb0 + lane == t + q) ? bits : extra;
                    }
                }
            }
            if (hasrec) {
                rec.flags |= extra | ((rec.flags & (RT_PORTS | RT_UNCOUNTED)) ? 0xA0000000u : 0u);
                Tb[buf * TB + tb0 + lane] = rec;
            }
            if (lane == 0) flags_lds[5 + 2 * buf + lw] = ep;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(&flags_lds[buf], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // ONE increment per wave
        }
        return;
    }

    // =============================== RESOLVER REPLICAS (waves 0 .. G-1; wave 0 leads) ===============================
    const u32 w = wave;
    const bool leader = w == 0;
    u64* my_drow = drow_lds + (size_t)w * RS;
    u64* my_xfix = xfix_lds + (size_t)w * RS;
    uint4* my_ring = ring_lds + w * 64;
    u32 ncommit = rfl32(a.ctl->ncommit), ninf = rfl32(a.ctl->ninf);
    u32 applied = ncommit;   // commits [applied, ncommit) still live only in the ring
    u32 st_retries = 0, st_slow = 0, st_rebase = 0, st_generic = 0, st_spins = 0, st_rounds = 0, st_round_tasks = 0;
    const u32 idx_bits = 32 - __clz((Wn * 64) | 1u);
    const u32 idx_mask = (1u << idx_bits) - 1u;
    u32 NB = 1, base = 0, h = 0;
    u32 la_count = 0, epoch = 0, flush_seq = 0, delta_seq = 0;
    bool fatal = false, soft_stop = false;
    // replicated exact state = (planes, LA0, LB0, T0) as of the last fold + the D row in LDS:
    //   level(n) = planes(n) + [n in D];  LA = LA0 & ~D;  LB = LB0 ^ D;  touched = T0 | D
    u64 pl[R1_NBR][K];
    u64 T0[K], BELOW[K], LA0[K], LB0[K], VAL[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        T0[k] = 0;
        VAL[k] = (lane + 64 * k) < Wn ? a.valid[lane + 64 * k] : 0ull;
    }
    auto aborted = [&]() __attribute__((always_inline)) -> bool {
        return rfl32(__hip_atomic_load(&flags_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0;
    };
    // spin until *flag >= want (acquire); false on abort / timeout
    auto wait_ge = [&](u32* flag, u32 want) __attribute__((always_inline)) -> bool {
        u32 spins = 0;
        while (rfl32(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24) || aborted()) return false;
        }
        st_spins += spins;
        return true;
    };
    // D row → registers, row zeroed (one wave: LDS executes in issue order, an earlier ds_or is included)
    auto take_D = [&](u64 (&d)[K]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < K; ++k) d[k] = __hip_atomic_load(&my_drow[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < K; ++k) __hip_atomic_store(&my_drow[lane + 64 * k], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto load_word_totals = [&](int k, u32 (&v)[64]) __attribute__((always_inline)) {
        const u32 wi = lane + 64 * k;
        const u32* src = a.total + (size_t)(wi < Wn ? wi : 0) * 64;
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // (re)build the level planes from total[] — every replica reads the same memory (callers flush first)
    auto build_planes = [&]() __attribute__((always_inline)) -> bool {
        u32 lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u32 v[64];
            load_word_totals(k, v);
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const bool on = (VAL[k] >> i) & 1ull;
                lo = min(lo, on ? v[i] : 0xFFFFFFFFu);
                hi = max(hi, on ? v[i] : 0u);
            }
        }
        lo = rfl32(wave_min_u32(lo));
        hi = rfl32(wave_max_u32(hi));
        if (lo == 0xFFFFFFFFu) { lo = 0; hi = 0; }
        const u32 need = 32 - __clz((hi - lo) | 1u);
        const u32 cap = min((u32)R1_NBR, 32u - idx_bits);
        if (need > cap) return false;
        base = lo;
        NB = min(cap, need + 1);
        u64 d[K];
        take_D(d);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            T0[k] |= d[k];
            u32 v[64];
            load_word_totals(k, v);
            u64 p[R1_NBR];
#pragma unroll
            for (int b = 0; b < R1_NBR; ++b) p[b] = 0;
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const u32 lvl = ((VAL[k] >> i) & 1ull) ? v[i] - base : 0u;
#pragma unroll
                for (int b = 0; b < R1_NBR; ++b) p[b] |= (u64)((lvl >> b) & 1u) << i;
            }
#pragma unroll
            for (int b = 0; b < R1_NBR; ++b) pl[b][k] = p[b];
        }
        return true;
    };
    auto fold = [&]() __attribute__((always_inline)) {
        u64 d[K];
        take_D(d);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u64 c = d[k];
            LA0[k] &= ~c;
            LB0[k] ^= c;
            T0[k] |= c;
#pragma unroll
            for (int b = 0; b < R1_NBR; ++b) {
                const u64 t = pl[b][k] & c;
                pl[b][k] ^= c;
                c = t;
            }
        }
    };
    auto derive_masks = [&](u32 hh) __attribute__((always_inline)) {
        h = hh;
        const bool ok = hh + 2u <= (1u << NB) - 1u;
        u32 cnt = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u64 lt = 0, eq = VAL[k], eq1 = VAL[k];
            const u32 h1 = hh + 1;
#pragma unroll
            for (int b = R1_NBR - 1; b >= 0; --b) {
                const u64 p = pl[b][k];
                if (hh >> b & 1u) { lt |= eq & ~p; eq &= p; } else { eq &= ~p; }
                if (h1 >> b & 1u) eq1 &= p; else eq1 &= ~p;
            }
            BELOW[k] = ok ? lt : VAL[k];
            LA0[k] = ok ? eq : 0ull;
            LB0[k] = ok ? eq1 : 0ull;
            cnt += (u32)__popcll(LA0[k]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += (u32)__shfl_xor((int)cnt, off, 64);
        la_count = rfl32(cnt);
        if (leader) {   // publish BELOW for the loaders (sequence lock); the replicas only keep the epoch in step
            __hip_atomic_store(&flags_lds[4], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
            for (int k = 0; k < K; ++k) __hip_atomic_store(&below_lds[lane + 64 * k], BELOW[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        epoch += 2u;
        if (leader) __hip_atomic_store(&flags_lds[4], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto search = [&](const u64 (&mk)[K]) __attribute__((always_inline)) -> u32 {
        u64 m[K];
        u32 lv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { m[k] = mk[k]; lv[k] = 0; }
#pragma unroll
        for (int b = R1_NBR - 1; b >= 0; --b) {
            if ((u32)b < NB) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    u64 t = m[k] & ~pl[b][k];
                    bool nz = t != 0;
                    m[k] = nz ? t : m[k];
                    lv[k] |= nz ? 0u : (1u << b);
                }
            }
        }
        u32 best = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u32 wi = lane + 64 * k;
            u32 cand = (lv[k] << idx_bits) | (wi * 64 + (u32)(__ffsll((long long)m[k]) - 1));
            best = min(best, m[k] ? cand : 0xFFFFFFFFu);
        }
        return best;
    };
    u32 tin = 0, bdone = 0;   // task index inside the block, blocks finished
    // Memory must reflect every commit so far: wave 0 applies the pending commits of the current block, waits for the
    // committer's finished blocks and drains its own stores; the other replicas wait for its flag. Collective: every
    // replica calls it at the same points.
    auto flush = [&]() __attribute__((always_inline)) {
        ++flush_seq;
        if (leader) {
            const u32 pend = ncommit - applied;
            if (lane < pend) {
                const uint4 e = my_ring[(applied + lane) & 63u];
                apply_commit(e.y, e.z, e.w, applied + lane, bdone);
            }
            if (!wait_ge(&flags_lds[10], bdone)) fatal = true;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&flags_lds[15], flush_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (!wait_ge(&flags_lds[15], flush_seq)) fatal = true;
        applied = ncommit;
    };

    if (!build_planes()) {
        if (leader && lane == 0) {
            a.ctl->error = ERR_LEVEL_RANGE;
            a.ctl->resume = a.j0;   // nothing of this window was touched
            __hip_atomic_store(&flags_lds[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    derive_masks(0);

    auto wait_block = [&](u32 bi) __attribute__((always_inline)) {
        if (!wait_ge(&flags_lds[bi & 1], 2u * ((bi >> 1) + 1u))) fatal = true;
    };
    auto block_epoch = [&](u32 bf) __attribute__((always_inline)) -> u32 {
        const u32 e0 = rfl32(flags_lds[5 + 2 * bf]);
        const u32 e1 = rfl32(flags_lds[6 + 2 * bf]);
        return e0 == e1 ? e0 : 0xFFFFFFFFu;   // odd: never equals the (even) current epoch
    };
    // staged mk row of a slot with the X-freshness repair from this replica's commit ring (as k_resolve3)
    auto load_row = [&](u32 slot, u32 rsvc, u64 (&mk)[K], bool& ring_hit) __attribute__((always_inline)) {
        const u64* row = MK + (size_t)slot * RS + lane;
#pragma unroll
        for (int k = 0; k < K; ++k) mk[k] = row[64 * k];
        const uint2 rg = *reinterpret_cast<const uint2*>(&my_ring[lane]);   // {svc, node} of the commit ≡ lane (mod 64)
        const bool hit = rg.x == rsvc;
        ring_hit = ballot64(hit) != 0;
        if (__builtin_expect(ring_hit, 0)) {
            u64* cell = my_xfix + (rg.y >> 6);
            if (hit) __hip_atomic_fetch_or(cell, 1ull << (rg.y & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
            for (int k = 0; k < K; ++k) mk[k] &= ~__hip_atomic_load(&my_xfix[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (hit) __hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    u32 blk_ep = 0, round = 0, j = 0;
    wait_block(0);
    if (!fatal) blk_ep = block_epoch(0);
#if R4_OPT & 2
    // lane t holds {flags, svc} of task t of the current block (TB <= 16 tasks)
    auto load_block_records = [&](u32 bf, u32 ntasks) __attribute__((always_inline)) -> uint2 {
        return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(&Tb[bf * TB + min(lane, ntasks - 1u)]) + 16);
    };
    uint2 blk_rec = make_uint2(0u, 0u);
    if (!fatal) blk_rec = load_block_records(0u, min(TB, a.count));
#endif

    while (j < a.count && !fatal) {
        const u32 base_slot = (bdone & 1u) * TB;
        const u32 nt = min(TB, a.count - bdone * TB);   // tasks of this block
        const u32 g = min((u32)G, nt - tin);            // tasks of this round (a round never crosses a block)
        ++round;
#if R4_OPT & 2
        // lane v gets {flags, svc} of task v of the round from the block's register copy (no LDS access on the way to the pick)
        const uint2 crv = make_uint2((u32)__builtin_amdgcn_ds_bpermute((int)((tin + min(lane, g - 1u)) << 2), (int)blk_rec.x),
                                     (u32)__builtin_amdgcn_ds_bpermute((int)((tin + min(lane, g - 1u)) << 2), (int)blk_rec.y));
#else
        // lane v holds {flags, svc} of task v of the round
        const uint2 crv = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(&Tb[base_slot + tin + min(lane, g - 1u)]) + 16);
#endif

        // ---------------- 1. this replica's speculative pick ----------------
        u32 meta_hi = 0;
        u64 keep = 0;
        if (w < g) {
#if R4_OPT & 2
            const u32 flagw = rl32(blk_rec.x, tin + w), rsvc = rl32(blk_rec.y, tin + w);
#else
            const u32 flagw = rl32(crv.x, w), rsvc = rl32(crv.y, w);
#endif
            u64 mk[K];
            bool ring_hit;
#if R4_OPT & 1
            u64 dk_[K];   // issued together with the row reads
#pragma unroll
            for (int k = 0; k < K; ++k) dk_[k] = __hip_atomic_load(&my_drow[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
            load_row(base_slot + tin + w, rsvc, mk, ring_hit);
            bool generic;
            if (__builtin_expect(blk_ep == epoch, 1)) generic = (int)flagw < 0;   // staged: forced, or a candidate below h
            else {
                u64 sb = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) sb = bitop3_u64<BITOP_AB_OR_C>(mk[k], BELOW[k], sb);
                generic = ballot64(((u32)sb | (u32)(sb >> 32)) != 0) != 0 || (flagw & 0x20000000u) != 0;
            }
            meta_hi = 1u;   // active
#if R4_OPT & 32
            bool anym_ = false;
#pragma unroll
            for (int k = 0; k < K; ++k) anym_ = anym_ || (mk[k] != 0);
            const bool none_ = ballot64(anym_) == 0 && (flagw & 0x40000000u) == 0 && !ring_hit;   // k_resolve3's quick exit
            if (none_) meta_hi = 3u | 8u;   // bit 3: nothing to place, nothing changes
            else
#endif
            if (!generic) {
                u64 ca[K], ba[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
#if R4_OPT & 1
                    const u64 dk = dk_[k];
#else
                    const u64 dk = __hip_atomic_load(&my_drow[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                    ca[k] = bitop3_u64<BITOP_A_AND_B_ANDN_C>(mk[k], LA0[k], dk);
                    ba[k] = ballot64(ca[k] != 0);
                }
                // first slot with a hot-level candidate; keep the lowest w+1 candidates of its first word
#if R4_OPT & 8
#define R4_PICK(kk)                                                                                    \
    {                                                                                                  \
        const u32 l_ = (u32)__builtin_ctzll(ba[kk]);                                                   \
        const u64 word_ = rl64(ca[kk], l_), tw_ = rl64(T0[kk], l_);                                    \
        u64 kp_ = 0, rem_ = word_;                                                                     \
        u32 tf_ = 0; /* bit c: the c-th kept candidate was committed to in this window (F may be stale) */ \
        for (u32 c_ = 0; c_ <= w; ++c_) {                                                              \
            const u64 low_ = rem_ & (0ull - rem_);                                                     \
            kp_ |= low_;                                                                               \
            tf_ |= ((low_ & tw_) != 0 ? 1u : 0u) << c_;                                                \
            rem_ ^= low_;                                                                              \
        }                                                                                              \
        keep = kp_;                                                                                    \
        meta_hi = 3u | ((u32)(kk) << 4) | (l_ << 8) | (tf_ << 16);                                     \
    }
#else
#define R4_PICK(kk)                                                                                    \
    {                                                                                                  \
        const u32 l_ = (u32)__builtin_ctzll(ba[kk]);                                                   \
        const u64 word_ = rl64(ca[kk], l_), tw_ = rl64(T0[kk], l_);                                    \
        u64 kp_ = 0, rem_ = word_;                                                                     \
        for (u32 c_ = 0; c_ <= w; ++c_) {                                                              \
            const u64 low_ = rem_ & (0ull - rem_);                                                     \
            kp_ |= low_;                                                                               \
            rem_ ^= low_;                                                                              \
        }                                                                                              \
        if ((kp_ & tw_) == 0) { /* a kept candidate committed to in this window: F may be stale */     \
            keep = kp_;                                                                                \
            meta_hi = 3u | ((u32)(kk) << 4) | (l_ << 8);                                               \
        }                                                                                              \
    }
#endif
                if (ba[0] != 0) { R4_PICK(0) }
                else if constexpr (K > 1) {
                    if (ba[1] != 0) { R4_PICK(1) }
                    else if constexpr (K > 2) {
                        if (ba[2] != 0) { R4_PICK(2) }
                        else if constexpr (K > 3) {
                            if (ba[3] != 0) { R4_PICK(3) }
                            else if constexpr (K > 4) {
                                if (ba[4] != 0) { R4_PICK(4) }
                                else if constexpr (K > 5) {
                                    if (ba[5] != 0) { R4_PICK(5) }
                                    else if constexpr (K > 6) {
                                        if (ba[6] != 0) { R4_PICK(6) }
                                        else if constexpr (K > 7) {
                                            if (ba[7] != 0) { R4_PICK(7) }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
#undef R4_PICK
#if R4_OPT & 16
                {
                    u64 anya = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) anya |= ba[k];
                    if (anya == 0) {   // no hot-level candidate at all: the lowest w+1 candidates at h+1 (LB = LB0 ^ D; D nodes are touched)
                        u64 cb[K], bb[K];
                        bool found_ = false;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
#if R4_OPT & 1
                            const u64 dkb = dk_[k];
#else
                            const u64 dkb = __hip_atomic_load(&my_drow[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                            cb[k] = bitop3_u64<BITOP_A_AND_BXORC>(mk[k], LB0[k], dkb);
                            bb[k] = ballot64(cb[k] != 0);
                            if (!found_ && bb[k] != 0) {
                                found_ = true;
                                const u32 l_ = (u32)__builtin_ctzll(bb[k]);
                                const u64 word_ = rl64(cb[k], l_), tw_ = rl64(T0[k] | dkb, l_);
                                u64 kp_ = 0, rem_ = word_;
                                u32 tf_ = 0;
                                for (u32 c_ = 0; c_ <= w; ++c_) {
                                    const u64 low_ = rem_ & (0ull - rem_);
                                    kp_ |= low_;
                                    tf_ |= ((low_ & tw_) != 0 ? 1u : 0u) << c_;
                                    rem_ ^= low_;
                                }
                                if ((R4_OPT & 8) != 0) {
                                    keep = kp_;
                                    meta_hi = 3u | 4u | ((u32)k << 4) | (l_ << 8) | (tf_ << 16);   // bit 2: an h+1 pick
                                } else if ((kp_ & tw_) == 0) {
                                    keep = kp_;
                                    meta_hi = 3u | 4u | ((u32)k << 4) | (l_ << 8);
                                }
                            }
                        }
                    }
                }
#endif
            }
        }
        // ---------------- 2. exchange ----------------
        const u32 rbase = (round & 1u) * G;
        if (lane == 0) lds_write_pair_ordered(lds_addr(&recW[rbase + w]), keep, lds_addr(&recM[rbase + w]), ((u64)meta_hi << 32) | round);
        u64 M = 0, Wd = 0;
        {
            const u32 ri = rbase + (lane < (u32)G ? lane : 0u);
            const u32 aM = lds_addr(&recM[ri]), aW = lds_addr(&recW[ri]);
            u32 spins = 0;
            for (;;) {
                lds_read_pair_ordered(aM, aW, M, Wd);   // meta first: a current round number implies a current word
                if (ballot64(lane < (u32)G && (u32)M != round) == 0) break;
#if R4_OPT & 4
                if ((++spins & 255u) == 0 && (spins > (1u << 22) || aborted())) { fatal = true; break; }
#else
                if (++spins > (1u << 22) || aborted()) { fatal = true; break; }
#endif
            }
            st_spins += spins;
        }
        if (fatal) break;
#if R4_OPT & 32
        // ---------------- 3'. resolution (same scalar code on every replica); no-candidate tasks pass through ----------------
        u32 n_round = 0, n_commits = 0, n_none = 0, n_round_a = 0;
        u32 cpos[G], crank[G], nrank[G];
        u64 cbit[G];
        {
            bool ended = false;
#pragma unroll
            for (int v = 0; v < G; ++v) {
                const u32 mh = rl32((u32)(M >> 32), (u32)v);
                const u64 wd = rl64(Wd, (u32)v);
                u64 tk = 0;   // bits of the same word taken by earlier tasks of the round
#pragma unroll
                for (int u = 0; u < v; ++u) tk |= (cpos[u] == R4_POS(mh)) ? cbit[u] : 0ull;
                const u64 avail = wd & ~tk;
                const bool none_v = (mh & 8u) != 0;
#if R4_OPT & 8
                const u64 low = avail & (0ull - avail);
                const u32 rank = (u32)__popcll(wd & (low - 1ull));
                const bool stale = ((mh >> (16u + rank)) & 1u) != 0;
                const bool cand_ok = avail != 0 && !stale;
#else
                const bool cand_ok = avail != 0;
#endif
                const bool ok = !ended && (mh & 3u) == 3u && (none_v || cand_ok);
                cpos[v] = R4_POS(mh);
                cbit[v] = (ok && !none_v) ? (avail & (0ull - avail)) : 0ull;
                crank[v] = n_commits;   // commits / no-ops of the round before task v
                nrank[v] = n_none;
                if (ok) {
                    n_round = (u32)v + 1u;
                    if (none_v) ++n_none;
                    else {
                        ++n_commits;
                        if ((mh & 4u) == 0) ++n_round_a;
                    }
                } else ended = true;
            }
        }
        if (leader) { ++st_rounds; st_round_tasks += n_round; }

        if (__builtin_expect(n_round != 0, 1)) {
            // ---------------- 4'. apply the round to this replica ----------------
            const u32 mypos = R4_POS((u32)(M >> 32));   // slot | lane << 4 of task `lane`'s word
            const u32 mywi = (mypos >> 4) + 64u * (mypos & 15u);
            u64 mybit = 0;
            u32 myrank = 0, mynr = 0;
#pragma unroll
            for (int v = 0; v < G; ++v) {
                mybit = (lane == (u32)v) ? cbit[v] : mybit;
                myrank = (lane == (u32)v) ? crank[v] : myrank;
                mynr = (lane == (u32)v) ? nrank[v] : mynr;
            }
            if (lane < n_round && mybit != 0) {
                __hip_atomic_fetch_or(&my_drow[mywi], mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const u32 node = (mywi << 6) + (u32)__builtin_ctzll(mybit);
                my_ring[(ncommit + myrank) & 63u] = make_uint4(crv.y, node, (tin + lane) | 0x100u, 0u);
            }
            if (leader && lane < n_round && mybit == 0) {   // "no suitable node": remembered for the explain pass with its moment
                a.inf_task[ninf + mynr] = a.j0 + j + lane;
                a.inf_pos[ninf + mynr] = ncommit + myrank;
            }
            ncommit += n_commits;
            ninf += n_none;
#if R4_OPT & 16
            la_count -= n_round_a;
            if (n_commits != n_round_a && la_count == 0 && h + 3u <= (1u << NB) - 1u) {   // h+1 picks and level h is exhausted
                fold();
                derive_masks(h + 1);
            }
#else
            la_count -= n_commits;
#endif
            tin += n_round;
            j += n_round;
#else
        // ---------------- 3. resolution: the same scalar code on every replica ----------------
        u32 n_round = 0;
#if R4_OPT & 16
        u32 n_round_a = 0;   // hot-level picks among them
#endif
        u32 cpos[G];
        u64 cbit[G];
        {
            bool ended = false;
#pragma unroll
            for (int v = 0; v < G; ++v) {
                const u32 mh = rl32((u32)(M >> 32), (u32)v);
                const u64 wd = rl64(Wd, (u32)v);
                u64 tk = 0;   // bits of the same word taken by earlier tasks of the round
#pragma unroll
                for (int u = 0; u < v; ++u) tk |= (cpos[u] == R4_POS(mh)) ? cbit[u] : 0ull;
                const u64 avail = wd & ~tk;
#if R4_OPT & 8
                const u64 low = avail & (0ull - avail);
                const u32 rank = (u32)__popcll(wd & (low - 1ull));            // how many kept candidates sit below the one taken
                const bool stale = ((mh >> (16u + rank)) & 1u) != 0;          // … and is that one touched (its F bit may be stale)
                const bool ok = !ended && (mh & 3u) == 3u && avail != 0 && !stale;
#else
                const bool ok = !ended && (mh & 3u) == 3u && avail != 0;
#endif
                cpos[v] = R4_POS(mh);
                cbit[v] = ok ? (avail & (0ull - avail)) : 0ull;
                if (ok) n_round = (u32)v + 1u;
                else ended = true;
#if R4_OPT & 16
                if (ok && (mh & 4u) == 0) ++n_round_a;
#endif
            }
        }
        if (leader) { ++st_rounds; st_round_tasks += n_round; }

        if (__builtin_expect(n_round != 0, 1)) {
            // ---------------- 4. apply the round to this replica ----------------
#if R4_OPT & 8
            const u32 mypos = R4_POS((u32)(M >> 32));   // slot | lane << 4 of task `lane`'s word
#else
            const u32 mypos = (u32)(M >> 36);   // slot | lane << 4 of task `lane`'s word
#endif
            const u32 mywi = (mypos >> 4) + 64u * (mypos & 15u);
            u64 mybit = 0;
#pragma unroll
            for (int v = 0; v < G; ++v) mybit = (lane == (u32)v) ? cbit[v] : mybit;
            if (lane < n_round) {
                __hip_atomic_fetch_or(&my_drow[mywi], mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const u32 node = (mywi << 6) + (u32)__builtin_ctzll(mybit);
                my_ring[(ncommit + lane) & 63u] = make_uint4(crv.y, node, (tin + lane) | 0x100u, 0u);
            }
            ncommit += n_round;
#if R4_OPT & 16
            la_count -= n_round_a;
            // picks came from h+1: once level h is exhausted for everybody, advance the hot level (as k_resolve3 does on a B pick)
            if (n_round_a != n_round && la_count == 0 && h + 3u <= (1u << NB) - 1u) {
                fold();
                derive_masks(h + 1);
            }
#else
            la_count -= n_round;
#endif
            tin += n_round;
            j += n_round;
#endif
        } else {
            // ---------------- task j is not simple: k_resolve3's full iteration, redundantly on every replica ----------------
            const u32 slot0 = base_slot + tin;
#if R4_OPT & 2
            const u32 flag0 = rl32(blk_rec.x, tin), rsvc = rl32(blk_rec.y, tin);
#else
            const u32 flag0 = rl32(crv.x, 0u), rsvc = rl32(crv.y, 0u);
#endif
            u64 mk[K], d[K];
            bool ring_hit;
            load_row(slot0, rsvc, mk, ring_hit);
#pragma unroll
            for (int k = 0; k < K; ++k) d[k] = __hip_atomic_load(&my_drow[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bool generic;
            if (blk_ep == epoch) generic = (int)flag0 < 0;
            else {
                u64 sb = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) sb = bitop3_u64<BITOP_AB_OR_C>(mk[k], BELOW[k], sb);
                generic = ballot64(((u32)sb | (u32)(sb >> 32)) != 0) != 0 || (flag0 & 0x20000000u) != 0;
            }
            bool placed = false, recorded = false;
            u32 n = 0;
#define R4_TAKE(kk, ISB, BAL, CW)                                                                                   \
    {                                                                                                               \
        const u32 l_ = (u32)__builtin_ctzll(BAL);                                                                   \
        const u64 word_ = rl64((CW), l_), tw_ = rl64(T0[kk] | d[kk], l_);                                           \
        const u32 bpos_ = (u32)__builtin_ctzll(word_);                                                              \
        const u64 bit_ = 1ull << bpos_;                                                                             \
        if ((tw_ & bit_) != 0) generic = true; /* committed to in this window: F may be stale */                    \
        else {                                                                                                      \
            n = ((l_ + 64u * (kk)) << 6) + bpos_;                                                                   \
            if (lane == l_) __hip_atomic_fetch_or(&my_drow[l_ + 64u * (kk)], bit_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            if (!(ISB)) --la_count;                                                                                 \
            placed = true;                                                                                          \
        }                                                                                                           \
    }
            if (!generic) {
                u64 ca[K], ba[K];
                u64 anya = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    ca[k] = bitop3_u64<BITOP_A_AND_B_ANDN_C>(mk[k], LA0[k], d[k]);
                    ba[k] = ballot64(ca[k] != 0);
                    anya |= ba[k];
                }
                if (anya != 0) {
                    bool done_ = false;
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        if (!done_ && ba[k] != 0) { R4_TAKE(k, false, ba[k], ca[k]) done_ = true; }
                } else {
                    u64 cb[K], bb[K];
                    u64 anyb = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        cb[k] = bitop3_u64<BITOP_A_AND_BXORC>(mk[k], LB0[k], d[k]);
                        bb[k] = ballot64(cb[k] != 0);
                        anyb |= bb[k];
                    }
                    if (anyb == 0) generic = true;
                    else {
                        bool done_ = false;
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            if (!done_ && bb[k] != 0) { R4_TAKE(k, true, bb[k], cb[k]) done_ = true; }
                        // picks come from h+1: once level h is exhausted for everybody, advance the hot level
                        if (placed && la_count == 0 && h + 3u <= (1u << NB) - 1u) {
                            fold();
                            derive_masks(h + 1);
                        }
                    }
                }
            }
#undef R4_TAKE
            if (!placed && generic) {
                // ---------------- generic path: exact planes; the memory-dependent decision is wave 0's ----------------
                if (leader) ++st_generic;
                bool anym = false;
#pragma unroll
                for (int k = 0; k < K; ++k) anym = anym || (mk[k] != 0);
                const bool listp = (flag0 & 0x40000000u) != 0 || ring_hit;
                if (ballot64(anym) != 0 || listp) {
                    const R2Rec rec = Tb[slot0];
                    const u32 rflags = rfl32(rec.flags);
                    const i64 rcpu = rec.cpu, rmem = rec.mem;
                    const u32 rpset = rec.pset;
                    const u32 gj = a.j0 + j;
                    fold();
                    flush();   // collective: memory now reflects every commit so far
                    ++delta_seq;
                    if (leader) {
                        u32 lvl = 0, entry = 0;
                        bool via_list = false, got = false;
                        u64 gk[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) gk[k] = mk[k];
                        for (;;) {
                            // every lane proposes the best candidate of its own words; lanes whose proposal sits at the winning
                            // level re-check it against memory in the same round trip (k_resolve3's vectorised verify)
                            const u32 mine = search(gk);
                            const u32 gmin = wave_min_u32_dpp(mine);
                            if (gmin == 0xFFFFFFFFu) break;
                            const u32 glvl = gmin >> idx_bits;
                            const bool act = mine != 0xFFFFFFFFu && (mine >> idx_bits) == glvl;
                            const u32 mn = mine & idx_mask, mw = mn >> 6, mko = mw >> 6;
                            const u64 mbit = 1ull << (mn & 63);
                            bool mine_ok = true;
                            if (act) {
                                u64 tsel = 0;
#pragma unroll
                                for (int k = 0; k < K; ++k) tsel = ((u32)k == mko) ? T0[k] : tsel;
                                if (tsel & mbit) {
                                    if (rflags & RT_RES) {
                                        i64 c = __hip_atomic_load(&a.cpu[mn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        i64 m = __hip_atomic_load(&a.mem[mn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        mine_ok = (rcpu <= c) && (rmem <= m);
                                    }
                                    if (mine_ok && (rflags & RT_PORTS)) {
                                        for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p)
                                            if (__hip_atomic_load(&a.portmap[(size_t)a.pset_ids[p] * Wn + mw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mbit) mine_ok = false;
                                    }
                                }
                                if (!mine_ok) {
#pragma unroll
                                    for (int k = 0; k < K; ++k) gk[k] &= ~(((u32)k == mko) ? mbit : 0ull);
                                }
                            }
                            st_retries += (u32)__popcll(ballot64(act && !mine_ok));
                            if (ballot64(act && mine == gmin && mine_ok) != 0) {
                                n = gmin & idx_mask;
                                lvl = glvl;
                                got = true;
                                break;
                            }
                        }
                        if (!got && listp) {
                            // exception list of the service: nodes with svcCount>0 or ≥5 recent failures
                            const u32 e0 = a.list_off[rsvc], e1 = a.list_off[rsvc + 1];
                            const u64 maxrep = a.rt[gj].maxrep;
                            u64 bhi = KEY_NONE, blo = KEY_NONE;
                            u32 be = 0;
                            for (u32 e = e0 + lane; e < e1; e += 64) {
                                u32 nn = __hip_atomic_load(&a.list_node[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (nn == LIST_EMPTY) continue;
                                u32 ww = nn >> 6;
                                u64 bb2 = 1ull << (nn & 63);
                                if (!(a.F[(size_t)j * Wn + ww] & bb2)) continue;
                                if (rflags & RT_RES) {
                                    i64 c = __hip_atomic_load(&a.cpu[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    i64 m = __hip_atomic_load(&a.mem[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    if (!(rcpu <= c && rmem <= m)) continue;
                                }
                                if (rflags & RT_PORTS) {
                                    bool used = false;
                                    for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p)
                                        if (__hip_atomic_load(&a.portmap[(size_t)a.pset_ids[p] * Wn + ww], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bb2) used = true;
                                    if (used) continue;
                                }
                                u32 svn = __hip_atomic_load(&a.list_svc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                u32 fl = __hip_atomic_load(&a.list_fail[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if ((rflags & RT_MAXREP) && !((u64)svn < maxrep)) continue;   // filter.go:373-375
                                u32 fcl = fl >= MAX_FAILURES ? fl - (MAX_FAILURES - 1) : 0u;    // nodeLess, scheduler.go:708-735
                                u32 tot = __hip_atomic_load(&a.total[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                u64 hi = ((u64)fcl << 32) | svn, lo = ((u64)tot << 32) | nn;
                                if (hi < bhi || (hi == bhi && lo < blo)) { bhi = hi; blo = lo; be = e; }
                            }
                            u64 ghi = wave_min_u64(bhi);
                            if (__builtin_amdgcn_readfirstlane((int)(ghi != KEY_NONE))) {
                                u64 glo = wave_min_u64(bhi == ghi ? blo : KEY_NONE);
                                u64 who = ballot64(bhi == ghi && blo == glo);
                                entry = rl32(be, (u32)(__ffsll((long long)who) - 1));
                                n = rfl32((u32)glo);
                                lvl = rfl32((u32)(glo >> 32)) - base;
                                got = true;
                                via_list = true;
                                ++st_slow;
                            }
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) {
                            delta_lds[0] = (got ? 1u : 0u) | (via_list ? 2u : 0u);
                            delta_lds[1] = n;
                            delta_lds[2] = lvl;
                            delta_lds[3] = entry;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __hip_atomic_store(&flags_lds[16], delta_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (!wait_ge(&flags_lds[16], delta_seq)) fatal = true;
                    // every replica applies the broadcast outcome
                    const u32 dflags = rfl32(__hip_atomic_load(&delta_lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    n = rfl32(__hip_atomic_load(&delta_lds[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const u32 lvl = rfl32(__hip_atomic_load(&delta_lds[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const u32 entry = rfl32(__hip_atomic_load(&delta_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const bool via_list = (dflags & 2u) != 0;
                    placed = (dflags & 1u) != 0 && !fatal;
                    if (placed) {
                        const u32 wi = n >> 6, ko = wi >> 6;
                        const u64 bit = 1ull << (n & 63);
                        const bool owner = (wi & 63) == lane;
                        const bool counted = !(rflags & RT_UNCOUNTED);
                        bool want_rebase = false;
                        u64 xk[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            xk[k] = (owner && (u32)k == ko) ? bit : 0ull;
                            T0[k] |= xk[k];
                        }
                        if (counted) {
                            if (lvl >= (1u << NB) - 1u) want_rebase = true;
                            else {
                                const u32 flip = lvl ^ (lvl + 1);
#pragma unroll
                                for (int b = 0; b < R1_NBR; ++b) {
                                    if (flip >> b & 1u) {
#pragma unroll
                                        for (int k = 0; k < K; ++k) pl[b][k] ^= xk[k];
                                    }
                                }
                            }
                        }
                        if (leader && (rflags & RT_PORTS)) {
                            if (owner)
                                for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p) atomicOr(&a.portmap[(size_t)a.pset_ids[p] * Wn + wi], bit);
                        }
                        if (lane == 0)
                            my_ring[ncommit & 63u] = make_uint4((counted && !via_list) ? rsvc : 0xFFFFFFFFu, n, tin | (counted ? 0x100u : 0u) | (via_list ? 0x200u : 0u), entry);
                        ++ncommit;
                        recorded = true;
                        if (want_rebase) {
                            // the commit is applied to memory first (total[n] + 1), then every replica rebuilds its planes
                            if (leader) ++st_rebase;
                            flush();
                            if (!build_planes()) {
                                // the spread outgrew the register planes: stop after this (committed) task; the host
                                // continues from `resume` with the 16-plane workgroup resolver
                                if (leader && lane == 0) { a.ctl->error = ERR_LEVEL_RANGE; a.ctl->resume = a.j0 + j + 1; }
                                fatal = true;
                                soft_stop = true;
                            } else derive_masks(0);
                        } else if (counted) {
                            derive_masks((!via_list && lvl + 2 < (1u << NB)) ? lvl : h);
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no VMEM result pending into the common path
                }
            }
            if (placed && !recorded) {
                if (lane == 0) my_ring[ncommit & 63u] = make_uint4(rsvc, n, tin | 0x100u, 0u);
                ++ncommit;
            } else if (!placed) {
                if (leader && lane == 0) {
                    a.inf_task[ninf] = a.j0 + j;
                    a.inf_pos[ninf] = ncommit;
                }
                ++ninf;
            }
            ++tin;
            ++j;
        }

        if (__builtin_expect(tin == nt, 0)) {
            // ---------------- block end: wave 0 hands the block's commits (its ring) to the committer ----------------
            if (leader) {
                bool ok = true;
#pragma unroll
                for (int v = 1; v < G; ++v) ok = ok && wait_ge(&flags_lds[17 + v], bdone + 1u);   // every replica has left the block's rows
                if (bdone >= 2) ok = ok && wait_ge(&flags_lds[9], bdone - 1u);                  // ring entries / records of block bdone-2 consumed
                if (!ok && !soft_stop) fatal = true;
                if (lane == 0) {
                    flags_lds[11 + (bdone & 1u)] = ncommit - applied;
                    flags_lds[13 + (bdone & 1u)] = applied;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&flags_lds[2], bdone + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (lane == 0) __hip_atomic_store(&flags_lds[17 + w], bdone + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            applied = ncommit;
            ++bdone;
            tin = 0;
            if (j < a.count && !fatal) {
                wait_block(bdone);
                if (!fatal) blk_ep = block_epoch(bdone & 1u);
#if R4_OPT & 2
                if (!fatal) blk_rec = load_block_records(bdone & 1u, min(TB, a.count - bdone * TB));
#endif
            }
        }
    }
    // pending commits of a partial block + the committer's finished blocks (collective; after a soft stop the helper
    // waves are still alive: release them only afterwards)
    const bool timed_out = fatal && !soft_stop;
    if (!timed_out) { fatal = false; flush(); }
    if (!leader) {
        if (timed_out || fatal) __hip_atomic_store(&flags_lds[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // let the others stop spinning
        return;
    }
    if (timed_out || fatal || soft_stop) __hip_atomic_store(&flags_lds[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (u32 n2 = lane; n2 < a.n_nodes; n2 += 64) a.last[n2] = last_lds[n2];
    if (lane == 0) {
        if (timed_out || (fatal && !soft_stop)) a.ctl->error = ERR_PROTOCOL;   // fail loudly on the host
        a.ctl->ncommit = ncommit;
        a.ctl->ninf = ninf;
        a.ctl->verify_retries += st_retries;
        a.ctl->slow_tasks += st_slow;
        a.ctl->rebases += st_rebase;
        a.ctl->generic_tasks += st_generic;
        a.ctl->spin_waits += st_spins;
        a.ctl->cyc[0] += st_rounds;        // diagnostics (SWP_DBG=16 prints them): rounds, tasks committed inside rounds
        a.ctl->cyc[1] += st_round_tasks;
    }
}
