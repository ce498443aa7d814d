// swp_scan.hpp — the SCAN resolver: the sequential argmin of the tick taken literally (nodeSet.tree with a heap of one, nodeset.go:50-124:
// the node with the least nodeLess key, scheduler.go:708-735, the lowest index among equals; NodeInfo.addTask, nodeinfo.go:108-154),
// one task after the other, every task by ALL threads of one workgroup over ALL nodes.
//
// What it is for. The block resolver (swp_resolve6.hpp) decides hundreds of tasks a round as long as tasks find PLAIN nodes (the task's
// service does not run there yet, fewer than five recent failures). A task without one must take the best node of its service's
// exception list by the full key, and that order moves with every placement of the service: the block resolver lets such a task be only
// a block's first (one task per ~50 us round). A stretch of such tasks — a service with more tasks than the cluster has nodes: every
// small cluster, BASELINE configs[0]'s 1 000 tasks on 10 nodes once they belong to several services — goes through this kernel instead:
// ~1 us per task on up to SCAN_MAXN nodes, whatever the tasks are. The engine switches to it when the block resolver's rounds decide
// fewer than a handful of tasks each, and back after the stretch (swp_engine.hip, run_blocks).
//
// State: the node rows (cpu, mem, total) live in LDS for the stretch; per (service, node) the key's upper half — failures >= 5 in bits
// 24-31, ActiveTasksCountByService in bits 0-23 — and the service's list entry of the node sit in two dense matrices in global memory
// (built from the lists by k_scan_prep; a node is only ever looked at by the one thread that owns it, so nothing there is shared).
// Written against swp_wave.hpp only (tests/emu/emu_scan.cpp runs it on CPU fibers against the sequential model).
#pragma once
#include "swp_resolve6.hpp"

namespace swpdev {

// A task's turn is a dependent chain — evaluate the thread's nodes, the wave's least key (two DPP reductions: the key carries the node), one
// LDS atomic min per wave, ONE barrier, the owner applies — and a wave issues an instruction every ~5 cycles whatever its neighbours do:
// the chain is shortest with one node per thread (four waves over 1 000 nodes were no faster than sixteen; what cost 3 800 cycles a task
// was the second reduction level behind a barrier and the second barrier).
#define SCAN_THREADS 1024
#define SCAN_MAXN 4096      // nodes: (8 + 8 + 4 + 4) bytes of LDS each
inline __host__ __device__ u32 scan_nq(u32 n_nodes) {   // nodes per thread of the instance for this node count: 1, 2 or 4
    u32 q = 1;
    while (q * SCAN_THREADS < n_nodes) q *= 2;
    return q;
}

struct ScanArgs {
    R6Args a;          // the block resolver's argument record: node rows, static class rows, lists, host ports, generic sets, logs
    u32 j0, j1;        // the stretch: tasks [j0, j1)
    u32 n_svc, n_sc;   // services of the batch; static class rows (k_scanb keeps them in LDS)
    u32* hmat;         // [n_svc][n_nodes] (failures >= 5 ? failures : 0) << 24 | ActiveTasksCountByService
    u32* emat;         // [n_svc][n_nodes] list entry of (service, node), LIST_EMPTY: none
};
#define SCAN_RTQ 32         // task records staged through LDS at a time (their loads are misses: a record is read once)
inline __host__ __device__ size_t scan_lds(u32 n_nodes) { return (size_t)n_nodes * 24 + 16 * 16 + 64 + 2 * SCAN_RTQ * 64; }
// ... plus the two (service, node) matrices when they fit next to it (SCAN_LM instances): a task then reads nothing but its static-class words from global memory
inline __host__ __device__ size_t scan_lds_lm(u32 n_nodes, u32 n_svc) { return ((scan_lds(n_nodes) + 15) & ~(size_t)15) + (size_t)2 * n_svc * n_nodes * 4; }

// The BATCHED instance (k_scanb, round 6): SCAN_B tasks share one barrier. Every thread evaluates its nodes for the next SCAN_B tasks
// against the state as it is NOW; the workgroup's SCAN_B argmins meet in SCAN_B words; behind the one barrier everybody accepts the longest
// prefix of tasks whose picks are pairwise different. That is exact: a placement changes ONE node — its residuals, task counts, the
// service's count there — and only for the worse, so the argmin of task j + 1 taken before task j was placed is its argmin afterwards
// too unless both are the same node (then task j + 1 and everything behind it is evaluated again in the next batch). No task of the
// stretch may reserve generic resources, publish host ports or mount cluster volumes (their inputs live in global memory, and a
// volume's state is not node-local): the launcher checks, and everything a task reads — node rows, the (service, node) matrices, the
// static class rows — is in LDS.
//
// Tasks nobody needs to look at. Inside a batch nothing is ever given back: a (task, node) pair that fails a filter fails it for the rest
// of the tick. So once a task found NO node, every later task with the same descriptor (R6Args.tmpl: the same service, class
// row, reservations, replica limit) finds none either — a saturated cluster answers its whole backlog that way. The stretch is walked in
// WINDOWS of SCAN_W tasks: a thread per task asks a small table of such descriptors (exact ids, one per cell: a cell taken by another
// descriptor just means no short cut), the tasks that are left form the window's queue — a bit mask, every thread holds it — and only they
// are evaluated, SCAN_B at a time. A skipped task changes nothing, so the others' order among themselves is all that matters; what the
// sequence of ALL tasks fixes — a task's place in the list of unplaceable tasks, the number of commits in front of it (Explain's
// moment) — is written at the window's end from the mask of placed tasks: place = unplaced tasks in front, commits = placed ones in front.
#define SCAN_B 4
#define SCAN_W 256          // tasks of a window: their records (16 KB) and descriptor ids in LDS, the next window's in registers
#define SCAN_DT 1024        // cells of the table of descriptors found unplaceable
// Threads of k_scanb. A SIMD issues one wave instruction every four cycles whoever it belongs to, and a wave on its own gets one out every
// ~5: ONE wave per SIMD nearly fills it. Sixteen waves (a node per thread at 1 000 nodes) made every wave repeat the batch's bookkeeping
// — ~450 instructions a wave and batch, four waves to a SIMD: 3.1 us a batch; four waves with four nodes a thread do the same evaluations
// and the bookkeeping once per SIMD.
#ifndef SCANB_THREADS
#define SCANB_THREADS 256
#endif
static_assert(SCANB_THREADS >= SCAN_W && SCANB_THREADS % 64 == 0 && (SCAN_W * 4) % SCANB_THREADS == 0, "a thread per task of a window");
inline __host__ __device__ u32 scanb_nq(u32 n_nodes) {   // nodes per thread of the instance for this node count: a power of two up to SCAN_MAXN / SCANB_THREADS
    u32 q = 1;
    while (q * SCANB_THREADS < n_nodes) q *= 2;
    return q;
}
inline __host__ __device__ u32 scan_dt_cell(u32 tm) { return (tm * 2654435761u) >> 22; }
inline __host__ __device__ size_t scan_lds_b(u32 n_nodes, u32 n_svc, u32 n_sc) {
    return (((size_t)n_nodes * 24 + 15) & ~(size_t)15) + (size_t)2 * n_svc * n_nodes * 4 + (size_t)n_sc * ((n_nodes + 63) / 64) * 8 + 3 * SCAN_B * 8 +
           (SCAN_W / 64) * 8 + 16 + (size_t)SCAN_W * 64 + SCAN_W * 4 + SCAN_DT * 4;
}

#ifdef SWP_SCAN_KERNELS
// hmat / emat from the per-service lists: grid (entries of the longest list / 256, services)
WV_KERNEL(256) void k_scan_fill(ScanArgs s) {
    const size_t n = (size_t)s.n_svc * s.a.n_nodes;
    for (size_t i = (size_t)wv::block() * 256 + wv::tid(); i < n; i += (size_t)256 * 1024) {
        s.hmat[i] = 0;
        s.emat[i] = LIST_EMPTY;
    }
}
WV_KERNEL(256) void k_scan_lists(ScanArgs s) {
    const u32 svc = wv::block_y();
    const u32 e0 = s.a.list_off[svc], e1 = s.a.list_off[svc + 1];
    for (u32 e = e0 + wv::block() * 256 + wv::tid(); e < e1; e += 256u * 64u) {
        const u32 n = s.a.list_node[e];
        if (n == LIST_EMPTY) continue;
        const u32 fl = s.a.list_fail[e], sv = s.a.list_svc[e];
        if (fl >= 256u || sv >= (1u << 24)) s.a.blk->error = ERR_GROUP_RANGE;
        s.hmat[(size_t)svc * s.a.n_nodes + n] = ((fl >= MAX_FAILURES ? fl : 0u) << 24) | sv;
        s.emat[(size_t)svc * s.a.n_nodes + n] = e;
    }
}

// SCAN_NQ: nodes per thread (the launcher instantiates 1, 2 and 4 — scan_nq: a thread of a 1 000-node scan carries no code for nodes it has not)
template <int SCAN_NQ, bool SCAN_LM>
WV_KERNEL(SCAN_THREADS) void k_scan(ScanArgs s) {
    const R6Args& a = s.a;
    const u32 tid = wv::tid(), lane = wv::lane(), N = a.n_nodes, Wn = a.n_words;
    unsigned char* l = reinterpret_cast<unsigned char*>(wv::lds());
    i64* cpu = reinterpret_cast<i64*>(l);
    i64* mem = cpu + N;
    u32* tot = reinterpret_cast<u32*>(mem + N);
    int32_t* lastc = reinterpret_cast<int32_t*>(tot + N);                               // the node's last commit (Explain's chains)
    u64* red_k = reinterpret_cast<u64*>(l + (((size_t)N * 24 + 15) & ~(size_t)15));   // [3] the least key of a task, in rotation: the waves' minima meet here (atomic min)
    u32* red_n = reinterpret_cast<u32*>(red_k + 16);                                     // (spare)
    u32* rtq = red_n + 16;                                                               // [2][SCAN_RTQ] task records, as dwords
    u32* hm = reinterpret_cast<u32*>(l + ((scan_lds(N) + 15) & ~(size_t)15));           // SCAN_LM: [n_svc][N] the key halves ...
    u32* em = hm + (size_t)s.n_svc * N;                                                  // ... and the list entries, here instead of in global memory
    if (a.blk->error != ERR_NONE) return;
    for (u32 n = tid; n < N; n += SCAN_THREADS) { cpu[n] = a.cpu[n]; mem[n] = a.mem[n]; tot[n] = a.total[n]; lastc[n] = a.last[n]; }
    if (SCAN_LM)
        for (u32 x = tid; x < s.n_svc * N; x += SCAN_THREADS) { hm[x] = s.hmat[x]; em[x] = s.emat[x]; }
    u32 nc = a.ctl->ncommit, ni = a.ctl->ninf;
    wv::barrier();
    // What a task reads from global memory — its record, its static-class word and the key halves of this thread's nodes — is
    // requested one task AHEAD: a task's turn then starts with everything in registers (a global round trip per task would be most
    // of its time). The one value the task in between can change, the key half of the node it takes, is corrected by its owner.
    // The task records themselves stream through LDS, SCAN_RTQ at a time, a chunk ahead (each is read once: a miss all the way to HBM).
    const u32* rt32 = reinterpret_cast<const u32*>(a.rt);
    const u32 cdw = SCAN_RTQ * 16u;   // dwords of a chunk
    auto chunk_load = [&](u32 c) -> u32 {   // this thread's dword of chunk c (tasks j0 + c * SCAN_RTQ ...), 0 beyond the stretch
        const u32 t0 = s.j0 + c * SCAN_RTQ;
        return (tid < cdw && t0 + tid / 16u < s.j1) ? rt32[(size_t)t0 * 16u + tid] : 0u;
    };
    if (tid < cdw) rtq[tid] = chunk_load(0);
    if (tid < 3) red_k[tid] = KEY_NONE;
    u32 next_dw = chunk_load(1);
    u32 slot = 0;   // red_k[slot] takes this task's minimum
    wv::barrier();
    auto task_rec = [&](u32 t) -> RTask {
        const u32 i = t - s.j0;
        return *reinterpret_cast<const RTask*>(rtq + ((i / SCAN_RTQ) & 1u) * cdw + (i % SCAN_RTQ) * 16u);
    };
    RTask rn = task_rec(s.j0 < s.j1 ? s.j0 : 0);
    u64 scn[SCAN_NQ];
    u32 hin[SCAN_NQ], ein[SCAN_NQ];
    WV_UNROLL
    for (int q = 0; q < SCAN_NQ; ++q) {
        const u32 n = tid + (u32)q * SCAN_THREADS;
        scn[q] = n < N ? a.sc[(size_t)rn.sc * Wn + (n >> 6)] : 0ull;
        hin[q] = (!SCAN_LM && n < N) ? s.hmat[(size_t)rn.svc * N + n] : 0u;
        ein[q] = (!SCAN_LM && n < N) ? s.emat[(size_t)rn.svc * N + n] : LIST_EMPTY;
    }
    for (u32 t = s.j0; t < s.j1; ++t) {
        const RTask r = rn;
        u64 scw[SCAN_NQ];
        u32 hiw[SCAN_NQ], enw[SCAN_NQ];
        WV_UNROLL
        for (int q = 0; q < SCAN_NQ; ++q) { scw[q] = scn[q]; hiw[q] = hin[q]; enw[q] = ein[q]; }
        if (t + 1 < s.j1) {
            if ((t + 1 - s.j0) % SCAN_RTQ == 0) {   // the next task opens a chunk: it is in this thread's register since the chunk before
                const u32 c = (t + 1 - s.j0) / SCAN_RTQ;
                if (tid < cdw) rtq[(c & 1u) * cdw + tid] = next_dw;   // (that half was last read a chunk ago: a barrier per task lies in between)
                next_dw = chunk_load(c + 1);
                wv::barrier();
            }
            rn = task_rec(t + 1);
            WV_UNROLL
            for (int q = 0; q < SCAN_NQ; ++q) {
                const u32 n = tid + (u32)q * SCAN_THREADS;
                scn[q] = n < N ? a.sc[(size_t)rn.sc * Wn + (n >> 6)] : 0ull;
                hin[q] = (!SCAN_LM && n < N) ? s.hmat[(size_t)rn.svc * N + n] : 0u;
                ein[q] = (!SCAN_LM && n < N) ? s.emat[(size_t)rn.svc * N + n] : LIST_EMPTY;
            }
        }
        const u32 gset = a.n_rg ? a.tg[t] : 0u;
        // ---- every thread: the best of its own nodes
        u64 bk = KEY_NONE;   // nodeLess' key with the node index in its lowest 12 bits: the least key IS the pick (lowest node among equals)
        WV_UNROLL
        for (int q = 0; q < SCAN_NQ; ++q) {
            const u32 n = tid + (u32)q * SCAN_THREADS;
            if (n >= N) continue;
            const u32 w = n >> 6;
            const u64 bit = 1ull << (n & 63);
            if (!(scw[q] & bit)) continue;   // (the static class row holds valid & ready & constraints & platform & plugins)
            const u32 hi = SCAN_LM ? hm[(size_t)r.svc * N + n] : hiw[q];
            if ((r.flags & RT_RES) && !(r.cpu <= cpu[n] && r.mem <= mem[n])) continue;
            bool ok = true;
            if (gset)
                for (u32 g = a.gs_off[gset]; g < a.gs_off[gset + 1]; ++g) {
                    const u32 row = a.gs_row[g];
                    if (a.gcnt[(size_t)a.rg_kind[row] * a.gstride + n] < a.rg_val[row]) ok = false;   // HasEnough, validate.go:24-52
                }
            if (ok && (r.flags & RT_PORTS))
                for (u32 z = a.pset_off[r.pset]; z < a.pset_off[r.pset + 1]; ++z)
                    if (wv::g_fresh64(a.portmap + (size_t)a.pset_ids[z] * Wn + w) & bit) ok = false;
            if (ok && (r.flags & RT_MAXREP) && !((u64)(hi & 0xFFFFFFu) < r.maxrep)) ok = false;
            if (!ok) continue;
            const u32 tn = tot[n];
            if (tn >> 20) a.blk->error = ERR_LEVEL_RANGE;   // (a million tasks on one node: beyond the 20 bits the packed key has for the count)
            const u64 key = ((u64)hi << 32) | ((u64)tn << 12) | n;
            if (key < bk) bk = key;
        }
        // ---- the workgroup's argmin: every wave's least key goes into the task's LDS word by an atomic min; behind the one barrier everybody
        // reads the result. The words rotate (three): the one two tasks ahead is reset behind this barrier — everybody is past reading it
        // (it was the previous task's) and nobody writes it before the next barrier.
        const u64 wk = r6_wave_min64(bk);
        if (lane == 0 && wk != KEY_NONE) wv::lds_min64(red_k + slot, wk);
        wv::barrier();
        const u64 gk = wv::lds_read64(red_k + slot);
        const u32 gn = gk == KEY_NONE ? R6_NONE : (u32)gk & 0xFFFu;
        if (tid == 0) red_k[slot == 0 ? 2 : slot - 1] = KEY_NONE;
        slot = slot == 2 ? 0 : slot + 1;
        // ---- the owner of the node applies the placement (NodeInfo.addTask); everybody counts
        if (gn == R6_NONE) {
            if (tid == 0) {
                a.inf_task[ni] = t;
                a.inf_pos[ni] = nc;
                a.out_node[t] = -1;
            }
            ++ni;
        } else {
            if ((gn & (SCAN_THREADS - 1u)) == tid) {
                const u32 nd = gn, w = nd >> 6;
                const u64 bit = 1ull << (nd & 63);
                if (r.cpu) cpu[nd] -= r.cpu;
                if (r.mem) mem[nd] -= r.mem;
                if (gset)
                    for (u32 g = a.gs_off[gset]; g < a.gs_off[gset + 1]; ++g) a.gcnt[(size_t)a.rg_kind[a.gs_row[g]] * a.gstride + nd] -= a.rg_val[a.gs_row[g]];   // Claim
                if (r.flags & RT_PORTS)
                    for (u32 q = a.pset_off[r.pset]; q < a.pset_off[r.pset + 1]; ++q) wv::g_or64(a.portmap + (size_t)a.pset_ids[q] * Wn + w, bit);
                if (!(r.flags & RT_UNCOUNTED)) {
                    tot[nd] += 1;
                    u32 hi = 0, entry = LIST_EMPTY;   // the node's key half and list entry for this service: read a task ago, in registers
                    if (SCAN_LM) {
                        hi = hm[(size_t)r.svc * N + nd];
                        entry = em[(size_t)r.svc * N + nd];
                    } else {
                        WV_UNROLL
                        for (int q = 0; q < SCAN_NQ; ++q)
                            if (tid + (u32)q * SCAN_THREADS == nd) { hi = hiw[q]; entry = enw[q]; }
                    }
                    hi += 1u;
                    if ((hi & 0xFFFFFFu) == 0) a.blk->error = ERR_GROUP_RANGE;
                    if (SCAN_LM) hm[(size_t)r.svc * N + nd] = hi;
                    else s.hmat[(size_t)r.svc * N + nd] = hi;
                    if (entry == LIST_EMPTY) {
                        wv::g_or64(a.X + (size_t)r.svc * a.xs + w, bit);
                        a.list_node[r.slot] = nd;
                        a.list_svc[r.slot] = 1;
                        a.list_fail[r.slot] = 0;
                        if (SCAN_LM) em[(size_t)r.svc * N + nd] = r.slot;
                        else s.emat[(size_t)r.svc * N + nd] = r.slot;
                        entry = r.slot;
                    } else
                        a.list_svc[entry] = hi & 0xFFFFFFu;   // (the entry's count is the key half's low 24 bits)
                    if (!SCAN_LM && rn.svc == r.svc) {   // the next task is of the same service: what it read of this node is a placement old
                        WV_UNROLL
                        for (int q = 0; q < SCAN_NQ; ++q)
                            if (tid + (u32)q * SCAN_THREADS == nd) { hin[q] = hi; ein[q] = entry; }
                    }
                }
                const int32_t prev = lastc[nd];
                a.log_node[nc] = nd;
                a.log_task[nc] = t;
                a.log_prev[nc] = prev;
                lastc[nd] = (int32_t)nc;
                a.out_node[t] = (int32_t)nd;
            }
            ++nc;
        }
        // (no second barrier: a node's row is read and written by its owner only, the records of a chunk are staged behind a barrier of their own)
    }
    for (u32 n = tid; n < N; n += SCAN_THREADS) { a.cpu[n] = cpu[n]; a.mem[n] = mem[n]; a.total[n] = tot[n]; a.last[n] = lastc[n]; }
    if (tid == 0) {
        a.ctl->ncommit = nc;
        a.ctl->ninf = ni;
        a.blk->pos = s.j1;
    }
}
template <int SCAN_NQ>
WV_KERNEL(SCANB_THREADS) void k_scanb(ScanArgs s) {
    const R6Args& a = s.a;
    const u32 tid = wv::tid(), lane = wv::lane(), N = a.n_nodes, Wn = a.n_words;
    unsigned char* l = reinterpret_cast<unsigned char*>(wv::lds());
    i64* cpu = reinterpret_cast<i64*>(l);
    i64* mem = cpu + N;
    u32* tot = reinterpret_cast<u32*>(mem + N);
    int32_t* lastc = reinterpret_cast<int32_t*>(tot + N);
    u32* hm = reinterpret_cast<u32*>(l + (((size_t)N * 24 + 15) & ~(size_t)15));
    u32* em = hm + (size_t)s.n_svc * N;
    u64* scl = reinterpret_cast<u64*>(em + (size_t)s.n_svc * N);                                   // [n_sc][Wn] the static class rows
    u64* red = scl + (size_t)s.n_sc * Wn;                                                            // [3][SCAN_B] the argmins, three sets in rotation
    u64* wlive = red + 3 * SCAN_B;                                                                   // [SCAN_W / 64] the window's queue: tasks to be looked at
    u32* wrec = reinterpret_cast<u32*>(l + ((reinterpret_cast<unsigned char*>(wlive + SCAN_W / 64) - l + 15) & ~(size_t)15));   // [SCAN_W][16] the window's task records
    u32* wtm = wrec + SCAN_W * 16;                                                                   // [SCAN_W] ... and descriptor ids
    u32* dtab = wtm + SCAN_W;                                                                        // [SCAN_DT] descriptors found unplaceable (id + 1; 0: free)
    if (a.blk->error != ERR_NONE) return;
    for (u32 n = tid; n < N; n += SCANB_THREADS) { cpu[n] = a.cpu[n]; mem[n] = a.mem[n]; tot[n] = a.total[n]; lastc[n] = a.last[n]; }
    for (u32 x = tid; x < s.n_svc * N; x += SCANB_THREADS) { hm[x] = s.hmat[x]; em[x] = s.emat[x]; }
    for (u32 x = tid; x < s.n_sc * Wn; x += SCANB_THREADS) scl[x] = a.sc[x];
    for (u32 x = tid; x < SCAN_DT; x += SCANB_THREADS) dtab[x] = 0;
    if (tid < 3 * SCAN_B) red[tid] = KEY_NONE;
    u32 nc = a.ctl->ncommit, ni = a.ctl->ninf;
    const u32* rt32 = reinterpret_cast<const u32*>(a.rt);
    // a window's records are 16 KB: SCAN_WX times sixteen bytes a thread, requested a window ahead (each is read once: a miss all the way to HBM)
    constexpr int SCAN_WX = SCAN_W * 4 / SCANB_THREADS;
    u32 nr[SCAN_WX][4], ntm;
    auto window_load = [&](u32 base) {
        WV_UNROLL
        for (int x = 0; x < SCAN_WX; ++x) {
            const u32 v = tid + (u32)x * SCANB_THREADS;   // the window's v-th sixteen bytes
            const bool have = base < s.j1 && base + v / 4u < s.j1;
            WV_UNROLL
            for (int k = 0; k < 4; ++k) nr[x][k] = have ? rt32[(size_t)base * 16u + v * 4u + (u32)k] : 0u;
        }
        ntm = (tid < SCAN_W && base < s.j1 && base + tid < s.j1) ? (a.tmpl ? a.tmpl[base + tid] : base + tid) : 0u;
    };
    window_load(s.j0);
    // this thread's nodes (a node beyond the last one tests no class bit: it is never a candidate)
    u32 nn[SCAN_NQ], cbit[SCAN_NQ];
    WV_UNROLL
    for (int q = 0; q < SCAN_NQ; ++q) {
        const u32 n = tid + (u32)q * SCANB_THREADS;
        nn[q] = min(n, N - 1u);
        cbit[q] = n < N ? 1u << (n & 31u) : 0u;
    }
    const u32* scl32 = reinterpret_cast<const u32*>(scl);
    u32 slot = 0, skipped = 0, batches = 0, over = 0;
#ifdef SWP_SCAN_PROF   // section timers (make EXTRA=-DSWP_SCAN_PROF; SWP_DBG=16 prints them): cycles of wave 0 per batch
    u64 pc[6] = {0, 0, 0, 0, 0, 0}, pt = wv::clock64();
#define SCAN_TICK(q) do { const u64 n_ = wv::clock64(); pc[q] += n_ - pt; pt = n_; } while (0)
#else
#define SCAN_TICK(q) do { } while (0)
#endif
    bool any_none = false;
    for (u32 base = s.j0; base < s.j1; base += SCAN_W) {
        const u32 wn = min((u32)SCAN_W, s.j1 - base);
        wv::barrier();   // everybody is past the window before (its records, its ids), and what it added to the table is in
        WV_UNROLL
        for (int x = 0; x < SCAN_WX; ++x) {
            WV_UNROLL
            for (int k = 0; k < 4; ++k) wrec[(tid + (u32)x * SCANB_THREADS) * 4u + (u32)k] = nr[x][k];
        }
        const u32 tm = ntm;
        if (tid < SCAN_W) wtm[tid] = tm;
        window_load(base + SCAN_W);
        const u64 lb = wv::ballot(tid < wn && dtab[scan_dt_cell(tm)] != tm + 1u);
        if (tid < SCAN_W && lane == 0) wlive[tid >> 6] = lb;
        wv::barrier();
        const u32 nc0 = nc;
        u64 placed[SCAN_W / 64], looked[SCAN_W / 64];
        WV_UNROLL
        for (int cw = 0; cw < SCAN_W / 64; ++cw) {
            const u64 lv = wv::lds_read64(wlive + cw);
            u64 cur = ((u64)wv::readfirstlane((u32)(lv >> 32)) << 32) | wv::readfirstlane((u32)lv);   // (the same on every thread: the loop below is a scalar one)
            u64 pl = 0;
            looked[cw] = cur;
            while (cur) {
                SCAN_TICK(5);
                // ---- the batch: the next SCAN_B tasks of the queue (a batch stays inside one word of it). Everything that is the same on
                // every thread — the queue, the picks, the counters — is kept in scalar registers (readfirstlane where a value came through
                // LDS): the loop's bookkeeping then runs on the scalar unit, and its branches are scalar ones.
                u32 ib[SCAN_B];
                u32 nb = 0;
                {
                    u64 c2 = cur;
                    WV_UNROLL
                    for (int b = 0; b < SCAN_B; ++b) {
                        ib[b] = c2 ? (u32)cw * 64u + (u32)wv::ffs64(c2) : ib[0];   // (a batch that is short evaluates its first task again: nobody reads the result)
                        if (c2) ++nb;
                        c2 &= c2 - 1ull;
                    }
                }
                // the records' fields and this thread's node rows: one batch of LDS reads
                i64 rcpu[SCAN_B], rmem[SCAN_B];
                u32 rfl[SCAN_B], rsc[SCAN_B], rsvc[SCAN_B], rslot[SCAN_B];
                u64 rmax[SCAN_B];
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) {
                    const RTask* rp = reinterpret_cast<const RTask*>(wrec + ib[b] * 16u);
                    rcpu[b] = rp->cpu;
                    rmem[b] = rp->mem;
                    rfl[b] = rp->flags;
                    rsc[b] = rp->sc;
                    rsvc[b] = rp->svc;
                    rslot[b] = rp->slot;
                    rmax[b] = rp->maxrep;
                }
                i64 ncpu[SCAN_NQ], nmem[SCAN_NQ];
                u32 ntl[SCAN_NQ];
                WV_UNROLL
                for (int q = 0; q < SCAN_NQ; ++q) {
                    ncpu[q] = cpu[nn[q]];
                    nmem[q] = mem[nn[q]];
                    const u32 tn = tot[nn[q]];
                    over |= tn >> 20;   // (a million tasks on one node: beyond the 20 bits the packed key has for the count; reported at the end)
                    ntl[q] = (tn << 12) | nn[q];
                }
                // what a flag switches off is switched off in the task's numbers (the same on every lane): no ResourceFilter = reservations
                // nothing can undercut, no replica limit = one no count reaches (a count has 24 bits: the limit is cut to 32)
                i64 ecpu[SCAN_B], emem[SCAN_B];
                u32 emax[SCAN_B], scoff[SCAN_B], svoff[SCAN_B];
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) {
                    const u32 fl = wv::readfirstlane(rfl[b]);
                    ecpu[b] = (fl & RT_RES) ? rcpu[b] : INT64_MIN;
                    emem[b] = (fl & RT_RES) ? rmem[b] : INT64_MIN;
                    emax[b] = (fl & RT_MAXREP) ? (rmax[b] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)rmax[b]) : 0xFFFFFFFFu;
                    scoff[b] = wv::readfirstlane(rsc[b]) * Wn * 2u;
                    svoff[b] = wv::readfirstlane(rsvc[b]) * N;
                }
                // ---- every thread: the best of its own nodes, for each of the batch's tasks. First every word the evaluations read — a task's
                // class word (32 nodes of it) and the service's key half per node — requested together: one wait, not one per word
                u32 cwv[SCAN_B][SCAN_NQ], hiv[SCAN_B][SCAN_NQ];
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) {
                    WV_UNROLL
                    for (int q = 0; q < SCAN_NQ; ++q) {
                        cwv[b][q] = scl32[scoff[b] + (nn[q] >> 5)];   // valid & ready & constraints & platform & plugins
                        hiv[b][q] = hm[svoff[b] + nn[q]];
                    }
                }
                // ... then straight-line code, no branch
                u64 bk[SCAN_B];
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) {
                    bk[b] = KEY_NONE;
                    WV_UNROLL
                    for (int q = 0; q < SCAN_NQ; ++q) {
                        const bool ok = ((cwv[b][q] & cbit[q]) != 0) & (ecpu[b] <= ncpu[q]) & (emem[b] <= nmem[q]) & ((hiv[b][q] & 0xFFFFFFu) < emax[b]);
                        const u64 key = ((u64)hiv[b][q] << 32) | ntl[q];
                        bk[b] = (ok & (key < bk[b])) ? key : bk[b];
                    }
                }
                SCAN_TICK(0);
                // the waves' least keys: the four upper halves together, then the lower halves of the lanes that hold the least upper one
                u32 kh[SCAN_B], kl[SCAN_B];
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) kh[b] = (u32)(bk[b] >> 32);
                static_assert(SCAN_B == 4, "wv::min4_u32 takes four values");
                u32 mh0 = kh[0], mh1 = kh[1], mh2 = kh[2], mh3 = kh[3];
                wv::min4_u32(mh0, mh1, mh2, mh3);
                const u32 mh[SCAN_B] = {mh0, mh1, mh2, mh3};
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) kl[b] = kh[b] == mh[b] ? (u32)bk[b] : 0xFFFFFFFFu;
                wv::min4_u32(kl[0], kl[1], kl[2], kl[3]);
                if (lane == 0) {
                    WV_UNROLL
                    for (int b = 0; b < SCAN_B; ++b) wv::lds_min64(red + slot * SCAN_B + b, ((u64)mh[b] << 32) | kl[b]);
                }
                SCAN_TICK(1);
                wv::barrier();
                SCAN_TICK(2);
                u32 gn[SCAN_B];
                bool none[SCAN_B];
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) {
                    const u64 g = wv::lds_read64(red + slot * SCAN_B + b);
                    const u32 glo = wv::readfirstlane((u32)g), ghi = wv::readfirstlane((u32)(g >> 32));
                    none[b] = (glo & ghi) == 0xFFFFFFFFu;
                    gn[b] = glo & 0xFFFu;
                }
                if (tid < SCAN_B) red[(slot == 0 ? 2u : slot - 1u) * SCAN_B + tid] = KEY_NONE;   // (the set of the batch before: everybody is past reading it)
                slot = slot == 2 ? 0 : slot + 1;
                // ---- the longest prefix of tasks whose picks differ (the same on every thread: the same words) — selects, no branch
                bool acc[SCAN_B], mine[SCAN_B];
                u32 ncb[SCAN_B];
                u32 any_mine = 0;
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) {
                    bool clash = (u32)b >= nb;
                    if (b) clash = clash | !acc[b - 1];   // (behind a task that was not accepted nothing is)
                    WV_UNROLL
                    for (int c = 0; c < b; ++c) clash = clash | (!none[b] & !none[c] & (gn[c] == gn[b]));
                    acc[b] = !clash;
                    const bool put = acc[b] & !none[b];
                    cur = acc[b] ? cur & (cur - 1ull) : cur;   // the task leaves the queue
                    pl |= put ? 1ull << (ib[b] & 63u) : 0ull;
                    ncb[b] = nc;
                    nc += put ? 1u : 0u;
                    any_none = any_none | (acc[b] & none[b]);
                    mine[b] = put & ((gn[b] & (SCANB_THREADS - 1u)) == tid);
                }
                ++batches;
                SCAN_TICK(3);
                // no node: nor for any later task with this descriptor (its place among the unplaceable ones: the window's end)
                if (any_none) {
                    any_none = false;
                    if (tid == 0) {
                        WV_UNROLL
                        for (int b = 0; b < SCAN_B; ++b)
                            if (acc[b] && none[b]) {
                                const u32 dm = wtm[ib[b]], cell = scan_dt_cell(dm);
                                if (dtab[cell] == 0) dtab[cell] = dm + 1u;
                            }
                    }
                }
                // ---- the owners apply (NodeInfo.addTask)
                WV_UNROLL
                for (int b = 0; b < SCAN_B; ++b) any_mine |= mine[b] ? 1u : 0u;
                if (any_mine) {
                    WV_UNROLL
                    for (int b = 0; b < SCAN_B; ++b) {
                        if (!mine[b]) continue;
                        const u32 nd = gn[b], w = nd >> 6, tb = base + ib[b];
                        const u64 bit = 1ull << (nd & 63);
                        cpu[nd] -= rcpu[b];
                        mem[nd] -= rmem[b];
                        if (!(rfl[b] & RT_UNCOUNTED)) {
                            tot[nd] += 1;
                            u32 hi = hm[(size_t)rsvc[b] * N + nd] + 1u;
                            u32 entry = em[(size_t)rsvc[b] * N + nd];
                            if ((hi & 0xFFFFFFu) == 0) a.blk->error = ERR_GROUP_RANGE;
                            hm[(size_t)rsvc[b] * N + nd] = hi;
                            if (entry == LIST_EMPTY) {
                                wv::g_or64(a.X + (size_t)rsvc[b] * a.xs + w, bit);
                                a.list_node[rslot[b]] = nd;
                                a.list_svc[rslot[b]] = 1;
                                a.list_fail[rslot[b]] = 0;
                                em[(size_t)rsvc[b] * N + nd] = rslot[b];
                            } else
                                a.list_svc[entry] = hi & 0xFFFFFFu;
                        }
                        const int32_t prev = lastc[nd];
                        a.log_node[ncb[b]] = nd;
                        a.log_task[ncb[b]] = tb;
                        a.log_prev[ncb[b]] = prev;
                        lastc[nd] = (int32_t)ncb[b];
                        a.out_node[tb] = (int32_t)nd;
                    }
                }
                SCAN_TICK(4);
            }
            placed[cw] = pl;
        }
        // ---- the window's unplaceable tasks, looked at or not: a thread per task
        u32 un_before = 0, un_all = 0;   // unplaced tasks of the words in front of this thread's; of the window
        u64 un_mine = 0;
        WV_UNROLL
        for (int cw = 0; cw < SCAN_W / 64; ++cw) {
            const u32 left = wn > (u32)cw * 64u ? wn - (u32)cw * 64u : 0u;
            const u64 un = ~placed[cw] & (left >= 64u ? ~0ull : (1ull << left) - 1ull);
            if ((tid >> 6) == (u32)cw) { un_before = un_all; un_mine = un; }
            un_all += (u32)wv::popc64(un);
        }
        if (tid < wn && ((un_mine >> lane) & 1ull)) {
            const u32 rank = un_before + (u32)wv::popc64(un_mine & ((1ull << lane) - 1ull));
            a.inf_task[ni + rank] = base + tid;
            a.inf_pos[ni + rank] = nc0 + (tid - rank);   // (the commits in front of it: the window's placed tasks in front of it)
            a.out_node[base + tid] = -1;
        }
        ni += un_all;
        WV_UNROLL
        for (int cw = 0; cw < SCAN_W / 64; ++cw) skipped += min(wn > (u32)cw * 64u ? wn - (u32)cw * 64u : 0u, 64u) - (u32)wv::popc64(looked[cw]);
    }
    if (tid == 0) {
        a.blk->scan_skipped += skipped;
        a.blk->scan_batches += batches;
#ifdef SWP_SCAN_PROF
        for (int q = 0; q < 6; ++q) a.ctl->cyc[q] += pc[q];
#endif
    }
    if (over) a.blk->error = ERR_LEVEL_RANGE;
    wv::barrier();
    for (u32 n = tid; n < N; n += SCANB_THREADS) { a.cpu[n] = cpu[n]; a.mem[n] = mem[n]; a.total[n] = tot[n]; a.last[n] = lastc[n]; }
    if (tid == 0) {
        a.ctl->ncommit = nc;
        a.ctl->ninf = ni;
        a.blk->pos = s.j1;
    }
}
#endif   // SWP_SCAN_KERNELS

}  // namespace swpdev
