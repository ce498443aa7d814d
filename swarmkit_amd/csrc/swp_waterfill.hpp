// swp_waterfill.hpp — k_waterfill: a RUN of identical one-off tasks (same service, same filters, same reservations — the
// reference's own benchmark shape is one such run of 100 000, scheduler_test.go:3375-3465; a service-major queue is a sequence of
// them) placed without walking the tasks one by one.
//
// For identical tasks the sequential tick (scheduler.go:694-748 with a heap of one, nodeLess :708-735) is water-filling over
// the key (failure class, svcCount, ActiveTasksCount, node index): each placement takes the minimum and moves that node to
// (svcCount + 1, total + 1). All nodes that share the minimum (failure class, svcCount) — call them A — are therefore served
// once each, in (total, index) order, before any of them is served again: the next min(|A|, tasks left) tasks map one to one
// onto the sorted prefix of A. A phase finds the minimum primary key (a block reduction), then walks A's distinct task counts
// in ascending order; inside one count the order is the node index, i.e. an exclusive scan over the nodes (threads own
// contiguous node ranges). A node leaves the game when its residuals or MaxReplicas allow no further task: its capacity is
// known up front (floor(residual / reservation)), because nothing else touches the nodes during the run. When no node has
// capacity left, the rest of the run is "no suitable node" — all at the same commit position.
//
// Side effects are exactly those of the resolvers' commit (NodeInfo.addTask, nodeinfo.go:108-154): node rows, task counts, the
// service's exception bitmap and list, the commit log with its per-node chains, placements, the residuals in resource units.
// One workgroup; the per-node scratch (primary key, capacity, list entry) lives in global memory so that any node count works.
#pragma once
#include "swp_types.hpp"

namespace swpdev {

struct WaterArgs {
    u32 n_nodes, n_words, xs;
    u32 j0, count;           // the run: tasks [j0, j0 + count), all equal to rt[j0] except for their list slot
    const RTask* rt;
    const u64* sc;           // [n_sc][n_words]
    i64* cpu;
    i64* mem;
    u32* total;
    u64* X;
    u32* list_node;
    u32* list_svc;
    u32* list_fail;
    const u32* list_off;
    int32_t* out_node;
    u32* log_node;
    u32* log_task;
    int32_t* log_prev;
    int32_t* last;
    u32* inf_task;
    u32* inf_pos;
    Ctl* ctl;
    int32_t* qres;           // residuals in resource units (k_resolve5's), or nullptr
    u32* ps;                 // scratch [n_nodes]: failure class << 24 | svcCount; 0xFFFFFFFF = not in the game
    u32* cap;                // scratch [n_nodes]: tasks the node can still take
    u32* ent;                // scratch [n_nodes]: the node's entry in the service's exception list, or LIST_EMPTY
};

#ifdef SWP_WATERFILL_KERNEL
#define WF_THREADS 1024

__device__ __forceinline__ u64 wf_block_min(u64 v, u64* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u64 o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    __syncthreads();   // the previous use of `red` is over
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 g = red[0];
    for (u32 i = 1; i < WF_THREADS / 64; ++i) g = red[i] < g ? red[i] : g;
    return g;
}
// exclusive prefix of `v` over the threads in thread order; *total = the sum
__device__ __forceinline__ u32 wf_block_scan(u32 v, u64* red, u32* total) {
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 o = (u32)__shfl_up((int)inc, off, 64);
        if (lane >= (u32)off) inc += o;
    }
    __syncthreads();
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    u32 base = 0, sum = 0;
    for (u32 i = 0; i < WF_THREADS / 64; ++i) {
        const u32 t = (u32)red[i];
        if (i < wave) base += t;
        sum += t;
    }
    *total = sum;
    return base + inc - v;
}

__global__ __launch_bounds__(WF_THREADS) void k_waterfill(WaterArgs a) {
    __shared__ u64 red[WF_THREADS / 64];
    if (a.ctl->error != ERR_NONE) return;
    const u32 tid = threadIdx.x;
    const u32 C = (a.n_nodes + WF_THREADS - 1) / WF_THREADS;
    const u32 n0 = min(tid * C, a.n_nodes), n1 = min(n0 + C, a.n_nodes);
    const RTask r = a.rt[a.j0];
    const u32 R = a.count;
    const u64* scrow = a.sc + (size_t)r.sc * a.n_words;
    // the service's exception list → per-node entry
    for (u32 n = n0; n < n1; ++n) a.ent[n] = LIST_EMPTY;
    __syncthreads();
    for (u32 e = a.list_off[r.svc] + tid; e < a.list_off[r.svc + 1]; e += WF_THREADS) {
        const u32 n = a.list_node[e];
        if (n != LIST_EMPTY) a.ent[n] = e;
    }
    __syncthreads();
    // every node's place in the game: primary key and capacity
    for (u32 n = n0; n < n1; ++n) {
        const bool elig = (scrow[n >> 6] >> (n & 63)) & 1ull;
        const u32 e = a.ent[n];
        u32 s = 0, fcl = 0;
        if (e != LIST_EMPTY) {
            s = a.list_svc[e];
            const u32 fl = a.list_fail[e];
            fcl = fl >= MAX_FAILURES ? fl - (MAX_FAILURES - 1) : 0u;
        }
        u64 cap = elig ? (u64)R : 0ull;
        if (cap && (r.flags & RT_RES)) {   // ResourceFilter.Check (filter.go:77-84) k more times: k * reservation <= residual
            const i64 c = a.cpu[n], m = a.mem[n];
            if (c < 0 || m < 0) cap = 0;   // a reservation of 0 still fails against a negative residual
            if (cap && r.cpu > 0) cap = min(cap, (u64)(c / r.cpu));
            if (cap && r.mem > 0) cap = min(cap, (u64)(m / r.mem));
        }
        if (cap && (r.flags & RT_MAXREP)) cap = r.maxrep > (u64)s ? min(cap, r.maxrep - (u64)s) : 0ull;   // filter.go:373-375
        if (s >= (1u << 24) || fcl >= 255u) cap = 0;   // beyond the packed key (never in practice): such a node is simply skipped... and flagged
        a.ps[n] = (fcl << 24) | s;
        a.cap[n] = (u32)min(cap, (u64)0xFFFFFFFFull);
    }
    const u32 ncommit0 = a.ctl->ncommit, ninf0 = a.ctl->ninf;
    u32 done = 0;
    while (done < R) {   // uniform
        u64 pm = KEY_NONE;
        for (u32 n = n0; n < n1; ++n)
            if (a.cap[n]) pm = min(pm, (u64)a.ps[n]);
        const u64 P = wf_block_min(pm, red);
        if (P == KEY_NONE) break;   // nobody can take another task
        u64 tlast = KEY_NONE;       // task counts of A already served in this phase: all below or equal to tlast (none yet)
        while (done < R) {
            u64 vm = KEY_NONE;
            for (u32 n = n0; n < n1; ++n)
                if (a.cap[n] && a.ps[n] == (u32)P) {
                    const u64 t = a.total[n];
                    if (tlast == KEY_NONE || t > tlast) vm = min(vm, t);
                }
            const u64 v = wf_block_min(vm, red);
            if (v == KEY_NONE) break;   // A is served once over
            u32 cnt = 0;
            for (u32 n = n0; n < n1; ++n) cnt += (a.cap[n] && a.ps[n] == (u32)P && a.total[n] == (u32)v) ? 1u : 0u;
            u32 nB = 0;
            const u32 excl = wf_block_scan(cnt, red, &nB);
            const u32 m = min(nB, R - done);
            u32 local = 0;
            for (u32 n = n0; n < n1; ++n) {
                if (!(a.cap[n] && a.ps[n] == (u32)P && a.total[n] == (u32)v)) continue;
                const u32 rank = excl + local++;
                if (rank >= m) break;
                const u32 gj = a.j0 + done + rank, ci = ncommit0 + done + rank, w = n >> 6;
                const u64 bit = 1ull << (n & 63);
                if (r.cpu) a.cpu[n] -= r.cpu;
                if (r.mem) a.mem[n] -= r.mem;
                if (a.qres) {
                    a.qres[2 * n] -= (int32_t)r.kc;
                    a.qres[2 * n + 1] -= (int32_t)r.km;
                }
                a.total[n] = (u32)v + 1;
                const u32 e = a.ent[n];
                if (e == LIST_EMPTY) {   // first task of the service here: the node joins its exception list (this task's own slot)
                    const u32 slot = a.rt[gj].slot;
                    atomicOr(a.X + (size_t)r.svc * a.xs + w, bit);
                    a.list_node[slot] = n;
                    a.list_svc[slot] = 1;
                    a.list_fail[slot] = 0;
                    a.ent[n] = slot;
                } else
                    a.list_svc[e] += 1;
                a.log_node[ci] = n;
                a.log_task[ci] = gj;
                a.log_prev[ci] = a.last[n];
                a.last[n] = (int32_t)ci;
                a.out_node[gj] = (int32_t)n;
                a.ps[n] += 1;
                a.cap[n] -= 1;
            }
            done += m;
            tlast = v;
        }
    }
    // the rest of the run has no suitable node: all at the same moment
    for (u32 i = done + tid; i < R; i += WF_THREADS) {
        a.inf_task[ninf0 + (i - done)] = a.j0 + i;
        a.inf_pos[ninf0 + (i - done)] = ncommit0 + done;
    }
    __syncthreads();
    if (tid == 0) {
        a.ctl->ncommit = ncommit0 + done;
        a.ctl->ninf = ninf0 + (R - done);
        a.ctl->slow_tasks += 0;
    }
}
#endif

}  // namespace swpdev
