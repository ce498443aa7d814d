// swp_shard.hpp — kernels of the node-range sharded scan (include/swp.h "node-range shards").
//
//   k_propose      one wavefront per pending task of the block: Pipeline.Process over THIS shard's nodes against the state
//                  as it is (static class row, ResourceFilter on the live residuals filter.go:77-84, HostPortFilter :336-347,
//                  exception nodes excluded), the minimum ActiveTasksCount among the survivors (nodeLess with svcCount == 0,
//                  scheduler.go:708-735), the first SWP_SHARD_CAND survivors of that level in node order — what a heap of one
//                  over this shard's range would end on, plus the runners-up the merge needs inside a block — and the best
//                  node of the service's exception list (nodes where it already runs / that failed it) by the full key.
//   k_shard_apply  NodeInfo.addTask (nodeinfo.go:108-154) for the picks this shard owns + the commit log / unplaceable-task
//                  records of the explain pass, numbered globally over all shards.
//
// The node rows are read straight from memory (L2-resident: 20 B per node and task); a word whose static class word is empty
// costs two scalar loads. One launch covers a block of tasks, so the grid is as wide as the block.
#pragma once
#include "swp_types.hpp"

namespace swpdev {

struct Proposal {   // == swp_proposal (include/swp.h)
    u32 level, n_cand;
    u32 word[4];
    u64 bits[4];
    u64 exc_hi, exc_lo;
    u32 exc_entry, flags;   // flags bit 0: an uncounted task (its pick ends the block)
};
static_assert(sizeof(Proposal) == 80, "swp_proposal layout");

struct ProposeArgs {
    u32 n_nodes, n_words, j0, count, xs;
    const i64* cpu;
    const i64* mem;
    const u32* total;
    const RTask* rt;
    const u64* sc;        // [n_sc][n_words]
    const u64* X;         // [n_svc][xs]
    const u64* portmap;
    const u32* pset_off;
    const u32* pset_ids;
    const u32* list_node;
    const u32* list_svc;
    const u32* list_fail;
    const u32* list_off;
    Proposal* out;        // [count]
};

struct ShardPickDev { u32 task, node, entry, ci; };   // a pick this shard owns: batch task index, local node, list entry, global commit index
struct ShardInfDev { u32 task, pos; };                // an unplaceable task and the number of commits (all shards) before it

struct ShardApplyArgs {
    u32 n_picks, n_inf, inf_base, n_words, xs;
    const ShardPickDev* picks;
    const ShardInfDev* infs;
    const RTask* rt;
    i64* cpu;
    i64* mem;
    u32* total;
    u64* X;
    u32* list_node;
    u32* list_svc;
    u32* list_fail;
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
};

#ifdef SWP_SHARD_KERNELS   // the kernels themselves: swp_shard.hip only (the engine TU shares the argument records)
__device__ __forceinline__ u32 sh_wave_min32(u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, (u32)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ __forceinline__ u64 sh_wave_min64(u64 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u64 o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
template <class T>
__device__ __forceinline__ T sh_uload(const T* p) {   // wave-uniform read-only load through the scalar cache
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

__global__ __launch_bounds__(64) void k_propose(ProposeArgs a) {
    const u32 lane = threadIdx.x, t = blockIdx.x;
    if (t >= a.count) return;
    const RTask* rt = a.rt + a.j0 + t;
    const i64 rcpu = sh_uload(&rt->cpu), rmem = sh_uload(&rt->mem);
    const u32 flags = sh_uload(&rt->flags), svc = sh_uload(&rt->svc), scid = sh_uload(&rt->sc), pset = sh_uload(&rt->pset);
    const u64 maxrep = sh_uload(&rt->maxrep);
    const u64* scrow = a.sc + (size_t)scid * a.n_words;
    const u64* xrow = a.X + (size_t)svc * a.xs;
    u32 p0 = 0, p1 = 0;
    if (flags & RT_PORTS) {
        p0 = sh_uload(a.pset_off + pset);
        p1 = sh_uload(a.pset_off + pset + 1);
    }
    // survivors of one node word: static class, not an exception node, host ports free — as a wave-uniform mask — then the
    // ResourceFilter per lane on the live residuals
    auto word_mask = [&](u32 w) -> u64 {
        u64 m = sh_uload(scrow + w) & ~xrow[w];   // X is written by k_shard_apply between launches: a plain load sees it
        for (u32 p = p0; p < p1 && m; ++p) m &= ~a.portmap[(size_t)sh_uload(a.pset_ids + p) * a.n_words + w];
        return m;
    };
    auto node_ok = [&](u32 n) -> bool { return !(flags & RT_RES) || (rcpu <= a.cpu[n] && rmem <= a.mem[n]); };

    // pass 1: the minimum level among the survivors
    u32 lmin = 0xFFFFFFFFu;
    for (u32 w = 0; w < a.n_words; ++w) {
        const u64 m = word_mask(w);
        if (!m) continue;   // uniform
        const u32 n = w * 64 + lane;
        if (((m >> lane) & 1) && n < a.n_nodes && node_ok(n)) lmin = min(lmin, a.total[n]);
    }
    lmin = sh_wave_min32(lmin);
    // pass 2: the first non-empty words of that level's survivors, in node order (the ballot IS the word)
    u32 cnt = 0, more = 0, cw[4] = {0, 0, 0, 0};
    u64 cb[4] = {0, 0, 0, 0};
    if (lmin != 0xFFFFFFFFu)
        for (u32 w = 0; w < a.n_words && !more; ++w) {
            const u64 m = word_mask(w);
            if (!m) continue;
            const u32 n = w * 64 + lane;
            const bool hit = ((m >> lane) & 1) && n < a.n_nodes && node_ok(n) && a.total[n] == lmin;
            const u64 bal = __ballot(hit);
            if (!bal) continue;
            if (cnt == 4) { more = 1; break; }
            cw[cnt] = w;
            cb[cnt] = bal;
            ++cnt;
        }
    // the service's exception list by the full key (scheduler.go:708-735): lanes stride over the entries
    u64 bhi = KEY_NONE, blo = KEY_NONE;
    u32 be = 0;
    const u32 e0 = sh_uload(a.list_off + svc), e1 = sh_uload(a.list_off + svc + 1);
    for (u32 e = e0 + lane; e < e1; e += 64) {
        const u32 n = a.list_node[e];
        if (n == LIST_EMPTY) continue;
        const u32 w = n >> 6;
        const u64 bit = 1ull << (n & 63);
        if (!(scrow[w] & bit)) continue;
        if (!node_ok(n)) continue;
        bool used = false;
        for (u32 p = p0; p < p1; ++p)
            if (a.portmap[(size_t)a.pset_ids[p] * a.n_words + w] & bit) used = true;
        if (used) continue;
        const u32 sv = a.list_svc[e], fl = a.list_fail[e];
        if ((flags & RT_MAXREP) && !((u64)sv < maxrep)) continue;   // filter.go:373-375
        const u32 fcl = fl >= MAX_FAILURES ? fl - (MAX_FAILURES - 1) : 0u;
        const u64 hi = ((u64)fcl << 32) | sv, lo = ((u64)a.total[n] << 32) | n;
        if (hi < bhi || (hi == bhi && lo < blo)) { bhi = hi; blo = lo; be = e; }
    }
    const u64 ghi = sh_wave_min64(bhi);
    const u64 glo = sh_wave_min64(bhi == ghi ? blo : KEY_NONE);
    const u64 who = __ballot(bhi == ghi && blo == glo && ghi != KEY_NONE);
    const u32 gentry = who ? (u32)__shfl((int)be, __ffsll((long long)who) - 1, 64) : 0u;
    if (lane == 0) {
        Proposal pr;
        pr.level = lmin;
        pr.n_cand = cnt | (more ? 0x80000000u : 0u);
        for (int i = 0; i < 4; ++i) {
            pr.word[i] = cw[i];
            pr.bits[i] = cb[i];
        }
        pr.exc_hi = ghi;
        pr.exc_lo = ghi == KEY_NONE ? KEY_NONE : glo;
        pr.exc_entry = gentry;
        pr.flags = (flags & RT_UNCOUNTED) ? 1u : 0u;
        a.out[t] = pr;
    }
}

__global__ __launch_bounds__(256) void k_shard_apply(ShardApplyArgs a) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_inf) {
        a.inf_task[a.inf_base + i] = a.infs[i].task;
        a.inf_pos[a.inf_base + i] = a.infs[i].pos;
    }
    if (i >= a.n_picks) return;
    const ShardPickDev p = a.picks[i];
    const RTask r = a.rt[p.task];
    const u32 n = p.node, w = n >> 6;
    const u64 bit = 1ull << (n & 63);
    // no two picks of one block share a node (a taken node is struck from every later list), so plain read-modify-write
    // of the node row is safe; words of X / the port bitmaps are shared between nodes: atomics
    if (r.cpu) a.cpu[n] -= r.cpu;
    if (r.mem) a.mem[n] -= r.mem;
    if (r.flags & RT_PORTS)
        for (u32 q = a.pset_off[r.pset]; q < a.pset_off[r.pset + 1]; ++q) atomicOr(a.portmap + (size_t)a.pset_ids[q] * a.n_words + w, bit);
    if (!(r.flags & RT_UNCOUNTED)) {
        a.total[n] += 1;
        if (p.entry == LIST_EMPTY) {
            atomicOr(a.X + (size_t)r.svc * a.xs + w, bit);
            a.list_node[r.slot] = n;
            a.list_svc[r.slot] = 1;
            a.list_fail[r.slot] = 0;
        } else
            a.list_svc[p.entry] += 1;
    }
    a.log_node[p.ci] = n;
    a.log_task[p.ci] = p.task;
    a.log_prev[p.ci] = a.last[n];
    a.last[n] = (int32_t)p.ci;
    a.out_node[p.task] = (int32_t)n;
}
#endif

}  // namespace swpdev
