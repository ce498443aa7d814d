// swp_device.hpp — CDNA4 (gfx950) kernels of the batch task-placement engine.
//
// Data layout in HBM (all struct-of-arrays, node index = canonical scan order):
//   cpu[n], mem[n] : int64   AvailableResources (nodeinfo.go:35)
//   total[n]       : uint32  ActiveTasksCount
//   flags[n]       : uint32  SWP_NODE_* | DEV_VALID
//   attr[c][n]     : uint32  folded-string ids per constraint column (id, hostname, os, arch, labels…)
//   bitmaps        : uint64 words, bit i of word w = node 64*w+i  ("a wave's ballot IS a word")
//     ready[w], con[class][w], plat[class][w], plug[class][w], sc[static class][w]
//     X[svc][w]    nodes that are NOT "plain" for a service (svcCount>0 or recent failures ≥5)
//
// This header holds the kernels around the resolvers: predicate classes, the explain pass, the
// event-handler residual updates, the enforcer sweep and the pair check. The sequential argmin + commit pass itself lives in
// swp_resolve5.hpp (round resolver, node sets that fit one workgroup's LDS), swp_resolve6.hpp (block resolver, bitmap rows in
// global memory), swp_waterfill.hpp (runs of identical tasks) and swp_shard.hpp (node-range shards). Why parallel kernels
// reproduce the reference's strictly sequential tick (scheduler.go:464-469) — inside one batch feasibility only goes 1→0 and a
// node's score (max(fail,4), svcCount, total, index) only grows — is argued in DESIGN.md §2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "swp_types.hpp"

namespace swpdev {

// Uniform (wave-invariant) read-only loads go through the constant address space so that the
// backend emits s_load (scalar cache) instead of 64 identical vector loads.
template <class T>
__device__ __forceinline__ T cload(const T* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }

// ---------------------------------------------------------------------------------------------
// class bitmaps: one launch evaluates [classes × nodes]
// ---------------------------------------------------------------------------------------------
struct NodeView {
    u32 n_nodes, n_words, ncap;
    const u32* flags;
    const i64* cpu;
    const i64* mem;
    const u32* total;
    const u32* os;      // SWP_SPACE_OS ids
    const u32* arch;    // SWP_SPACE_ARCH ids
    const u32* attr;    // [ncols][ncap]
    const u32* ip;      // [ncap][4]
    const u32* plug_off;   // [n+1]
    const u32* plug_ids;
    u32 role_worker, role_manager;   // FOLDED ids of "WORKER"/"MANAGER"
};

// ready[w] = READY && valid (ReadyFilter.Check, filter.go:41-44); valid[w] = slot present
__global__ void k_ready(NodeView nv, u64* __restrict__ ready, u64* __restrict__ valid) {
    u32 n = blockIdx.x * blockDim.x + threadIdx.x;
    u32 f = n < nv.n_nodes ? nv.flags[n] : 0u;
    u64 r = ballot64((f & DEV_VALID) && (f & NF_READY));
    u64 v = ballot64((f & DEV_VALID) != 0);
    if ((threadIdx.x & 63) == 0 && (n >> 6) < nv.n_words) {
        ready[n >> 6] = r;
        valid[n >> 6] = v;
    }
}

// constraint.NodeMatches (constraint.go:107-207) on interned ids. grid.y = class (row 0 unused).
__global__ void k_constraint_classes(NodeView nv, const u32* __restrict__ con_off, const DevConstraint* __restrict__ cons,
                                     u64* __restrict__ out /* [ncls][n_words] */) {
    u32 cls = blockIdx.y + 1;
    u32 n = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = n < nv.n_nodes;
    u32 f = in ? nv.flags[n] : 0u;
    bool ok = in && (f & DEV_VALID);
    u32 c0 = cload(con_off + cls), c1 = cload(con_off + cls + 1);
    for (u32 c = c0; c < c1 && __any(ok); ++c) {
        DevConstraint k = cons[c];
        bool pass = false;
        if (ok) {
            u32 val = 0;
            bool is_ip = false;
            switch (k.kind) {
            case 0: val = nv.attr[0 * (size_t)nv.ncap + n]; break;                                                   // node.id
            case 1: val = (f & NF_HAS_DESC) ? nv.attr[1 * (size_t)nv.ncap + n] : 0u; break;                          // node.hostname
            case 2: is_ip = true; break;
            case 3: val = (f & NF_MANAGER) ? nv.role_manager : nv.role_worker; break;                                // node.role
            case 4: val = ((f & NF_HAS_DESC) && (f & NF_HAS_PLATFORM)) ? nv.attr[2 * (size_t)nv.ncap + n] : 0u; break;   // node.platform.os
            case 5: val = ((f & NF_HAS_DESC) && (f & NF_HAS_PLATFORM)) ? nv.attr[3 * (size_t)nv.ncap + n] : 0u; break;   // node.platform.arch
            case 6: val = (f & NF_HAS_LABELS) ? nv.attr[k.col * (size_t)nv.ncap + n] : 0u; break;                    // node.labels.*
            case 7: val = ((f & NF_HAS_DESC) && (f & NF_HAS_ENGINE) && (f & NF_HAS_ELABELS)) ? nv.attr[k.col * (size_t)nv.ncap + n] : 0u; break;
            default: break;
            }
            if (k.kind <= 7 && !is_ip) {
                bool eq = (val == k.value);
                pass = (k.op == 0) ? eq : !eq;
            } else if (is_ip) {
                // constraint.go:127-146
                const u32* a = nv.ip + (size_t)n * 4;
                bool valid_ip = (f & NF_IP_VALID) != 0;
                if (k.ip_kind == 0) {
                    bool eq = valid_ip && a[0] == k.ip[0] && a[1] == k.ip[1] && a[2] == k.ip[2] && a[3] == k.ip[3];
                    pass = (k.op == 0) ? eq : !eq;
                } else if (k.ip_kind == 1) {
                    bool within = valid_ip && (((f & NF_IP_V4) != 0) == (k.ip_is_v4 != 0));
                    if (within) {
                        // compare the first prefix_len bits (ip words are big-endian packed)
                        u32 bits = k.prefix_len;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            u32 nb = bits >= 32 ? 32 : bits;
                            u32 mask = nb == 0 ? 0u : (nb == 32 ? 0xFFFFFFFFu : ~((1u << (32 - nb)) - 1u));
                            within = within && ((a[q] & mask) == (k.ip[q] & mask));
                            bits -= nb;
                        }
                    }
                    pass = (k.op == 0) ? within : !within;
                } else {
                    pass = false;   // malformed: both operators fail
                }
            }
        }
        ok = ok && pass;
    }
    u64 word = ballot64(ok);
    if ((threadIdx.x & 63) == 0 && (n >> 6) < nv.n_words) out[(size_t)cls * nv.n_words + (n >> 6)] = word;
}

// PlatformFilter.Check (filter.go:266-306) on (OS id, normalised ARCH id) pairs
__global__ void k_platform_classes(NodeView nv, const u32* __restrict__ off, const uint2* __restrict__ plats, u64* __restrict__ out) {
    u32 cls = blockIdx.y + 1;
    u32 n = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = n < nv.n_nodes;
    u32 f = in ? nv.flags[n] : 0u;
    bool elig = in && (f & DEV_VALID) && (f & NF_HAS_DESC) && (f & NF_HAS_PLATFORM);
    u32 os = elig ? nv.os[n] : 0u, arch = elig ? nv.arch[n] : 0u;
    bool ok = false;
    u32 c0 = cload(off + cls), c1 = cload(off + cls + 1);
    for (u32 c = c0; c < c1; ++c) {
        uint2 p = plats[c];   // x = os, y = arch ; 0 = wildcard
        ok = ok || (elig && (p.y == 0u || p.y == arch) && (p.x == 0u || p.x == os));
    }
    u64 word = ballot64(ok);
    if ((threadIdx.x & 63) == 0 && (n >> 6) < nv.n_words) out[(size_t)cls * nv.n_words + (n >> 6)] = word;
}

// PluginFilter.Check (filter.go:135-176). req list per class: [log_plugin, required...]
__global__ void k_plugin_classes(NodeView nv, const u32* __restrict__ off, const u32* __restrict__ req, u64* __restrict__ out) {
    u32 cls = blockIdx.y + 1;
    u32 n = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = n < nv.n_nodes;
    u32 f = in ? nv.flags[n] : 0u;
    bool present = in && (f & DEV_VALID);
    bool ok = present;
    if (present && (f & NF_HAS_DESC) && (f & NF_HAS_ENGINE)) {
        u32 p0 = nv.plug_off[n], p1 = nv.plug_off[n + 1];
        u32 c0 = off[cls], c1 = off[cls + 1];
        u32 logp = req[c0];
        for (u32 c = c0 + 1; c < c1 && ok; ++c) {
            u32 want = req[c];
            bool found = false;
            for (u32 q = p0; q < p1; ++q) found = found || (nv.plug_ids[q] == want);
            ok = found;
        }
        if (ok && logp != 0u) {
            bool found = false;
            for (u32 q = p0; q < p1; ++q) found = found || (nv.plug_ids[q] == logp);
            if (!found && (f & NF_HAS_LOGPLUG)) ok = false;
        }
    }
    u64 word = ballot64(ok);
    if ((threadIdx.x & 63) == 0 && (n >> 6) < nv.n_words) out[(size_t)cls * nv.n_words + (n >> 6)] = word;
}

// sc[s][w] = ready & plug[g] & con[c] & plat[p]; class row 0 of each table means "filter disabled".
__global__ void k_static_combine(u32 n_words, u32 n_sc, const uint4* __restrict__ triples /* x=con,y=plat,z=plug */,
                                 const u64* __restrict__ ready, const u64* __restrict__ con, const u64* __restrict__ plat,
                                 const u64* __restrict__ plug, u64* __restrict__ sc) {
    u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    u32 s = blockIdx.y;
    if (w >= n_words || s >= n_sc) return;
    uint4 t = triples[s];
    u64 v = ready[w];
    if (t.x) v &= con[(size_t)t.x * n_words + w];
    if (t.y) v &= plat[(size_t)t.y * n_words + w];
    if (t.z) v &= plug[(size_t)t.z * n_words + w];
    sc[(size_t)s * n_words + w] = v;
}

// scatter sparse (node) lists into bitmaps: X[svc] from exception lists, portmap[p] from port users
__global__ void k_scatter_bits(u32 n_entries, const u32* __restrict__ row, const u32* __restrict__ node, u32 n_words, u64* __restrict__ bm) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    u32 n = node[i];
    if (n == LIST_EMPTY) return;
    atomicOr(&bm[(size_t)row[i] * n_words + (n >> 6)], 1ull << (n & 63));
}

// X[s] = the nodes of service s's exception list (the entries the batch starts with; a task's reserved slot is LIST_EMPTY): read off
// the lists themselves — grid (entries of the longest list / 256 at most 64, services) — instead of a second (row, node) copy of them
__global__ void k_scatter_lists(const u32* __restrict__ list_off, const u32* __restrict__ list_node, u32 n_words, u64* __restrict__ X) {
    const u32 s = blockIdx.y;
    const u32 e1 = list_off[s + 1];
    for (u32 e = list_off[s] + blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += gridDim.x * blockDim.x) {
        const u32 n = list_node[e];
        if (n != LIST_EMPTY) atomicOr(&X[(size_t)s * n_words + (n >> 6)], 1ull << (n & 63));
    }
}

// ---------------------------------------------------------------------------------------------
// k_explain — per-filter first-failure histogram for every task that found no node, evaluated
// against the node state AT THE MOMENT that task was tried (Pipeline.Process counters,
// pipeline.go:56-68, read by Explain :84-103). The state is rebuilt per node by walking that
// node's chain of commits backwards from the end of the batch.
// ---------------------------------------------------------------------------------------------
struct ExplainArgs {
    u32 n_nodes, n_words, n_inf;
    const u32* inf_task;
    const u32* inf_pos;
    const RTask* rt;
    const u64* valid;
    const u64* ready;
    const u64* con;
    const u64* plat;
    const u64* plug;
    const i64* cpu;
    const i64* mem;
    const u64* portmap;
    const u32* pset_off;
    const u32* pset_ids;
    const u32* list_node;
    const u32* list_svc;
    const u32* list_off;
    const u32* log_task;
    const int32_t* log_prev;
    const int32_t* last;
    u32* hist;   // [T][8]
    // per-node commit segments, ascending commit index, with suffix sums of the reservations (k_chain_segments)
    const u32* seg_off;   // [n_nodes]
    const u32* seg_len;   // [n_nodes]
    const u32* ent_ci;    // [ncommit]
    const i64* ent_scpu;  // [ncommit] sum of cpu over this entry and every later one of the node
    const i64* ent_smem;
    // generic reservations (n_rg == 0: none): the rows and sets of the batch, the END-of-batch counts
    u32 n_rg, gstride;
    const int32_t* gcnt;
    const u32* tg;
    const u32* gs_off;
    const u32* gs_row;
    const u32* rg_kind;
    const int32_t* rg_val;
    // tasks with cluster mounts (nullptr: none): the VolumesFilter row of the task's deciding round — the volumes as they stood at its moment
    const u32* csi_of;
    const u64* vrows;
};

// One thread per node: the node's chain of commits (arbitrary order) → a contiguous segment sorted by commit index
// with suffix sums, so that the explain pass finds "residuals at the task's moment" with a short contiguous scan
// instead of three dependent loads per chain step.
struct SegArgs {
    u32 n_nodes;
    const RTask* rt;
    const u32* log_task;
    const int32_t* log_prev;
    const int32_t* last;
    u32* alloc;   // one counter
    u32* seg_off;
    u32* seg_len;
    u32* ent_ci;
    i64* ent_scpu;
    i64* ent_smem;
};
__global__ __launch_bounds__(256) void k_chain_segments(SegArgs a) {
    const u32 n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.n_nodes) return;
    u32 len = 0;
    for (int32_t ci = a.last[n]; ci >= 0; ci = a.log_prev[ci]) ++len;
    const u32 off = len ? atomicAdd(a.alloc, len) : 0u;
    a.seg_off[n] = off;
    a.seg_len[n] = len;
    if (!len) return;
    // A chain runs from the node's last commit back: descending commit indices whenever commits were linked in order (every
    // resolver does), so the segment is written back to front and is sorted; a chain that is not (linked out of order) gets an
    // insertion sort afterwards. (Sorting while copying was quadratic on the long chains of a one-service batch: 180 commits a node.)
    u32 k = 0;
    bool descending = true;
    int32_t before = 0x7FFFFFFF;
    for (int32_t ci = a.last[n]; ci >= 0; ci = a.log_prev[ci], ++k) {
        const RTask* tk = a.rt + a.log_task[ci];
        const u32 p = len - 1u - k;
        a.ent_ci[off + p] = (u32)ci;
        a.ent_scpu[off + p] = tk->cpu;
        a.ent_smem[off + p] = tk->mem;
        descending = descending && ci < before;
        before = ci;
    }
    if (!descending)
        for (u32 q = 1; q < len; ++q) {
            const u32 ci = a.ent_ci[off + q];
            const i64 tc = a.ent_scpu[off + q], tm = a.ent_smem[off + q];
            u32 p = q;
            while (p > 0 && a.ent_ci[off + p - 1] > ci) {
                a.ent_ci[off + p] = a.ent_ci[off + p - 1];
                a.ent_scpu[off + p] = a.ent_scpu[off + p - 1];
                a.ent_smem[off + p] = a.ent_smem[off + p - 1];
                --p;
            }
            a.ent_ci[off + p] = ci;
            a.ent_scpu[off + p] = tc;
            a.ent_smem[off + p] = tm;
        }
    i64 sc = 0, sm = 0;
    for (u32 p = len; p-- > 0;) {
        sc += a.ent_scpu[off + p];
        sm += a.ent_smem[off + p];
        a.ent_scpu[off + p] = sc;
        a.ent_smem[off + p] = sm;
    }
}

#define EX_TCH 32   // unplaceable tasks per block: the node row is loaded once and stays in registers
__global__ __launch_bounds__(256) void k_explain(ExplainArgs a) {
    const u32 e0 = blockIdx.y * EX_TCH, e1 = min(a.n_inf, e0 + EX_TCH);
    const u32 n = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 w = n >> 6;   // one word per wave
    const u64 bit = 1ull << (n & 63);
    const bool inw = w < a.n_words;
    const bool present = n < a.n_nodes && inw && (cload(a.valid + (inw ? w : 0)) & bit);
    const bool is_ready = inw && (cload(a.ready + (inw ? w : 0)) & bit);
    const i64 c_end = present ? a.cpu[n] : 0, m_end = present ? a.mem[n] : 0;   // end-of-batch residuals
    __shared__ u32 cnt[EX_TCH][8];
    for (u32 q = threadIdx.x; q < EX_TCH * 8; q += blockDim.x) (&cnt[0][0])[q] = 0;
    __syncthreads();
    // the node's commit segment is walked ONCE per block: the unplaceable tasks are recorded in batch order, so their moments
    // (commits before them) ascend and the cursor only moves forward (a moment that steps back restarts it)
    const u32 seg_o = present ? a.seg_off[n] : 0u, seg_n = present ? a.seg_len[n] : 0u;
    u32 seg_p = 0, seg_cached = 0xFFFFFFFFu;
    int32_t seg_pos = 0;
    i64 seg_c = 0, seg_m = 0;
    for (u32 e = e0; e < e1; ++e) {
        const u32 gj = cload(a.inf_task + e);
        const int32_t pos = (int32_t)cload(a.inf_pos + e);
        const RTask* rp = a.rt + gj;
        const u32 rflags = cload(&rp->flags), rsvc = cload(&rp->svc), rpset = cload(&rp->pset);
        const u32 cls_con = cload(&rp->cls_con), cls_plat = cload(&rp->cls_plat), cls_plug = cload(&rp->cls_plug);
        const i64 rcpu = cload(&rp->cpu), rmem = cload(&rp->mem);
        const u64 rmaxrep = cload(&rp->maxrep);
        int ff = -1;
        if (present) {
            // state of node n at the task's moment = end-of-batch state minus the commits with index >= pos; the chain
            // is only walked when a filter's verdict can depend on those commits (resources only shrink inside a
            // batch: a node that still fits at the end fitted at the task's moment)
            i64 c = c_end, m = m_end;
            u32 svc_later = 0;
            u32 port_later = 0;   // bit q: the q-th port of this task's set was taken on n by a LATER commit
            u32 pp0 = 0, pp1 = 0;
            if (rflags & RT_PORTS) { pp0 = a.pset_off[rpset]; pp1 = a.pset_off[rpset + 1]; }
            bool walked = false;
            auto walk = [&]() {
                if (walked) return;
                walked = true;
                for (int32_t ci = a.last[n]; ci >= 0; ci = a.log_prev[ci]) {   // chain order is arbitrary: filter, don't stop early
                    if (ci < pos) continue;
                    const RTask* tk = a.rt + a.log_task[ci];
                    c += tk->cpu;
                    m += tk->mem;
                    if (tk->svc == rsvc && !(tk->flags & RT_UNCOUNTED)) ++svc_later;
                    if ((rflags & RT_PORTS) && (tk->flags & RT_PORTS)) {
                        for (u32 q = pp0; q < pp1 && q - pp0 < 32; ++q)
                            for (u32 z = a.pset_off[tk->pset]; z < a.pset_off[tk->pset + 1]; ++z)
                                if (a.pset_ids[z] == a.pset_ids[q]) port_later |= 1u << (q - pp0);
                    }
                }
            };
            bool res_fail = false;
            if ((rflags & RT_RES) && is_ready && !(rcpu <= c_end && rmem <= m_end)) {
                // residuals at the task's moment = end state + reservations of the node's commits with index >= pos
                if (pos < seg_pos) seg_p = 0;
                seg_pos = pos;
                while (seg_p < seg_n && a.ent_ci[seg_o + seg_p] < (u32)pos) ++seg_p;
                if (seg_p != seg_cached) {
                    seg_cached = seg_p;
                    seg_c = seg_p < seg_n ? a.ent_scpu[seg_o + seg_p] : 0;
                    seg_m = seg_p < seg_n ? a.ent_smem[seg_o + seg_p] : 0;
                }
                res_fail = !(rcpu <= c_end + seg_c && rmem <= m_end + seg_m);
            }
            if (a.n_rg && (rflags & RT_RES) && is_ready && !res_fail) {
                // generic reservations (filter.go:86-91): count at the task's moment = end-of-batch count + what the node's commits
                // with index >= pos claimed of the kind. Counts only shrink inside a batch: the chain is walked only for a kind the
                // node lacks at the end.
                const u32 gset = cload(a.tg + gj);
                for (u32 g = cload(a.gs_off + gset); g < cload(a.gs_off + gset + 1) && !res_fail; ++g) {
                    const u32 row = cload(a.gs_row + g), kind = cload(a.rg_kind + row);
                    const int32_t want = cload(a.rg_val + row);
                    int32_t have = a.gcnt[(size_t)kind * a.gstride + n];
                    if (have >= want) continue;
                    for (int32_t ci = a.last[n]; ci >= 0; ci = a.log_prev[ci]) {
                        if (ci < pos) continue;
                        const u32 ts = a.tg[a.log_task[ci]];
                        for (u32 z = a.gs_off[ts]; z < a.gs_off[ts + 1]; ++z)
                            if (a.rg_kind[a.gs_row[z]] == kind) have += a.rg_val[a.gs_row[z]];
                    }
                    if (have < want) res_fail = true;
                }
            }
            if (!is_ready) ff = 0;
            else if (res_fail) ff = 1;
            else if (cls_plug && !(cload(a.plug + (size_t)cls_plug * a.n_words + w) & bit)) ff = 2;
            else if (cls_con && !(cload(a.con + (size_t)cls_con * a.n_words + w) & bit)) ff = 3;
            else if (cls_plat && !(cload(a.plat + (size_t)cls_plat * a.n_words + w) & bit)) ff = 4;
            else {
                bool port_busy = false;
                if (rflags & RT_PORTS) {
                    bool any = false;
                    for (u32 q = pp0; q < pp1; ++q) any = any || (a.portmap[(size_t)a.pset_ids[q] * a.n_words + w] & bit);
                    if (any) {
                        walk();
                        for (u32 q = pp0; q < pp1; ++q)
                            if ((a.portmap[(size_t)a.pset_ids[q] * a.n_words + w] & bit) && !((q - pp0 < 32) && (port_later >> (q - pp0) & 1u))) port_busy = true;
                    }
                }
                if (port_busy) ff = 5;
                else if (rflags & RT_MAXREP) {
                    u32 sv = 0;
                    for (u32 z = a.list_off[rsvc]; z < a.list_off[rsvc + 1]; ++z)
                        if (a.list_node[z] == n) { sv = a.list_svc[z]; break; }
                    if (!((u64)sv < rmaxrep)) {   // fails at the end of the batch: it may have passed at the task's moment
                        walk();
                        u32 at = sv - svc_later;
                        if (!((u64)at < rmaxrep)) ff = 6;
                    }
                }
            }
        }
        if (present && ff < 0 && a.csi_of) {   // VolumesFilter, the pipeline's last entry (scheduler.go:132)
            const u32 ck = cload(a.csi_of + gj);
            if (ck != 0xFFFFFFFFu && !(a.vrows[(size_t)ck * a.n_words + w] & bit)) ff = 7;
        }
        for (int f = 0; f < 8; ++f) {
            u64 bm = ballot64(ff == f);
            if (bm && (threadIdx.x & 63) == 0) atomicAdd(&cnt[e - e0][f], (u32)__popcll(bm));
        }
    }
    __syncthreads();
    for (u32 q = threadIdx.x; q < (e1 - e0) * 8; q += blockDim.x) {
        const u32 v = (&cnt[0][0])[q];
        if (v) atomicAdd(&a.hist[(size_t)cload(a.inf_task + e0 + (q >> 3)) * 8 + (q & 7)], v);
    }
}

// ---------------------------------------------------------------------------------------------
// The explain pass BY GROUP. Inside a batch residuals only shrink, host ports are only taken and a service's count on a node only
// grows: for every filter whose verdict depends on the task's moment (Resource, HostPort, MaxReplicas) a node has ONE commit from
// which on it fails a given request, and that commit is one of the node's own. Unplaceable tasks that share their predicate classes,
// reservations, port set and (with MaxReplicas) service form a group whose entries are sorted by moment; per (group, node) the
// kernel finds the thresholds in the node's commit segment, bisects them into the group's moments and books +1 / -1 into
// difference arrays over the entries: first-failing-filter counts for ALL entries of the group at the cost of one node pass per
// GROUP instead of one per task (cfg4 at 1M x 100k: 104k unplaceable tasks, most of them from services with MaxReplicas).
//   k_xg_nodes    grid (nodes / 256, groups)           difference arrays for every filter but MaxReplicas
//   k_xg_maxrep   grid (list chunks, MaxReplicas groups)  the service's own (node, count) list instead of the node set
//   k_xg_write    one workgroup per group              prefix sums -> hist[task][filter]
// Tasks with generic reservations keep the per-task pass (k_explain).
// ---------------------------------------------------------------------------------------------
#define XG_RES 1u      // ResourceFilter enabled
#define XG_PORTS 2u    // HostPortFilter enabled (pset)
#define XG_MAXREP 4u   // MaxReplicasFilter enabled (svc, maxrep)
struct XGroup {   // 64 B
    i64 cpu, mem;                      // the group's reservations
    u32 cls_con, cls_plat, cls_plug;   // its predicate class rows (0 = filter disabled)
    u32 flags;                         // XG_*
    u32 off, cnt;                      // its entries in xpos / xtask: [off, off + cnt), moments ascending
    u32 doff, pset;                    // its cnt + 1 difference slots start here; its host-port set
    u32 svc, pad;                      // batch-local service (XG_MAXREP)
    u64 maxrep;
};
static_assert(sizeof(XGroup) == 64, "XGroup layout");
#define XG_PLANES 6
#define XG_LDS_CNT 1024   // a group of up to this many entries is bisected and booked in LDS
#define XG_NPT 4          // nodes per thread of k_xg_nodes: one flush of the LDS planes per 1024 nodes
#define XG_NEVER 0xFFFFFFFFu
struct XGArgs {
    u32 n_nodes, n_words, n_groups, dstride;   // dstride: slots per filter plane of the difference arrays
    const XGroup* g;
    const u32* gorder;    // the groups ordered by reservation pair
    const uint2* chunks;  // [grid.y of k_xg_nodes] (first, count) in gorder: groups that share their reservations
    const u32* xpos;      // [entries] commits before the task (its moment)
    const u32* xtask;     // [entries] the task
    const RTask* rt;
    const u64* valid;
    const u64* ready;
    const u64* con;
    const u64* plat;
    const u64* plug;
    const i64* cpu;       // end-of-batch residuals
    const i64* mem;
    const u64* portmap;   // end-of-batch host ports
    const u32* pset_off;
    const u32* pset_ids;
    const u32* list_node; // end-of-batch per-service (node, count) lists
    const u32* list_svc;
    const u32* list_off;
    const u32* log_task;
    const u32* seg_off;
    const u32* seg_len;
    const u32* ent_ci;
    const i64* ent_scpu;
    const i64* ent_smem;
    const u32* mr;        // [grid.y of k_xg_maxrep] the groups with XG_MAXREP
    const uint2* wchunks; // [grid of k_xg_write] (group, first entry)
    int32_t* diff;        // [XG_PLANES][dstride]: plane f = hist column 1 + f (Resource, Plugin, Constraint, Platform, HostPort, MaxReplicas)
    u32* notready;        // one counter: nodes that fail the ReadyFilter (the same for every entry of every group)
    u32* hist;            // [T][8]
};

// Thresholds are commits: 0 = "from the start of the batch", XG_NEVER = "not by the end of the batch", else 1 + the index of the
// commit behind which the node fails. xg_index turns one into the first entry (of moments pos[0..cnt), ascending) that sees the failure.
template <class P> __device__ __forceinline__ u32 xg_index(P pos, u32 cnt, u32 thr) {
    if (thr == 0) return 0;
    if (thr == XG_NEVER) return cnt;
    const u32 ci = thr - 1u;   // first entry whose moment lies behind commit ci (pos > ci)
    u32 lo = 0, hi = cnt;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (pos[mid] > ci) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// ResourceFilter threshold of node n for a reservation pair (resources only shrink: a node that fits at the end fitted all along)
__device__ __forceinline__ u32 xg_res_thr(const XGArgs& a, i64 cpu, i64 mem, u32 n) {
    const i64 c_end = a.cpu[n], m_end = a.mem[n];
    if (cpu <= c_end && mem <= m_end) return XG_NEVER;
    const u32 so = a.seg_off[n], sl = a.seg_len[n];
    u32 k = 0;   // commits of this node that must have happened: the smallest k whose residual no longer fits (k == sl does not)
    for (; k < sl; ++k)
        if (!(cpu <= c_end + a.ent_scpu[so + k] && mem <= m_end + a.ent_smem[so + k])) break;
    return k == 0 ? 0u : a.ent_ci[so + k - 1] + 1u;
}

// HostPortFilter threshold: a port of the set that is taken at the end of the batch was taken by ONE commit of this node (the filter
// keeps a second one away) or before the batch; the earliest taking decides
__device__ __forceinline__ u32 xg_port_thr(const XGArgs& a, u32 pset, u32 n) {
    const u32 w = n >> 6;
    const u64 bit = 1ull << (n & 63);
    const u32 so = a.seg_off[n], sl = a.seg_len[n];
    u32 first = XG_NEVER;
    for (u32 q = a.pset_off[pset]; q < a.pset_off[pset + 1]; ++q) {
        const u32 port = a.pset_ids[q];
        if (!(a.portmap[(size_t)port * a.n_words + w] & bit)) continue;
        u32 tq = 0;
        for (u32 k = 0; k < sl && !tq; ++k) {
            const u32 ci = a.ent_ci[so + k];
            const RTask* tk = a.rt + a.log_task[ci];
            if (!(tk->flags & RT_PORTS)) continue;
            for (u32 z = a.pset_off[tk->pset]; z < a.pset_off[tk->pset + 1]; ++z)
                if (a.pset_ids[z] == port) tq = ci + 1u;
        }
        first = min(first, tq);
    }
    return first;
}

// first failing static filter behind the ResourceFilter, in pipeline order (pipeline.go:9-20): 0 = none, 1 Plugin, 2 Constraint, 3 Platform
__device__ __forceinline__ u32 xg_static(const XGArgs& a, const XGroup& G, u32 w, u64 bit) {
    if (G.cls_plug && !(cload(a.plug + (size_t)G.cls_plug * a.n_words + w) & bit)) return 1;
    if (G.cls_con && !(cload(a.con + (size_t)G.cls_con * a.n_words + w) & bit)) return 2;
    if (G.cls_plat && !(cload(a.plat + (size_t)G.cls_plat * a.n_words + w) & bit)) return 3;
    return 0;
}

// What a READY node shows a group's entries, filter by filter: entries [idx_r, cnt) see it short of resources, the ones before see
// the static verdict s, and when that passes entries [idx_p, idx_r) see a host port taken.
__global__ __launch_bounds__(256) void k_xg_nodes(XGArgs a) {
    __shared__ u32 sh_pos[XG_LDS_CNT];
    __shared__ int32_t sh_diff[5][XG_LDS_CNT + 1];
    const uint2 ch = a.chunks[blockIdx.y];
    const u32 tid = threadIdx.x;
    const XGroup G0 = a.g[a.gorder[ch.x]];
    u32 thr_r[XG_NPT];
    u32 rdy = 0;   // bit j: node j of this thread is ready
    for (int j = 0; j < XG_NPT; ++j) {
        const u32 n = (blockIdx.x * XG_NPT + j) * 256 + tid;
        const u32 w = n >> 6;
        const u64 bit = 1ull << (n & 63);
        const bool present = n < a.n_nodes && (cload(a.valid + w) & bit);
        const bool is_ready = present && (cload(a.ready + w) & bit);
        if (blockIdx.y == 0) {
            const u64 nr = ballot64(present && !is_ready);
            if (nr && (tid & 63) == 0) atomicAdd(a.notready, (u32)__popcll(nr));
        }
        thr_r[j] = XG_NEVER;
        if (is_ready) {
            rdy |= 1u << j;
            if (G0.flags & XG_RES) thr_r[j] = xg_res_thr(a, G0.cpu, G0.mem, n);   // the chunk's groups share the pair
        }
    }
    for (u32 i = 0; i < ch.y; ++i) {
        const XGroup G = a.g[a.gorder[ch.x + i]];
        const bool in_lds = G.cnt <= XG_LDS_CNT;
        const u32* gpos = a.xpos + G.off;
        if (in_lds) {
            for (u32 q = tid; q < G.cnt; q += 256) sh_pos[q] = gpos[q];
            for (int f = 0; f < 5; ++f)
                for (u32 q = tid; q <= G.cnt; q += 256) sh_diff[f][q] = 0;
            __syncthreads();
        }
        auto book = [&](u32 plane, u32 at, int32_t v) {
            if (in_lds) atomicAdd(&sh_diff[plane][at], v);
            else atomicAdd(a.diff + (size_t)plane * a.dstride + G.doff + at, v);
        };
        for (int j = 0; j < XG_NPT; ++j) {
            const u32 n = (blockIdx.x * XG_NPT + j) * 256 + tid;
            const u32 w = n >> 6;
            const u64 bit = 1ull << (n & 63);
            const bool is_ready = (rdy >> j) & 1u;
            u32 s = 0, idx_r = G.cnt, idx_p = G.cnt;
            if (is_ready) {
                s = xg_static(a, G, w, bit);
                idx_r = in_lds ? xg_index(sh_pos, G.cnt, thr_r[j]) : xg_index(gpos, G.cnt, thr_r[j]);
                if ((G.flags & XG_PORTS) && s == 0 && idx_r > 0) {
                    const u32 tp = xg_port_thr(a, G.pset, n);
                    idx_p = in_lds ? xg_index(sh_pos, G.cnt, tp) : xg_index(gpos, G.cnt, tp);
                }
            }
            // the common cases are booked once per wave (by lane 0 whatever its own node is): short from the start; a static
            // verdict from the start (its end at idx_r is booked by the node itself unless idx_r == cnt, a slot nobody reads)
            const u64 always = ballot64(is_ready && idx_r == 0);
            if (always && (tid & 63) == 0) book(0, 0, (int32_t)__popcll(always));
            for (u32 f = 1; f <= 3; ++f) {
                const u64 bm = ballot64(is_ready && idx_r > 0 && s == f);
                if (bm && (tid & 63) == 0) book(f, 0, (int32_t)__popcll(bm));
            }
            if (is_ready && idx_r > 0) {
                if (idx_r != G.cnt) {
                    book(0, idx_r, 1);
                    if (s) book(s, idx_r, -1);
                }
                if (!s && idx_p < idx_r) {   // entries [idx_p, idx_r) see the port taken
                    book(4, idx_p, 1);
                    book(4, idx_r, -1);
                }
            }
        }
        if (in_lds) {
            __syncthreads();
            for (int f = 0; f < 5; ++f)
                for (u32 q = tid; q <= G.cnt; q += 256) {
                    const int32_t v = sh_diff[f][q];
                    if (v) atomicAdd(a.diff + (size_t)f * a.dstride + G.doff + q, v);
                }
            __syncthreads();
        }
    }
}

// MaxReplicasFilter (the last of the pipeline): only a node ON the service's list can fail it, so the list is walked instead of the
// node set. The service's count on a node only grows inside a batch: from the commit that brought it to MaxReplicas on, the entries
// see this filter — as far as an earlier filter does not claim them (evaluated again for this one node).
__global__ __launch_bounds__(256) void k_xg_maxrep(XGArgs a) {
    const XGroup G = a.g[a.mr[blockIdx.y]];
    const u32* gpos = a.xpos + G.off;
    const u32 l0 = a.list_off[G.svc], l1 = a.list_off[G.svc + 1];
    for (u32 z = l0 + blockIdx.x * blockDim.x + threadIdx.x; z < l1; z += gridDim.x * blockDim.x) {
        const u32 n = a.list_node[z];
        if (n == LIST_EMPTY || n >= a.n_nodes) continue;
        const u32 sv = a.list_svc[z];
        if ((u64)sv < G.maxrep) continue;
        const u32 w = n >> 6;
        const u64 bit = 1ull << (n & 63);
        if (!(a.valid[w] & bit) || !(a.ready[w] & bit)) continue;
        if (xg_static(a, G, w, bit)) continue;
        u32 hi = (G.flags & XG_RES) ? xg_index(gpos, G.cnt, xg_res_thr(a, G.cpu, G.mem, n)) : G.cnt;
        if (hi && (G.flags & XG_PORTS)) hi = min(hi, xg_index(gpos, G.cnt, xg_port_thr(a, G.pset, n)));
        if (hi == 0) continue;
        // the service's counted commits on this node, latest first: the count at the end is sv; stepping back over `over` of them
        // brings it below MaxReplicas
        const u32 so = a.seg_off[n], sl = a.seg_len[n];
        const u64 over = (u64)sv - G.maxrep + 1u;   // >= 1
        u64 seen = 0;
        u32 thr = 0;                                // reached before the batch
        for (u32 k = sl; k-- > 0;) {
            const u32 ci = a.ent_ci[so + k];
            const RTask* tk = a.rt + a.log_task[ci];
            if (tk->svc != G.svc || (tk->flags & RT_UNCOUNTED)) continue;
            if (++seen == over) { thr = ci + 1u; break; }
        }
        const u32 idx_m = xg_index(gpos, G.cnt, thr);
        if (idx_m < hi) {
            atomicAdd(a.diff + (size_t)5 * a.dstride + G.doff + idx_m, 1);
            atomicAdd(a.diff + (size_t)5 * a.dstride + G.doff + hi, -1);
        }
    }
}

#define XG_WCH 2048   // entries of a group one workgroup of k_xg_write turns into counters (eight per thread)
__global__ __launch_bounds__(256) void k_xg_write(XGArgs a) {
    const uint2 ch = a.wchunks[blockIdx.x];   // (group, first entry): a long group (one service's 90k unplaceable tasks) is many chunks
    const XGroup G = a.g[ch.x];
    __shared__ int32_t part[XG_PLANES][256];
    __shared__ int32_t base[XG_PLANES];
    const u32 tid = threadIdx.x, s0 = ch.y;
    if (tid < XG_PLANES) base[tid] = 0;
    __syncthreads();
    if (s0) {   // the differences in front of the chunk
        int32_t acc[XG_PLANES] = {};
        for (u32 e = tid; e < s0; e += 256)
            for (int f = 0; f < XG_PLANES; ++f) acc[f] += a.diff[(size_t)f * a.dstride + G.doff + e];
        for (int f = 0; f < XG_PLANES; ++f)
            if (acc[f]) atomicAdd(&base[f], acc[f]);
    }
    const u32 e0 = min(s0 + tid * 8u, G.cnt), e1 = min(min(e0 + 8u, s0 + (u32)XG_WCH), G.cnt);
    int32_t sum[XG_PLANES] = {};
    for (u32 e = e0; e < e1; ++e)
        for (int f = 0; f < XG_PLANES; ++f) sum[f] += a.diff[(size_t)f * a.dstride + G.doff + e];
    for (int f = 0; f < XG_PLANES; ++f) part[f][tid] = sum[f];
    __syncthreads();
    int32_t run[XG_PLANES];
    for (int f = 0; f < XG_PLANES; ++f) run[f] = base[f];
    for (u32 t = 0; t < tid; ++t)
        for (int f = 0; f < XG_PLANES; ++f) run[f] += part[f][t];
    const u32 nr = a.notready[0];
    for (u32 e = e0; e < e1; ++e) {
        u32* h = a.hist + (size_t)a.xtask[G.off + e] * 8;
        for (int f = 0; f < XG_PLANES; ++f) {
            run[f] += a.diff[(size_t)f * a.dstride + G.doff + e];
            h[1 + f] = (u32)run[f];
        }
        h[0] = nr;
    }
}

// the Explain rows of the unplaceable tasks, gathered for one short copy to the host (hist is [T][8], mostly zeros)
__global__ __launch_bounds__(256) void k_gather_rows(const u32* idx, const u32* src, u32* dst, u32 n) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n * 8u) dst[g] = src[(size_t)idx[g >> 3] * 8u + (g & 7u)];
}

// ---------------------------------------------------------------------------------------------
// k_commit — NodeInfo.addTask / removeTask arithmetic for placements decided outside the engine
// (nodeinfo.go:66-154). Several placements may hit one node: integer atomics commute.
// ---------------------------------------------------------------------------------------------
// Residuals in the batch's resource units for k_resolve5: floor division, so that need <= residual <=> need/unit <= q for
// every need that is a multiple of the unit (a negative residual fits nothing, not even a zero reservation: filter.go:78-84).
__global__ void k_units(u32 n, const i64* __restrict__ cpu, const i64* __restrict__ mem, i64 uc, i64 um, int32_t* __restrict__ q) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto fd = [](i64 a, i64 b) {
        i64 d = a / b;
        if (a % b != 0 && a < 0) --d;
        return d < -(1ll << 30) ? -(1ll << 30) : d > (1ll << 30) ? (1ll << 30) : d;   // present nodes are range-checked by the host
    };
    q[2 * i] = (int32_t)fd(cpu[i], uc);
    q[2 * i + 1] = (int32_t)fd(mem[i], um);
}

// k_scatter_rows — the rows swp_node_update_dynamic changed (flush_nodes): flags word, residuals, task count of each
struct DevRow {
    u32 node, flags, total, pad;
    i64 cpu, mem;
};
static_assert(sizeof(DevRow) == 32, "DevRow layout");
__global__ void k_scatter_rows(u32 n, const DevRow* __restrict__ r, u32* flags, i64* cpu, i64* mem, u32* total) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevRow x = r[i];
    flags[x.node] = x.flags;
    cpu[x.node] = x.cpu;
    mem[x.node] = x.mem;
    total[x.node] = x.total;
}
struct DevPlacement { u32 node; u32 counted; i64 cpu, mem; };

__global__ void k_commit(u32 n, const DevPlacement* __restrict__ p, int add, i64* cpu, i64* mem, u32* total) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DevPlacement q = p[i];
    i64 dc = add ? -q.cpu : q.cpu, dm = add ? -q.mem : q.mem;
    atomicAdd(reinterpret_cast<u64*>(cpu + q.node), (u64)dc);
    atomicAdd(reinterpret_cast<u64*>(mem + q.node), (u64)dm);
    if (q.counted) atomicAdd(total + q.node, add ? 1u : 0xFFFFFFFFu);
}

// ---------------------------------------------------------------------------------------------
// k_enforce — constraintenforcer.rejectNoncompliantTasks (constraint_enforcer.go:65-196), one thread per node:
// the node's tasks in store order, constraint verdicts from the class bitmaps (the same k_constraint_classes the
// scheduler's ConstraintFilter uses), reservations accounted sequentially against Description.Resources.
// ---------------------------------------------------------------------------------------------
struct EnfNode { u32 node, first, count, pad; i64 cpu, mem; };
struct EnfTask { i64 cpu, mem; u32 cls_con, flags, desired, state; };
static_assert(sizeof(EnfNode) == 32 && sizeof(EnfTask) == 32, "enforcer record layout");
#define TASK_STATE_ASSIGNED 192u
#define TASK_STATE_COMPLETE 576u

__global__ __launch_bounds__(256) void k_enforce(u32 n_enf, u32 n_words, const EnfNode* __restrict__ nodes, const EnfTask* __restrict__ tasks,
                                                 const u64* __restrict__ con, unsigned char* __restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_enf) return;
    const EnfNode nd = nodes[i];
    const u32 w = nd.node >> 6;
    const u64 bit = 1ull << (nd.node & 63);
    i64 cpu = nd.cpu, mem = nd.mem;
    for (u32 t = nd.first; t < nd.first + nd.count; ++t) {
        const EnfTask tk = tasks[t];
        unsigned char rej = 0;
        if (tk.desired < TASK_STATE_ASSIGNED || tk.desired > TASK_STATE_COMPLETE) { out[t] = 0; continue; }   // :118-120
        if (tk.state >= TASK_STATE_COMPLETE) { out[t] = 0; continue; }                                       // :124-126
        if (tk.cls_con && !(con[(size_t)tk.cls_con * n_words + w] & bit)) rej = 1;                            // :162-168
        else if (tk.flags & 1u) {                                                                            // :172-184
            if (tk.mem > mem) rej = 1;
            else if (tk.cpu > cpu) rej = 1;
            else { mem -= tk.mem; cpu -= tk.cpu; }
        }
        out[t] = rej;
    }
}

// ---------------------------------------------------------------------------------------------
// k_check_pair — Pipeline.Process on one (task,node) pair (taskFitNode, scheduler.go:646-654)
// ---------------------------------------------------------------------------------------------
struct CheckArgs {
    u32 node, n_words;
    RTask rt;
    const u64* valid;
    const u64* ready;
    const u64* con;
    const u64* plat;
    const u64* plug;
    const i64* cpu;
    const i64* mem;
    u32 port_busy;    // host-evaluated (port sets live on the host between batches)
    u32 svc_count;    // host-evaluated ActiveTasksCountByService[service]
    int32_t* out;
    u32 n_gen, gstride;        // the task's generic reservations (at most 8) against the device's counts
    u32 gkind[8];
    int32_t gval[8];
    const int32_t* gcnt;
};
__global__ void k_check_pair(CheckArgs a) {
    if (threadIdx.x != 0) return;
    u32 n = a.node, w = n >> 6;
    u64 bit = 1ull << (n & 63);
    int ff = -1;
    if (!(a.valid[w] & bit)) ff = -2;
    else if (!(a.ready[w] & bit)) ff = 0;
    else if ((a.rt.flags & RT_RES) && !(a.rt.cpu <= a.cpu[n] && a.rt.mem <= a.mem[n])) ff = 1;
    else if ((a.rt.flags & RT_RES) && [&] {
                 for (u32 g = 0; g < a.n_gen; ++g)
                     if (a.gcnt[(size_t)a.gkind[g] * a.gstride + n] < a.gval[g]) return true;   // HasEnough, validate.go:24-52
                 return false;
             }())
        ff = 1;
    else if (a.rt.cls_plug && !(a.plug[(size_t)a.rt.cls_plug * a.n_words + w] & bit)) ff = 2;
    else if (a.rt.cls_con && !(a.con[(size_t)a.rt.cls_con * a.n_words + w] & bit)) ff = 3;
    else if (a.rt.cls_plat && !(a.plat[(size_t)a.rt.cls_plat * a.n_words + w] & bit)) ff = 4;
    else if ((a.rt.flags & RT_PORTS) && a.port_busy) ff = 5;
    else if ((a.rt.flags & RT_MAXREP) && !((u64)a.svc_count < a.rt.maxrep)) ff = 6;
    *a.out = ff;
}

}  // namespace swpdev
