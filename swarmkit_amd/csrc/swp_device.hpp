// swp_device.hpp — CDNA4 (gfx950) kernels of the batch task-placement engine.
//
// Data layout in HBM (all struct-of-arrays, node index = canonical scan order):
//   cpu[n], mem[n] : int64   AvailableResources (nodeinfo.go:35)
//   total[n]       : uint32  ActiveTasksCount
//   flags[n]       : uint32  SWP_NODE_* | DEV_VALID
//   attr[c][n]     : uint32  folded-string ids per constraint column (id, hostname, os, arch, labels…)
//   bitmaps        : uint64 words, bit i of word w = node 64*w+i  ("a wave's ballot IS a word")
//     ready[w], con[class][w], plat[class][w], plug[class][w], sc[static class][w]
//     F[task][w]   feasibility of (task,node) against a snapshot — superset of the truth, see below
//     X[svc][w]    nodes that are NOT "plain" for a service (svcCount>0 or recent failures ≥5)
//
// Exactness argument (why a parallel scan + one sequential resolver reproduces the reference's
// strictly sequential tick, scheduler.go:464-469):
//   inside one batch nothing is ever freed, so per (task,node) feasibility only goes 1→0 and a
//   node's score (max(fail,4), svcCount, total, index) only grows. Hence (a) a feasibility bit
//   computed against ANY earlier state is a superset of the current truth and only nodes touched
//   since need a re-check; (b) the resolver, which owns the live per-node level (= total) in LDS
//   bit-planes, finds argmin(level, index) over F & ~X word-parallel — the same node the
//   reference's heap of size 1 keeps (nodeset.go:111-120: a later equal node never displaces it).
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

// ---------------------------------------------------------------------------------------------
// k_scan — the tasks × nodes feasibility grid.
//   One wave owns WPW consecutive node words (WPW*64 node rows held in VGPRs for the whole task
//   chunk: each node row is read from HBM once per chunk, coalesced 8 B/lane) and walks a chunk
//   of tasks. Per (task, word): the resource compare is a v_cmp whose lane mask IS the output
//   word; the static filters arrive as one precomputed class word through the scalar cache.
//   Output words are parked one per lane so that one store instruction writes
//   8 tasks × 64 B fully-covered segments.
//   ResourceFilter.Check filter.go:77-84 (int64 signed compares), HostPortFilter.Check :336-347.
// ---------------------------------------------------------------------------------------------
struct ScanArgs {
    u32 n_nodes, n_words;
    u32 j0, count;          // window of tasks
    const i64* cpu;
    const i64* mem;
    const RTask* rt;        // whole batch
    const u64* sc;          // [n_sc][n_words]
    const u64* portmap;     // [n_ports][n_words]
    const u32* pset_off;
    const u32* pset_ids;
    u64* F;                 // [count][n_words]
};

#define SCAN_WPW 8
#define SCAN_TCH 64

__global__ __launch_bounds__(64) void k_scan(ScanArgs a) {
    const u32 lane = threadIdx.x;
    const u32 w0 = blockIdx.x * SCAN_WPW;
    const u32 t0 = blockIdx.y * SCAN_TCH;
    const u32 t1 = min(t0 + (u32)SCAN_TCH, a.count);
    i64 ncpu[SCAN_WPW], nmem[SCAN_WPW];
#pragma unroll
    for (int k = 0; k < SCAN_WPW; ++k) {
        u32 n = (w0 + k) * 64 + lane;
        bool in = n < a.n_nodes;
        ncpu[k] = in ? a.cpu[n] : INT64_MIN;
        nmem[k] = in ? a.mem[n] : INT64_MIN;
    }
    for (u32 t = t0; t < t1; t += 8) {
        u32 acc_lo = 0, acc_hi = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u32 tt = t + u;
            if (tt < t1) {   // uniform
                const RTask* r = a.rt + a.j0 + tt;
                i64 rc = cload(&r->cpu), rm = cload(&r->mem);
                u32 fl = cload(&r->flags), scid = cload(&r->sc);
                const u64* scrow = a.sc + (size_t)scid * a.n_words;
#pragma unroll
                for (int k = 0; k < SCAN_WPW; ++k) {
                    u64 word = 0;
                    if (w0 + k < a.n_words) {   // uniform
                        u64 fit = ballot64(rc <= ncpu[k] && rm <= nmem[k]);
                        if (!(fl & RT_RES)) fit = ~0ull;
                        word = fit & cload(scrow + w0 + k);
                        if (fl & RT_PORTS) {
                            u32 ps = cload(&r->pset);
                            u32 p0 = cload(a.pset_off + ps), p1 = cload(a.pset_off + ps + 1);
                            for (u32 p = p0; p < p1; ++p) word &= ~cload(a.portmap + (size_t)cload(a.pset_ids + p) * a.n_words + w0 + k);
                        }
                    }
                    // park the (wave-uniform) word in lane u*8+k — plain select: hipcc schedules and pads it
                    const bool mine = lane == (u32)(u * 8 + k);
                    acc_lo = mine ? (u32)word : acc_lo;
                    acc_hi = mine ? (u32)(word >> 32) : acc_hi;
                }
            }
        }
        u32 tt = t + (lane >> 3), w = w0 + (lane & 7);
        if (tt < t1 && w < a.n_words) a.F[(size_t)tt * a.n_words + w] = ((u64)acc_hi << 32) | acc_lo;
    }
}

// ---------------------------------------------------------------------------------------------
// k_resolve — the sequential part of the tick, ONE workgroup.
//   Thread `tid` owns node words {tid + k*B}: every mutable per-node quantity (cpu, mem, total,
//   last commit, X bit, port bits) is only ever read or written by its word's owner thread, so
//   program order is the only ordering the global-memory state needs.
//   LDS: level bit-planes planes[b][w] (bit i = bit b of (total[64w+i] - base)) and the
//   touched-since-scan bitmap. argmin(level, index) over a candidate word is the classic
//   bit-sliced minimum: NB AND/ANDN steps, no data-dependent loop.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        u64 o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, (u32)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, (u32)__shfl_xor((int)v, off, 64));
    return v;
}

// Workgroup barrier that orders LDS only: the waves' outstanding global loads (the next task's prefetched rows) and
// fire-and-forget stores/atomics stay in flight. __syncthreads() would drain them (vmcnt(0)) — ~2 us per task.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ u32 wave_min_u32_dpp(u32 v);

template <int KMAX>
struct Resolver {
    const ResolveArgs& a;
    u64* planes;      // [nb_alloc][n_words]
    u64* touched;     // [n_words]
    u64* red;         // [2][16]
    u32* sh;          // small shared scalars
    u32 tid, B, nw, lane, wave;
    u32 NB, base;
    u32 par;

    enum { SH_OK = 0, SH_REBASE = 1, SH_ENTRY = 2, SH_ERR = 3, SH_MIN = 4, SH_MAX = 5 };

    __device__ Resolver(const ResolveArgs& args, u64* lds) : a(args) {
        tid = threadIdx.x;
        B = blockDim.x;
        nw = B >> 6;
        lane = tid & 63;
        wave = tid >> 6;
        planes = lds;
        touched = planes + (size_t)a.nb_alloc * a.n_words;
        red = touched + a.n_words;
        sh = reinterpret_cast<u32*>(red + 2 * 16);
        par = 0;
        NB = 1;
        base = 0;
    }

    __device__ u64 block_min(u64 v) {
        u64 wv = wave_min_u64(v);
        if (lane == 0) red[par * 16 + wave] = wv;
        lds_barrier();
        u64 g = red[par * 16];
        for (u32 i = 1; i < nw; ++i) {
            u64 o = red[par * 16 + i];
            g = o < g ? o : g;
        }
        par ^= 1;
        return g;
    }

    // the same for a 32-bit key: six DPP steps instead of twelve ds_bpermute round trips
    __device__ u32 block_min32(u32 v) {
        u32 wv = wave_min_u32_dpp(v);
        if (lane == 0) reinterpret_cast<u32*>(red)[par * 32 + wave] = wv;
        lds_barrier();
        u32 g = reinterpret_cast<u32*>(red)[par * 32];
        for (u32 i = 1; i < nw; ++i) g = min(g, reinterpret_cast<u32*>(red)[par * 32 + i]);
        par ^= 1;
        return g;
    }

    // (re)build the level planes from total[] — called at window start and on level overflow.
    // Returns false (uniformly) when the level span does not fit nb_alloc planes.
    __device__ bool build_planes() {
        u32 lo = 0xFFFFFFFFu, hi = 0;
        for (int k = 0; k < KMAX; ++k) {
            u32 w = tid + k * B;
            if (w >= a.n_words) break;
            u64 vm = a.valid[w];
            while (vm) {
                int i = __ffsll((long long)vm) - 1;
                vm &= vm - 1;
                u32 t = __hip_atomic_load(&a.total[w * 64 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo = min(lo, t);
                hi = max(hi, t);
            }
        }
        lo = wave_min_u32(lo);
        hi = wave_max_u32(hi);
        __syncthreads();   // previous users of sh[] / red[] are done
        if (lane == 0) {
            reinterpret_cast<u32*>(red)[wave] = lo;
            reinterpret_cast<u32*>(red)[16 + wave] = hi;
        }
        __syncthreads();
        for (u32 i = 0; i < nw; ++i) {
            lo = min(lo, reinterpret_cast<u32*>(red)[i]);
            hi = max(hi, reinterpret_cast<u32*>(red)[16 + i]);
        }
        __syncthreads();
        if (lo == 0xFFFFFFFFu) { lo = 0; hi = 0; }   // no valid node at all
        u32 span = hi - lo;
        u32 need = 32 - __clz(span | 1u);            // bits to hold span (≥1)
        const u32 idx_bits = 32 - __clz((a.n_words * 64) | 1u);
        const u32 capb = min(a.nb_alloc, 32u - idx_bits);   // (level << idx_bits | node) must fit 32 bits (block_min32)
        if (need > capb) return false;
        base = lo;
        NB = min(capb, need + 1);                    // one spare bit: room to double before the next rebase
        for (int k = 0; k < KMAX; ++k) {
            u32 w = tid + k * B;
            if (w >= a.n_words) break;
            u64 pl[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) pl[b] = 0;
            u64 vm = a.valid[w];
            while (vm) {
                int i = __ffsll((long long)vm) - 1;
                vm &= vm - 1;
                u32 lvl = __hip_atomic_load(&a.total[w * 64 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
#pragma unroll
                for (int b = 0; b < 16; ++b) pl[b] |= (u64)((lvl >> b) & 1u) << i;
            }
#pragma unroll
            for (int b = 0; b < 16; ++b)
                if ((u32)b < NB) planes[(size_t)b * a.n_words + w] = pl[b];
        }
        __syncthreads();
        return true;
    }

    // level of one node read back from the planes (owner thread)
    __device__ u32 level_of(u32 w, u64 bit) {
        u32 lvl = 0;
        for (u32 b = 0; b < NB; ++b)
            if (planes[(size_t)b * a.n_words + w] & bit) lvl |= 1u << b;
        return lvl;
    }

    // ripple-carry +1 on one node's level; returns false when the level was saturated
    __device__ bool bump_level(u32 w, u64 bit) {
        for (u32 b = 0; b < NB; ++b) {
            u64 p = planes[(size_t)b * a.n_words + w];
            planes[(size_t)b * a.n_words + w] = p ^ bit;
            if (!(p & bit)) return true;
        }
        return false;   // wrapped to 0: planes for this node are wrong until the rebuild
    }
};

template <int KMAX>
__global__ __launch_bounds__(1024) void k_resolve(ResolveArgs a) {
    extern __shared__ u64 lds[];
    Resolver<KMAX> R(a, lds);
    const u32 tid = R.tid, B = R.B;
    u32* sh = R.sh;
    typedef Resolver<KMAX> RS;

    u32 ncommit = a.ctl->ncommit, ninf = a.ctl->ninf;
    u64 st_retries = 0, st_slow = 0, st_rebase = 0;
    if (a.ctl->error != ERR_NONE) return;

    for (int k = 0; k < KMAX; ++k) {
        u32 w = tid + k * B;
        if (w < a.n_words) R.touched[w] = 0;
    }
    if (tid == 0) { sh[RS::SH_OK] = 0; sh[RS::SH_REBASE] = 0; sh[RS::SH_ERR] = 0; }
    bool fits = R.build_planes();
    if (!fits) {
        if (tid == 0) a.ctl->error = ERR_LEVEL_RANGE;
        return;
    }

    const u32 idx_bits = 32 - __clz((a.n_words * 64) | 1u), idx_mask = (1u << idx_bits) - 1u;
    u32 pend_idx = 0xFFFFFFFFu;   // this thread's last commit whose chain link (log_prev) is still in flight
    int32_t pend_prev = -1;
    // software prefetch of the next task's rows
    RTask rt_next = a.rt[a.j0];
    u64 Fnx[KMAX], Xnx[KMAX];
    for (int k = 0; k < KMAX; ++k) {
        u32 w = tid + k * B;
        bool in = w < a.n_words;
        Fnx[k] = in ? a.F[w] : 0;
        Xnx[k] = in ? a.X[(size_t)rt_next.svc * a.n_words + w] : 0;
    }

    for (u32 j = 0; j < a.count; ++j) {
        const RTask rt = rt_next;
        u64 Xc[KMAX], mk[KMAX];
        for (int k = 0; k < KMAX; ++k) {
            Xc[k] = Xnx[k];
            mk[k] = Fnx[k] & ~Xnx[k];
        }
        const bool have_next = j + 1 < a.count;
        if (have_next) {
            rt_next = a.rt[a.j0 + j + 1];
            for (int k = 0; k < KMAX; ++k) {
                u32 w = tid + k * B;
                bool in = w < a.n_words;
                Fnx[k] = in ? a.F[(size_t)(j + 1) * a.n_words + w] : 0;
                Xnx[k] = in ? a.X[(size_t)rt_next.svc * a.n_words + w] : 0;
            }
        }
        const bool counted = !(rt.flags & RT_UNCOUNTED);
        const u32 gj = a.j0 + j;
        bool placed = false;

        // ---------------- plain path: nodes with svcCount == 0 and < MAX_FAILURES failures ----------------
        for (;;) {
            u32 best = 0xFFFFFFFFu;
            for (int k = 0; k < KMAX; ++k) {
                u64 m = mk[k];
                if (m) {
                    u32 w = tid + k * B;
                    u32 lvl = 0;
                    for (int b = (int)R.NB - 1; b >= 0; --b) {
                        u64 t = m & ~R.planes[(size_t)b * a.n_words + w];
                        if (t) m = t;
                        else lvl |= 1u << b;
                    }
                    u32 cand = (lvl << idx_bits) | (w * 64 + (u32)(__ffsll((long long)m) - 1));
                    best = min(best, cand);
                }
            }
            u32 g = R.block_min32(best);
            if (g == 0xFFFFFFFFu) break;
            u32 n = g & idx_mask, w = n >> 6;
            u64 bit = 1ull << (n & 63);
            bool owner = (w % B) == tid;
            int ko = (int)(w / B);
            if (owner) {
                bool ok = true;
                if (R.touched[w] & bit) {   // F may be stale for this node: re-check the dynamic filters
                    if (rt.flags & RT_RES)
                        ok = (rt.cpu <= __hip_atomic_load(&a.cpu[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) &&
                             (rt.mem <= __hip_atomic_load(&a.mem[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (ok && (rt.flags & RT_PORTS)) {
                        for (u32 p = a.pset_off[rt.pset]; p < a.pset_off[rt.pset + 1]; ++p)
                            if (a.portmap[(size_t)a.pset_ids[p] * a.n_words + w] & bit) ok = false;
                    }
                }
                if (ok) {
                    // residual update == NodeInfo.addTask (nodeinfo.go:108-154); no-return atomics: nothing to wait for
                    if (rt.cpu) atomicAdd(reinterpret_cast<u64*>(a.cpu + n), (u64)(-rt.cpu));
                    if (rt.mem) atomicAdd(reinterpret_cast<u64*>(a.mem + n), (u64)(-rt.mem));
                    R.touched[w] |= bit;
                    if (rt.flags & RT_PORTS)
                        for (u32 p = a.pset_off[rt.pset]; p < a.pset_off[rt.pset + 1]; ++p)
                            a.portmap[(size_t)a.pset_ids[p] * a.n_words + w] |= bit;
                    if (counted) {
                        atomicAdd(a.total + n, 1u);
                        if (!R.bump_level(w, bit)) sh[RS::SH_REBASE] = 1;
                        u64 nx = 0;
                        for (int k = 0; k < KMAX; ++k)
                            if (k == ko) nx = Xc[k] | bit;
                        a.X[(size_t)rt.svc * a.n_words + w] = nx;
                        if (have_next && rt_next.svc == rt.svc)
                            for (int k = 0; k < KMAX; ++k)
                                if (k == ko) Xnx[k] |= bit;
                        a.list_node[rt.slot] = n;
                        a.list_svc[rt.slot] = 1;
                        a.list_fail[rt.slot] = 0;
                    }
                    a.log_node[ncommit] = n;
                    a.log_task[ncommit] = gj;
                    // chain link: the exchange's result is stored when this thread commits next (or at the end), so that
                    // its latency never sits between two barriers
                    if (pend_idx != 0xFFFFFFFFu) a.log_prev[pend_idx] = pend_prev;
                    pend_prev = (int32_t)atomicExch(reinterpret_cast<u32*>(&a.last[n]), ncommit);
                    pend_idx = ncommit;
                    a.out_node[gj] = (int32_t)n;
                }
                sh[RS::SH_OK] = ok ? 1u : 0u;
            }
            lds_barrier();
            bool ok = sh[RS::SH_OK] != 0;
            if (ok) { placed = true; break; }
            if (owner)
                for (int k = 0; k < KMAX; ++k)
                    if (k == ko) mk[k] &= ~bit;
            ++st_retries;
        }

        // ---------------- slow path: the service's exception list (svcCount>0 or failures≥5) ----------------
        if (!placed) {
            u32 e0 = a.list_off[rt.svc], e1 = a.list_off[rt.svc + 1];
            u64 bhi = KEY_NONE, blo = KEY_NONE;
            u32 be = 0;
            __syncthreads();   // list entries / cpu / mem / total written by other threads become visible
            for (u32 e = e0 + tid; e < e1; e += B) {
                u32 n = __hip_atomic_load(&a.list_node[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n == LIST_EMPTY) continue;
                u32 w = n >> 6;
                u64 bit = 1ull << (n & 63);
                if (!(a.F[(size_t)j * a.n_words + w] & bit)) continue;
                if (rt.flags & RT_RES) {
                    i64 c = __hip_atomic_load(&a.cpu[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    i64 m = __hip_atomic_load(&a.mem[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(rt.cpu <= c && rt.mem <= m)) continue;
                }
                if (rt.flags & RT_PORTS) {
                    bool used = false;
                    for (u32 p = a.pset_off[rt.pset]; p < a.pset_off[rt.pset + 1]; ++p)
                        if (__hip_atomic_load(&a.portmap[(size_t)a.pset_ids[p] * a.n_words + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) used = true;
                    if (used) continue;
                }
                u32 sv = __hip_atomic_load(&a.list_svc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u32 fl = __hip_atomic_load(&a.list_fail[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((rt.flags & RT_MAXREP) && !((u64)sv < rt.maxrep)) continue;   // filter.go:373-375
                u32 fcl = fl >= MAX_FAILURES ? fl - (MAX_FAILURES - 1) : 0u;       // nodeLess, scheduler.go:708-735
                u32 tot = __hip_atomic_load(&a.total[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u64 hi = ((u64)fcl << 32) | sv, lo = ((u64)tot << 32) | n;
                if (hi < bhi || (hi == bhi && lo < blo)) { bhi = hi; blo = lo; be = e; }
            }
            u64 ghi = R.block_min(bhi);
            if (ghi != KEY_NONE) {
                u64 glo = R.block_min(bhi == ghi ? blo : KEY_NONE);
                if (bhi == ghi && blo == glo) sh[RS::SH_ENTRY] = be;
                __syncthreads();
                u32 e = sh[RS::SH_ENTRY];
                u32 n = (u32)glo, w = n >> 6;
                u64 bit = 1ull << (n & 63);
                if ((w % B) == tid) {
                    if (rt.cpu) atomicAdd(reinterpret_cast<u64*>(a.cpu + n), (u64)(-rt.cpu));
                    if (rt.mem) atomicAdd(reinterpret_cast<u64*>(a.mem + n), (u64)(-rt.mem));
                    R.touched[w] |= bit;
                    if (rt.flags & RT_PORTS)
                        for (u32 p = a.pset_off[rt.pset]; p < a.pset_off[rt.pset + 1]; ++p)
                            a.portmap[(size_t)a.pset_ids[p] * a.n_words + w] |= bit;
                    if (counted) {
                        atomicAdd(a.total + n, 1u);
                        if (!R.bump_level(w, bit)) sh[RS::SH_REBASE] = 1;
                        u32 sv = __hip_atomic_load(&a.list_svc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&a.list_svc[e], sv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    a.log_node[ncommit] = n;
                    a.log_task[ncommit] = gj;
                    if (pend_idx != 0xFFFFFFFFu) a.log_prev[pend_idx] = pend_prev;
                    pend_prev = (int32_t)atomicExch(reinterpret_cast<u32*>(&a.last[n]), ncommit);
                    pend_idx = ncommit;
                    a.out_node[gj] = (int32_t)n;
                }
                placed = true;
                ++st_slow;
                __syncthreads();
            }
        }

        if (placed) {
            ++ncommit;
            if (sh[RS::SH_REBASE]) {   // uniform: written before the barrier that ended the commit
                __syncthreads();
                if (tid == 0) sh[RS::SH_REBASE] = 0;
                ++st_rebase;
                if (!R.build_planes()) {
                    if (tid == 0) a.ctl->error = ERR_LEVEL_RANGE;
                    break;
                }
            }
        } else {
            if (tid == 0) {
                a.out_node[gj] = -1;
                a.inf_task[ninf] = gj;
                a.inf_pos[ninf] = ncommit;
            }
            ++ninf;
        }
    }
    if (pend_idx != 0xFFFFFFFFu) a.log_prev[pend_idx] = pend_prev;
    if (tid == 0) {
        a.ctl->ncommit = ncommit;
        a.ctl->ninf = ninf;
        a.ctl->verify_retries += st_retries;
        a.ctl->slow_tasks += st_slow;
        a.ctl->rebases += st_rebase;
    }
}

// ---------------------------------------------------------------------------------------------
// k_resolve1 — the same sequential pass as k_resolve for node sets that fit ONE wavefront
// (n_words ≤ 64*K): no workgroup barrier anywhere, so nothing ever forces vmcnt(0) and the global
// loads of the next tasks' rows stay in flight under the current task's work.
//   * lane l owns words {l + 64k}; F/X rows are prefetched D tasks ahead into a register ring
//     (static indexing: the task loop is unrolled by D);
//   * argmin key = (level << idx_bits) | node packed in 32 bits and reduced with 6 DPP steps;
//   * the commit is fire-and-forget: ds_xor on exactly the planes whose bit flips in level+1,
//     no-return global atomics for cpu/mem/total, plain stores for X / list / log; the one value that
//     must come back (the node's previous commit, for the explain chain) is consumed one commit later.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ u32 dpp_u32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// min over the 64 lanes of a wave; every lane returns the result
__device__ __forceinline__ u32 wave_min_u32_dpp(u32 v) {
    v = min(v, dpp_u32<0x111>(v));        // row_shr:1
    v = min(v, dpp_u32<0x112>(v));        // row_shr:2
    v = min(v, dpp_u32<0x114>(v));        // row_shr:4
    v = min(v, dpp_u32<0x118>(v));        // row_shr:8   → lane 15 of every row holds the row minimum
    v = min(v, dpp_u32<0x142, 0xa>(v));   // row_bcast:15 into rows 1,3
    v = min(v, dpp_u32<0x143, 0xc>(v));   // row_bcast:31 into rows 2,3 → lane 63 holds the wave minimum
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// NBR = level planes held in registers per owned word (levels 0 .. 2^NBR-1 above `base`).
#define R1_NBR 8

template <int K, int D>
__global__ __launch_bounds__(64) void k_resolve1(ResolveArgs a) {
    const u32 lane = threadIdx.x;
    const u32 Wn = a.n_words;
    if (a.ctl->error != ERR_NONE) return;
    u32 ncommit = a.ctl->ncommit, ninf = a.ctl->ninf;
    u32 st_retries = 0, st_slow = 0, st_rebase = 0;
    // last commit per node (head of the explain pass's per-node chain) lives in LDS for the whole
    // window: the hot loop must never consume a VMEM result younger than the D-deep prefetch ring —
    // vmcnt retires in order, so waiting on one recent load/atomic would drain the whole ring.
    extern __shared__ int32_t last_lds[];
    for (u32 n = lane; n < a.n_nodes; n += 64) last_lds[n] = a.last[n];

    const u32 idx_bits = 32 - __clz((Wn * 64) | 1u);   // node indices < 64*64*K
    const u32 idx_mask = (1u << idx_bits) - 1u;
    u32 NB = 1, base = 0;

    // Per-lane state, all in registers: lane l owns words {l + 64k}.
    u64 pl[R1_NBR][K];   // level bit-planes of the owned words
    u64 tch[K];          // nodes committed to since the scan (their F bits may be stale)
#pragma unroll
    for (int k = 0; k < K; ++k) tch[k] = 0;

    // (re)build the level planes from total[]; false when the level span needs more than R1_NBR planes.
    // The bit loop is the dynamic one; k and b stay fully unrolled so that pl[][] never leaves registers.
    auto build_planes = [&]() __attribute__((always_inline)) -> bool {
        u64 vm[K];
#pragma unroll
        for (int k = 0; k < K; ++k) vm[k] = (lane + 64 * k) < Wn ? a.valid[lane + 64 * k] : 0ull;
        u32 lo = 0xFFFFFFFFu, hi = 0;
        for (int i = 0; i < 64; ++i) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if ((vm[k] >> i) & 1ull) {
                    u32 t = __hip_atomic_load(&a.total[(lane + 64 * k) * 64 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lo = min(lo, t);
                    hi = max(hi, t);
                }
            }
        }
        lo = wave_min_u32(lo);
        hi = wave_max_u32(hi);
        if (lo == 0xFFFFFFFFu) { lo = 0; hi = 0; }
        u32 span = hi - lo;
        u32 need = 32 - __clz(span | 1u);
        u32 cap = min((u32)R1_NBR, 32u - idx_bits);   // the packed (level, node) key must fit 32 bits
        if (need > cap) return false;
        base = lo;
        NB = min(cap, need + 1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int b = 0; b < R1_NBR; ++b) pl[b][k] = 0;
        }
        for (int i = 0; i < 64; ++i) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                u32 lvl = 0;
                if ((vm[k] >> i) & 1ull)
                    lvl = __hip_atomic_load(&a.total[(lane + 64 * k) * 64 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
#pragma unroll
                for (int b = 0; b < R1_NBR; ++b) pl[b][k] |= (u64)((lvl >> b) & 1u) << i;
            }
        }
        return true;
    };
    if (!build_planes()) {
        if (lane == 0) { a.ctl->error = ERR_LEVEL_RANGE; a.ctl->resume = a.j0; }
        return;
    }

    // bit-sliced argmin(level, index) over the candidate words; 0xFFFFFFFF when there is none
    auto search = [&](const u64 (&mk)[K]) __attribute__((always_inline)) -> u32 {
        u64 m[K];
        u32 lv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { m[k] = mk[k]; lv[k] = 0; }
#pragma unroll
        for (int b = R1_NBR - 1; b >= 0; --b) {
            if ((u32)b < NB) {   // uniform
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
            u32 w = lane + 64 * k;
            u32 cand = (lv[k] << idx_bits) | (w * 64 + (u32)(__ffsll((long long)m[k]) - 1));
            best = min(best, m[k] ? cand : 0xFFFFFFFFu);
        }
        return wave_min_u32_dpp(best);
    };

    // level+1 for node (owner lane, word slot ko): flip exactly the planes whose bit changes
    auto bump = [&](bool owner, u32 ko, u64 bit, u32 lvl) __attribute__((always_inline)) {
        const u32 flip = lvl ^ (lvl + 1);
        const u64 xb = owner ? bit : 0ull;
#pragma unroll
        for (int b = 0; b < R1_NBR; ++b) {
            if (flip >> b & 1u) {   // uniform
#pragma unroll
                for (int k = 0; k < K; ++k) pl[b][k] ^= ((u32)k == ko) ? xb : 0ull;   // value select keeps pl[][] in registers
            }
        }
    };

    // register ring: slot u holds the rows of the task with (index ≡ u mod D)
    u64 Fr[D][K], Xr[D][K];
    u32 sv[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        bool have = (u32)u < a.count;
        sv[u] = have ? cload(&a.rt[a.j0 + u].svc) : 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u32 w = lane + 64 * k;
            bool in = have && w < Wn;
            Fr[u][k] = in ? a.F[(size_t)u * Wn + w] : 0;
            Xr[u][k] = in ? __hip_atomic_load(&a.X[(size_t)sv[u] * Wn + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
    }
    // service / flags of upcoming tasks are fetched one iteration before they are needed
    u32 sv_ahead = (u32)D < a.count ? cload(&a.rt[a.j0 + D].svc) : 0u;
    u32 fl_next = cload(&a.rt[a.j0].flags);

    u32 j = 0;          // next task (window-relative); its rows sit in ring slot j % D
    int ustart = 0;     // == j % D
    bool fatal = false;
    u64 gk[K], gXc[K];  // candidate / X words of a task handed to the generic path
#pragma unroll
    for (int k = 0; k < K; ++k) { gk[k] = 0; gXc[k] = 0; }
    u32 g_svc = 0;

    while (j < a.count && !fatal) {
        bool generic = false, want_rebase = false;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (u < ustart) continue;                               // uniform
            if (j >= a.count || generic || want_rebase) continue;   // uniform
            const u32 gj = a.j0 + j;
            const RTask* r = a.rt + gj;
            u64 mk[K], Xc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Xc[k] = Xr[u][k];
                mk[k] = Fr[u][k] & ~Xr[u][k];
            }
            const u32 rsvc = sv[u];
            const u32 rflags = fl_next;
            // refill this ring slot with task j+D; fetch flags of task j+1
            {
                const u32 jn = j + D;
                const bool have = jn < a.count;
                const u32 sn = sv_ahead;
                sv_ahead = (jn + 1 < a.count) ? cload(&a.rt[a.j0 + jn + 1].svc) : 0u;
                fl_next = (j + 1 < a.count) ? cload(&a.rt[gj + 1].flags) : 0u;
                sv[u] = sn;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    u32 w = lane + 64 * k;
                    const bool in = have && w < Wn;
                    Fr[u][k] = in ? a.F[(size_t)jn * Wn + w] : 0;
                    Xr[u][k] = in ? __hip_atomic_load(&a.X[(size_t)sn * Wn + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                }
            }
            const i64 rcpu = cload(&r->cpu), rmem = cload(&r->mem);
            const u32 rslot = cload(&r->slot);

            u32 g = 0xFFFFFFFFu;
            bool fast = !(rflags & (RT_PORTS | RT_UNCOUNTED));
            u32 n = 0, lvl = 0, w = 0, ko = 0;
            u64 bit = 0;
            bool owner = false;
            if (fast) {
                g = search(mk);
                fast = g != 0xFFFFFFFFu;
            }
            if (fast) {
                n = g & idx_mask;
                lvl = g >> idx_bits;
                w = n >> 6;
                ko = w >> 6;
                bit = 1ull << (n & 63);
                owner = (w & 63) == lane;
                u64 tsel = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) tsel = ((u32)k == ko) ? tch[k] : tsel;
                // stale-F hazard or level overflow → generic path
                if (ballot64(owner && (tsel & bit)) != 0 || lvl == (1u << NB) - 1u) fast = false;
            }
            if (!fast) {   // hand the task to the generic path below (one copy of the heavy code)
#pragma unroll
                for (int k = 0; k < K; ++k) { gk[k] = mk[k]; gXc[k] = Xc[k]; }
                g_svc = rsvc;
                generic = true;
                ustart = u;
                continue;
            }

            // ------------- lean commit == NodeInfo.addTask (nodeinfo.go:108-154), counted task, no host ports -------------
            bump(owner, ko, bit, lvl);
#pragma unroll
            for (int k = 0; k < K; ++k) tch[k] |= (owner && (u32)k == ko) ? bit : 0ull;
            if (owner) {
                if (rcpu) atomicAdd(reinterpret_cast<u64*>(a.cpu + n), (u64)(-rcpu));
                if (rmem) atomicAdd(reinterpret_cast<u64*>(a.mem + n), (u64)(-rmem));
                atomicAdd(a.total + n, 1u);
                u64 nx = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) nx = ((u32)k == ko) ? (Xc[k] | bit) : nx;
                __hip_atomic_store(&a.X[(size_t)rsvc * Wn + w], nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.list_node[rslot] = n;
                a.list_svc[rslot] = 1;
                a.list_fail[rslot] = 0;
                a.log_node[ncommit] = n;
                a.log_task[ncommit] = gj;
                a.log_prev[ncommit] = last_lds[n];
                last_lds[n] = (int32_t)ncommit;
                a.out_node[gj] = (int32_t)n;
            }
            // a prefetched X row of the same service must see this node as taken
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const u32 held = j - (u32)u + (u32)d + ((u32)d <= (u32)u ? (u32)D : 0u);   // task whose rows sit in slot d now
                if (sv[d] == rsvc && held < a.count) {   // uniform, rare
#pragma unroll
                    for (int k = 0; k < K; ++k) Xr[d][k] |= (owner && (u32)k == ko) ? bit : 0ull;
                }
            }
            ++ncommit;
            ++j;
        }
        if (!generic && !want_rebase) { ustart = 0; continue; }   // a full round of D tasks went through the fast path

        // =========================== generic path: one task, every feature ===========================
        {
            const u32 gj = a.j0 + j;
            const RTask* r = a.rt + gj;
            const u32 rflags = r->flags, rsvc = g_svc, rslot = r->slot, rpset = r->pset;
            const i64 rcpu = r->cpu, rmem = r->mem;
            const bool counted = !(rflags & RT_UNCOUNTED);
            bool placed = false, via_list = false;
            u32 n = 0, lvl = 0, entry = 0;
            for (;;) {   // plain candidates, re-checking nodes committed to since the scan
                u32 g = search(gk);
                if (g == 0xFFFFFFFFu) break;
                n = g & idx_mask;
                lvl = g >> idx_bits;
                const u32 w = n >> 6, ko = w >> 6;
                const u64 bit = 1ull << (n & 63);
                const bool owner = (w & 63) == lane;
                u64 tsel = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) tsel = ((u32)k == ko) ? tch[k] : tsel;
                bool ok = true;
                if (ballot64(owner && (tsel & bit)) != 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                    if (rflags & RT_RES) {
                        i64 c = __hip_atomic_load(&a.cpu[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        i64 m = __hip_atomic_load(&a.mem[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (rcpu <= c) && (rmem <= m);
                    }
                    if (ok && (rflags & RT_PORTS)) {
                        for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p)
                            if (__hip_atomic_load(&a.portmap[(size_t)a.pset_ids[p] * Wn + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) ok = false;
                    }
                }
                if (ok) { placed = true; break; }
#pragma unroll
                for (int k = 0; k < K; ++k) gk[k] &= ~((owner && (u32)k == ko) ? bit : 0ull);
                ++st_retries;
            }
            if (!placed) {
                // exception list of the service: nodes with svcCount>0 or ≥5 recent failures
                const u64 maxrep = r->maxrep;
                const u32 e0 = a.list_off[rsvc], e1 = a.list_off[rsvc + 1];
                u64 bhi = KEY_NONE, blo = KEY_NONE;
                u32 be = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                for (u32 e = e0 + lane; e < e1; e += 64) {
                    u32 nn = __hip_atomic_load(&a.list_node[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (nn == LIST_EMPTY) continue;
                    u32 w = nn >> 6;
                    u64 bit = 1ull << (nn & 63);
                    if (!(a.F[(size_t)j * Wn + w] & bit)) continue;
                    if (rflags & RT_RES) {
                        i64 c = __hip_atomic_load(&a.cpu[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        i64 m = __hip_atomic_load(&a.mem[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!(rcpu <= c && rmem <= m)) continue;
                    }
                    if (rflags & RT_PORTS) {
                        bool used = false;
                        for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p)
                            if (__hip_atomic_load(&a.portmap[(size_t)a.pset_ids[p] * Wn + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) used = true;
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
                if (ghi != KEY_NONE) {
                    u64 glo = wave_min_u64(bhi == ghi ? blo : KEY_NONE);
                    u64 who = ballot64(bhi == ghi && blo == glo);
                    entry = (u32)__builtin_amdgcn_readlane((int)be, __ffsll((long long)who) - 1);
                    n = (u32)glo;
                    lvl = (u32)(glo >> 32) - base;
                    placed = true;
                    via_list = true;
                    ++st_slow;
                }
            }
            if (placed) {
                const u32 w = n >> 6, ko = w >> 6;
                const u64 bit = 1ull << (n & 63);
                const bool owner = (w & 63) == lane;
                if (counted) {
                    if (lvl == (1u << NB) - 1u) want_rebase = true;   // level+1 leaves the planes: rebuild from total[]
                    else bump(owner, ko, bit, lvl);
                }
#pragma unroll
                for (int k = 0; k < K; ++k) tch[k] |= (owner && (u32)k == ko) ? bit : 0ull;
                if (owner) {
                    if (rcpu) atomicAdd(reinterpret_cast<u64*>(a.cpu + n), (u64)(-rcpu));
                    if (rmem) atomicAdd(reinterpret_cast<u64*>(a.mem + n), (u64)(-rmem));
                    if (rflags & RT_PORTS)
                        for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p) atomicOr(&a.portmap[(size_t)a.pset_ids[p] * Wn + w], bit);
                    if (counted) {
                        atomicAdd(a.total + n, 1u);
                        if (via_list) atomicAdd(a.list_svc + entry, 1u);
                        else {
                            u64 nx = 0;
#pragma unroll
                            for (int k = 0; k < K; ++k) nx = ((u32)k == ko) ? (gXc[k] | bit) : nx;
                            __hip_atomic_store(&a.X[(size_t)rsvc * Wn + w], nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            a.list_node[rslot] = n;
                            a.list_svc[rslot] = 1;
                            a.list_fail[rslot] = 0;
                        }
                    }
                    a.log_node[ncommit] = n;
                    a.log_task[ncommit] = gj;
                    a.log_prev[ncommit] = last_lds[n];
                    last_lds[n] = (int32_t)ncommit;
                    a.out_node[gj] = (int32_t)n;
                }
                if (counted && !via_list) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        // slot d holds task j - ustart + d (+D when it was already consumed and refilled this round)
                        const u32 held = j - (u32)ustart + (u32)d + ((int)d <= ustart ? (u32)D : 0u);
                        if (sv[d] == rsvc && held < a.count) {
#pragma unroll
                            for (int k = 0; k < K; ++k) Xr[d][k] |= (owner && (u32)k == ko) ? bit : 0ull;
                        }
                    }
                }
                ++ncommit;
            } else {
                if (lane == 0) {
                    a.out_node[gj] = -1;
                    a.inf_task[ninf] = gj;
                    a.inf_pos[ninf] = ncommit;
                }
                ++ninf;
            }
            ++j;
            ustart = ustart + 1;
        }
        if (want_rebase) {
            ++st_rebase;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            if (!build_planes()) {
                if (lane == 0) { a.ctl->error = ERR_LEVEL_RANGE; a.ctl->resume = a.j0 + j; }   // j tasks are done
                fatal = true;
            }
        }
        if (ustart >= D) ustart = 0;
    }
    for (u32 n = lane; n < a.n_nodes; n += 64) a.last[n] = last_lds[n];
    if (lane == 0) {
        a.ctl->ncommit = ncommit;
        a.ctl->ninf = ninf;
        a.ctl->verify_retries += st_retries;
        a.ctl->slow_tasks += st_slow;
        a.ctl->rebases += st_rebase;
    }
}

// ---------------------------------------------------------------------------------------------
// k_resolve2 — two cooperating wavefronts in one workgroup.
//   wave 1 (LOADER)   streams, one block of R2_TB tasks ahead, the tasks' F rows, their services' X rows
//                     and a compact task record from HBM into an LDS double buffer;
//   wave 0 (RESOLVER) runs the sequential argmin/commit loop touching LDS and registers only: it never
//                     consumes a VMEM or SMEM result, so no memory latency sits on the critical path and
//                     its stores/atomics are fire-and-forget.
// A lone wave issues roughly one dependent instruction per 4-5 cycles, so the resolver is built for a
// SHORT instruction stream: a small non-unrolled loop (I-cache resident) and "hot level" masks —
//   BELOW = nodes with level < h, LA = level == h, LB = level == h+1 (h = the level most picks come from) —
// which turn the common-case argmin into ANDs plus wave ballots (no reduction at all: the lowest node index
// is the lowest (k, lane, bit)). The bit-planes stay the ground truth; anything unusual (no candidate at
// h/h+1, a candidate below h, host ports, a node already committed to in this window, level overflow)
// takes the generic path = the full bit-sliced search + re-check + exception list.
// X freshness: an X row staged in LDS may miss the commits of the last ≤2 blocks; the resolver keeps the
// last 64 commits (service, node), one per lane, and ORs the matching ones in at consumption.
// ---------------------------------------------------------------------------------------------
#define R2_TB_MAX 16
struct R2Rec {   // 32 B, staged per task
    i64 cpu, mem;
    u32 flags, svc, slot, pset;
};

template <int K, bool PROF>
__global__ __launch_bounds__(128) void k_resolve2(ResolveArgs a) {
    extern __shared__ unsigned char r2_lds[];
    const u32 Wn = a.n_words, XS = a.xs, R2_TB = a.tb;
    // LDS carve-up (all 16-byte aligned)
    int32_t* last_lds = reinterpret_cast<int32_t*>(r2_lds);                                   // [n_nodes]
    const size_t off_f = (((size_t)a.n_nodes * 4 + 15) / 16) * 16;
    const size_t fblk = (size_t)R2_TB * Wn * 8, xblk = (size_t)R2_TB * XS * 8;
    u64* Fb = reinterpret_cast<u64*>(r2_lds + off_f);                                          // [2][TB][Wn]
    u64* Xb = reinterpret_cast<u64*>(r2_lds + off_f + 2 * fblk);                               // [2][TB][XS]
    R2Rec* Tb = reinterpret_cast<R2Rec*>(r2_lds + off_f + 2 * fblk + 2 * xblk);                // [2][TB]
    u32* flags_lds = reinterpret_cast<u32*>(r2_lds + off_f + 2 * fblk + 2 * xblk + 2 * R2_TB * sizeof(R2Rec));
    // flags_lds[0..1] = ready[buf] (block index + 1), [2] = done (blocks finished by the resolver), [3] = abort
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 nblk = (a.count + R2_TB - 1) / R2_TB;
    if (a.ctl->error != ERR_NONE) return;
    if (tid < 4) flags_lds[tid] = 0;
    for (u32 n = tid; n < a.n_nodes; n += 128) last_lds[n] = a.last[n];
    __syncthreads();

    if (wave == 1) {
        // =============================== LOADER ===============================
        for (u32 b = 0; b < nblk; ++b) {
            const u32 buf = b & 1;
            if (b >= 2) {   // buffer is free (and every commit of blocks ≤ b-2 is visible) once block b-2 is done
                u32 spins = 0;
                while (__hip_atomic_load(&flags_lds[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b - 1) {
                    if (__hip_atomic_load(&flags_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 26)) return;   // bounded: never hang the GPU
                }
            }
            const u32 t0 = b * R2_TB, nt = min((u32)R2_TB, a.count - t0);
            // task records
            if (lane < nt) {
                const RTask* r = a.rt + a.j0 + t0 + lane;
                R2Rec rec;
                rec.cpu = r->cpu;
                rec.mem = r->mem;
                rec.flags = r->flags;
                rec.svc = r->svc;
                rec.slot = r->slot;
                rec.pset = r->pset;
                Tb[buf * R2_TB + lane] = rec;
            }
            // F rows: one contiguous run of nt*Wn words; 8 loads in flight per lane
            {
                const u64* src = a.F + (size_t)t0 * Wn;
                u64* dst = Fb + (size_t)buf * R2_TB * Wn;
                const u32 nw = nt * Wn;
                for (u32 i0 = 0; i0 < nw; i0 += 64 * 8) {
                    u64 v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        u32 i = i0 + 64 * q + lane;
                        v[q] = i < nw ? src[i] : 0ull;
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        u32 i = i0 + 64 * q + lane;
                        if (i < nw) dst[i] = v[q];
                    }
                }
            }
            // X rows (mutable: read past the L1); 4 rows = 4*K loads in flight per lane
            for (u32 t = 0; t < nt; t += 4) {
                u64 v[4][K];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ht = t + q < nt;
                    const u32 svc = ht ? cload(&a.rt[a.j0 + t0 + t + q].svc) : 0u;
                    const u64* src = a.X + (size_t)svc * XS;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const u32 i = lane + 64 * k;
                        v[q][k] = (ht && i < Wn) ? __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (t + q < nt) {
                        u64* dst = Xb + ((size_t)buf * R2_TB + t + q) * XS;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const u32 i = lane + 64 * k;
                            if (i < Wn) dst[i] = v[q][k];
                        }
                    }
                }
            }
            __hip_atomic_store(&flags_lds[buf], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }

    // =============================== RESOLVER ===============================
    u32 ncommit = a.ctl->ncommit, ninf = a.ctl->ninf;
    u32 applied = ncommit;   // commits [applied, ncommit) still live only in the lane ring
    u32 st_retries = 0, st_slow = 0, st_rebase = 0, st_generic = 0, st_spins = 0;
    const u32 idx_bits = 32 - __clz((Wn * 64) | 1u);
    const u32 idx_mask = (1u << idx_bits) - 1u;
    u32 NB = 1, base = 0, h = 0;
    u64 pl[R1_NBR][K];   // level bit-planes of the owned words {lane + 64k}
    u64 tch[K];          // nodes committed to since the scan
    u64 BELOW[K], LA[K], LB[K], VAL[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        tch[k] = 0;
        VAL[k] = (lane + 64 * k) < Wn ? a.valid[lane + 64 * k] : 0ull;
    }
    // commit ring: lane e holds the commit whose index ≡ e (mod 64). It serves (1) X freshness for rows
    // staged before the commit and (2) the deferred, lane-parallel application of the side effects.
    u32 rg_svc = 0xFFFFFFFFu, rg_node = 0, rg_task = 0, rg_slot = 0, rg_flags = 0;
    i64 rg_cpu = 0, rg_mem = 0;

    auto build_planes = [&]() __attribute__((always_inline)) -> bool {
        u32 lo = 0xFFFFFFFFu, hi = 0;
        for (int i = 0; i < 64; ++i) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if ((VAL[k] >> i) & 1ull) {
                    u32 t = __hip_atomic_load(&a.total[(lane + 64 * k) * 64 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lo = min(lo, t);
                    hi = max(hi, t);
                }
            }
        }
        lo = wave_min_u32(lo);
        hi = wave_max_u32(hi);
        if (lo == 0xFFFFFFFFu) { lo = 0; hi = 0; }
        u32 need = 32 - __clz((hi - lo) | 1u);
        u32 cap = min((u32)R1_NBR, 32u - idx_bits);
        if (need > cap) return false;
        base = lo;
        NB = min(cap, need + 1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int b = 0; b < R1_NBR; ++b) pl[b][k] = 0;
        }
        for (int i = 0; i < 64; ++i) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                u32 lvl = 0;
                if ((VAL[k] >> i) & 1ull)
                    lvl = __hip_atomic_load(&a.total[(lane + 64 * k) * 64 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
#pragma unroll
                for (int b = 0; b < R1_NBR; ++b) pl[b][k] |= (u64)((lvl >> b) & 1u) << i;
            }
        }
        return true;
    };
    // hot masks from the planes for level hh: BELOW = level < hh, LA = level == hh, LB = level == hh+1
    auto derive_masks = [&](u32 hh) __attribute__((always_inline)) {
        h = hh;
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
            BELOW[k] = lt;
            LA[k] = eq;
            LB[k] = eq1;
        }
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
            u32 w = lane + 64 * k;
            u32 cand = (lv[k] << idx_bits) | (w * 64 + (u32)(__ffsll((long long)m[k]) - 1));
            best = min(best, m[k] ? cand : 0xFFFFFFFFu);
        }
        return wave_min_u32_dpp(best);
    };
    // level+1 for one node: planes 0..ctz(~lvl) flip (always a run of low bits → early-exit chain)
    auto bump = [&](const u64 (&xk)[K], u32 lvl) __attribute__((always_inline)) {
        const u32 flip = lvl ^ (lvl + 1);
#pragma unroll
        for (int b = 0; b < R1_NBR; ++b) {
            if (__builtin_expect(!(flip >> b & 1u), b > 0)) break;
#pragma unroll
            for (int k = 0; k < K; ++k) pl[b][k] ^= xk[k];
        }
    };
    // Apply the side effects of commits [applied, ncommit) — one commit per lane: residual update of the
    // node row (NodeInfo.addTask, nodeinfo.go:108-154), exception-list entry, commit log + per-node chain.
    auto flush = [&]() __attribute__((always_inline)) {
        if (ncommit != applied) {
            const u32 last_c = ncommit - 1;
            const u32 ce = last_c - ((last_c - lane) & 63u);   // this lane's newest commit index
            if (ce >= applied && ce <= last_c) {
                const u32 n = rg_node;
                if (rg_cpu) atomicAdd(reinterpret_cast<u64*>(a.cpu + n), (u64)(-rg_cpu));
                if (rg_mem) atomicAdd(reinterpret_cast<u64*>(a.mem + n), (u64)(-rg_mem));
                if (rg_flags & 1u) {   // counted
                    atomicAdd(a.total + n, 1u);
                    if (rg_flags & 2u) atomicAdd(a.list_svc + rg_slot, 1u);   // placed through the exception list
                    else {
                        atomicOr(&a.X[(size_t)rg_svc * XS + (n >> 6)], 1ull << (n & 63));   // the loader restages X rows from memory
                        a.list_node[rg_slot] = n;
                        a.list_svc[rg_slot] = 1;
                        a.list_fail[rg_slot] = 0;
                    }
                }
                a.log_node[ce] = n;
                a.log_task[ce] = rg_task;
                a.log_prev[ce] = (int32_t)atomicExch(reinterpret_cast<u32*>(&last_lds[n]), ce);   // chain order within a flush is arbitrary
                a.out_node[rg_task] = (int32_t)n;
            }
            applied = ncommit;
        }
    };

    if (!build_planes()) {
        if (lane == 0) {
            a.ctl->error = ERR_LEVEL_RANGE;
            a.ctl->resume = a.j0;   // nothing of this window was touched
            __hip_atomic_store(&flags_lds[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    derive_masks(0);
    bool fatal = false;
    constexpr bool prof = PROF;
    u64 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 tk = prof ? wall_clock64() : 0;
    const u64 c_start = prof ? clock64() : 0, w_start = tk;
#define R2_TICK(slot) do { if constexpr (PROF) { u64 _n = wall_clock64(); cyc[slot] += _n - tk; tk = _n; } } while (0)

    auto wait_block = [&](u32 bi) __attribute__((always_inline)) {
        u32 spins = 0;
        while (__hip_atomic_load(&flags_lds[bi & 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != bi + 1) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 26)) { fatal = true; break; }
        }
        st_spins += spins;
    };
    // rows + record of the task about to be processed (read one task ahead)
    u64 cF[K], cX[K];
    R2Rec crec;
    auto read_task_at = [&](u32 bi, u32 tt) __attribute__((always_inline)) {
        const u32 bf = bi & 1;
        crec = Tb[bf * R2_TB + tt];
        const u64* frow = Fb + ((size_t)bf * R2_TB + tt) * Wn;
        const u64* xrow = Xb + ((size_t)bf * R2_TB + tt) * XS;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const u32 w = lane + 64 * k;
            const bool in = w < Wn;
            cF[k] = in ? frow[w] : 0ull;
            cX[k] = in ? xrow[w] : 0ull;
        }
    };
    wait_block(0);
    if (!fatal) read_task_at(0, 0);
    R2_TICK(0);

    u32 tin = 0, bdone = 0;   // task index inside the block, blocks finished
    for (u32 j = 0; j < a.count && !fatal; ++j) {
        const u32 gj = a.j0 + j;
        const R2Rec rec = crec;
        const u32 rflags = (u32)__builtin_amdgcn_readfirstlane((int)rec.flags);
        const u32 rsvc = (u32)__builtin_amdgcn_readfirstlane((int)rec.svc);
        u64 mk[K];
        {
            u64 Xc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) Xc[k] = cX[k];
            // commits younger than the staged X row
            u64 match = ballot64(rg_svc == rsvc);
            while (__builtin_expect(match != 0, 0)) {
                int e = __ffsll((long long)match) - 1;
                match &= match - 1;
                u32 nn = (u32)__builtin_amdgcn_readlane((int)rg_node, e);
                u32 ww = nn >> 6;
#pragma unroll
                for (int k = 0; k < K; ++k) Xc[k] |= (lane + 64u * k == ww) ? (1ull << (nn & 63)) : 0ull;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) mk[k] = cF[k] & ~Xc[k];
        }
        R2_TICK(1);

        // ---------------- fast pick: lowest node at the hot level h (LA), else at h+1 (LB); nothing below h ----------------
        // preference order of the 2K candidate words: A0..A(K-1), L0..L(K-1); inside a word: lowest lane, lowest bit
        u64 cw[2 * K];
        u64 bal[2 * K];
        bool anyb = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            anyb = anyb || ((mk[k] & BELOW[k]) != 0);
            cw[k] = mk[k] & LA[k];
            cw[K + k] = mk[k] & LB[k];
        }
#pragma unroll
        for (int q = 0; q < 2 * K; ++q) bal[q] = ballot64(cw[q] != 0);
        const u64 below = ballot64(anyb);
        int qsel = -1;
        u64 bsel = 0;
#pragma unroll
        for (int q = 2 * K - 1; q >= 0; --q) {
            const bool nz = bal[q] != 0;
            qsel = nz ? q : qsel;
            bsel = nz ? bal[q] : bsel;
        }
        bool fast = !(rflags & (RT_PORTS | RT_UNCOUNTED)) && below == 0 && qsel >= 0;
        u32 n = 0, lvl = 0, w = 0, ko = 0;
        u64 bit = 0;
        bool owner = false;
        if (__builtin_expect(fast, 1)) {
            const int l = __ffsll((long long)bsel) - 1;
            u64 wsel = cw[0];
#pragma unroll
            for (int q = 1; q < 2 * K; ++q) wsel = (q == qsel) ? cw[q] : wsel;
            const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)wsel, l);
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(wsel >> 32), l);
            const u64 word = ((u64)hi << 32) | lo;
            ko = (u32)qsel % (u32)K;
            lvl = h + ((u32)qsel >= (u32)K ? 1u : 0u);
            w = (u32)l + 64u * ko;
            const u32 bpos = (u32)(__ffsll((long long)word) - 1);
            n = w * 64u + bpos;
            bit = 1ull << bpos;
            owner = (u32)l == lane;
            u64 tsel = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) tsel = ((u32)k == ko) ? tch[k] : tsel;
            // node already committed to in this window (its F bit may be stale) or level overflow → generic path
            if (ballot64(owner && (tsel & bit)) != 0 || lvl >= (1u << NB) - 1u) fast = false;
        }
        R2_TICK(2);

        // ---------------- read the next task's rows now: their LDS latency hides under the commit ----------------
        if (j + 1 < a.count) {
            if (__builtin_expect(tin + 1 == R2_TB, 0)) wait_block(bdone + 1);
            if (!fatal) read_task_at(tin + 1 == R2_TB ? bdone + 1 : bdone, tin + 1 == R2_TB ? 0u : tin + 1);
        }

        bool placed = fast, via_list = false;
        u32 entry = 0;
        bool want_rebase = false;
        if (__builtin_expect(!fast, 0)) {
            // ---------------- generic path: every feature, full bit-sliced search ----------------
            ++st_generic;
            bool anym = false;
#pragma unroll
            for (int k = 0; k < K; ++k) anym = anym || (mk[k] != 0);
            const u32 e0 = a.list_off[rsvc], e1 = a.list_off[rsvc + 1];
            placed = false;
            if (ballot64(anym) != 0 || e1 > e0) {
                const i64 rcpu = rec.cpu, rmem = rec.mem;
                const u32 rpset = rec.pset;
                flush();   // the re-checks below read cpu/mem/total/lists: bring them up to date
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                u64 gk[K];
#pragma unroll
                for (int k = 0; k < K; ++k) gk[k] = mk[k];
                for (;;) {
                    u32 g = search(gk);
                    if (g == 0xFFFFFFFFu) break;
                    n = g & idx_mask;
                    lvl = g >> idx_bits;
                    w = n >> 6;
                    ko = w >> 6;
                    bit = 1ull << (n & 63);
                    owner = (w & 63) == lane;
                    u64 tsel = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) tsel = ((u32)k == ko) ? tch[k] : tsel;
                    bool ok = true;
                    if (ballot64(owner && (tsel & bit)) != 0) {
                        if (rflags & RT_RES) {
                            i64 c = __hip_atomic_load(&a.cpu[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            i64 m = __hip_atomic_load(&a.mem[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = (rcpu <= c) && (rmem <= m);
                        }
                        if (ok && (rflags & RT_PORTS)) {
                            for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p)
                                if (__hip_atomic_load(&a.portmap[(size_t)a.pset_ids[p] * Wn + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) ok = false;
                        }
                    }
                    if (ok) { placed = true; break; }
#pragma unroll
                    for (int k = 0; k < K; ++k) gk[k] &= ~((owner && (u32)k == ko) ? bit : 0ull);
                    ++st_retries;
                }
                if (!placed && e1 > e0) {
                    // exception list of the service: nodes with svcCount>0 or ≥5 recent failures
                    const u64 maxrep = a.rt[gj].maxrep;
                    u64 bhi = KEY_NONE, blo = KEY_NONE;
                    u32 be = 0;
                    for (u32 e = e0 + lane; e < e1; e += 64) {
                        u32 nn = __hip_atomic_load(&a.list_node[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (nn == LIST_EMPTY) continue;
                        u32 ww = nn >> 6;
                        u64 bb = 1ull << (nn & 63);
                        if (!(a.F[(size_t)j * Wn + ww] & bb)) continue;
                        if (rflags & RT_RES) {
                            i64 c = __hip_atomic_load(&a.cpu[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            i64 m = __hip_atomic_load(&a.mem[nn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (!(rcpu <= c && rmem <= m)) continue;
                        }
                        if (rflags & RT_PORTS) {
                            bool used = false;
                            for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p)
                                if (__hip_atomic_load(&a.portmap[(size_t)a.pset_ids[p] * Wn + ww], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bb) used = true;
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
                    if (ghi != KEY_NONE) {
                        u64 glo = wave_min_u64(bhi == ghi ? blo : KEY_NONE);
                        u64 who = ballot64(bhi == ghi && blo == glo);
                        entry = (u32)__builtin_amdgcn_readlane((int)be, __ffsll((long long)who) - 1);
                        n = (u32)glo;
                        lvl = (u32)(glo >> 32) - base;
                        w = n >> 6;
                        ko = w >> 6;
                        bit = 1ull << (n & 63);
                        owner = (w & 63) == lane;
                        placed = true;
                        via_list = true;
                        ++st_slow;
                    }
                }
            }
        }
        R2_TICK(3);

        if (__builtin_expect(placed, 1)) {
            // ---------------- commit: registers now, memory side effects at the next flush ----------------
            const bool counted = !(rflags & RT_UNCOUNTED);
            u64 xk[K];
#pragma unroll
            for (int k = 0; k < K; ++k) xk[k] = (owner && (u32)k == ko) ? bit : 0ull;
            if (counted) {
                if (__builtin_expect(lvl >= (1u << NB) - 1u, 0)) want_rebase = true;
                else {
                    bump(xk, lvl);
                    if (fast) {   // node moves h→h+1 (LA→LB) or h+1→h+2 (leaves LB)
                        const bool from_a = lvl == h;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            LA[k] &= ~xk[k];
                            LB[k] = (LB[k] & ~xk[k]) | (from_a ? xk[k] : 0ull);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) tch[k] |= xk[k];
            if (__builtin_expect((rflags & RT_PORTS) != 0, 0)) {
                if (owner)
                    for (u32 p = a.pset_off[rec.pset]; p < a.pset_off[rec.pset + 1]; ++p) atomicOr(&a.portmap[(size_t)a.pset_ids[p] * Wn + w], bit);
            }
            {   // remember the commit: X freshness + deferred side effects
                const bool me = lane == (ncommit & 63u);
                rg_svc = me ? ((counted && !via_list) ? rsvc : 0xFFFFFFFFu) : rg_svc;
                rg_node = me ? n : rg_node;
                rg_task = me ? gj : rg_task;
                rg_slot = me ? (via_list ? entry : rec.slot) : rg_slot;
                rg_flags = me ? ((counted ? 1u : 0u) | (via_list ? 2u : 0u)) : rg_flags;
                rg_cpu = me ? rec.cpu : rg_cpu;
                rg_mem = me ? rec.mem : rg_mem;
            }
            ++ncommit;
            if (__builtin_expect(!fast && !want_rebase && counted, 0)) {
                // a generic commit may have moved a node across the hot levels; a plain pick at another level
                // re-centres the hot level there (nodes below it stay exact through BELOW)
                derive_masks((!via_list && lvl + 2 < (1u << NB)) ? lvl : h);
            }
            if (__builtin_expect(fast && lvl == h + 1, 0)) {
                // picks come from h+1: if level h is exhausted for everybody, advance the hot level
                bool anya = false;
#pragma unroll
                for (int k = 0; k < K; ++k) anya = anya || (LA[k] != 0);
                if (!ballot64(anya) && h + 2 < (1u << NB) - 1u) derive_masks(h + 1);
            }
        } else {
            if (lane == 0) {
                a.inf_task[ninf] = gj;
                a.inf_pos[ninf] = ncommit;
            }
            ++ninf;
        }
        if (__builtin_expect(want_rebase, 0)) {
            ++st_rebase;
            flush();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!build_planes()) {
                if (lane == 0) { a.ctl->error = ERR_LEVEL_RANGE; a.ctl->resume = a.j0 + j + 1; }   // this task is committed
                fatal = true;
            } else derive_masks(0);
        }
        R2_TICK(4);
        if (__builtin_expect(++tin == R2_TB || j + 1 == a.count, 0)) {
            // block end: apply the (≤ R2_TB) pending commits lane-parallel; every X atomic of this block must be in
            // L2 before the loader may restage rows that depend on it
            flush();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&flags_lds[2], ++bdone, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            tin = 0;
            R2_TICK(5);
        }
    }
    if (fatal) __hip_atomic_store(&flags_lds[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    flush();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (u32 n = lane; n < a.n_nodes; n += 64) a.last[n] = last_lds[n];
    if (lane == 0) {
        a.ctl->ncommit = ncommit;
        a.ctl->ninf = ninf;
        a.ctl->verify_retries += st_retries;
        a.ctl->slow_tasks += st_slow;
        a.ctl->rebases += st_rebase;
        a.ctl->generic_tasks += st_generic;
        a.ctl->spin_waits += st_spins;
        if (prof) { cyc[6] = clock64() - c_start; cyc[7] = wall_clock64() - w_start; }
        for (int q = 0; q < 8; ++q) a.ctl->cyc[q] += cyc[q];
    }
}

// three-input bit operations on 64-bit words as two v_bitop3_b32 (gfx950); imm bit index = a*4 + b*2 + c
template <int IMM>
__device__ __forceinline__ u64 bitop3_u64(u64 a, u64 b, u64 c) {
    const u32 lo = (u32)__builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, IMM);
    const u32 hi = (u32)__builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), IMM);
    return ((u64)hi << 32) | lo;
}
#define BITOP_A_AND_B_ANDN_C 0x40   // a & b & ~c
#define BITOP_A_AND_BXORC 0x60      // a & (b ^ c)
#define BITOP_AB_OR_C 0xEA          // (a & b) | c

// ---------------------------------------------------------------------------------------------
// k_resolve3 — the sequential argmin + commit pass as ONE workgroup of four wavefronts. A lone wave issues roughly one
// instruction per 4 ns, so the task rate IS the resolver wave's instruction count per task; everything that is not the
// decision itself lives on the three helper waves:
//   * waves 1,2 (LOADERS) stage, one block of TB tasks ahead, mk = F & ~X per task (rows padded to 64*K words so that the
//     resolver's reads are unconditional), the task record, and per-task bits: "must take the generic path" (host ports /
//     uncounted / a candidate below the hot level — computed against the BELOW mask the resolver publishes under a
//     sequence lock; a block staged under an older epoch is recomputed by the resolver) and "the exception list may matter";
//   * wave 3 (COMMITTER) applies the memory side effects of every finished block from an LDS hand-over, so the resolver's
//     common path issues no VMEM and never waits for one;
//   * wave 0 (RESOLVER): level bit-planes, hot-level masks and the touched set in registers over the words {lane + 64k}.
//     The pick is specialised per slot k (a scalar branch): one s_ff1 on the slot's ballot, readlanes of that slot's
//     candidate / touched word — nothing goes through v_cndmask chains. A fast commit changes ONE register pair (D, the
//     nodes fast-committed since the last fold: LA = LA0 & ~D, LB = LB0 ^ D, touched = T0 | D; a node takes at most one fast
//     commit per window); the bit-sliced "+1 on D" brings the planes up to date when the generic path or a hot-level
//     change needs them. Level-range and hot-level-exhausted checks are scalars maintained at derive time.
//   * X rows staged before a commit are repaired from a 64-entry commit ring through a zeroed LDS row (constant cost).
// Semantics (pick order, exception lists, commit log, counters) are those of k_resolve2.
// ---------------------------------------------------------------------------------------------
template <int K, bool PROF>
__global__ __launch_bounds__(256) void k_resolve3(ResolveArgs a) {
    extern __shared__ unsigned char r3_lds[];
    const u32 Wn = a.n_words, XS = a.xs, TB = a.tb;
    constexpr u32 RS = K * 64;   // staged row stride in words
    int32_t* last_lds = reinterpret_cast<int32_t*>(r3_lds);                                    // [n_nodes]
    const size_t off_f = (((size_t)a.n_nodes * 4 + 15) / 16) * 16;
    u64* MK = reinterpret_cast<u64*>(r3_lds + off_f);                                          // [2*TB + 1][RS]  F & ~X
    u64* below_lds = MK + (size_t)(2 * TB + 1) * RS;                                           // [RS] published BELOW
    u64* xfix_lds = below_lds + RS;                                                            // [RS] scratch, all zero between uses
    R2Rec* Tb = reinterpret_cast<R2Rec*>(xfix_lds + RS);                                       // [2*TB + 1]
    uint4* dump_lds = reinterpret_cast<uint4*>(Tb + (2 * TB + 2));                             // [2][R2_TB_MAX] commits handed to the committer
    u32* flags_lds = reinterpret_cast<u32*>(dump_lds + 2 * R2_TB_MAX);
    // flags_lds[0..1] = ready[buf]: +1 per loader wave and staged block (block b is ready at 2*(b/2 + 1)),
    // [2] = done (blocks finished by the resolver), [3] = abort, [4] = BELOW epoch (odd while being rewritten),
    // [5 + 2*buf + loader] = epoch that loader's below-flags of the staged block were computed with,
    // [9] = issued (blocks whose side effects the committer has issued), [10] = completed (… and that are visible
    // in memory), [11 + buf] = number of commits in dump_lds[buf]
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 nblk = (a.count + TB - 1) / TB;
    if (a.ctl->error != ERR_NONE) return;
    if (tid < 16) flags_lds[tid] = 0;
    for (u32 n = tid; n < a.n_nodes; n += 256) last_lds[n] = a.last[n];
    for (u32 w = tid; w < RS; w += 256) xfix_lds[w] = 0;
    __syncthreads();

    // Side effects of one commit: residual update of the node row (NodeInfo.addTask, nodeinfo.go:108-154),
    // exception bitmap + list entry, commit log + per-node chain, placement. blk = the block the task belongs to.
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
                atomicOr(&a.X[(size_t)r.svc * XS + (n >> 6)], 1ull << (n & 63));   // the loaders restage X rows from memory
                a.list_node[r.slot] = n;
                a.list_svc[r.slot] = 1;
                a.list_fail[r.slot] = 0;
            }
        }
        a.log_node[ce] = n;
        a.log_task[ce] = gj;
        a.log_prev[ce] = (int32_t)atomicExch(reinterpret_cast<u32*>(&last_lds[n]), ce);   // chain order is arbitrary
        a.out_node[gj] = (int32_t)n;
    };

    if (wave == 3) {
        // =============================== COMMITTER ===============================
        // Applies the memory side effects of every finished block (handed over through dump_lds) so that the
        // resolver never issues or waits for VMEM on its common path.
        for (u32 b = 0; b < nblk; ++b) {
            u32 spins = 0;
            while (__hip_atomic_load(&flags_lds[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b + 1) {
                if (__hip_atomic_load(&flags_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 26)) return;   // bounded: never hang the GPU
            }
            const u32 cnt = flags_lds[11 + (b & 1u)];
            if (lane < cnt) {
                const uint4 e = dump_lds[(b & 1u) * R2_TB_MAX + lane];
                apply_commit(e.x, e.y, e.z, e.w, b);
            }
            // the dump and the task records of this block have been read: the loaders may restage the buffer
            __hip_atomic_store(&flags_lds[9], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&flags_lds[10], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }

    if (wave != 0) {
        // =============================== LOADERS (two waves, half a block each) ===============================
        // Stage, one block ahead: mk = F & ~X per task (rows padded to RS words), the task record, and three
        // per-task bits in the record's flags word so that the resolver's common case does no mask arithmetic:
        //   bit 31  the task must take the generic path: host ports / uncounted, or a candidate BELOW the hot level
        //   bit 30  a feasible node is an exception node of the service (F & X != 0): the exception list may matter
        //   bit 29  host ports / uncounted alone (used when the BELOW snapshot of the block is stale)
        const u32 lw = wave - 1;
        const u32 half = (TB + 1) / 2;
        constexpr int LB = K <= 4 ? 8 : 4;   // rows in flight per loader wave (2*K*LB loads per lane)
        for (u32 b = 0; b < nblk; ++b) {
            const u32 buf = b & 1;
            if (b >= 2) {   // buffer is free once the committer has read block b-2's records; commits of blocks ≤ b-3 are visible
                u32 spins = 0;
                while (__hip_atomic_load(&flags_lds[9], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b - 1) {
                    if (__hip_atomic_load(&flags_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 26)) return;   // bounded: never hang the GPU
                }
            }
            const u32 t0 = b * TB, nt = min(TB, a.count - t0);
            const u32 tb0 = min(nt, lw * half), tb1 = min(nt, (lw + 1) * half);   // this wave's tasks of the block
            // task records first: their latency hides under the row loads
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
            // BELOW snapshot under a sequence lock
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
                        x[q][k] = ok ? __hip_atomic_load(&xs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;   // mutable: past the L1
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // rows + record of every lane before the count
            if (lane == 0) __hip_atomic_fetch_add(&flags_lds[buf], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // ONE increment per wave
        }
        return;
    }

    // =============================== RESOLVER ===============================
    u32 ncommit = (u32)__builtin_amdgcn_readfirstlane((int)a.ctl->ncommit), ninf = (u32)__builtin_amdgcn_readfirstlane((int)a.ctl->ninf);
    u32 applied = ncommit;   // commits [applied, ncommit) still live only in the lane ring
    u32 st_retries = 0, st_slow = 0, st_rebase = 0, st_generic = 0, st_spins = 0;
    const u32 idx_bits = 32 - __clz((Wn * 64) | 1u);
    const u32 idx_mask = (1u << idx_bits) - 1u;
    u32 NB = 1, base = 0, h = 0;
    u32 la_count = 0;    // nodes left at the hot level (scalar)
    u32 epoch = 0;       // BELOW epoch (even), bumped by 2 at every publish
    // Exact state = (planes, LA0, LB0, T0) as of the last fold, plus D = nodes fast-committed since then:
    //   level(n) = planes(n) + [n in D];  LA = LA0 & ~D;  LB = LB0 ^ D;  touched = T0 | D.
    // A node takes at most one fast commit per window (a touched pick goes generic), so one bit per node suffices
    // and a fast commit changes D only.
    u64 pl[R1_NBR][K];   // level bit-planes of the owned words {lane + 64k}
    u64 D[K];
    u64 T0[K];           // nodes committed to since the scan (their F bits may be stale)
    u64 BELOW[K], LA0[K], LB0[K], VAL[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        T0[k] = 0;
        D[k] = 0;
        VAL[k] = (lane + 64 * k) < Wn ? a.valid[lane + 64 * k] : 0ull;
    }
    // commit ring: lane e holds the commit whose index ≡ e (mod 64): X freshness for rows staged before the
    // commit, and the deferred lane-parallel application of the side effects (≤ TB pending at any time)
    u32 rg_svc = 0xFFFFFFFFu, rg_node = 0, rg_meta = 0, rg_slot = 0;   // meta = task-in-block | counted<<8 | via_list<<9

    // (Re)build the level planes from total[]. Each lane owns 64 consecutive counters per word: all 64 loads of a word
    // are issued before the first use (one memory round trip per word instead of one per node: the serialised version
    // cost ~0.25 ms per launch). The loads bypass the L1: a rebase re-reads counters this kernel has just updated.
    auto load_word_totals = [&](int k, u32 (&v)[64]) __attribute__((always_inline)) {
        const u32 w = lane + 64 * k;
        const u32* src = a.total + (size_t)(w < Wn ? w : 0) * 64;
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
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
        lo = (u32)__builtin_amdgcn_readfirstlane((int)wave_min_u32(lo));
        hi = (u32)__builtin_amdgcn_readfirstlane((int)wave_max_u32(hi));
        if (lo == 0xFFFFFFFFu) { lo = 0; hi = 0; }
        u32 need = 32 - __clz((hi - lo) | 1u);
        u32 cap = min((u32)R1_NBR, 32u - idx_bits);
        if (need > cap) return false;
        base = lo;
        NB = min(cap, need + 1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            T0[k] |= D[k];
            D[k] = 0;
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
    // fold D into the exact state: planes += 1 on D (bit-sliced ripple carry), masks, touched; D = 0
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u64 c = D[k];
            LA0[k] &= ~c;
            LB0[k] ^= c;
            T0[k] |= c;
#pragma unroll
            for (int b = 0; b < R1_NBR; ++b) {
                const u64 t = pl[b][k] & c;
                pl[b][k] ^= c;
                c = t;
            }
            D[k] = 0;
        }
    };
    // hot masks from the planes (D must be empty) for level hh: BELOW = level < hh, LA = level == hh, LB = level == hh+1.
    // When hh+1 cannot be bumped inside the planes every candidate is routed to the generic path (BELOW = all).
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
        la_count = (u32)__builtin_amdgcn_readfirstlane((int)cnt);
        // publish BELOW for the loader (sequence lock: odd epoch while the words are rewritten)
        __hip_atomic_store(&flags_lds[4], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
        for (int k = 0; k < K; ++k) __hip_atomic_store(&below_lds[lane + 64 * k], BELOW[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        epoch += 2u;
        __hip_atomic_store(&flags_lds[4], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
            u32 w = lane + 64 * k;
            u32 cand = (lv[k] << idx_bits) | (w * 64 + (u32)(__ffsll((long long)m[k]) - 1));
            best = min(best, m[k] ? cand : 0xFFFFFFFFu);
        }
        return best;   // per lane: (level << idx_bits) | node of its best candidate, or 0xFFFFFFFF
    };
    u32 tin = 0, bdone = 0;   // task index inside the block, blocks finished
    // Generic path / rebase only: memory must reflect every commit so far. The pending commits of the current block
    // are applied here (one per lane); the finished blocks are the committer's — wait until it reports them visible.
    bool fatal = false, soft_stop = false;
    auto flush = [&]() __attribute__((always_inline)) {
        if (ncommit != applied) {
            const u32 last_c = ncommit - 1;
            const u32 ce = last_c - ((last_c - lane) & 63u);   // this lane's newest commit index
            if (ce >= applied && ce <= last_c) apply_commit(rg_node, rg_meta, rg_slot, ce, bdone);
            applied = ncommit;
        }
        u32 spins = 0;
        while ((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&flags_lds[10], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < bdone) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 26)) { fatal = true; break; }
        }
    };

    if (!build_planes()) {
        if (lane == 0) {
            a.ctl->error = ERR_LEVEL_RANGE;
            a.ctl->resume = a.j0;   // nothing of this window was touched
            __hip_atomic_store(&flags_lds[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    derive_masks(0);
    u64 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 tk = PROF ? wall_clock64() : 0;
    const u64 c_start = PROF ? clock64() : 0, w_start = tk;

    auto wait_block = [&](u32 bi) __attribute__((always_inline)) {
        u32 spins = 0;
        while ((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&flags_lds[bi & 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < 2u * ((bi >> 1) + 1u)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 26)) { fatal = true; break; }
        }
        st_spins += spins;
    };
    // rows + record of the task about to be processed: read at the END of the previous iteration into the same
    // registers the pick works on (no copies)
    u64 mk[K];
    uint2 cr;   // {flags word, svc}
    u32 blk_ep = 0;   // BELOW epoch the current block's bit 31 was computed with
    auto read_slot = [&](u32 slot) __attribute__((always_inline)) {   // slot = buf*TB + task-in-block (slot 2*TB = padding)
        cr = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(&Tb[slot]) + 16);
        const u64* row = MK + (size_t)slot * RS + lane;
#pragma unroll
        for (int k = 0; k < K; ++k) mk[k] = row[64 * k];
    };
    wait_block(0);
    if (!fatal) read_slot(0);
    auto block_epoch = [&](u32 bf) __attribute__((always_inline)) -> u32 {
        const u32 e0 = (u32)__builtin_amdgcn_readfirstlane((int)flags_lds[5 + 2 * bf]);
        const u32 e1 = (u32)__builtin_amdgcn_readfirstlane((int)flags_lds[6 + 2 * bf]);
        return e0 == e1 ? e0 : 0xFFFFFFFFu;   // odd: never equals the (even) current epoch
    };
    blk_ep = block_epoch(0);
    u32 nslot = 1;   // staged slot of the NEXT task
    R2_TICK(0);

    for (u32 j = 0; j < a.count && !fatal; ++j) {
        const u32 flagw = cr.x;   // per-lane copy of a uniform word
        const u32 rsvc = (u32)__builtin_amdgcn_readfirstlane((int)cr.y);
        {   // commits younger than the staged X row (the ring spans the last 64 commits ≥ 3 blocks): every matching
            // ring lane ORs its node's bit into a zeroed LDS row, all lanes subtract their words, the row is zeroed
            // again — constant cost however many commits match (service-major task order: all of them)
            const bool hit = rg_svc == rsvc;
            if (__builtin_expect(ballot64(hit) != 0, 0)) {
                u64* cell = xfix_lds + (rg_node >> 6);
                if (hit) __hip_atomic_fetch_or(cell, 1ull << (rg_node & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // one wave: LDS executes in issue order
#pragma unroll
                for (int k = 0; k < K; ++k) mk[k] &= ~__hip_atomic_load(&xfix_lds[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                if (hit) __hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        R2_TICK(1);

        // ---------------- fast pick: lowest node at the hot level h (LA), else at h+1 (LB); nothing below h ----------------
        bool generic;
        if (__builtin_expect(blk_ep == epoch, 1)) generic = ballot64((int)flagw < 0) != 0;   // staged: forced or a candidate below h
        else {
            u64 sb = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) sb = bitop3_u64<BITOP_AB_OR_C>(mk[k], BELOW[k], sb);
            generic = ballot64(((u32)sb | (u32)(sb >> 32) | (flagw & 0x20000000u)) != 0) != 0;
        }
        u64 ca[K], ba[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ca[k] = bitop3_u64<BITOP_A_AND_B_ANDN_C>(mk[k], LA0[k], D[k]);
            ba[k] = ballot64(ca[k] != 0);
        }
        bool placed = false;
        u32 n = 0;

        // one specialised copy per (class, slot): scalar pick, then D gains the node's bit on its owner lane
#define R3_TAKE(kk, ISB, BAL, CW)                                                                                  \
    {                                                                                                              \
        const int l_ = __builtin_ctzll(BAL);                                                                       \
        const u64 tv_ = T0[kk] | D[kk];                                                                            \
        const u32 wlo_ = (u32)__builtin_amdgcn_readlane((int)(u32)(CW), l_);                                       \
        const u32 whi_ = (u32)__builtin_amdgcn_readlane((int)(u32)((CW) >> 32), l_);                               \
        const u32 tlo_ = (u32)__builtin_amdgcn_readlane((int)(u32)tv_, l_);                                        \
        const u32 thi_ = (u32)__builtin_amdgcn_readlane((int)(u32)(tv_ >> 32), l_);                                \
        const u64 word_ = ((u64)whi_ << 32) | wlo_, tw_ = ((u64)thi_ << 32) | tlo_;                                \
        const u32 bpos_ = (u32)__builtin_ctzll(word_);                                                             \
        const u64 bit_ = 1ull << bpos_;                                                                            \
        if (__builtin_expect((tw_ & bit_) != 0, 0)) generic = true; /* committed to in this window: F may be stale */ \
        else {                                                                                                     \
            n = (((u32)l_ + 64u * kk) << 6) + bpos_;                                                               \
            D[kk] |= (lane == (u32)l_) ? bit_ : 0ull;                                                              \
            if (!ISB) --la_count;                                                                                  \
            placed = true;                                                                                         \
        }                                                                                                          \
    }
        if (__builtin_expect(!generic, 1)) {
            bool tryb = false;
            if (ba[0] != 0) { R3_TAKE(0, false, ba[0], ca[0]) }
            else if constexpr (K > 1) {
                if (ba[1] != 0) { R3_TAKE(1, false, ba[1], ca[1]) }
                else if constexpr (K > 2) {
                    if (ba[2] != 0) { R3_TAKE(2, false, ba[2], ca[2]) }
                    else if constexpr (K > 3) {
                        if (ba[3] != 0) { R3_TAKE(3, false, ba[3], ca[3]) }
                        else if constexpr (K > 4) {
                            if (ba[4] != 0) { R3_TAKE(4, false, ba[4], ca[4]) }
                            else if constexpr (K > 5) {
                                if (ba[5] != 0) { R3_TAKE(5, false, ba[5], ca[5]) }
                                else if constexpr (K > 6) {
                                    if (ba[6] != 0) { R3_TAKE(6, false, ba[6], ca[6]) }
                                    else if constexpr (K > 7) {
                                        if (ba[7] != 0) { R3_TAKE(7, false, ba[7], ca[7]) }
                                        else tryb = true;
                                    }
                                    else tryb = true;
                                }
                                else tryb = true;
                            }
                            else tryb = true;
                        }
                        else tryb = true;
                    }
                    else tryb = true;
                }
                else tryb = true;
            }
            else tryb = true;
            if (tryb) {
                u64 cb[K], bb[K];
                u64 anyb = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    cb[k] = bitop3_u64<BITOP_A_AND_BXORC>(mk[k], LB0[k], D[k]);
                    bb[k] = ballot64(cb[k] != 0);
                    anyb |= bb[k];
                }
                if (anyb == 0) generic = true;
                else {
                    if (K == 1 || bb[0] != 0) { R3_TAKE(0, true, bb[0], cb[0]) }
                    else if constexpr (K > 1) {
                        if (K == 2 || bb[1] != 0) { R3_TAKE(1, true, bb[1], cb[1]) }
                        else if constexpr (K > 2) {
                            if (K == 3 || bb[2] != 0) { R3_TAKE(2, true, bb[2], cb[2]) }
                            else if constexpr (K > 3) {
                                if (K == 4 || bb[3] != 0) { R3_TAKE(3, true, bb[3], cb[3]) }
                                else if constexpr (K > 4) {
                                    if (K == 5 || bb[4] != 0) { R3_TAKE(4, true, bb[4], cb[4]) }
                                    else if constexpr (K > 5) {
                                        if (K == 6 || bb[5] != 0) { R3_TAKE(5, true, bb[5], cb[5]) }
                                        else if constexpr (K > 6) {
                                            if (K == 7 || bb[6] != 0) { R3_TAKE(6, true, bb[6], cb[6]) }
                                            else if constexpr (K > 7) {
                                                if (K == 8 || bb[7] != 0) { R3_TAKE(7, true, bb[7], cb[7]) }
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                    // picks come from h+1: once level h is exhausted for everybody, advance the hot level
                    if (__builtin_expect(placed && la_count == 0 && h + 3u <= (1u << NB) - 1u, 0)) {
                        fold();
                        derive_masks(h + 1);
                    }
                }
            }
        }
#undef R3_TAKE
        R2_TICK(2);

        bool recorded = false;   // the generic path records its own commit
        if (__builtin_expect(!placed && generic, 0)) {
            // ---------------- generic path: every feature, full bit-sliced search on exact planes ----------------
            ++st_generic;
            // the exception list can only matter if a feasible node is an exception node of the service (list nodes ⊆ X:
            // staged bit 30) or a commit of the service is still in flight
            bool anym = false;
#pragma unroll
            for (int k = 0; k < K; ++k) anym = anym || (mk[k] != 0);
            const bool listp = ballot64((flagw & 0x40000000u) != 0) != 0 || ballot64(rg_svc == rsvc) != 0;
            if (ballot64(anym) != 0 || listp) {
                const R2Rec rec = Tb[(bdone & 1u) * TB + tin];
                const u32 rflags = (u32)__builtin_amdgcn_readfirstlane((int)rec.flags);
                const i64 rcpu = rec.cpu, rmem = rec.mem;
                const u32 rpset = rec.pset;
                const u32 gj = a.j0 + j;
                fold();
                flush();   // the re-checks below read cpu/mem/total/lists: bring them up to date
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                u32 lvl = 0, w = 0, ko = 0;
                u64 bit = 0;
                bool owner = false, via_list = false;
                u32 entry = 0;
                u64 gk[K];
#pragma unroll
                for (int k = 0; k < K; ++k) gk[k] = mk[k];
                for (;;) {
                    // Every lane proposes the best candidate of its own words; the wave's minimum is THE candidate of the
                    // reference (lowest level, lowest index). Lanes whose proposal sits at that same level re-check it
                    // against memory in the same round trip (a node committed to in this window may no longer fit: its F
                    // bit is stale) and drop it if it fails — so a storm of stale candidates costs one memory latency per
                    // 64 of them, not one each. Dropping is safe: resources only shrink inside a batch.
                    const u32 mine = search(gk);
                    const u32 g = wave_min_u32_dpp(mine);
                    if (g == 0xFFFFFFFFu) break;
                    const u32 glvl = g >> idx_bits;
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
                    if (ballot64(act && mine == g && mine_ok) != 0) {
                        n = g & idx_mask;
                        lvl = glvl;
                        w = n >> 6;
                        ko = w >> 6;
                        bit = 1ull << (n & 63);
                        owner = (w & 63) == lane;
                        placed = true;
                        break;
                    }
                }
                if (!placed && listp) {
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
                        entry = (u32)__builtin_amdgcn_readlane((int)be, __ffsll((long long)who) - 1);
                        n = (u32)__builtin_amdgcn_readfirstlane((int)(u32)glo);
                        lvl = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(glo >> 32)) - base;
                        w = n >> 6;
                        ko = w >> 6;
                        bit = 1ull << (n & 63);
                        owner = (w & 63) == lane;
                        placed = true;
                        via_list = true;
                        ++st_slow;
                    }
                }
                if (placed) {
                    // generic commit: exact planes bumped in place, hot masks re-derived
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
                    if (rflags & RT_PORTS) {
                        if (owner)
                            for (u32 p = a.pset_off[rpset]; p < a.pset_off[rpset + 1]; ++p) atomicOr(&a.portmap[(size_t)a.pset_ids[p] * Wn + w], bit);
                    }
                    {
                        const bool me = lane == (ncommit & 63u);
                        rg_svc = me ? ((counted && !via_list) ? rsvc : 0xFFFFFFFFu) : rg_svc;
                        rg_node = me ? n : rg_node;
                        rg_meta = me ? (tin | (counted ? 0x100u : 0u) | (via_list ? 0x200u : 0u)) : rg_meta;
                        rg_slot = me ? entry : rg_slot;
                        ++ncommit;
                        recorded = true;
                    }
                    if (want_rebase) {
                        // the commit is applied to memory first (total[n] + 1), then the planes are rebuilt
                        ++st_rebase;
                        flush();
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (!build_planes()) {
                            // the spread outgrew the register planes: stop after this (committed) task; the host
                            // continues from `resume` with the 16-plane workgroup resolver
                            if (lane == 0) { a.ctl->error = ERR_LEVEL_RANGE; a.ctl->resume = a.j0 + j + 1; }
                            fatal = true;
                            soft_stop = true;
                        } else derive_masks(0);
                    } else if (counted) {
                        // a plain pick at another level re-centres the hot level there (nodes below stay exact through BELOW)
                        derive_masks((!via_list && lvl + 2 < (1u << NB)) ? lvl : h);
                    }
                }
                // leave no VMEM result pending into the common path: the waitcnt pass would otherwise guard the loop
                // top with a vmcnt(0) that also drains every fire-and-forget store. (Only here: the quick exit above —
                // a task without any candidate — issues no load, and must not wait for the stores in flight.)
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            }
        }
        R2_TICK(3);

        if (__builtin_expect(placed && !recorded, 1)) {
            // remember the fast commit: X freshness + deferred side effects (list slot comes from the task record)
            const bool me = lane == (ncommit & 63u);
            rg_svc = me ? rsvc : rg_svc;
            rg_node = me ? n : rg_node;
            rg_meta = me ? (tin | 0x100u) : rg_meta;
            ++ncommit;
        } else if (!placed) {
            if (lane == 0) {
                a.inf_task[ninf] = a.j0 + j;
                a.inf_pos[ninf] = ncommit;
            }
            ++ninf;
        }
        R2_TICK(4);
        // next task's rows and record (at a block end this reads a stale/padding slot that the block-end code re-reads)
        read_slot(nslot);
        ++nslot;
        if (__builtin_expect(++tin == TB || j + 1 == a.count, 0)) {
            // block end: hand the (≤ TB) pending commits to the committer wave through LDS. No VMEM here. The dump slot
            // was last used two blocks ago; the committer has normally long consumed it.
            {
                u32 spins = 0;
                while (bdone >= 2 && (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&flags_lds[9], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < bdone - 1) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 26)) { fatal = true; break; }
                }
                st_spins += spins;
            }
            {
                const u32 pend = ncommit - applied;
                if (pend) {
                    const u32 last_c = ncommit - 1;
                    const u32 ce = last_c - ((last_c - lane) & 63u);
                    if (ce >= applied && ce <= last_c) dump_lds[(bdone & 1u) * R2_TB_MAX + (ce - applied)] = make_uint4(rg_node, rg_meta, rg_slot, ce);
                }
                if (lane == 0) flags_lds[11 + (bdone & 1u)] = pend;
                applied = ncommit;
            }
            __hip_atomic_store(&flags_lds[2], ++bdone, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            tin = 0;
            if (j + 1 < a.count) {
                wait_block(bdone);
                nslot = (bdone & 1u) * TB;
                if (!fatal) read_slot(nslot);
                ++nslot;
                blk_ep = block_epoch(bdone & 1u);
            }
            R2_TICK(5);
        }
    }
    // pending commits of a partial block + the committer's finished blocks (after a soft stop the helper waves are
    // still alive: release them only afterwards)
    if (!fatal || soft_stop) { fatal = false; flush(); fatal = fatal || soft_stop; }
    if (fatal) __hip_atomic_store(&flags_lds[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (u32 n2 = lane; n2 < a.n_nodes; n2 += 64) a.last[n2] = last_lds[n2];
    if (lane == 0) {
        a.ctl->ncommit = ncommit;
        a.ctl->ninf = ninf;
        a.ctl->verify_retries += st_retries;
        a.ctl->slow_tasks += st_slow;
        a.ctl->rebases += st_rebase;
        a.ctl->generic_tasks += st_generic;
        a.ctl->spin_waits += st_spins;
        if (PROF) { cyc[6] = clock64() - c_start; cyc[7] = wall_clock64() - w_start; }
        for (int q = 0; q < 8; ++q) a.ctl->cyc[q] += cyc[q];
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
    // insertion sort by commit index while copying (chains are short and nearly descending already)
    u32 k = 0;
    for (int32_t ci = a.last[n]; ci >= 0; ci = a.log_prev[ci], ++k) {
        const RTask* tk = a.rt + a.log_task[ci];
        const i64 tc = tk->cpu, tm = tk->mem;
        u32 p = k;
        while (p > 0 && a.ent_ci[off + p - 1] > (u32)ci) {
            a.ent_ci[off + p] = a.ent_ci[off + p - 1];
            a.ent_scpu[off + p] = a.ent_scpu[off + p - 1];
            a.ent_smem[off + p] = a.ent_smem[off + p - 1];
            --p;
        }
        a.ent_ci[off + p] = (u32)ci;
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
        for (int f = 0; f < 7; ++f) {
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
// k_groups — grouped tasks (SpecVersion != nil): scheduleTaskGroup with k = len(group)
// (scheduler.go:694-748), nodeSet.tree with a bounded max-heap per leaf (nodeset.go:50-124,
// container/heap mechanics reproduced step for step: nodeheap.go, decision_tree.go:24-52),
// scheduleNTasksOnSubtree (:772-825) and the fill loop scheduleNTasksOnNodes (:844-924).
// One workgroup walks the groups of a tick in order (a group sees the previous group's commits).
// Per group: (A) all threads evaluate Pipeline.Process + the nodeLess key for every node in parallel;
// (B) nodes are admitted to their leaf's heap in index order, 256 at a time: threads pre-filter
// against the heap root at chunk start (the root key only decreases, so the pre-filter is a superset),
// thread 0 replays the survivors exactly; (C) thread 0 runs the tree walk and the fill loops on the
// heap nodes' state held in LDS; (D) results are written back and the service's (node, count) list is
// rebuilt. The Explain counters of a group that could not be placed completely are recovered by
// replaying the Process call sequence of tree() and appending the fill phase's logged calls.
// ---------------------------------------------------------------------------------------------
#define G_HCAP 1536      // heap slots over all leaves of one group
#define G_MAXT 512       // tree nodes of one spread set
#define G_LOG 8192       // Process results logged by the fill phase
#define FF_PASS 255u

struct GroupRec {   // one per group
    i64 cpu, mem;
    u32 flags;          // RT_*
    u32 k;              // group size
    u32 svc;            // batch-local service
    u32 out_off;        // first output index
    u32 pset;
    u32 cls_con, cls_plat, cls_plug;
    u64 maxrep;
    u32 tree;           // spread set (0 = no preferences)
    u32 pad;
};
static_assert(sizeof(GroupRec) == 64, "GroupRec layout");

struct GroupArgs {
    u32 n_nodes, n_words, n_groups, n_trees;
    const GroupRec* g;
    const u64* valid;
    const u64* ready;
    const u64* con;
    const u64* plat;
    const u64* plug;
    i64* cpu;
    i64* mem;
    u32* total;
    u64* portmap;
    const u32* pset_off;
    const u32* pset_ids;
    u32* list_node;
    u32* list_svc;
    u32* list_fail;
    const u32* list_off;     // [n_svc+1]
    // tree topology per spread set: tnodes of tree t are [tree_off[t], tree_off[t+1]) ; indices are relative
    const u32* tree_off;
    const u32* tn_parent;    // 0xFFFFFFFF for the root
    const u32* tn_first;     // first child or 0xFFFFFFFF
    const u32* tn_next;      // next sibling or 0xFFFFFFFF
    const u32* tn_nchild;
    const u32* tn_nodes;     // nodes whose leaf this tnode is
    const u32* leaf_of_node; // [n_trees][n_nodes] tnode of the node's leaf
    unsigned char* ff;       // scratch [n_nodes]
    u32* svc_dense;          // scratch [n_nodes]
    u32* fail_dense;         // scratch [n_nodes]
    int32_t* out_node;
    u32* hist;               // [n_groups][8]
    Ctl* ctl;
};

// nodeLess, scheduler.go:708-735
__device__ __forceinline__ bool g_less_vals(u32 fa, u32 sa, u32 ta, u32 fb, u32 sb, u32 tb) {
    if (fa >= MAX_FAILURES || fb >= MAX_FAILURES) {
        if (fa > fb) return false;
        if (fb > fa) return true;
    }
    if (sa < sb) return true;
    if (sa > sb) return false;
    return ta < tb;
}

// nodeLess as ONE integer compare: key = (failures if >= 5 else 0, svcCount, total) packed 8 | 24 | 32 bits
// (both sides below 5 failures skip the failure compare, scheduler.go:713-722; a side at >= 5 beats any side below).
__device__ __forceinline__ u64 g_key(u32 fail, u32 svc, u32 total) {
    const u32 fc = fail >= MAX_FAILURES ? fail : 0u;
    return ((u64)fc << 56) | ((u64)svc << 32) | total;
}
__device__ __forceinline__ bool g_key_ok(u32 fail, u32 svc) { return fail < 256u && svc < (1u << 24); }

struct GHeap {   // heap positions hold (key, state id); the node state itself never moves
    u64* key;
    u32* pay;
    u32 *node, *total, *svc, *fail, *placed;   // state, indexed by state id
    i64 *cpu, *mem;
    __device__ __forceinline__ bool less(u32 a, u32 b) const { return key[a] < key[b]; }
    __device__ __forceinline__ void swap(u32 a, u32 b) {
        const u64 k = key[a]; key[a] = key[b]; key[b] = k;
        const u32 t = pay[a]; pay[a] = pay[b]; pay[b] = t;
    }
};
// container/heap (go stdlib) over nodeMaxHeap: Less(i,j) = lessFunc(nodes[j], nodes[i]) (nodeheap.go:17-20).
// up / down move ONE element along a path and swap it with what it meets: the element rides in registers and every step copies
// the other one into the hole — the same comparisons, the same final arrangement as the swap sequence, but one LDS round trip
// per level (both children's keys and payloads are requested together) instead of three. The walk is the serial chain of
// k_groups: LDS latency of a single thread.
template <class HP> __device__ inline void g_up(HP& h, u32 base, int j0) {
    int j = j0;
    const u64 kv = h.key[base + j];
    const u32 pv = h.pay[base + j];
    for (;;) {
        const int i = (j - 1) / 2;   // j == 0: i == 0 (Go's integer division truncates, too)
        if (i == j) break;
        const u64 ki = h.key[base + i];
        const u32 pi = h.pay[base + i];
        if (!(ki < kv)) break;       // !Less(j, i)
        h.key[base + j] = ki;
        h.pay[base + j] = pi;
        j = i;
    }
    if (j != j0) {
        h.key[base + j] = kv;
        h.pay[base + j] = pv;
    }
}
template <class HP> __device__ inline bool g_down(HP& h, u32 base, int i0, int n) {
    int i = i0;
    const u64 kv = h.key[base + i];
    const u32 pv = h.pay[base + i];
    for (;;) {
        const int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        const int j2 = j1 + 1;
        const bool two = j2 < n;
        const u64 k1 = h.key[base + j1], k2 = two ? h.key[base + j2] : 0ull;
        const u32 p1 = h.pay[base + j1], p2 = two ? h.pay[base + j2] : 0u;
        const bool right = two && k1 < k2;   // Less(j2, j1)
        const u64 kj = right ? k2 : k1;
        if (!(kv < kj)) break;               // !Less(j, i)
        h.key[base + i] = kj;
        h.pay[base + i] = right ? p2 : p1;
        i = right ? j2 : j1;
    }
    if (i > i0) {
        h.key[base + i] = kv;
        h.pay[base + i] = pv;
    }
    return i > i0;
}

#define G_THREADS 1024   // one workgroup; every N-long loop strides by it
__global__ __launch_bounds__(G_THREADS) void k_groups(GroupArgs a) {
    extern __shared__ unsigned char g_lds[];
    GHeap H;
    unsigned char* p = g_lds;
    H.key = reinterpret_cast<u64*>(p); p += G_HCAP * 8;
    H.cpu = reinterpret_cast<i64*>(p); p += G_HCAP * 8;
    H.mem = reinterpret_cast<i64*>(p); p += G_HCAP * 8;
    i64* tsum = reinterpret_cast<i64*>(p); p += G_MAXT * 8;              // decisionTree.tasks
    i64* e_cpu = reinterpret_cast<i64*>(p); p += G_THREADS * 8;
    i64* e_mem = reinterpret_cast<i64*>(p); p += G_THREADS * 8;
    H.node = reinterpret_cast<u32*>(p); p += G_HCAP * 4;
    H.total = reinterpret_cast<u32*>(p); p += G_HCAP * 4;
    H.svc = reinterpret_cast<u32*>(p); p += G_HCAP * 4;
    H.fail = reinterpret_cast<u32*>(p); p += G_HCAP * 4;
    H.placed = reinterpret_cast<u32*>(p); p += G_HCAP * 4;
    H.pay = reinterpret_cast<u32*>(p); p += G_HCAP * 4;
    u32* h_off = reinterpret_cast<u32*>(p); p += G_MAXT * 4;              // first heap slot of a leaf
    int32_t* h_len = reinterpret_cast<int32_t*>(p); p += G_MAXT * 4;      // nodeMaxHeap.length
    int32_t* h_cnt = reinterpret_cast<int32_t*>(p); p += G_MAXT * 4;      // len(nodeMaxHeap.nodes)
    int32_t* h_adm = reinterpret_cast<int32_t*>(p); p += G_MAXT * 4;      // slots ever filled (write-back range)
    u32* e_node = reinterpret_cast<u32*>(p); p += G_THREADS * 4;
    u32* e_total = reinterpret_cast<u32*>(p); p += G_THREADS * 4;
    u32* e_svc = reinterpret_cast<u32*>(p); p += G_THREADS * 4;
    u32* e_fail = reinterpret_cast<u32*>(p); p += G_THREADS * 4;
    u32* e_leaf = reinterpret_cast<u32*>(p); p += G_THREADS * 4;
    p += G_THREADS * 4;   // (spare staging column)
    u32* wcnt = reinterpret_cast<u32*>(p); p += 16 * 4;
    u32* shv = reinterpret_cast<u32*>(p); p += 16 * 4;
    u32* cntx = reinterpret_cast<u32*>(p); p += 8 * 4;                    // Explain counters
    u64* failed = reinterpret_cast<u64*>(p); p += (G_HCAP / 64) * 8;     // fill loop: slots that failed Process
    u64* rootkey = reinterpret_cast<u64*>(p); p += G_MAXT * 8;             // per leaf: heap root key after tree()
    unsigned char* plog = p; p += G_LOG;
    enum { S_NENT = 0, S_ERR = 1, S_LEFT = 2, S_NLOG = 3, S_LISTPOS = 4, S_LASTPASS = 5 };

    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 N = a.n_nodes, Wn = a.n_words;
    if (a.ctl->error != ERR_NONE) return;

    u64 gt[6] = {0, 0, 0, 0, 0, 0};
    u64 gtk = wall_clock64();
#define G_TICK(q) do { u64 _n = wall_clock64(); gt[q] += _n - gtk; gtk = _n; } while (0)
    for (u32 gi = 0; gi < a.n_groups; ++gi) {
        const GroupRec G = a.g[gi];
        const u32 tbase = a.tree_off[G.tree], ntn = a.tree_off[G.tree + 1] - tbase;
        const u32* leaf_of = a.leaf_of_node + (size_t)G.tree * N;
        const u32 k = G.k;
        // ---------- per-group reset ----------
        for (u32 i = tid; i < ntn; i += G_THREADS) { tsum[i] = 0; h_off[i] = 0; h_len[i] = 0; h_cnt[i] = 0; h_adm[i] = 0; }
        for (u32 n = tid; n < N; n += G_THREADS) { a.svc_dense[n] = 0; a.fail_dense[n] = 0; }
        if (tid == 0) { shv[S_ERR] = 0; shv[S_NLOG] = 0; shv[S_LEFT] = 0; shv[S_LASTPASS] = 0; }
        __syncthreads();
        // the service's (node, svcCount, failures) list → dense per-node columns
        for (u32 e = a.list_off[G.svc] + tid; e < a.list_off[G.svc + 1]; e += G_THREADS) {
            u32 n = a.list_node[e];
            if (n != LIST_EMPTY) { a.svc_dense[n] = a.list_svc[e]; a.fail_dense[n] = a.list_fail[e]; }
        }
        __syncthreads();

        G_TICK(0);
        // ---------- (A) Pipeline.Process on every node: first failing filter or FF_PASS ----------
        for (u32 n0 = 0; n0 < N; n0 += G_THREADS) {
            const u32 n = n0 + tid;
            if (n < N) {
                const u32 w = n >> 6;
                const u64 bit = 1ull << (n & 63);
                if (a.valid[w] & bit) {
                    u32 ff = FF_PASS;
                    if (!(a.ready[w] & bit)) ff = 0;
                    else if ((G.flags & RT_RES) && !(G.cpu <= a.cpu[n] && G.mem <= a.mem[n])) ff = 1;
                    else if (G.cls_plug && !(a.plug[(size_t)G.cls_plug * Wn + w] & bit)) ff = 2;
                    else if (G.cls_con && !(a.con[(size_t)G.cls_con * Wn + w] & bit)) ff = 3;
                    else if (G.cls_plat && !(a.plat[(size_t)G.cls_plat * Wn + w] & bit)) ff = 4;
                    else {
                        bool busy = false;
                        if (G.flags & RT_PORTS)
                            for (u32 q = a.pset_off[G.pset]; q < a.pset_off[G.pset + 1]; ++q)
                                if (a.portmap[(size_t)a.pset_ids[q] * Wn + w] & bit) busy = true;
                        if (busy) ff = 5;
                        else if ((G.flags & RT_MAXREP) && !((u64)a.svc_dense[n] < G.maxrep)) ff = 6;
                    }
                    a.ff[n] = (unsigned char)ff;
                    // tree(): the node's service count is added at its leaf (and, below, at every ancestor)
                    // whether or not the node is feasible (nodeset.go:88-90,103-105)
                    u32 sv = a.svc_dense[n];
                    if (sv) atomicAdd(reinterpret_cast<u64*>(&tsum[leaf_of[n]]), (u64)sv);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            for (int i = (int)ntn - 1; i > 0; --i) tsum[a.tn_parent[tbase + i]] += tsum[i];   // children follow their parent
            u32 off = 0;
            for (u32 i = 0; i < ntn; ++i)
                if (a.tn_nchild[tbase + i] == 0) { h_off[i] = off; off += min(k, a.tn_nodes[tbase + i]); }   // a leaf's heap holds ≤ its node count
            if (off > G_HCAP || ntn > G_MAXT) shv[S_ERR] = 1;
        }
        __syncthreads();
        if (shv[S_ERR]) {
            if (tid == 0) a.ctl->error = ERR_GROUP_RANGE;
            return;
        }

        G_TICK(1);
        // ---------- (B) heap admission in node order, a chunk of nodes at a time (nodeset.go:107-120) ----------
        // The pre-filter compares with the heap roots as they are at the chunk's start, so it is the sharper the shorter the chunk:
        // the first G_THREADS nodes (while the heaps fill and their roots still drop fast) go in chunks of 128.
        for (u32 n0 = 0; n0 < N;) {
            const u32 step = n0 < G_THREADS ? 128u : (u32)G_THREADS;
            const u32 n = n0 + tid;
            bool cand = false;
            u32 leaf = 0, sv = 0, fl = 0, tot = 0;
            if (tid < step && n < N && ((a.valid[n >> 6] >> (n & 63)) & 1ull) && a.ff[n] == FF_PASS) {
                leaf = leaf_of[n];
                sv = a.svc_dense[n];
                fl = a.fail_dense[n];
                tot = a.total[n];
                if (!g_key_ok(fl, sv)) shv[S_ERR] = 1;
                if (h_len[leaf] < (int)k) cand = true;
                else cand = g_key(fl, sv, tot) < H.key[h_off[leaf]];
            }
            const u64 bal = ballot64(cand);
            if (lane == 0) wcnt[wave] = (u32)__popcll(bal);
            __syncthreads();
            u32 before = 0;
            for (u32 q = 0; q < wave; ++q) before += wcnt[q];
            const u32 pos = before + (u32)__popcll(bal & ((1ull << lane) - 1ull));
            if (cand) {
                e_node[pos] = n; e_leaf[pos] = leaf; e_svc[pos] = sv; e_fail[pos] = fl; e_total[pos] = tot;
                e_cpu[pos] = a.cpu[n]; e_mem[pos] = a.mem[n];
            }
            if (tid == G_THREADS - 1) shv[S_NENT] = before + (u32)__popcll(bal);
            __syncthreads();
            if (tid == 0) {
                const u32 ne = shv[S_NENT];
                // the leaf of the last entry, its heap's base / length / root key ride in registers: a run of entries of one leaf
                // (every group without spread preferences) pays one LDS round trip per entry that does not enter the heap
                u32 c_lf = 0xFFFFFFFFu, base = 0;
                int len = 0;
                u64 root = 0;
                for (u32 i = 0; i < ne; ++i) {
                    const u32 lf = e_leaf[i];
                    const u64 ek = g_key(e_fail[i], e_svc[i], e_total[i]);
                    if (lf != c_lf) {
                        c_lf = lf;
                        base = h_off[lf];
                        len = h_len[lf];
                        root = len ? H.key[base] : 0ull;
                    }
                    u32 sid;
                    if (len < (int)k) sid = base + (u32)len;          // heap.Push: a fresh state slot
                    else if (ek < root) sid = H.pay[base];             // replaces the root: the evicted node's state slot is reused
                    else continue;
                    shv[S_LASTPASS] = e_node[i] + 1;   // the last Process that returned true inside tree()
                    H.node[sid] = e_node[i]; H.total[sid] = e_total[i]; H.svc[sid] = e_svc[i]; H.fail[sid] = e_fail[i];
                    H.cpu[sid] = e_cpu[i]; H.mem[sid] = e_mem[i]; H.placed[sid] = 0;
                    if (len < (int)k) {
                        H.key[sid] = ek; H.pay[sid] = sid;
                        h_len[lf] = len + 1; h_cnt[lf] = len + 1; h_adm[lf] = len + 1;
                        g_up(H, base, len);
                        ++len;
                    } else {
                        H.key[base] = ek;
                        if (!g_down(H, base, 0, len)) g_up(H, base, 0);   // heap.Fix(0)
                    }
                    root = H.key[base];
                }
            }
            __syncthreads();
            n0 += step;
        }

        G_TICK(2);
        // heap roots and lengths as tree() left them: the Explain pass needs them after (C) has spent the heaps
        for (u32 i = tid; i < ntn; i += G_THREADS) {
            rootkey[i] = H.key[h_off[i]];
            h_adm[i] = h_len[i];   // == slots ever filled (pushes only grow the heap)
        }
        __syncthreads();
        // ---------- (C) tree walk + fill loops: thread 0, on LDS state only ----------
        if (tid == 0) {
            u32 next_task = 0, nlog = 0;
            bool bad_key = false;
            const bool has_ports = (G.flags & RT_PORTS) != 0;
            // Pipeline.Process on a heap slot: the static filters passed at admission and cannot change
            auto process = [&](u32 pos) -> bool {
                const u32 sl = H.pay[pos];
                u32 ff = FF_PASS;
                if ((G.flags & RT_RES) && !(G.cpu <= H.cpu[sl] && G.mem <= H.mem[sl])) ff = 1;
                else if (has_ports && H.placed[sl] > 0) ff = 5;
                else if ((G.flags & RT_MAXREP) && !((u64)H.svc[sl] < G.maxrep)) ff = 6;
                if (nlog < G_LOG) plog[nlog] = (unsigned char)ff;
                ++nlog;
                return ff == FF_PASS;
            };
            // scheduleNTasksOnNodes, scheduler.go:844-924, on the leaf's slots [base, base+cnt)
            auto fill = [&](int want, u32 base, int cnt) -> int {
                int scheduled = 0, iter = 0;
                for (int q = 0; q <= (cnt >> 6); ++q) failed[q] = 0;
                while (next_task < k) {
                    const u32 pos = base + (u32)(iter % cnt);
                    const u32 sl = H.pay[pos];
                    a.out_node[G.out_off + next_task] = (int32_t)H.node[sl];
                    ++next_task;
                    H.cpu[sl] -= G.cpu;   // NodeInfo.addTask (nodeinfo.go:108-154)
                    H.mem[sl] -= G.mem;
                    H.placed[sl] += 1;
                    if (!(G.flags & RT_UNCOUNTED)) {
                        H.total[sl] += 1; H.svc[sl] += 1;
                        if (!g_key_ok(H.fail[sl], H.svc[sl])) bad_key = true;
                        H.key[pos] = g_key(H.fail[sl], H.svc[sl], H.total[sl]);
                    }
                    ++scheduled;
                    if (scheduled == want) return scheduled;
                    if (iter + 1 < cnt) {
                        if (H.less(base + (u32)((iter + 1) % cnt), pos)) ++iter;   // first pass
                    } else ++iter;                                                 // later passes: round robin
                    const int orig = iter;
                    for (;;) {
                        const int ix = iter % cnt;
                        const bool bad = (failed[ix >> 6] >> (ix & 63)) & 1ull;
                        if (!bad && process(base + (u32)ix)) break;
                        failed[ix >> 6] |= 1ull << (ix & 63);
                        ++iter;
                        if (iter - orig == cnt) return scheduled;
                    }
                }
                return scheduled;
            };
            // decisionTree.orderedNodes, decision_tree.go:24-52
            auto ordered = [&](u32 lf) -> int {
                const u32 base = h_off[lf];
                if (h_len[lf] != h_cnt[lf]) {
                    int cnt = h_cnt[lf];
                    for (int i = 0; i < cnt;) {
                        if (process(base + (u32)i)) ++i;
                        else {
                            --cnt;
                            if (i != cnt) H.swap(base + (u32)i, base + (u32)cnt);   // nodes[i] = nodes[last]; the dropped
                        }                                                            // node keeps its slot for the write-back
                    }
                    h_cnt[lf] = cnt;
                    h_len[lf] = cnt;
                    for (int i = cnt / 2 - 1; i >= 0; --i) g_down(H, base, i, cnt);   // heap.Init
                }
                while (h_len[lf] > 0) {   // heap.Pop: Swap(0,n-1); down(0,n-1); length--
                    const int nn = h_len[lf] - 1;
                    H.swap(base, base + (u32)nn);
                    g_down(H, base, 0, nn);
                    h_len[lf] = nn;
                }
                return h_cnt[lf];
            };
            // scheduleNTasksOnSubtree, scheduler.go:772-825, as an explicit stack machine
            struct Frame { u32 tn; int n; int scheduled; i64 usable; u64 noroom; i64 desired; i64 rem; u32 child; int child_pos; int assign; bool converging; int phase; };
            Frame st[8];
            int sp = 0, ret = 0;
            bool bad = false;
            st[0] = Frame{0u, (int)k, 0, 0, 0ull, 0, 0, 0xFFFFFFFFu, 0, 0, true, 0};
            while (sp >= 0 && !bad) {
                Frame& f = st[sp];
                const u32 nch = a.tn_nchild[tbase + f.tn];
                if (f.phase == 0) {
                    if (nch == 0) {   // leaf
                        const int cnt = ordered(f.tn);
                        ret = cnt == 0 ? 0 : fill(f.n, h_off[f.tn], cnt);
                        --sp;
                        continue;
                    }
                    if (nch > 64) { bad = true; break; }
                    f.scheduled = 0;
                    f.usable = tsum[f.tn];
                    f.noroom = 0;
                    f.converging = true;
                    f.phase = 1;
                }
                if (f.phase == 3) {   // a child call returned `ret`
                    if (ret < f.assign) {
                        f.noroom |= 1ull << f.child_pos;
                        f.usable -= tsum[f.child];
                    } else if (f.rem > 0) f.rem--;
                    f.scheduled += ret;
                    f.child = a.tn_next[tbase + f.child];
                    f.child_pos++;
                    f.phase = 2;
                }
                if (f.phase == 1) {   // while condition + per-round quantities
                    const int room = (int)nch - __popcll(f.noroom);
                    if (!(f.scheduled != f.n && room != 0 && f.converging)) {
                        ret = f.scheduled;
                        --sp;
                        continue;
                    }
                    const i64 tot = f.usable + f.n - f.scheduled;
                    f.desired = tot / room;
                    f.rem = tot % room;
                    f.converging = false;
                    f.child = a.tn_first[tbase + f.tn];
                    f.child_pos = 0;
                    f.phase = 2;
                }
                // phase 2: `for _, subtree := range tree.next`
                bool called = false;
                while (f.child != 0xFFFFFFFFu) {
                    if (!((f.noroom >> f.child_pos) & 1ull)) {
                        const i64 sub = tsum[f.child];
                        if (sub < f.desired || (sub == f.desired && f.rem > 0)) {
                            f.converging = true;
                            f.assign = (int)(f.desired - sub) + (f.rem > 0 ? 1 : 0);
                            if (sp + 1 >= 8) { bad = true; break; }
                            f.phase = 3;
                            st[sp + 1] = Frame{f.child, f.assign, 0, 0, 0ull, 0, 0, 0xFFFFFFFFu, 0, 0, true, 0};
                            ++sp;
                            called = true;
                            break;
                        }
                    }
                    f.child = a.tn_next[tbase + f.child];
                    f.child_pos++;
                }
                if (!called && !bad) f.phase = 1;
            }
            if (bad || bad_key) shv[S_ERR] = 1;
            shv[S_LEFT] = k - next_task;
            shv[S_NLOG] = nlog;
            for (u32 i = next_task; i < k; ++i) a.out_node[G.out_off + i] = -1;
        }
        __syncthreads();
        if (shv[S_ERR]) {
            if (tid == 0) a.ctl->error = ERR_GROUP_RANGE;
            return;
        }

        G_TICK(3);
        // ---------- Explain counters for a group with leftovers (pipeline.go:56-68 call sequence) ----------
        if (shv[S_LEFT] > 0) {
            // Every passing Process zeroes the counters (pipeline.go:64-66), so only the calls AFTER the last passing one
            // count. Inside tree() the heaps stop changing after that call: a later node was "called" (nodeset.go:108-116)
            // iff its leaf's heap was not full or the node is less than the final root — evaluated in parallel.
            if (tid < 8) cntx[tid] = 0;
            __syncthreads();
            const u32 lastp = shv[S_LASTPASS];   // node index + 1 of the last passing call (0: none)
            for (u32 n0 = 0; n0 < N; n0 += G_THREADS) {
                const u32 n = n0 + tid;
                u32 f = 0xFFu;
                if (n < N && n >= lastp && ((a.valid[n >> 6] >> (n & 63)) & 1ull)) {
                    const u32 ffn = a.ff[n];
                    if (ffn != FF_PASS) {
                        const u32 lf = leaf_of[n];
                        const bool called = h_adm[lf] < (int)k ||
                                            g_key(a.fail_dense[n], a.svc_dense[n], a.total[n]) < rootkey[lf];   // (D) has not run yet: tree()-time values
                        if (called) f = ffn;
                    }
                }
                for (u32 q = 0; q < 7; ++q) {
                    const u64 bm = ballot64(f == q);
                    if (bm && lane == 0) atomicAdd(&cntx[q], (u32)__popcll(bm));
                }
            }
            __syncthreads();
            if (tid == 0) {
                const u32 nl = min(shv[S_NLOG], (u32)G_LOG);
                for (u32 i = 0; i < nl; ++i) {
                    if (plog[i] == FF_PASS) { for (int q = 0; q < 8; ++q) cntx[q] = 0; }
                    else cntx[plog[i]]++;
                }
                for (int q = 0; q < 8; ++q) a.hist[(size_t)gi * 8 + q] = cntx[q];
                if (shv[S_NLOG] > G_LOG) a.ctl->error = ERR_GROUP_RANGE;
            }
            __syncthreads();
        }

        // ---------- (D) write-back: node rows, host ports, the service's (node, count) list ----------
        for (u32 i = tid; i < ntn; i += G_THREADS) {
            if (a.tn_nchild[tbase + i] != 0) continue;
            const u32 base = h_off[i];
            for (int q = 0; q < h_adm[i]; ++q) {
                const u32 sl = base + (u32)q;
                if (H.placed[sl] == 0) continue;
                const u32 n = H.node[sl];
                a.cpu[n] = H.cpu[sl];
                a.mem[n] = H.mem[sl];
                a.total[n] = H.total[sl];
                a.svc_dense[n] = H.svc[sl];
                if (G.flags & RT_PORTS)
                    for (u32 z = a.pset_off[G.pset]; z < a.pset_off[G.pset + 1]; ++z)
                        atomicOr(&a.portmap[(size_t)a.pset_ids[z] * Wn + (n >> 6)], 1ull << (n & 63));
            }
        }
        if (tid == 0) shv[S_LISTPOS] = a.list_off[G.svc];
        __syncthreads();
        const u32 lend = a.list_off[G.svc + 1];
        for (u32 n0 = 0; n0 < N; n0 += G_THREADS) {
            const u32 n = n0 + tid;
            const bool keep = n < N && (a.svc_dense[n] > 0 || a.fail_dense[n] >= MAX_FAILURES);
            const u64 bal = ballot64(keep);
            if (lane == 0) wcnt[wave] = (u32)__popcll(bal);
            __syncthreads();
            u32 before = shv[S_LISTPOS];
            for (u32 q = 0; q < wave; ++q) before += wcnt[q];
            const u32 pos = before + (u32)__popcll(bal & ((1ull << lane) - 1ull));
            if (keep && pos < lend) { a.list_node[pos] = n; a.list_svc[pos] = a.svc_dense[n]; a.list_fail[pos] = a.fail_dense[n]; }
            __syncthreads();
            if (tid == 0) { u32 t_ = 0; for (u32 q = 0; q < G_THREADS / 64; ++q) t_ += wcnt[q]; shv[S_LISTPOS] += t_; }
            __syncthreads();
        }
        for (u32 e = shv[S_LISTPOS] + tid; e < lend; e += G_THREADS) a.list_node[e] = LIST_EMPTY;
        __syncthreads();
        G_TICK(5);
    }
    if (tid == 0) for (int q = 0; q < 6; ++q) a.ctl->cyc[q] = gt[q];
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
