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

namespace swpdev {

typedef unsigned long long u64;
typedef long long i64;
typedef uint32_t u32;

#define DEV_VALID 0x80000000u   // node slot is present in the nodeSet

// mirror of SWP_NODE_* (include/swp.h)
#define NF_READY 0x001u
#define NF_HAS_DESC 0x002u
#define NF_HAS_PLATFORM 0x004u
#define NF_HAS_ENGINE 0x008u
#define NF_HAS_LABELS 0x010u
#define NF_HAS_ELABELS 0x020u
#define NF_MANAGER 0x040u
#define NF_HAS_LOGPLUG 0x080u
#define NF_IP_VALID 0x100u
#define NF_IP_V4 0x200u

// RTask.flags
#define RT_RES 0x1u        // resource filter enabled
#define RT_PORTS 0x2u      // host-port filter enabled
#define RT_MAXREP 0x4u     // max-replicas filter enabled
#define RT_UNCOUNTED 0x8u  // DesiredState > COMPLETED: placement does not bump the task counts

#define LIST_EMPTY 0xFFFFFFFFu
#define KEY_NONE 0xFFFFFFFFFFFFFFFFull
#define MAX_FAILURES 5u   // scheduler.go:23

struct RTask {   // 64 B per task, batch order
    i64 cpu, mem;
    u32 flags;
    u32 sc;        // static class (ready & plugin & constraint & platform bitmap row)
    u32 svc;       // batch-local service index
    u32 slot;      // absolute index of this task's own entry in the per-service exception list
    u32 pset;      // batch-local port set
    u32 cls_con, cls_plat, cls_plug;   // batch-local class rows (0 = filter disabled) — explain pass
    u64 maxrep;
    u32 pad[2];
};
static_assert(sizeof(RTask) == 64, "RTask layout");

struct DevConstraint {   // 48 B
    u32 kind, op, col, value;
    u32 ip[4];
    u32 ip_kind, prefix_len, ip_is_v4, pad;
};

struct Ctl {
    u32 ncommit, ninf, error, pad0;
    u64 verify_retries, slow_tasks, rebases, pad1;
};

enum { ERR_NONE = 0, ERR_LEVEL_RANGE = 1 };

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
struct ResolveArgs {
    u32 n_nodes, n_words;
    u32 j0, count;
    u32 nb_alloc;            // planes that fit in LDS
    const u64* F;            // [count][n_words] for this window
    const u64* valid;        // [n_words]
    u64* X;                  // [n_svc][n_words]
    const RTask* rt;
    i64* cpu;
    i64* mem;
    u32* total;
    u32* list_node;
    u32* list_svc;
    u32* list_fail;
    const u32* list_off;     // [n_svc+1]
    u64* portmap;
    const u32* pset_off;
    const u32* pset_ids;
    int32_t* out_node;       // [T]
    u32* log_node;
    u32* log_task;
    int32_t* log_prev;
    int32_t* last;           // [n_nodes]
    u32* inf_task;
    u32* inf_pos;
    Ctl* ctl;
};

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
        __syncthreads();
        u64 g = red[par * 16];
        for (u32 i = 1; i < nw; ++i) {
            u64 o = red[par * 16 + i];
            g = o < g ? o : g;
        }
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
                u32 t = a.total[w * 64 + i];
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
        if (need > a.nb_alloc) return false;
        base = lo;
        NB = min(a.nb_alloc, need + 1);              // one spare bit: room to double before the next rebase
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
                u32 lvl = a.total[w * 64 + i] - base;
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
            u64 best = KEY_NONE;
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
                    u64 cand = ((u64)lvl << 32) | (u64)(w * 64 + (u32)(__ffsll((long long)m) - 1));
                    best = cand < best ? cand : best;
                }
            }
            u64 g = R.block_min(best);
            if (g == KEY_NONE) break;
            u32 n = (u32)g, w = n >> 6;
            u64 bit = 1ull << (n & 63);
            bool owner = (w % B) == tid;
            int ko = (int)(w / B);
            if (owner) {
                bool ok = true;
                if (R.touched[w] & bit) {   // F may be stale for this node: re-check the dynamic filters
                    if (rt.flags & RT_RES) ok = (rt.cpu <= a.cpu[n]) && (rt.mem <= a.mem[n]);
                    if (ok && (rt.flags & RT_PORTS)) {
                        for (u32 p = a.pset_off[rt.pset]; p < a.pset_off[rt.pset + 1]; ++p)
                            if (a.portmap[(size_t)a.pset_ids[p] * a.n_words + w] & bit) ok = false;
                    }
                }
                if (ok) {
                    // residual update == NodeInfo.addTask (nodeinfo.go:108-154)
                    a.cpu[n] -= rt.cpu;
                    a.mem[n] -= rt.mem;
                    R.touched[w] |= bit;
                    if (rt.flags & RT_PORTS)
                        for (u32 p = a.pset_off[rt.pset]; p < a.pset_off[rt.pset + 1]; ++p)
                            a.portmap[(size_t)a.pset_ids[p] * a.n_words + w] |= bit;
                    if (counted) {
                        a.total[n] += 1;
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
                    a.log_prev[ncommit] = a.last[n];
                    a.last[n] = (int32_t)ncommit;
                    a.out_node[gj] = (int32_t)n;
                }
                sh[RS::SH_OK] = ok ? 1u : 0u;
            }
            __syncthreads();
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
                    a.cpu[n] -= rt.cpu;
                    a.mem[n] -= rt.mem;
                    R.touched[w] |= bit;
                    if (rt.flags & RT_PORTS)
                        for (u32 p = a.pset_off[rt.pset]; p < a.pset_off[rt.pset + 1]; ++p)
                            a.portmap[(size_t)a.pset_ids[p] * a.n_words + w] |= bit;
                    if (counted) {
                        a.total[n] += 1;
                        if (!R.bump_level(w, bit)) sh[RS::SH_REBASE] = 1;
                        u32 sv = __hip_atomic_load(&a.list_svc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&a.list_svc[e], sv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    a.log_node[ncommit] = n;
                    a.log_task[ncommit] = gj;
                    a.log_prev[ncommit] = a.last[n];
                    a.last[n] = (int32_t)ncommit;
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
    if (tid == 0) {
        a.ctl->ncommit = ncommit;
        a.ctl->ninf = ninf;
        a.ctl->verify_retries += st_retries;
        a.ctl->slow_tasks += st_slow;
        a.ctl->rebases += st_rebase;
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
};

__global__ __launch_bounds__(256) void k_explain(ExplainArgs a) {
    u32 e = blockIdx.y;
    u32 n = blockIdx.x * blockDim.x + threadIdx.x;
    u32 gj = cload(a.inf_task + e);
    int32_t pos = (int32_t)cload(a.inf_pos + e);
    const RTask rt = a.rt[gj];
    u32 w = n >> 6;
    u64 bit = 1ull << (n & 63);
    bool present = n < a.n_nodes && (a.valid[w] & bit);
    int ff = -1;
    if (present) {
        i64 c = a.cpu[n], m = a.mem[n];
        u32 svc_later = 0;
        u32 port_later = 0;   // bit q: the q-th port of this task's set was taken on n by a LATER commit
        u32 pp0 = 0, pp1 = 0;
        if (rt.flags & RT_PORTS) { pp0 = a.pset_off[rt.pset]; pp1 = a.pset_off[rt.pset + 1]; }
        for (int32_t ci = a.last[n]; ci >= pos; ci = a.log_prev[ci]) {
            const RTask* tk = a.rt + a.log_task[ci];
            c += tk->cpu;
            m += tk->mem;
            if (tk->svc == rt.svc && !(tk->flags & RT_UNCOUNTED)) ++svc_later;
            if ((rt.flags & RT_PORTS) && (tk->flags & RT_PORTS)) {
                for (u32 q = pp0; q < pp1 && q - pp0 < 32; ++q)
                    for (u32 z = a.pset_off[tk->pset]; z < a.pset_off[tk->pset + 1]; ++z)
                        if (a.pset_ids[z] == a.pset_ids[q]) port_later |= 1u << (q - pp0);
            }
        }
        if (!(a.ready[w] & bit)) ff = 0;
        else if ((rt.flags & RT_RES) && !(rt.cpu <= c && rt.mem <= m)) ff = 1;
        else if (rt.cls_plug && !(a.plug[(size_t)rt.cls_plug * a.n_words + w] & bit)) ff = 2;
        else if (rt.cls_con && !(a.con[(size_t)rt.cls_con * a.n_words + w] & bit)) ff = 3;
        else if (rt.cls_plat && !(a.plat[(size_t)rt.cls_plat * a.n_words + w] & bit)) ff = 4;
        else {
            bool port_busy = false;
            if (rt.flags & RT_PORTS)
                for (u32 q = pp0; q < pp1; ++q)
                    if ((a.portmap[(size_t)a.pset_ids[q] * a.n_words + w] & bit) && !((q - pp0 < 32) && (port_later >> (q - pp0) & 1u))) port_busy = true;
            if (port_busy) ff = 5;
            else if (rt.flags & RT_MAXREP) {
                u32 sv = 0;
                for (u32 z = a.list_off[rt.svc]; z < a.list_off[rt.svc + 1]; ++z)
                    if (a.list_node[z] == n) { sv = a.list_svc[z]; break; }
                u32 at = sv - svc_later;
                if (!((u64)at < rt.maxrep)) ff = 6;
            }
        }
    }
    for (int f = 0; f < 7; ++f) {
        u64 bm = ballot64(ff == f);
        if (bm && (threadIdx.x & 63) == 0) atomicAdd(&a.hist[(size_t)gj * 8 + f], (u32)__popcll(bm));
    }
}

// ---------------------------------------------------------------------------------------------
// k_commit — NodeInfo.addTask / removeTask arithmetic for placements decided outside the engine
// (nodeinfo.go:66-154). Several placements may hit one node: integer atomics commute.
// ---------------------------------------------------------------------------------------------
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
};
__global__ void k_check_pair(CheckArgs a) {
    if (threadIdx.x != 0) return;
    u32 n = a.node, w = n >> 6;
    u64 bit = 1ull << (n & 63);
    int ff = -1;
    if (!(a.valid[w] & bit)) ff = -2;
    else if (!(a.ready[w] & bit)) ff = 0;
    else if ((a.rt.flags & RT_RES) && !(a.rt.cpu <= a.cpu[n] && a.rt.mem <= a.mem[n])) ff = 1;
    else if (a.rt.cls_plug && !(a.plug[(size_t)a.rt.cls_plug * a.n_words + w] & bit)) ff = 2;
    else if (a.rt.cls_con && !(a.con[(size_t)a.rt.cls_con * a.n_words + w] & bit)) ff = 3;
    else if (a.rt.cls_plat && !(a.plat[(size_t)a.rt.cls_plat * a.n_words + w] & bit)) ff = 4;
    else if ((a.rt.flags & RT_PORTS) && a.port_busy) ff = 5;
    else if ((a.rt.flags & RT_MAXREP) && !((u64)a.svc_count < a.rt.maxrep)) ff = 6;
    *a.out = ff;
}

}  // namespace swpdev
