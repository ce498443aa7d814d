// swp_wave.hpp — the wave64 / workgroup primitives the round resolver (swp_resolve5.hpp) is written against, gfx950 build.
//
// The kernel source uses ONLY these wrappers for everything that is not plain C++ on registers and pointers
// (cross-lane traffic, LDS / global atomics, barriers, uniform loads). tests/emu/wv_emu.hpp implements the same
// interface on CPU fibers, so the kernel's control flow, indexing and protocol can be run against a sequential model
// without a GPU (tests/test_emu_resolve5.py). That harness is test infrastructure: the product only ever builds this
// header. Rule the kernel follows so that both agree: collectives (ballot, readlane, min, barrier) are only called
// from wave-uniform control flow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "swp_types.hpp"

#define WV_DEV __device__ __forceinline__
#define WV_KERNEL(bounds) __global__ __launch_bounds__(bounds)
#define WV_UNROLL _Pragma("unroll")

namespace wv {
using swpdev::i64;
using swpdev::u32;
using swpdev::u64;

WV_DEV u32 tid() { return threadIdx.x; }
WV_DEV u32 nthreads() { return blockDim.x; }
WV_DEV u32 lane() { return threadIdx.x & 63u; }
// wave index as a scalar (threadIdx.x >> 6 is uniform, the compiler does not always know)
WV_DEV u32 wave() { return (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
WV_DEV u32 block() { return blockIdx.x; }   // workgroup index of a multi-workgroup launch (swp_resolve6.hpp)
WV_DEV u32 block_z() { return blockIdx.z; }
WV_DEV u32 block_y() { return blockIdx.y; } // second grid dimension: the shard of a launch that covers several (swp_resolve7.hpp)
WV_DEV u64* lds() {
    extern __shared__ u64 wv_lds_[];
    return wv_lds_;
}

WV_DEV u64 ballot(bool p) { return __ballot(p); }
WV_DEV u32 readfirstlane(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
WV_DEV u32 readlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
WV_DEV u64 readlane64(u64 v, u32 l) {
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, (int)l);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), (int)l);
    return ((u64)hi << 32) | lo;
}
// v with lane l replaced by the uniform value s
WV_DEV u32 writelane(u32 v, u32 s, u32 l) { return (threadIdx.x & 63u) == l ? s : v; }   // v_cmp + v_cndmask (no builtin for v_writelane here)
// number of set bits of `mask` that belong to lanes below this one
WV_DEV u32 mbcnt(u64 mask) { return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u)); }

template <int CTRL, int ROW_MASK = 0xf>
WV_DEV u32 dpp_(u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false); }
// min over the 64 lanes; every lane returns it (six DPP steps + one readlane)
WV_DEV u32 min_u32(u32 v) {
    v = min(v, dpp_<0x111>(v));
    v = min(v, dpp_<0x112>(v));
    v = min(v, dpp_<0x114>(v));
    v = min(v, dpp_<0x118>(v));
    v = min(v, dpp_<0x142, 0xa>(v));
    v = min(v, dpp_<0x143, 0xc>(v));
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// the wave's minimum of FOUR values at once; every lane gets the four minima. Four independent DPP chains, interleaved: one fused
// instruction per step and value (dst = min(dst, its DPP source); a lane without a source is left alone), and no chain ever waits for the
// VALU-write -> DPP-read hazard because three other instructions lie in between. (The compiler's own code for dpp_ + min is a mov, a
// mov_dpp, a min and a nop per step: k_scanb's eight reductions a batch were 200 of its instructions.)
WV_DEV void min4_u32(u32& a, u32& b, u32& c, u32& d) {
#define WV_M4(ctrl)                                                                                                                     \
    "v_min_u32_dpp %0, %0, %0 " ctrl "\n\tv_min_u32_dpp %1, %1, %1 " ctrl "\n\tv_min_u32_dpp %2, %2, %2 " ctrl "\n\tv_min_u32_dpp %3, %3, %3 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t" WV_M4("row_shr:1 row_mask:0xf bank_mask:0xf") WV_M4("row_shr:2 row_mask:0xf bank_mask:0xf") WV_M4("row_shr:4 row_mask:0xf bank_mask:0xf")
                     WV_M4("row_shr:8 row_mask:0xf bank_mask:0xf") WV_M4("row_bcast:15 row_mask:0xa bank_mask:0xf") WV_M4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef WV_M4
    a = (u32)__builtin_amdgcn_readlane((int)a, 63);
    b = (u32)__builtin_amdgcn_readlane((int)b, 63);
    c = (u32)__builtin_amdgcn_readlane((int)c, 63);
    d = (u32)__builtin_amdgcn_readlane((int)d, 63);
}

// inclusive prefix sum over the 64 lanes (lane l returns v[0] + ... + v[l]): four DPP row shifts inside the rows of 16, then the three
// row totals by readlane (k_groups2's helpers: the offsets of the node words in a group's compact candidate list)
WV_DEV u32 scan_incl_u32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1 (a lane without a source adds `old` = 0)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    const u32 r0 = (u32)__builtin_amdgcn_readlane((int)v, 15), r1 = (u32)__builtin_amdgcn_readlane((int)v, 31), r2 = (u32)__builtin_amdgcn_readlane((int)v, 47);
    const u32 l = threadIdx.x & 63u;
    return v + (l >= 16u ? r0 : 0u) + (l >= 32u ? r1 : 0u) + (l >= 48u ? r2 : 0u);
}

// workgroup barrier that orders LDS only: outstanding global loads / fire-and-forget atomics stay in flight
WV_DEV void barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the wave's LDS operations issued so far are done before any lane goes on: what one lane wrote (or or-ed) is what another
// lane of the same wave reads next. The hardware executes one wave's LDS instructions in order; this pins the compiler too.
WV_DEV void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// a point every lane of the wave passes together. The hardware runs a wave in lockstep, so this is only a compiler fence here;
// the CPU emulation, whose lanes are separate fibers, rendezvous at it (e.g. all lanes have READ a word before one lane rewrites it).
WV_DEV void lockstep() { asm volatile("" ::: "memory"); }
// this wave's global stores / atomics / loads have completed (what a later wave behind a barrier may rely on)
WV_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- LDS atomics (no return value: nothing to wait for) ----
WV_DEV void lds_or64(u64* p, u64 v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void lds_xor64(u64* p, u64 v) { __hip_atomic_fetch_xor(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void lds_or32(u32* p, u32 v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void lds_andn64(u64* p, u64 v) { __hip_atomic_fetch_and(p, ~v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void lds_min64(u64* p, u64 v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// a read behind a barrier, not to be merged with an earlier one: the compiler fence does that — a `volatile` access would lose the
// pointer's address space and come out as a FLAT load with a wait of its own (found in k_scanb: four of them in a row, ~2 000 cycles a batch)
WV_DEV u64 lds_read64(const u64* p) {
    asm volatile("" ::: "memory");
    return *p;
}
WV_DEV void lds_add32(u32* p, u32 v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- flags between waves of one workgroup that do NOT meet at a barrier (k_r6_commit: the matching wave runs on while the others apply) ----
// publish: everything this wave wrote to LDS before is visible to a wave that polls the new value; poll + spin_pause: the waiting side
WV_DEV void lds_publish32(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV u32 lds_poll32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// a counter several waves bump when their share of a job is done (k_groups2's helper waves): everything the wave wrote before —
// LDS and global — is visible to the wave that polls the sum
WV_DEV void lds_add_release32(u32* p, u32 v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// issue priority of this wave among the waves of its SIMD (0 lowest .. 3 highest): the one wave that carries a serial chain asks for 3
template <int P> WV_DEV void setprio() { __builtin_amdgcn_s_setprio(P); }
WV_DEV void spin_pause() { __builtin_amdgcn_s_sleep(8); }   // ~0.2 µs off the CU's issue slots between two polls

// ---- global memory ----
// Everything the kernel exchanges through memory is exchanged between waves of ONE workgroup, so workgroup scope is all the
// coherence it needs: the operations stay in this XCD's L2 (device scope would send every load past it, ~1 µs each).
// What other kernels wrote is visible from the launch on, what this one writes at its end.
WV_DEV void g_add64(i64* p, i64 v) { __hip_atomic_fetch_add(reinterpret_cast<u64*>(p), (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void g_add32(u32* p, u32 v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void g_or64(u64* p, u64 v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void g_min64(u64* p, u64 v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void g_xor64(u64* p, u64 v) { __hip_atomic_fetch_xor(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_DEV void g_andn64(u64* p, u64 v) { __hip_atomic_fetch_and(p, ~v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_DEV void g_max32(u32* p, u32 v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_DEV u32 g_exch32(u32* p, u32 v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// loads that must see what other waves of this workgroup wrote through L2 (bypass the CU's vector L1)
WV_DEV u64 g_fresh64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV u32 g_fresh32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV i64 g_fresh64s(const i64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void g_store32_fresh(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// pull the cache line at p into this XCD's L2. An ordinary load whose value the caller must consume (sum it up and hand the
// sum to keep()): the compiler then orders the wait itself. (A bare global_load in inline asm returns into a register the
// compiler believes free — the late write corrupted whatever lived there next: found by the 12 400-node scan-mode case.)
WV_DEV u32 prefetch_l2(const void* p) { return *reinterpret_cast<const volatile u32*>(p); }
// keeps a value alive without doing anything with it
WV_DEV void keep(u32 v) { asm volatile("" ::"v"(v)); }
// wave-uniform read-only load: constant address space → s_load through the scalar cache
template <class T>
WV_DEV T uload(const T* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

// The matcher's walk (swp_resolve5.hpp, swp_resolve6.hpp, swp_resolve7.hpp), hand-scheduled: the serial chain of the whole engine.
// Round 2's version was a scalar loop of 17 instructions per task over a mask of lanes still to serve (s_ff1 for the next lane, m0 as
// the lane select of v_writelane); this one is unrolled over the 64 lanes.
// Lane indices are immediates, so the scalar side no longer
// finds the next lane (s_ff1 on a todo mask), clears it, moves it to m0 or branches back, and a lane carries TWO half-words — the
// current one (bits, w) and the next of its list (bits2, w2), both struck by every pick — so that a lane whose current half-word
// ran empty steps to the next one inside the walk (11 instructions, no LDS, no exit). 13 instructions per task, no exec-mask
// round trips (two compare -> select -> clear chains interleaved, so that no instruction waits for the one before it):
//   v_readlane sb ← bits[L];  v_readlane sw ← w[L];  s_sub t ← 0 − sb;  s_and sm ← sb & t  (the lowest candidate, SCC = any);
//   s_cbranch_scc0 step_L;  v_cmp vcc ← (w == sw);  v_cmp m2 ← (w2 == sw);  v_mov vsm ← sm;  v_cndmask t1 ← vcc ? vsm : 0;
//   v_cndmask t2 ← m2 ? vsm : 0;  v_bfi bits &= ~t1;  v_bfi bits2 &= ~t2;  v_writelane pickb[L] ← sb
//   step_L: every lane >= L with empty bits takes (bits2, w2) and leaves bits2 empty; lane L is tried again; still empty: stop.
// Every lane in [from, 64) is walked in order. A lane the caller does not want served carries a dummy (bits = 1, bits2 = 0,
// w = WV_DUMMY_W | lane: no real half-word index has bit 31, so nobody else is struck); the walk stops in front of the first lane whose
// two half-words are both empty and returns its index (64: walked to the end) — the caller's stop sentinel is simply bits = bits2 = 0
// in the lane behind its last task.
//   pickb[L] = lane L's candidate bits at its turn: the node it took is w[L] * 32 + ctz(pickb[L]) (a served lane never moves).
//   lid = the lane index in a register. Entry at lane `from` is a computed jump: every body is WV_MB_BYTES = 80 bytes
//   (tools/check_matcher_asm.sh disassembles and checks). s[94:95] holds the jump target (clobbered); exec is all ones afterwards.
#define WV_DUMMY_W 0x80000000u
#define WV_MB(L)                                                                                                                       \
    "3" #L ":\n\t"                                                                                                                     \
    "v_readlane_b32 %[sb], %[bits], " #L "\n\t"                                                                                        \
    "v_readlane_b32 %[sw], %[w], " #L "\n\t"                                                                                           \
    "s_sub_u32 %[st], 0, %[sb]\n\t"                                                                                                    \
    "s_and_b32 %[sm], %[sb], %[st]\n\t"                                                                                                \
    "s_cbranch_scc0 4" #L "f\n\t"                                                                                                      \
    "v_cmp_eq_u32_e32 vcc, %[sw], %[w]\n\t"                                                                                            \
    "v_cmp_eq_u32_e64 %[m2], %[sw], %[w2]\n\t"                                                                                         \
    "v_mov_b32_e32 %[vsm], %[sm]\n\t"                                                                                                  \
    "v_cndmask_b32_e32 %[t1], 0, %[vsm], vcc\n\t"                                                                                      \
    "v_cndmask_b32_e64 %[t2], 0, %[vsm], %[m2]\n\t"                                                                                    \
    "v_bfi_b32 %[bits], %[t1], 0, %[bits]\n\t"                                                                                         \
    "v_bfi_b32 %[bits2], %[t2], 0, %[bits2]\n\t"                                                                                       \
    "v_writelane_b32 %[pickb], %[sb], " #L "\n\t"
#define WV_MS(L)                                                                                                                       \
    "4" #L ":\n\t"                                                                                                                     \
    "v_cmpx_le_u32_e32 vcc, " #L ", %[lid]\n\t"                                                                                        \
    "v_cmpx_eq_u32_e32 vcc, 0, %[bits]\n\t"                                                                                            \
    "v_mov_b32_e32 %[bits], %[bits2]\n\t"                                                                                              \
    "v_mov_b32_e32 %[w], %[w2]\n\t"                                                                                                    \
    "v_mov_b32_e32 %[bits2], 0\n\t"                                                                                                    \
    "s_mov_b64 exec, -1\n\t"                                                                                                           \
    "v_readlane_b32 %[sb], %[bits], " #L "\n\t"                                                                                        \
    "s_cmp_eq_u32 %[sb], 0\n\t"                                                                                                        \
    "s_cbranch_scc0 3" #L "b\n\t"                                                                                                      \
    "s_movk_i32 %[at], " #L "\n\t"                                                                                                     \
    "s_branch 99f\n\t"
#define WV_REP64(F)                                                                                                                    \
    F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17) F(18) F(19) F(20) F(21) F(22)    \
    F(23) F(24) F(25) F(26) F(27) F(28) F(29) F(30) F(31) F(32) F(33) F(34) F(35) F(36) F(37) F(38) F(39) F(40) F(41) F(42) F(43)      \
    F(44) F(45) F(46) F(47) F(48) F(49) F(50) F(51) F(52) F(53) F(54) F(55) F(56) F(57) F(58) F(59) F(60) F(61) F(62) F(63)
#define WV_MB_BYTES 80
WV_DEV u32 match_seq64(u32& bits, u32& w, u32& bits2, u32 w2, u32& pickb, u32 lid, u32 from) {
    u32 at, sb, sw, sm, st, vsm, t1, t2;
    u64 m2;
    from = (u32)__builtin_amdgcn_readfirstlane((int)from);
    asm volatile(
        "s_setprio 3\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_getpc_b64 s[94:95]\n"
        "9:\n\t"
        "s_mul_i32 %[st], %[from], 80\n\t"
        "s_add_u32 %[st], %[st], 30f-9b\n\t"
        "s_add_u32 s94, s94, %[st]\n\t"
        "s_addc_u32 s95, s95, 0\n\t"
        "s_setpc_b64 s[94:95]\n\t"
        WV_REP64(WV_MB)
        "s_movk_i32 %[at], 64\n\t"
        "s_branch 99f\n\t"
        WV_REP64(WV_MS)
        "99:\n\t"
        "s_setprio 0\n\t"
        : [bits] "+v"(bits), [w] "+v"(w), [bits2] "+v"(bits2), [pickb] "+v"(pickb), [at] "=&s"(at), [sb] "=&s"(sb), [sw] "=&s"(sw), [sm] "=&s"(sm),
          [st] "=&s"(st), [m2] "=&s"(m2), [vsm] "=&v"(vsm), [t1] "=&v"(t1), [t2] "=&v"(t2)
        : [w2] "v"(w2), [lid] "v"(lid), [from] "s"(from)
        : "vcc", "scc", "s94", "s95", "memory");
    return at;
}

// shader clock (s_memtime); used by the kernel's section timers when ResolveArgs.dbg & 16
WV_DEV u64 clock64() { return __builtin_amdgcn_s_memtime(); }

WV_DEV int ffs64(u64 v) { return __ffsll((long long)v) - 1; }   // index of the lowest set bit (v != 0)
WV_DEV int popc64(u64 v) { return __popcll(v); }
WV_DEV int clz32(u32 v) { return __clz((int)v); }   // 32 for v == 0

}  // namespace wv
