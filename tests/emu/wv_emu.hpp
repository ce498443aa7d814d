// wv_emu.hpp — CPU implementation of the wave / workgroup primitives of swarmkit_amd/csrc/swp_wave.hpp.
//
// TEST INFRASTRUCTURE. One workgroup is run as cooperative fibers (ucontext), one per thread; a collective (ballot,
// readlane, min, wave_sync, barrier) parks the calling fiber until all 64 lanes of its wave (all threads of the
// workgroup for a barrier) have arrived at the SAME collective, then the last arriver computes the result and releases
// the others. That is enough to run the round resolver's source (swp_resolve5.hpp) unchanged and check its control
// flow, indexing and hand-shakes against a sequential model — not its timing, and not memory-ordering hazards between
// waves (fibers switch only at collectives). The product never includes this file.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <deque>
#include <functional>
#include <vector>

#define __host__
#define __device__
#define WV_DEV inline
#define WV_KERNEL(bounds)
#define WV_UNROLL

#include "../../swarmkit_amd/csrc/swp_types.hpp"

using std::max;
using std::min;

// deadlock reports name each thread's last collective: the caller's caller with -DEMU_DEEP_SITE (needs -O0 -fno-omit-frame-pointer;
// resolve with addr2line), else the wrapper itself
#ifdef EMU_DEEP_SITE
#define EMU_SITE() __builtin_return_address(1)
#else
#define EMU_SITE() __builtin_return_address(0)
#endif

namespace emu {
using swpdev::u32;
using swpdev::u64;

enum Op { OP_NONE = 0, OP_BALLOT, OP_READLANE, OP_READFIRST, OP_MIN, OP_SYNC, OP_BARRIER, OP_SCAN };

// Minimal x86-64 System V context switch (callee-saved registers + stack pointer). glibc's swapcontext makes a
// sigprocmask system call per switch, and a run makes tens of millions of switches.
struct Ctx { void* sp = nullptr; };
extern "C" void emu_switch(Ctx* from, Ctx* to);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

struct Fiber {
    Ctx ctx;
    char* stack = nullptr;
    bool done = false;
};
struct WaveSync {
    int arrived = 0;
    int op = OP_NONE;
    u64 vals[64];
    u64 aux = 0;
    u64 result[64];
};

struct Block {
    u32 nthreads = 0;
    std::vector<Fiber> fib;
    std::vector<WaveSync> waves;
    int bar_arrived = 0;
    std::deque<u32> runq;
    // EMU_SCHED_SEED=<n>: the runnable fibers are kept per WAVE and the next wave to run is drawn at random — but only at the points where
    // a wave is of one mind: when one of its collectives completes (every lane is at the same instruction), or when none of its lanes can
    // go on (they wait for another wave). In between, the wave drawn last runs all the lanes it has runnable, in their order: on the
    // device the lanes of a wave execute an instruction together, so another wave can never see the stores of lane 0's stretch without
    // those of lane 1's — the stretch between two collectives is the unit of interleaving here. At such a point the same wave is kept
    // with a probability drawn per launch (from "waves alternate" to "one wave runs as far as it can"). The default order — one queue,
    // first in first out — is ONE timing among the ones the device can produce; the hand-shakes between waves that never meet at a
    // barrier must hold under all of them.
    std::vector<std::deque<u32>> wq;
    u64 rng = 0;
    u32 sticky_wave = ~0u, stick_pct = 0;
    bool decide = true;   // the next dequeue is a point where another wave may be drawn
    size_t queued = 0;
    std::vector<u32> pollers;   // fibers parked in a polling loop (spin_pause) until the next lds_publish32
    std::vector<u64> lds;
    std::function<void()> body;
    Ctx sched;
    u32 cur = 0;
    u64 switches = 0;
    std::vector<void*> last_site;   // per thread: return address of its last collective / barrier call (deadlock report)
    std::vector<u64> ncoll;
};
inline u32& blockidx_z() {   // third grid dimension (wv::block_z())
    static u32 v = 0;
    return v;
}
inline u32& blockidx_y() {   // second grid dimension (wv::block_y()): set by the harness around a launch
    static u32 v = 0;
    return v;
}
inline u32& blockidx() {   // workgroup index seen by wv::block(): the harness runs the workgroups of a grid one after the other
    static u32 b = 0;
    return b;
}
inline Block*& B() {
    static Block* b = nullptr;
    return b;
}
inline u64 sched_seed() {
    static const u64 s = getenv("EMU_SCHED_SEED") ? strtoull(getenv("EMU_SCHED_SEED"), nullptr, 10) : 0;
    return s;
}
inline u64& launch_counter() {
    static u64 n = 0;
    return n;
}
inline u64 next_rand(Block* b) {   // xorshift64*
    b->rng ^= b->rng >> 12;
    b->rng ^= b->rng << 25;
    b->rng ^= b->rng >> 27;
    return b->rng * 0x2545F4914F6CDD1Dull;
}
inline void enqueue(Block* b, u32 t) {
    if (sched_seed() == 0) { b->runq.push_back(t); return; }
    b->wq[t >> 6].push_back(t);
    b->queued++;
}
inline bool nothing_runnable(Block* b) { return sched_seed() == 0 ? b->runq.empty() : b->queued == 0; }
inline u32 dequeue(Block* b) {
    if (sched_seed() == 0) {
        const u32 t = b->runq.front();
        b->runq.pop_front();
        return t;
    }
    u32 w = b->sticky_wave;
    if (b->decide || w >= b->wq.size() || b->wq[w].empty()) {
        if (w >= b->wq.size() || b->wq[w].empty() || next_rand(b) % 100 >= b->stick_pct) {
            u32 cand[64], n = 0;
            for (u32 k = 0; k < b->wq.size(); ++k)
                if (!b->wq[k].empty()) cand[n++] = k;
            w = cand[next_rand(b) % n];
            b->sticky_wave = w;
        }
        b->decide = false;
    }
    const u32 t = b->wq[w].front();
    b->wq[w].pop_front();
    b->queued--;
    return t;
}

inline void yield_until_publish() {   // a polling loop: parked until some fiber publishes a flag (wake_pollers), then it polls again
    Block* b = B();
    b->switches++;
    b->pollers.push_back(b->cur);
    emu_switch(&b->fib[b->cur].ctx, &b->sched);
}
inline void wake_pollers() {
    Block* b = B();
    for (u32 t : b->pollers) enqueue(b, t);
    b->pollers.clear();
}
inline void yield_blocked() {   // park the current fiber; somebody else re-queues it
    Block* b = B();
    b->switches++;
    emu_switch(&b->fib[b->cur].ctx, &b->sched);
}

inline void fiber_main() {
    Block* b = B();
    b->body();
    b->fib[b->cur].done = true;
    emu_switch(&b->fib[b->cur].ctx, &b->sched);
    abort();   // a finished fiber is never resumed
}

// run one workgroup of `nthreads` threads with `lds_bytes` of dynamic LDS
inline void launch(u32 nthreads, size_t lds_bytes, std::function<void()> body) {
    Block blk;
    B() = &blk;
    blk.nthreads = nthreads;
    blk.fib.resize(nthreads);
    blk.waves.resize((nthreads + 63) / 64);
    blk.last_site.assign(nthreads, nullptr);
    blk.ncoll.assign(nthreads, 0);
    blk.lds.assign((lds_bytes + 7) / 8 + 8, 0xCDCDCDCDCDCDCDCDull);   // LDS is NOT zero-initialised on the device either
    blk.body = body;
    if (sched_seed() != 0) {
        blk.wq.resize(blk.waves.size());
        blk.rng = (sched_seed() * 0x9E3779B97F4A7C15ull) ^ (++launch_counter() * 0xD1B54A32D192ED03ull) ^ 1;
        static const u32 pcts[6] = {0, 50, 90, 99, 100, 97};
        blk.stick_pct = pcts[next_rand(&blk) % 6];
        if (getenv("EMU_SCHED_STICK")) blk.stick_pct = (u32)atoi(getenv("EMU_SCHED_STICK"));   // (a fixed regime, for narrowing a failure down)
    }
    const size_t STK = 256 * 1024;
    // one stack area for the process, kept between launches: a run makes thousands of launches, and fresh mappings would be
    // paged in (zeroed) again every time
    static char* pool = nullptr;
    static size_t pool_bytes = 0;
    if (pool_bytes < STK * nthreads) {
        if (pool) munmap(pool, pool_bytes);
        pool_bytes = STK * nthreads;
        pool = (char*)mmap(nullptr, pool_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (pool == MAP_FAILED) { perror("mmap"); abort(); }
    }
    char* stacks = pool;
    for (u32 t = 0; t < nthreads; ++t) {
        Fiber& f = blk.fib[t];
        f.stack = stacks + STK * t;
        // initial frame: six zeroed callee-saved registers, then the entry point as the return address; the stack is
        // 16-byte aligned + 8 at function entry, as after a call
        void** sp = reinterpret_cast<void**>(f.stack + STK - 64);
        *--sp = nullptr;                       // fake return address of fiber_main (never used)
        *--sp = (void*)fiber_main;
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.ctx.sp = sp;
        enqueue(&blk, t);
    }
    u32 alive = nthreads;
    while (alive) {
        if (nothing_runnable(&blk)) {
            fprintf(stderr, "emu: DEADLOCK — %u threads alive, none runnable\n", alive);
            for (size_t w = 0; w < blk.waves.size(); ++w)
                fprintf(stderr, "  wave %zu: %d lanes waiting in op %d\n", w, blk.waves[w].arrived, blk.waves[w].op);
            fprintf(stderr, "  barrier: %d arrived\n", blk.bar_arrived);
            for (u32 t = 0; t < nthreads; ++t)
                if ((t & 63) < 2 || (t & 63) > 61 || blk.last_site[t] != blk.last_site[t & ~63u])
                    fprintf(stderr, "  thread %u: %llu collectives, last at %p\n", t, (unsigned long long)blk.ncoll[t], blk.last_site[t]);
            abort();
        }
        blk.cur = dequeue(&blk);
        emu_switch(&blk.sched, &blk.fib[blk.cur].ctx);
        if (blk.fib[blk.cur].done) --alive;
    }
    B() = nullptr;
}

// wave collective: every lane contributes `v` (and the uniform `aux`); returns this lane's result
__attribute__((noinline)) inline u64 collective(int op, u64 v, u64 aux) {
    Block* b = B();
    b->last_site[b->cur] = EMU_SITE();
    b->ncoll[b->cur]++;
    const u32 t = b->cur, lane = t & 63, wave = t >> 6;
    WaveSync& ws = b->waves[wave];
    const u32 lanes = std::min<u32>(64, b->nthreads - wave * 64);
    if (ws.arrived == 0) { ws.op = op; ws.aux = aux; }
    if (ws.op != op || ws.aux != aux) {
        fprintf(stderr, "emu: wave %u diverged at a collective: lane %u calls op %d (aux %llu) while others wait in op %d (aux %llu)\n", wave, lane, op,
                (unsigned long long)aux, ws.op, (unsigned long long)ws.aux);
        abort();
    }
    ws.vals[lane] = v;
    if (++ws.arrived < (int)lanes) {
        yield_blocked();
        return ws.result[lane];
    }
    // last arriver computes
    switch (op) {
    case OP_BALLOT: {
        u64 m = 0;
        for (u32 l = 0; l < lanes; ++l)
            if (ws.vals[l]) m |= 1ull << l;
        for (u32 l = 0; l < lanes; ++l) ws.result[l] = m;
        break;
    }
    case OP_READLANE:
        for (u32 l = 0; l < lanes; ++l) ws.result[l] = ws.vals[aux & 63];
        break;
    case OP_READFIRST:
        for (u32 l = 0; l < lanes; ++l) ws.result[l] = ws.vals[0];
        break;
    case OP_MIN: {
        u64 m = ~0ull;
        for (u32 l = 0; l < lanes; ++l) m = std::min(m, ws.vals[l]);
        for (u32 l = 0; l < lanes; ++l) ws.result[l] = m;
        break;
    }
    case OP_SCAN: {
        u64 run = 0;
        for (u32 l = 0; l < lanes; ++l) { run += ws.vals[l]; ws.result[l] = run; }
        break;
    }
    default:
        break;
    }
    ws.arrived = 0;
    ws.op = OP_NONE;
    if (sched_seed() != 0) {   // the wave is of one mind here: another wave may get the next turn (this lane still goes first when its wave does)
        enqueue(b, t);
        for (u32 l = 0; l < lanes; ++l)
            if (wave * 64 + l != t) enqueue(b, wave * 64 + l);
        b->decide = true;
        yield_blocked();
        return ws.result[lane];
    }
    for (u32 l = 0; l < lanes; ++l)
        if (wave * 64 + l != t) enqueue(b, wave * 64 + l);
    return ws.result[lane];
}

__attribute__((noinline)) inline void block_barrier() {
    Block* b = B();
    b->last_site[b->cur] = EMU_SITE();
    b->ncoll[b->cur]++;
    if (b->waves[b->cur >> 6].arrived) {
        fprintf(stderr, "emu: thread %u (wave %u lane %u) reaches a barrier while %d lanes of its wave wait in collective op %d (aux %llu)\n", b->cur, b->cur >> 6,
                b->cur & 63, b->waves[b->cur >> 6].arrived, b->waves[b->cur >> 6].op, (unsigned long long)b->waves[b->cur >> 6].aux);
        abort();
    }
    if (++b->bar_arrived < (int)b->nthreads) {
        yield_blocked();
        return;
    }
    b->bar_arrived = 0;
    for (u32 t = 0; t < b->nthreads; ++t)
        if (t != b->cur) enqueue(b, t);
    if (sched_seed() != 0) {
        enqueue(b, b->cur);
        b->decide = true;
        yield_blocked();
    }
}
}  // namespace emu

namespace wv {
using swpdev::i64;
using swpdev::u32;
using swpdev::u64;

inline u32 tid() { return emu::B()->cur; }
inline u32 nthreads() { return emu::B()->nthreads; }
inline u32 lane() { return emu::B()->cur & 63u; }
inline u32 wave() { return emu::B()->cur >> 6; }
inline u32 block() { return emu::blockidx(); }
inline u32 block_y() { return emu::blockidx_y(); }
inline u32 block_z() { return emu::blockidx_z(); }
inline u64* lds() { return emu::B()->lds.data(); }

inline u64 ballot(bool p) { return emu::collective(emu::OP_BALLOT, p ? 1 : 0, 0); }
inline u32 readfirstlane(u32 v) { return (u32)emu::collective(emu::OP_READFIRST, v, 0); }
inline u32 readlane(u32 v, u32 l) { return (u32)emu::collective(emu::OP_READLANE, v, l); }
inline u64 readlane64(u64 v, u32 l) { return emu::collective(emu::OP_READLANE, v, l); }
inline u32 writelane(u32 v, u32 s, u32 l) { return lane() == l ? s : v; }
inline u32 mbcnt(u64 mask) { return (u32)__builtin_popcountll(mask & ((1ull << lane()) - 1ull)); }
inline u32 min_u32(u32 v) { return (u32)emu::collective(emu::OP_MIN, v, 0); }
inline void min4_u32(u32& a, u32& b, u32& c, u32& d) { a = min_u32(a); b = min_u32(b); c = min_u32(c); d = min_u32(d); }
inline u32 scan_incl_u32(u32 v) { return (u32)emu::collective(emu::OP_SCAN, v, 0); }
inline void barrier() { emu::block_barrier(); }
inline void wave_sync() { (void)emu::collective(emu::OP_SYNC, 0, 0); }
inline void lockstep() { (void)emu::collective(emu::OP_SYNC, 1, 1); }
inline void wait_vm() {}

inline void lds_publish32(u32* p, u32 v) { *reinterpret_cast<volatile u32*>(p) = v; emu::wake_pollers(); }
inline u32 lds_poll32(const u32* p) { return *reinterpret_cast<const volatile u32*>(p); }
template <int P> inline void setprio() {}
inline void spin_pause() { emu::yield_until_publish(); }
inline void lds_or64(u64* p, u64 v) { *p |= v; }
inline void lds_xor64(u64* p, u64 v) { *p ^= v; }
inline void lds_or32(u32* p, u32 v) { *p |= v; }
inline void lds_andn64(u64* p, u64 v) { *p &= ~v; }
inline void lds_add32(u32* p, u32 v) { *p += v; }
inline void lds_min64(u64* p, u64 v) { if (v < *p) *p = v; }
inline u64 lds_read64(const u64* p) { return *reinterpret_cast<const volatile u64*>(p); }
inline void lds_add_release32(u32* p, u32 v) { *reinterpret_cast<volatile u32*>(p) += v; emu::wake_pollers(); }

inline void g_add64(i64* p, i64 v) { *p += v; }
inline void g_add32(u32* p, u32 v) { *p += v; }
inline void g_or64(u64* p, u64 v) { *p |= v; }
inline void g_min64(u64* p, u64 v) { if (v < *p) *p = v; }
inline void g_xor64(u64* p, u64 v) { *p ^= v; }
inline void g_andn64(u64* p, u64 v) { *p &= ~v; }
inline void g_max32(u32* p, u32 v) { if (v > *p) *p = v; }
inline u32 g_exch32(u32* p, u32 v) { u32 o = *p; *p = v; return o; }
inline u64 g_fresh64(const u64* p) { return *p; }
inline u32 g_fresh32(const u32* p) { return *p; }
inline i64 g_fresh64s(const i64* p) { return *p; }
inline void g_store32_fresh(u32* p, u32 v) { *p = v; }
inline u32 prefetch_l2(const void* p) { return *reinterpret_cast<const volatile u32*>(p); }
inline void keep(u32) {}
template <class T>
inline T uload(const T* p) { return *p; }

// reference semantics of the hand-scheduled matcher walk of swp_wave.hpp, on the collectives above: every lane in [from, 64) in order; a lane whose current half-word is empty steps to its next one
// (together with every later lane in the same state); stop in front of the first lane that has no candidate in either
#define WV_DUMMY_W 0x80000000u
inline u32 match_seq64(u32& bits, u32& w, u32& bits2, u32 w2, u32& pickb, u32 lid, u32 from) {
    (void)lid;
    for (u32 l = from; l < 64; ++l) {
        u32 sb = readlane(bits, l);
        if (sb == 0) {
            if (lane() >= l && bits == 0) { bits = bits2; w = w2; bits2 = 0; }
            sb = readlane(bits, l);
            if (sb == 0) return l;
        }
        const u32 sw = readlane(w, l);
        if (lane() == l) pickb = sb;
        const u32 sm = sb & (0u - sb);
        if (w == sw) bits &= ~sm;
        if (w2 == sw) bits2 &= ~sm;
    }
    return 64;
}

inline u64 clock64() { return 0; }

inline int ffs64(u64 v) { return __builtin_ffsll((long long)v) - 1; }
inline int popc64(u64 v) { return __builtin_popcountll(v); }
inline int clz32(u32 v) { return v ? __builtin_clz(v) : 32; }
}  // namespace wv
