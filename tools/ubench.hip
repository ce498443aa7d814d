// tools/ubench.hip — cost model of single-wave primitives on gfx950 (developer aid, not product).
// hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench && ./tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define KEEP(x) asm volatile("" : "+v"(x))
#define KEEPS(x) asm volatile("" : "+s"(x))

template <int MODE>
__global__ void k(u64* out, u32* gmem, int iters, u32 seed) {
    __shared__ u64 lds[1024];
    const u32 lane = threadIdx.x & 63;
    u32 a = seed + lane, b = seed * 3 + lane, c = lane, d = seed;
    u64 p = ((u64)a << 32) | b, q = ~p;
    lds[threadIdx.x & 1023] = p;
    __syncthreads();
    u32 s = __builtin_amdgcn_readfirstlane(seed);
    u64 t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) {   // 8 dependent VALU
#pragma unroll
            for (int j = 0; j < 8; ++j) { a = a + b; KEEP(a); }
        } else if (MODE == 2) {   // 8 independent VALU
            u32 x0 = a, x1 = b, x2 = c, x3 = d;
#pragma unroll
            for (int j = 0; j < 2; ++j) { x0 += 1; x1 += 2; x2 += 3; x3 += 4; KEEP(x0); KEEP(x1); KEEP(x2); KEEP(x3); }
            a = x0; b = x1; c = x2; d = x3;
        } else if (MODE == 3) {   // 8 dependent 64-bit and/andn
#pragma unroll
            for (int j = 0; j < 8; ++j) { p = p & ~q; q = q ^ p; asm volatile("" : "+v"(p), "+v"(q)); }
        } else if (MODE == 4) {   // uniform taken branch
            s = s * 1664525u + 1013904223u; KEEPS(s);
            if (s & 0x10000u) { a += 1; KEEP(a); } else { b += 1; KEEP(b); }
        } else if (MODE == 5) {   // ballot + ff1 + readlane
            u64 m = __ballot((a ^ (u32)i) & 1u);
            int l = m ? __ffsll((long long)m) - 1 : 0;
            u32 v = __builtin_amdgcn_readlane(b, l);
            a += v; KEEP(a);
        } else if (MODE == 6) {   // LDS write then dependent read
            lds[lane] = p; p = lds[(lane + 1) & 63] + 1; asm volatile("" : "+v"(p));
        } else if (MODE == 7) {   // 64-bit compare -> scalar branch
            p += 1; asm volatile("" : "+v"(p));
            if (__ballot(p != 0) != 0) { a += 1; KEEP(a); }
        } else if (MODE == 8) {   // DPP min reduce (6 steps) + readlane
            u32 v = a ^ (u32)i;
            v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
            v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
            v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
            v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
            v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
            v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
            a += __builtin_amdgcn_readlane(v, 63); KEEP(a);
        } else if (MODE == 9) {   // one fire-and-forget global atomic from one lane
            if (lane == (u32)(i & 63)) atomicAdd(gmem + (i & 1023), 1u);
        } else if (MODE == 10) {   // one fire-and-forget global store from one lane
            if (lane == (u32)(i & 63)) gmem[2048 + (i & 1023)] = a;
        } else if (MODE == 11) {   // workgroup barrier (launch with 256 threads)
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 12) {   // LDS word exchange between waves: write, barrier, read (256 threads)
            if (lane == 0) lds[(i & 1) * 8 + (threadIdx.x >> 6)] = p + i;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            u64 r = lds[(i & 1) * 8] + lds[(i & 1) * 8 + 1] + lds[(i & 1) * 8 + 2] + lds[(i & 1) * 8 + 3];
            p += r; asm volatile("" : "+v"(p));
        } else if (MODE == 13) {   // 8 SALU dependent
#pragma unroll
            for (int j = 0; j < 8; ++j) { s = s * 5u + 1u; KEEPS(s); }
        } else if (MODE == 14) {   // v_readfirstlane of a VALU result feeding a scalar branch
            a += 1; KEEP(a);
            u32 u = __builtin_amdgcn_readfirstlane(a);
            if (u & 1u) { b += 1; KEEP(b); }
        } else if (MODE == 15) {   // exec-masked block: if (lane == x) { valu }
            if (lane == (u32)(i & 63)) { a += b; KEEP(a); }
        } else if (MODE == 16) {   // v_cndmask chain with scalar condition (select by uniform index), 6 selects
            u32 sel = (u32)i % 3u;
            u32 w = sel == 0 ? a : (sel == 1 ? b : c);
            d += w; KEEP(d);
        }
    }
    u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + b + c + d + (u32)p + (u32)q + s; }
}

template <int MODE>
void run(const char* name, int threads, u64* d_out, u32* d_g) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d_out, d_g, iters, 12345u);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d_out, d_g, iters, 12345u);
    u64 h[2];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f cycles/iter\n", name, (double)h[0] / iters);
}

int main() {
    u64* d_out; u32* d_g;
    hipMalloc(&d_out, 64); hipMalloc(&d_g, 1 << 16); hipMemset(d_g, 0, 1 << 16);
    run<0>("0  empty loop", 64, d_out, d_g);
    run<1>("1  8 dependent v_add_u32", 64, d_out, d_g);
    run<2>("2  8 independent v_add_u32", 64, d_out, d_g);
    run<3>("3  8 dependent (64-bit and-not, xor)", 64, d_out, d_g);
    run<13>("13 8 dependent SALU mul-add", 64, d_out, d_g);
    run<4>("4  uniform if/else on scalar (random direction)", 64, d_out, d_g);
    run<14>("14 VALU -> readfirstlane -> scalar branch", 64, d_out, d_g);
    run<7>("7  v_cmp_u64 -> ballot -> scalar branch", 64, d_out, d_g);
    run<15>("15 exec-masked one-lane block", 64, d_out, d_g);
    run<16>("16 select by uniform index (2 cndmask)", 64, d_out, d_g);
    run<5>("5  ballot + ff1 + readlane + add", 64, d_out, d_g);
    run<8>("8  DPP min-reduce (6 steps) + readlane", 64, d_out, d_g);
    run<6>("6  LDS write + dependent read", 64, d_out, d_g);
    run<9>("9  one-lane global atomicAdd (no return)", 64, d_out, d_g);
    run<10>("10 one-lane global store", 64, d_out, d_g);
    run<11>("11 s_barrier, 4 waves", 256, d_out, d_g);
    run<12>("12 LDS publish + s_barrier + read 4 words, 4 waves", 256, d_out, d_g);
    run<11>("11 s_barrier, 2 waves", 128, d_out, d_g);
    return 0;
}
