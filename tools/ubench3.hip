// Cross-unit latencies of one wavefront on gfx950: the sequences k_resolve3's pick is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define REP8(x) x x x x x x x x
template <int T>
__global__ void k(u64* out, int iters, u32 seed) {
    u32 a = seed + threadIdx.x, b = seed * 3 + 1;
    u64 w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (T == 0) {   // dependent VALU
            REP8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if (T == 1) {   // VALU compare -> SGPR pair -> SALU -> VALU (round trip)
            REP8(asm volatile("v_cmp_ne_u32 s[4:5], %0, %1\n s_and_b32 s6, s4, 1\n v_add_u32 %0, s6, %0" : "+v"(a) : "v"(b) : "s4", "s5", "s6", "scc");)
        } else if (T == 2) {   // s_ff1 -> v_readlane (SGPR lane select) -> SALU
            REP8(asm volatile("s_ff1_i32_b64 s4, exec\n s_nop 3\n v_readlane_b32 s5, %0, s4\n s_add_u32 s6, s5, 1\n v_add_u32 %0, s6, %0" : "+v"(a) : : "s4", "s5", "s6", "scc");)
        } else if (T == 3) {   // v_cmp -> vcc -> not-taken branch
            REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_cbranch_vccnz 1f\n v_add_u32 %0, 1, %0\n1:" : "+v"(a) : "v"(b) : "vcc");)
        } else if (T == 4) {   // taken scalar branch
            REP8(asm volatile("s_cmp_eq_u32 s4, s4\n s_cbranch_scc1 1f\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n1: v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b) : "s4", "scc");)
        } else if (T == 5) {   // readfirstlane -> SALU -> VALU
            REP8(asm volatile("v_readfirstlane_b32 s4, %0\n s_add_u32 s5, s4, 1\n v_add_u32 %0, s5, %0" : "+v"(a) : : "s4", "s5", "scc");)
        } else if (T == 6) {   // v_cmp_ne_u64 -> s_cmp_eq_u64 -> s_cbranch_scc1 (not taken) -> v
            REP8(asm volatile("v_cmp_ne_u32 s[4:5], %0, %1\n s_cmp_eq_u64 s[4:5], 0\n s_cbranch_scc1 1f\n v_add_u32 %0, 1, %0\n1:" : "+v"(a) : "v"(b) : "s4", "s5", "scc");)
        } else if (T == 7) {   // independent SALU x4 between dependent VALU
            REP8(asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s4, s4, 1\n s_add_u32 s5, s5, 1\n s_add_u32 s6, s6, 1\n s_add_u32 s7, s7, 1" : "+v"(a) : "v"(b) : "s4", "s5", "s6", "s7", "scc");)
        } else if (T == 8) {   // 4 independent VALU
            u32 c = a + 1, d = a + 2, e = a + 3;
            REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
            a += c + d + e;
        } else if (T == 9) {   // v_cndmask chain with vcc from v_cmp
            REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, 1, %0" : "+v"(a) : "v"(b) : "vcc");)
        } else if (T == 10) {  // dependent SALU chain
            REP8(asm volatile("s_add_u32 s4, s4, 1\n s_add_u32 s4, s4, 1\n s_add_u32 s4, s4, 1\n s_add_u32 s4, s4, 1" : : : "s4", "scc");)
        } else if (T == 11) {  // v_readlane x4 independent -> s_ff1
            REP8(asm volatile("v_readlane_b32 s4, %0, 3\n v_readlane_b32 s5, %0, 4\n v_readlane_b32 s6, %0, 5\n v_readlane_b32 s7, %0, 6\n s_or_b32 s4, s4, s5\n s_ff1_i32_b32 s4, s4\n v_add_u32 %0, s4, %0" : "+v"(a) : : "s4", "s5", "s6", "s7", "scc");)
        }
    }
    u64 w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = a; }
}
template <int T>
void run(const char* name, int ninstr, u64* d) {
    const int iters = 200000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<T>, dim3(1), dim3(64), 0, 0, d, iters, 7u);
        hipDeviceSynchronize();
    }
    u64 h[2];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double ns = h[0] * 10.0 / (iters * 8.0);
    printf("%-58s %7.2f ns per group (%d instr) = %.2f ns/instr\n", name, ns, ninstr, ns / ninstr);
}
int main(int argc, char** argv) {
    u64* d;
    hipMalloc(&d, 64);
    int t = argc > 1 ? atoi(argv[1]) : -1;
    if (t == 0) run<0>("dependent v_add", 1, d);
    if (t == 1) run<1>("v_cmp->sgpr; s_and; v_add(sgpr)", 3, d);
    if (t == 2) run<2>("s_ff1; s_nop3; v_readlane(s); s_add; v_add", 5, d);
    if (t == 3) run<3>("v_cmp vcc; s_cbranch_vccnz (not taken); v_add", 3, d);
    if (t == 4) run<4>("s_cmp; s_cbranch_scc1 TAKEN; v_add", 3, d);
    if (t == 5) run<5>("v_readfirstlane; s_add; v_add", 3, d);
    if (t == 6) run<6>("v_cmp->s[4:5]; s_cmp_eq_u64; s_cbranch_scc1 (nt); v_add", 4, d);
    if (t == 7) run<7>("v_add + 4 independent s_add", 5, d);
    if (t == 8) run<8>("4 independent v_add", 4, d);
    if (t == 9) run<9>("v_cmp vcc; v_cndmask; v_add", 3, d);
    if (t == 10) run<10>("4 dependent s_add", 4, d);
    if (t == 11) run<11>("4 v_readlane; s_or; s_ff1; v_add", 7, d);
    fflush(stdout);
    return 0;
}
