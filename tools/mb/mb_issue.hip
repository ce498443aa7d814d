// Issue-rate / latency microbenchmark for single-wave code on gfx950: cycles (s_memtime) per instruction of short sequences.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb/mb_issue.hip -o /tmp/mb_issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define T0 "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
#define T1 "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
#define OUTS [t0] "=&s"(t0), [t1] "=&s"(t1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3)
#define CLOB "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "vcc", "scc", "memory"

#define CASE(ID, N, BODY)                                                                         \
    if (which == ID) {                                                                            \
        asm volatile(T0 ".rept " #N "\n\t" BODY ".endr\n\t" T1 : OUTS : : CLOB);                    \
    }

__global__ __launch_bounds__(64) void k(int which, unsigned long long* out, unsigned* sink) {
    unsigned long long t0 = 0, t1 = 0;
    unsigned v0 = threadIdx.x * 7 + 1, v1 = threadIdx.x, v2 = 3, v3 = 5;
    asm volatile("s_mov_b32 s40, 1\n\ts_mov_b32 s41, 2\n\ts_mov_b32 s42, 3\n\ts_mov_b32 s43, 4\n\ts_mov_b32 s44, 0\n\ts_mov_b32 s45, 0\n\ts_mov_b32 s46,0\n\ts_mov_b32 s47,0\n\t" ::: CLOB);
    CASE(0, 1024, "s_add_u32 s40, s40, 1\n\t")                                                                // dependent SALU
    CASE(1, 256, "s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\t")   // independent SALU
    CASE(2, 1024, "v_add_u32_e32 %[v0], 1, %[v0]\n\t")                                                         // dependent VALU
    CASE(3, 256, "v_add_u32_e32 %[v0], 1, %[v0]\n\tv_add_u32_e32 %[v1], 1, %[v1]\n\tv_add_u32_e32 %[v2], 1, %[v2]\n\tv_add_u32_e32 %[v3], 1, %[v3]\n\t")
    CASE(4, 256, "v_readlane_b32 s40, %[v0], 3\n\ts_add_u32 s41, s40, 1\n\tv_mov_b32_e32 %[v1], s41\n\tv_add_u32_e32 %[v0], %[v1], %[v0]\n\t")   // VALU->SGPR->SALU->VALU round trip, 4 instr
    CASE(5, 256, "v_readlane_b32 s40, %[v0], 3\n\tv_mov_b32_e32 %[v1], s40\n\tv_add_u32_e32 %[v0], %[v1], %[v0]\n\t")                        // VALU->SGPR->VALU, 3 instr
    CASE(6, 256, "s_mov_b32 m0, s44\n\ts_nop 0\n\ts_movrels_b32 s41, s48\n\ts_or_b32 s41, s41, 1\n\ts_movreld_b32 s48, s41\n\t")           // movrel read-modify-write, 5 instr
    CASE(7, 256, "v_readlane_b32 s40, %[v0], 3\n\tv_readlane_b32 s41, %[v1], 5\n\tv_readlane_b32 s42, %[v2], 7\n\tv_readlane_b32 s43, %[v3], 9\n\t")   // independent readlanes
    CASE(8, 256, "s_sub_u32 s41, 0, s40\n\ts_and_b32 s42, s40, s41\n\ts_or_b32 s43, s43, s42\n\ts_andn2_b32 s40, s40, s42\n\ts_or_b32 s40, s40, 1\n\t")   // dependent SALU mix, 5
    CASE(9, 256, "v_writelane_b32 %[v0], s40, 3\n\tv_writelane_b32 %[v1], s41, 5\n\t")                                                      // writelanes
    CASE(10, 256, "v_cmp_eq_u32_e32 vcc, s40, %[v0]\n\tv_cndmask_b32_e32 %[v1], 0, %[v2], vcc\n\tv_bfi_b32 %[v0], %[v1], 0, %[v0]\n\t")      // cmp->cndmask->bfi dependent VALU, 3
    CASE(11, 256, "s_cmp_eq_u32 s44, 1\n\ts_cbranch_scc1 .Lmb_exit\n\t")                                                                            // compare + untaken branch
    CASE(12, 256, "s_bitcmp1_b64 s[46:47], 5\n\ts_cbranch_scc1 .Lmb_exit\n\ts_add_u32 s40, s40, 1\n\t")
    // the proposed all-scalar pick: list bits + slot by readlane (independent of the chain), slot table in SGPRs s[48:55]
    CASE(13, 256,
         "v_readlane_b32 s40, %[v0], 3\n\tv_readlane_b32 s44, %[v2], 3\n\ts_mov_b32 m0, s44\n\ts_nop 0\n\ts_movrels_b32 s41, s48\n\ts_andn2_b32 s42, s40, s41\n\ts_cbranch_scc0 .Lmb_exit\n\t"
         "s_sub_u32 s43, 0, s42\n\ts_and_b32 s43, s42, s43\n\ts_or_b32 s41, s41, s43\n\ts_andn2_b32 s41, s41, s43\n\ts_movreld_b32 s48, s41\n\tv_writelane_b32 %[v1], s43, 3\n\t")   // 13 instr (the andn2 keeps the table from saturating)
    asm volatile(".Lmb_exit:\n\t" ::: "memory");
    if (threadIdx.x == 0) out[which] = t1 - t0;
    sink[threadIdx.x] = v0 + v1 + v2 + v3;
}

int main() {
    unsigned long long* out;
    unsigned* sink;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&sink, 64 * 4);
    hipMemset(out, 0, 64 * 8);
    const int n = 14;
    const int instr[n] = {1024, 1024, 1024, 1024, 1024, 768, 1280, 1024, 1280, 512, 768, 512, 768, 256 * 13};
    const char* name[n] = {"dependent SALU", "independent SALU", "dependent VALU", "independent VALU", "readlane->SALU->VALU->VALU chain", "readlane->VALU->VALU chain",
                           "movrels/or/movreld via m0", "independent readlanes", "dependent SALU mix", "writelanes", "v_cmp->v_cndmask->v_bfi chain", "s_cmp + untaken branch",
                           "s_bitcmp1_b64 + untaken branch + add", "all-scalar pick body (13 instr)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, i, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%-42s %8llu ticks / %5d instr = %6.2f per instr\n", name[i], h[i], instr[i], (double)h[i] / instr[i]);
    return 0;
}
