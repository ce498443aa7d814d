// Microbenchmark for k_scanb's evaluation (swarmkit_amd/csrc/swp_scan.hpp): what its instruction kinds cost a wave that has its SIMD to
// itself — 64-bit compares, compare -> SGPR mask -> s_and -> v_cndmask chains against compare -> vcc -> v_cndmask, fused DPP minima.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb/mb_scan.hip -o /tmp/mb_scan ; run on the GPU box. s_memtime ticks per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define T0 "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
#define T1 "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
#define OUTS [t0] "=&s"(t0), [t1] "=&s"(t1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [l] "+v"(l)
#define CLOB "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "vcc", "scc", "memory"
#define CASE(ID, N, BODY)                                                                         \
    if (which == ID) {                                                                            \
        asm volatile(T0 ".rept " #N "\n\t" BODY ".endr\n\t" T1 : OUTS : : CLOB);                    \
    }

__global__ __launch_bounds__(64) void k(int which, unsigned long long* out, unsigned* sink) {
    extern __shared__ unsigned lds[];
    unsigned long long t0 = 0, t1 = 0;
    unsigned v0 = threadIdx.x * 7 + 1, v1 = threadIdx.x, v2 = 3, v3 = 5;
    unsigned long long a = threadIdx.x * 1000003ull, b = 77777777777ull, c = 5;
    unsigned l = threadIdx.x * 4;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    CASE(0, 256, "v_cmp_le_i64_e64 s[40:41], %[a], %[b]\n\tv_cmp_le_i64_e64 s[42:43], %[b], %[c]\n\tv_cmp_le_i64_e64 s[44:45], %[c], %[a]\n\tv_cmp_le_i64_e64 s[46:47], %[b], %[a]\n\t")
    CASE(1, 256, "v_cmp_le_u32_e64 s[40:41], %[v0], %[v1]\n\tv_cmp_le_u32_e64 s[42:43], %[v1], %[v2]\n\tv_cmp_le_u32_e64 s[44:45], %[v2], %[v3]\n\tv_cmp_le_u32_e64 s[46:47], %[v3], %[v0]\n\t")
    CASE(2, 256, "v_cmp_le_u32_e64 s[40:41], %[v0], %[v1]\n\tv_cmp_le_u32_e64 s[42:43], %[v1], %[v2]\n\ts_and_b64 s[44:45], s[40:41], s[42:43]\n\tv_cndmask_b32_e64 %[v3], %[v3], %[v2], s[44:45]\n\t")   // cmp, cmp, s_and, cndmask (4)
    CASE(3, 256, "v_cmp_le_u32_e32 vcc, %[v0], %[v1]\n\tv_cndmask_b32_e32 %[v3], -1, %[v3], vcc\n\tv_cmp_le_u32_e32 vcc, %[v1], %[v2]\n\tv_cndmask_b32_e32 %[v3], -1, %[v3], vcc\n\t")               // cmp vcc, cndmask, cmp vcc, cndmask (4)
    CASE(4, 256, "v_min_u32_dpp %[v0], %[v0], %[v0] row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_u32_dpp %[v1], %[v1], %[v1] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_u32_dpp %[v2], %[v2], %[v2] row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_u32_dpp %[v3], %[v3], %[v3] row_shr:1 row_mask:0xf bank_mask:0xf\n\t")
    CASE(5, 256, "v_lshrrev_b64 %[a], 3, %[a]\n\tv_lshrrev_b64 %[b], 3, %[b]\n\t")
    CASE(6, 128, "ds_read_b32 %[v0], %[l]\n\tds_read_b32 %[v1], %[l] offset:256\n\tds_read_b32 %[v2], %[l] offset:512\n\tds_read_b32 %[v3], %[l] offset:768\n\ts_waitcnt lgkmcnt(0)\n\t")   // four reads + one wait (5)
    CASE(7, 256, "v_cmp_lt_u64_e32 vcc, %[a], %[b]\n\tv_cndmask_b32_e32 %[v0], %[v0], %[v1], vcc\n\tv_cndmask_b32_e32 %[v2], %[v2], %[v3], vcc\n\t")                                          // u64 min step: cmp + two cndmask (3)
    CASE(8, 256, "v_and_b32_e32 %[v2], %[v0], %[v1]\n\tv_cmp_ne_u32_e32 vcc, 0, %[v2]\n\tv_cndmask_b32_e32 %[v3], -1, %[v3], vcc\n\tv_cmp_le_i64_e32 vcc, %[a], %[b]\n\tv_cndmask_b32_e32 %[v3], -1, %[v3], vcc\n\t"
                 "v_cmp_le_i64_e32 vcc, %[c], %[b]\n\tv_cndmask_b32_e32 %[v3], -1, %[v3], vcc\n\tv_and_b32_e32 %[v2], 0xffffff, %[v3]\n\tv_cmp_lt_u32_e32 vcc, %[v2], %[v1]\n\tv_cndmask_b32_e32 %[v3], -1, %[v3], vcc\n\t")   // one evaluation as a vcc chain (10)
    CASE(9, 256, "v_readfirstlane_b32 s40, %[v0]\n\tv_readfirstlane_b32 s41, %[v1]\n\tv_readfirstlane_b32 s42, %[v2]\n\tv_readfirstlane_b32 s43, %[v3]\n\t")
    CASE(10, 128, "ds_read_b32 %[v0], %[l]\n\ts_waitcnt lgkmcnt(0)\n\t")   // a dependent LDS round trip (2)
    if (threadIdx.x == 0) out[which] = t1 - t0;
    sink[threadIdx.x] = v0 + v1 + v2 + v3 + (unsigned)a + (unsigned)b + (unsigned)c;
}

int main() {
    unsigned long long* out;
    unsigned* sink;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&sink, 64 * 4);
    hipMemset(out, 0, 64 * 8);
    const int n = 11;
    const int instr[n] = {1024, 1024, 1024, 1024, 1024, 512, 640, 768, 2560, 1024, 256};
    const char* name[n] = {"v_cmp_le_i64 -> SGPR pair, independent", "v_cmp_le_u32 -> SGPR pair, independent", "cmp, cmp, s_and_b64, v_cndmask(sgpr)", "cmp vcc, cndmask, cmp vcc, cndmask",
                           "four interleaved v_min_u32_dpp", "v_lshrrev_b64", "four ds_read_b32 + one wait", "v_cmp_lt_u64 + two cndmask", "one evaluation as a vcc chain (10 instr)",
                           "independent v_readfirstlane", "ds_read_b32 + wait"};
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, i, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%-44s %8llu ticks / %5d instr = %6.2f per instr\n", name[i], h[i], instr[i], (double)h[i] / instr[i]);
    return 0;
}
