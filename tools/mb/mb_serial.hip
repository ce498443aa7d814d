// What a strictly serial chain costs on one gfx950 wave: dependent LDS loads (b32 / b128), dependent global loads (L2 hits), taken
// branches — alone in the workgroup and with 15 other waves polling an LDS word between s_sleep's (k_groups2's helper waves).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb/mb_serial.hip -o tools/mb/mb_serial.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void k(int which, unsigned long long* out, unsigned* gchain, unsigned* sink) {
    __shared__ unsigned lds[4096];
    __shared__ unsigned flag;
    const unsigned tid = threadIdx.x;
    for (unsigned i = tid; i < 4096; i += blockDim.x) lds[i] = ((i * 7 + 5) & 1023) * 16;   // byte offsets, 16-aligned, a permutation cycle over 1024 slots
    if (tid == 0) flag = 0;
    __syncthreads();
    if (tid >= 64) {   // the other waves: poll until wave 0 is done
        while (__hip_atomic_load(&flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(8);
        return;
    }
    unsigned long long t0 = 0, t1 = 0;
    unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    unsigned base = (unsigned)(size_t)lds;   // LDS address of the table
    if (which == 0) {   // dependent ds_read_b32 chain
        v0 = base;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     ".rept 512\n\tds_read_b32 %[v0], %[v0]\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32_e32 %[v0], %[b], %[v0]\n\t.endr\n\t"
                     "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [v0] "+v"(v0) : [b] "v"(base) : "memory");
    } else if (which == 1) {   // dependent ds_read_b128 chain (the address comes from the first dword)
        v0 = base;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     ".rept 512\n\tds_read_b128 v[20:23], %[v0]\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32_e32 %[v0], %[b], v20\n\t.endr\n\t"
                     "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [v0] "+v"(v0) : [b] "v"(base) : "memory", "v20", "v21", "v22", "v23");
    } else if (which == 2) {   // taken branch loop: 3 instructions per iteration, 1024 iterations
        asm volatile("s_mov_b32 s40, 1024\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n"
                     "1:\n\ts_sub_u32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\t"
                     "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [t0] "=&s"(t0), [t1] "=&s"(t1) : : "memory", "s40", "scc");
    } else if (which == 3) {   // dependent global load chain (a 4 KB table: L2 / L1 hits)
        unsigned long long p = (unsigned long long)gchain;
        unsigned lo = (unsigned)p, hi = (unsigned)(p >> 32);
        asm volatile("v_mov_b32 v20, %[lo]\n\tv_mov_b32 v21, %[hi]\n\tv_mov_b32 v22, 0\n\t"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     ".rept 256\n\tglobal_load_dword v22, v[20:21], off\n\ts_waitcnt vmcnt(0)\n\tv_add_co_u32_e32 v20, vcc, %[lo], v22\n\tv_mov_b32 v21, %[hi]\n\t.endr\n\t"
                     "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [t0] "=&s"(t0), [t1] "=&s"(t1) : [lo] "v"(lo), [hi] "v"(hi) : "memory", "v20", "v21", "v22", "vcc");
    } else if (which == 4) {   // ds_write_b128 then dependent ds_read_b128 of the same address (store -> load round trip)
        v0 = base;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     ".rept 256\n\tds_read_b128 v[20:23], %[v0]\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b128 %[v0], v[20:23]\n\tds_read_b128 v[24:27], %[v0]\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32_e32 %[v0], %[b], v24\n\t.endr\n\t"
                     "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [v0] "+v"(v0) : [b] "v"(base)
                     : "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    } else if (which == 5) {   // exec = lane 0 only, dependent ds_read_b32 chain
        v0 = base;
        asm volatile("s_mov_b64 s[42:43], exec\n\ts_mov_b64 exec, 1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     ".rept 512\n\tds_read_b32 %[v0], %[v0]\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32_e32 %[v0], %[b], %[v0]\n\t.endr\n\t"
                     "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[42:43]\n\t"
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [v0] "+v"(v0) : [b] "v"(base) : "memory", "s42", "s43");
    } else if (which == 6) {   // not-taken / taken conditional branch over a block (if-then skipped): v_cmp + s_cbranch_vccz forward taken
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     ".rept 256\n\tv_cmp_eq_u32_e32 vcc, 12345, %[v0]\n\ts_cbranch_vccz 2f\n\tv_add_u32_e32 %[v1], 1, %[v1]\n\tv_add_u32_e32 %[v1], 1, %[v1]\n2:\n\t.endr\n\t"
                     "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [v0] "+v"(v0), [v1] "+v"(v1) : : "memory", "vcc");
    }
    if (tid == 0) {
        out[which] = t1 - t0;
        __hip_atomic_store(&flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    sink[tid] = v0 + v1 + v2 + v3;
}

int main() {
    unsigned long long* out;
    unsigned *sink, *gchain;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&sink, 64 * 4);
    hipMalloc(&gchain, 4096);
    std::vector<unsigned> h(1024);
    for (unsigned i = 0; i < 1024; ++i) h[i] = ((i * 7 + 5) & 1023) * 4;
    hipMemcpy(gchain, h.data(), 4096, hipMemcpyHostToDevice);
    const int n = 7;
    const int ops[n] = {512, 512, 1024, 256, 256, 512, 256};
    const char* name[n] = {"dependent ds_read_b32 (+1 valu)", "dependent ds_read_b128 (+1 valu)", "loop: s_sub, s_cmp, taken s_cbranch", "dependent global_load_dword (+2 valu)",
                           "ds_read_b128, ds_write_b128, ds_read_b128 same addr", "exec=1: dependent ds_read_b32 (+1 valu)", "v_cmp + taken forward s_cbranch_vccz over 2 valu"};
    for (int threads : {64, 1024}) {
        hipMemset(out, 0, 64 * 8);
        for (int rep = 0; rep < 2; ++rep)
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, i, out, gchain, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> r(64);
        hipMemcpy(r.data(), out, 64 * 8, hipMemcpyDeviceToHost);
        printf("--- workgroup of %d threads (%d other waves polling LDS between s_sleep 8)\n", threads, threads / 64 - 1);
        for (int i = 0; i < n; ++i) printf("%-55s %9llu ticks / %5d = %7.1f each\n", name[i], r[i], ops[i], (double)r[i] / ops[i]);
    }
    return 0;
}
