// DVFS probe: time a single-wave dependent VALU chain with and without a chip-wide heater kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
__global__ void chain(u64* out, int iters, u32 seed) {
    u32 a = seed + threadIdx.x, b = seed * 3;
    u64 w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a = a + b; asm volatile("" : "+v"(a)); }
    }
    u64 w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = a; }
}
__global__ void heater(float* out, int iters) {
    float x = threadIdx.x * 0.001f, y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = x * y + 0.5f;
    }
    if (x == 12345.f) out[0] = x;
}
int main() {
    u64* d; float* f; hipMalloc(&d, 64); hipMalloc(&f, 64);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int iters = 2000000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, iters, 7u);
        hipStreamSynchronize(s1);
        u64 h[3]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("alone   : %.2f ns per VALU (wall), %.2f clock64 ticks per VALU\n", h[0] * 10.0 / (iters * 8.0), (double)h[1] / (iters * 8.0));
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(heater, dim3(255 * 8), dim3(256), 0, s2, f, 4000000);
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, iters, 7u);
        hipStreamSynchronize(s1);
        u64 h[3]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("+heater : %.2f ns per VALU (wall), %.2f clock64 ticks per VALU\n", h[0] * 10.0 / (iters * 8.0), (double)h[1] / (iters * 8.0));
        hipStreamSynchronize(s2);
    }
    return 0;
}
