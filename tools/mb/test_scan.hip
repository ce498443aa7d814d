// checks wv::scan_incl_u32 (DPP row shifts + readlane row totals) against a serial prefix sum on the device
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../swarmkit_amd/csrc/swp_wave.hpp"
__global__ void k(const unsigned* in, unsigned* out) { out[threadIdx.x] = wv::scan_incl_u32(in[threadIdx.x]); }
int main() {
    unsigned h[64], o[64], *di, *dO;
    hipMalloc(&di, 256); hipMalloc(&dO, 256);
    int bad = 0;
    for (int t = 0; t < 50; ++t) {
        for (int i = 0; i < 64; ++i) h[i] = (unsigned)((i * 2654435761u + t * 40503u) >> (t % 28)) % 65u;
        hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dO);
        hipMemcpy(o, dO, 256, hipMemcpyDeviceToHost);
        unsigned run = 0;
        for (int i = 0; i < 64; ++i) { run += h[i]; if (o[i] != run) { if (bad++ < 5) printf("t %d lane %d: %u != %u\n", t, i, o[i], run); } }
    }
    printf("scan_incl_u32: %s\n", bad ? "FAILED" : "OK");
    return bad != 0;
}
