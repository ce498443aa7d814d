// swp_groups.hip — translation unit of the task-group kernel (k_groups2, swp_groups.hpp) and its launcher.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "swp_launch.hpp"
#include "swp_wave.hpp"
#ifdef SWP_G2_STATS   // experiments only (make EXTRA=-DSWP_G2_STATS): which admission path the machine took, counted on the device
#include <stdio.h>
namespace swpdev { __device__ unsigned long long g2_stat_dev[12]; }
#define G2_STAT(i, v) do { if (wv::lane() == 0) atomicAdd(&swpdev::g2_stat_dev[i], (unsigned long long)(v)); } while (0)
#endif
#define SWP_G2_KERNELS
#include "swp_groups.hpp"

namespace swpdev {

// k_g2_static: a wave per static class; then one workgroup for the whole tick: the machine wave + 15 helper waves (swp_groups.hpp)
hipError_t launch_groups2(const Groups2Args& a, hipStream_t s, int dev) {
    hipError_t r = ensure_big_lds(reinterpret_cast<const void*>(&k_groups2), dev);
    if (r != hipSuccess) return r;
    unsigned threads = G2_THREADS;
    if (const char* t = getenv("SWP_G2_THREADS")) {   // experiments only: fewer helper waves (a multiple of 64, at least 128)
        const unsigned v = (unsigned)atoi(t);
        if (v >= 128 && v <= G2_THREADS && v % 64 == 0) threads = v;
    }
    // the static class lists first (one wave per class, all over the chip), then the tick
    hipLaunchKernelGGL(k_g2_static, dim3(a.n_scls), dim3(64), 0, s, a);
    if ((r = hipGetLastError()) != hipSuccess) return r;
    hipLaunchKernelGGL(k_groups2, dim3(1), dim3(threads), g2_lds_bytes(), s, a);
#ifdef SWP_G2_STATS
    {
        unsigned long long h[12] = {0}, z[12] = {0};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g2_stat_dev), sizeof h);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g2_stat_dev), z, sizeof z);
        fprintf(stderr, "[swp] k_groups2 paths: %llu heaps in flat mode, %llu candidates by the post-order scatter, %llu by a flush's replay, %llu flushes forced by a third key, %llu pipelined root replacements, %llu chunks taken whole in flat mode, %llu chunks appended whole while filling, %llu pushes sifted up by the wave\n",
                h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        fprintf(stderr, "[swp] k_groups2 sort: %llu heaps popped by lane 0, %llu elements in them, %llu of them with the root's key\n", h[9], h[11], h[10]);
    }
#endif
    return hipGetLastError();
}

}  // namespace swpdev
