// swp_groups.hip — translation unit of the task-group kernel (k_groups2, swp_groups.hpp) and its launcher.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "swp_launch.hpp"
#include "swp_wave.hpp"
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
    return hipGetLastError();
}

}  // namespace swpdev
