// swp_waterfill.hip — translation unit of k_waterfill (swp_waterfill.hpp) and its launcher.
#include <hip/hip_runtime.h>

#include "swp_launch.hpp"
#define SWP_WATERFILL_KERNEL
#include "swp_waterfill.hpp"

namespace swpdev {

hipError_t launch_waterfill(const WaterArgs& a, hipStream_t s) {
    if (a.count == 0 || a.n_nodes == 0) return hipSuccess;
    hipLaunchKernelGGL(k_waterfill, dim3(1), dim3(WF_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace swpdev
