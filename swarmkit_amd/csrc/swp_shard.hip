// swp_shard.hip — translation unit of the sharded-scan kernels (swp_shard.hpp) and their launchers.
#include <hip/hip_runtime.h>

#include "swp_launch.hpp"
#define SWP_SHARD_KERNELS
#include "swp_shard.hpp"

namespace swpdev {

hipError_t launch_propose(const ProposeArgs& a, hipStream_t s) {
    if (a.count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_propose, dim3(a.count), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_shard_apply(const ShardApplyArgs& a, hipStream_t s) {
    const uint32_t n = a.n_picks > a.n_inf ? a.n_picks : a.n_inf;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_shard_apply, dim3((n + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace swpdev
