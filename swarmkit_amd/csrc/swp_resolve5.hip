// swp_resolve5.hip — translation unit of the round resolver (k_resolve5, swp_resolve5.hpp) and its launcher.
// A separate object so that the resolver can be rebuilt without recompiling the engine's other kernels (swp_device.hpp).
#include <hip/hip_runtime.h>

#include "swp_launch.hpp"
#include "swp_wave.hpp"
#include "swp_resolve5.hpp"

namespace swpdev {

size_t r5_lds_size(uint32_t n_nodes, uint32_t n_words, uint32_t n_rr) { return r5_lds_bytes(n_nodes, n_words, n_rr); }
uint32_t r5_max_rows() { return R5_RRMAX; }
bool r5_supports(uint32_t n_words) { return n_words <= 64 * R5_KMAX; }

template <int K>
static hipError_t launch_k(const ResolveArgs& ra, size_t lds, hipStream_t s, int dev) {
    hipError_t r = ensure_big_lds(reinterpret_cast<const void*>(&k_resolve5<K>), dev);
    if (r != hipSuccess) return r;
    hipLaunchKernelGGL((k_resolve5<K>), dim3(1), dim3(R5_THREADS), lds, s, ra);
    return hipGetLastError();
}

hipError_t launch_resolve5(const ResolveArgs& ra, size_t lds, hipStream_t s, int dev) {
    switch ((ra.n_words + 63) / 64) {
    case 1: return launch_k<1>(ra, lds, s, dev);
    case 2: return launch_k<2>(ra, lds, s, dev);
    case 3: return launch_k<3>(ra, lds, s, dev);
    default: return launch_k<4>(ra, lds, s, dev);
    }
}

}  // namespace swpdev
