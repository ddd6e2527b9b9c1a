// NoisyLinear noise factors (kernels_noisy.hip) shared with reduce_kernel.
#pragma once
#include "net.hpp"

namespace frl {

// eps of set `s` of learner p: [eps_in0[k_pad], eps_out0[n_pad], eps_in1[k_pad], eps_out1[n_pad]]
__device__ __forceinline__ g_cf noisy_eps_of(const EngineDesc& D, const LayerDesc& H, int p, int s) {
    return as_global(D.noisy_eps + ((size_t)p * 3 + s) * 2 * (H.k_pad + H.n_pad));
}
__device__ __forceinline__ float noisy_eps_w(g_cf eps, const LayerDesc& H, int split, int k, int n) {
    const int sub = (n >= split) ? 1 : 0;
    g_cf e = eps + sub * (H.k_pad + H.n_pad);
    return e[k] * e[H.k_pad + n];
}
__device__ __forceinline__ float noisy_eps_b(g_cf eps, const LayerDesc& H, int split, int n) {
    const int sub = (n >= split) ? 1 : 0;
    return eps[sub * (H.k_pad + H.n_pad) + H.k_pad + n];
}

}  // namespace frl
