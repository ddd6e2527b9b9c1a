// NoisyLinear heads (DQN_file/Noisy_net.py:17-76, used by DQN_with_tricks.py:49-51,68-70): y = (mu_w + sigma_w * eps_w) x +
// (mu_b + sigma_b * eps_b) with factorised noise eps_w = eps_out (x) eps_in, eps_b = eps_out, redrawn at EVERY forward.
// The forward/backward kernels stay noise-agnostic: `noisy_materialise_kernel` writes the effective parameter set of one
// forward (trunk copied, head = mu + sigma * eps) into theta_eff, the gradient w.r.t. the effective head weights lands in
// the head's (= mu's) gradient slots, and `reduce_kernel` derives d/d sigma = d/d W_eff * eps while it sums the slabs.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/net.hpp"
#include "device/noisy.hpp"

namespace frl {

// grid (P, n_sets): set s of learner p <- (from_target[s] ? target : theta) with the head replaced by mu + sigma * eps_s.
// The sigma used is always the ONLINE net's for the online sets and the TARGET net's for the target set (deepcopy keeps
// its own sigma, soft-updated like every other parameter).
__global__ __launch_bounds__(256) void noisy_materialise_kernel(const EngineDesc* __restrict__ Dp, int set0, int n_sets, int target_mask) {
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x, s = set0 + blockIdx.y;
    if (blockIdx.y >= n_sets) return;
    const NetDesc& N = D.net[0];
    const LayerDesc& H = N.L[N.n_layers - 1];
    const LayerDesc& SG = N.L[N.n_layers];                 // shadow layer: sigma
    const size_t base = (size_t)p * D.learner_stride + D.net_off[0];
    g_cf src = as_global(((target_mask >> s) & 1 ? D.target : D.theta) + base);
    g_f dst = as_global(D.theta_eff + ((size_t)p * 3 + s) * D.learner_stride + D.net_off[0]);
    g_cf eps = noisy_eps_of(D, H, p, s);
    for (int i = threadIdx.x; i < H.w_off; i += kWG) dst[i] = src[i];                    // the trunk, unchanged
    for (int i = threadIdx.x; i < H.k_pad * H.n_pad; i += kWG) {                            // Wk[k][n]
        const int k = i / H.n_pad, n = i - k * H.n_pad;
        dst[H.w_off + i] = src[H.w_off + i] + src[SG.w_off + i] * noisy_eps_w(eps, H, D.noisy_split, k, n);
    }
    for (int n = threadIdx.x; n < H.n_pad; n += kWG)
        dst[H.b_off + n] = src[H.b_off + n] + src[SG.b_off + n] * noisy_eps_b(eps, H, D.noisy_split, n);
}

// device-drawn noise: f(x) = sign(x) sqrt(|x|) of standard normals (Noisy_net.py:72-76); padded slots stay zero
__global__ __launch_bounds__(256) void noisy_draw_kernel(const EngineDesc* __restrict__ Dp, int set0, int n_sets, unsigned long long counter) {
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x, s = set0 + blockIdx.y;
    if (blockIdx.y >= n_sets) return;
    const NetDesc& N = D.net[0];
    const LayerDesc& H = N.L[N.n_layers - 1];
    const int per = 2 * (H.k_pad + H.n_pad);
    float* eps = D.noisy_eps + ((size_t)p * 3 + s) * per;
    const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
    for (int i = threadIdx.x; i < per; i += kWG) {
        const int sub = i / (H.k_pad + H.n_pad), j = i - sub * (H.k_pad + H.n_pad);
        bool live;
        if (j < H.k_pad) live = j < H.k;
        else {
            const int n = j - H.k_pad;
            live = sub == 0 ? n < (D.dueling ? D.noisy_split : H.n) : (D.dueling && n >= D.noisy_split && n < H.n);
        }
        float v = 0.f;
        if (live) {
            float a, b;
            normal2(philox4x32_10(counter, 0x9000u + (unsigned)s, (unsigned)i, key), a, b);
            v = copysignf(sqrtf(fabsf(a)), a);
        }
        eps[i] = v;
    }
}

}  // namespace frl
