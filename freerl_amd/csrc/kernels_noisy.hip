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

// grid (P): the sets [set0, set0 + n_sets) of learner p; set s <- (from_target[s] ? target : theta) with the head replaced by
// mu + sigma * eps_s.  The sigma used is always the ONLINE net's for the online sets and the TARGET net's for the target set (deepcopy
// keeps its own sigma, soft-updated like every other parameter).  16-byte accesses; mu and sigma of a source net are read ONCE for
// all the sets that come from it (an update's sets 0 and 2 are both the online net under different noise) — round 4's form, one
// dword per thread and step with a division per element and every set reading its source again, ran at 4.7 TB/s of 1.23 MB per
// learner; this one moves 0.96 MB.
__global__ __launch_bounds__(256) void noisy_materialise_kernel(const EngineDesc* __restrict__ Dp, int set0, int n_sets, int target_mask) {
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x;
    const NetDesc& N = D.net[0];
    const LayerDesc& H = N.L[N.n_layers - 1];
    const LayerDesc& SG = N.L[N.n_layers];                 // shadow layer: sigma
    const size_t base = (size_t)p * D.learner_stride + D.net_off[0];
    const int kn = H.k_pad + H.n_pad, n4 = H.n_pad >> 2, split = D.noisy_split;
    for (int from_target = 0; from_target < 2; ++from_target) {
        int sets[3], ns = 0;
        for (int s = set0; s < set0 + n_sets && ns < 3; ++s)
            if (((target_mask >> s) & 1) == from_target) sets[ns++] = s;
        if (!ns) continue;
        g_cf src = as_global((from_target ? D.target : D.theta) + base);
        g_f dst[3];
        g_cf eps[3];
        for (int j = 0; j < 3; ++j) {
            const int s = sets[j < ns ? j : 0];
            dst[j] = as_global(D.theta_eff + ((size_t)p * 3 + s) * D.learner_stride + D.net_off[0]);
            eps[j] = noisy_eps_of(D, H, p, s);
        }
        // the trunk, unchanged (w_off of the head is a multiple of four floats: every block of the layout is)
        for (int i = threadIdx.x; i < (H.w_off >> 2); i += kWG) {
            const f32x4 v = ld4(src + 4 * i);
            for (int j = 0; j < ns; ++j) st4(dst[j] + 4 * i, v);
        }
        // the head Wk[k][n], four n per thread: eps_w = eps_in[k] * eps_out[n], the noise pair by which side of `split` n is on
        for (int i = threadIdx.x; i < H.k_pad * n4; i += kWG) {
            const int k = i / n4, n0 = (i - k * n4) << 2;
            const f32x4 mu = ld4(src + H.w_off + 4 * i), sg = ld4(src + SG.w_off + 4 * i);
            for (int j = 0; j < ns; ++j) {
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    g_cf e = eps[j] + ((n0 + c >= split) ? kn : 0);
                    o[c] = mu[c] + sg[c] * (e[k] * e[H.k_pad + n0 + c]);
                }
                st4(dst[j] + H.w_off + 4 * i, o);
            }
        }
        for (int n = threadIdx.x; n < H.n_pad; n += kWG) {
            const float mu = src[H.b_off + n], sg = src[SG.b_off + n];
            for (int j = 0; j < ns; ++j) dst[j][H.b_off + n] = mu + sg * noisy_eps_b(eps[j], H, split, n);
        }
    }
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
