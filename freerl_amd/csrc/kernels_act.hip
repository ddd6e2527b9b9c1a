// Batched policy / value forward: select_action, evaluate_action and the PPO value pass.
// DQN.py:70-84 (argmax), TD3.py:163-170 (tanh actor), SAC.py:192-204 (tanh-Gaussian sample /
// tanh(mean)), PPO_with_tricks.py:234-270 (Gaussian sample + per-dimension log-prob / mean).
// grid = (row chunks, learners); rows may be one observation (the reference's per-step call)
// or a whole vectorised-env batch staged by the env pool.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/act_common.hpp"

namespace frl {


__global__ __launch_bounds__(256) void act_kernel(const EngineDesc* __restrict__ Dp, ActArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int p = blockIdx.y, r0 = blockIdx.x * D.rc;
    const NetDesc& N = D.net[a.net];
    const Lds S = carve_lds(smem, D.rc, D.hidden, D.lds_kin_pad, D.lds_out_pad, D.lds_batch_pad, D.lds_act_pad, D.lds_hbufs);
    const int rc = D.rc, nv = min(rc, a.n_rows - r0);
    const size_t off = (size_t)p * D.learner_stride + D.net_off[a.net];
    g_cf theta = a.theta_alt ? as_global(a.theta_alt + (size_t)p * N.size)                                             // Wk copy of a fragment-image net
               : a.use_target == 2 ? as_global(D.theta_eff + (size_t)p * 3 * D.learner_stride + D.net_off[a.net])     // noisy set 0
                                   : as_global((a.use_target ? D.target : D.theta) + off);
    const int nl = N.n_layers / N.heads, l0 = a.head * nl;
    const int K = a.in_dim, kpad = N.L[l0].k_pad;
    g_cf in = as_global(a.in + ((size_t)p * a.n_rows + r0) * K);
    for (int e = threadIdx.x; e < rc * kpad; e += kWG) {
        const int r = e / kpad, c = e - r * kpad;
        S.xin[r * S.xp + c] = (r < nv && c < K) ? in[(size_t)r * K + c] : 0.f;
    }
    if (D.obs_norm_on && a.normalize) {          // select_action: norm(obs, update=False) (SAC.py:194-195, MADDPG.py:162-163)
        __syncthreads();
        const int nag = D.n_agents, j = nag > 1 ? a.net / 2 : 0;           // MADDPG: net 2j is agent j's actor
        normalize_cols(S.xin, S.xp, nv, 0, D.rec.obs_dim[j],
                       as_global(D.obsnorm + (((size_t)p * nag + (nag - 1)) * nag + j) * D.obsnorm_w), D.rec.obs_dim[j]);
    }
    __syncthreads();
    const int out_act = (a.mode == ACTM_TANH || a.mode == ACTM_PPO_SAMPLE) ? ACT_TANH : ACT_NONE;
    mlp_fwd(N, l0, nl, theta, S, out_act);
    act_epilogue(D, a, N, theta, S.outb, S.op, nv, r0, p, N.L[l0 + nl - 1].n);
}

// The same for the engines of the register-chained kernels (NetDesc::frag): device/act_common.hpp's act_frag_body;
// grid = (ceil(n_rows / 64), learners).
__global__ __launch_bounds__(256) void act_frag_kernel(const EngineDesc* __restrict__ Dp, ActArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    act_frag_body(*Dp, a, smem, blockIdx.y, blockIdx.x * 64);
}

}  // namespace frl
