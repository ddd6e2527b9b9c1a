// Fused replay-sample + batched-update kernels: ONE launch = one learn() of every learner.
//
// Replaces, per learner, the reference's ~60 (DQN) to ~700 (MADDPG) eager ATen launches per
// learn() and the 5 H2D copies of Buffer.sample (TD3_file/Buffer.py:50-55):
//   index draw (Philox, or host-supplied for parity) -> record gather from the HBM ring ->
//   target forward -> TD target -> online forward -> MSE delta -> backward (MFMA) ->
//   global-norm clip -> Adam -> soft target update.
// One workgroup owns one (learner, agent): batch-row chunks stream through LDS, weight
// gradients accumulate in the learner's global grad block, so no cross-workgroup traffic.
// Population mode (P learners = independent seeds) fills the 256 CUs; P = 1 is the
// reference-compatible single learner.
#include <hip/hip_runtime.h>

#include "device/net.hpp"

namespace frl {

namespace {

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;
constexpr float kLog2 = 0.69314718055994530942f;

__device__ __forceinline__ float softplus_t(float x) {     // F.softplus (beta 1, threshold 20)
    return x > 20.f ? x : log1pf(expf(x));
}

__device__ __forceinline__ Lds carve(const EngineDesc& D, float* smem) {
    return carve_lds(smem, D.rc, D.hidden, D.lds_kin_pad, D.lds_out_pad, D.lds_batch_pad, D.lds_act_pad);
}

}  // namespace

// ------------------------------------------------------------------------------------- DQN
// DQN.learn (DQN_file/DQN.py:104-128): y = r + gamma * max_a Q_t(s',a) * (1-d);
// loss = mean((Q(s)[a] - y)^2); Adam (no clipping, DQN.py:56-59); soft update.
__global__ __launch_bounds__(256) void dqn_update_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x;
    const NetDesc& N = D.net[0];
    const RecordDesc& R = D.rec;
    const Lds S = carve(D, smem);
    const int rc = D.rc, B = a.batch, nl = N.n_layers;
    const size_t base = (size_t)p * D.learner_stride + D.net_off[0];
    float* theta = D.theta + base;
    float* target = D.target + base;
    float* grad = D.grad + base;
    const float* ring = D.replay + (size_t)p * D.capacity * R.stride;
    int* idx = D.idx + (size_t)p * D.n_agents * D.batch_max;
    const int O = R.obs_dim[0], nA = N.L[nl - 1].n, npad = N.L[nl - 1].n_pad, k0pad = N.L[0].k_pad;

    const unsigned long long counter = a.rng_counter;
    if (a.device_rng) {
        draw_indices(idx, reinterpret_cast<int*>(S.y), B, a.size, counter, 0u, D.seed + 0x9E3779B97F4A7C15ull * (p + 1));
    }
    __syncthreads();

    // ---- TD targets with the target net
    for (int r0 = 0; r0 < B; r0 += rc) {
        const int nv = min(rc, B - r0);
        gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.nobs_off[0], O, 0);
        zero_cols(S.xin, S.xp, rc, O, k0pad);
        __syncthreads();
        mlp_fwd(N, 0, nl, target, S, ACT_NONE);
        for (int r = threadIdx.x; r < nv; r += kWG) {
            float mx = S.outb[r * S.op];
            for (int j = 1; j < nA; ++j) mx = fmaxf(mx, S.outb[r * S.op + j]);
            const float* rec = ring + (size_t)idx[r0 + r] * R.stride;
            S.y[r0 + r] = rec[R.rew_off] + a.gamma * mx * (1.f - rec[R.done_off]);
        }
        __syncthreads();
    }
    // ---- online forward, MSE delta on the taken action, backward
    float lossp = 0.f;
    for (int r0 = 0; r0 < B; r0 += rc) {
        const int nv = min(rc, B - r0);
        gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.obs_off[0], O, 0);
        zero_cols(S.xin, S.xp, rc, O, k0pad);
        __syncthreads();
        mlp_fwd(N, 0, nl, theta, S, ACT_NONE);
        for (int e = threadIdx.x; e < rc * npad; e += kWG) {
            const int r = e / npad, j = e - r * npad;
            float d = 0.f;
            if (r < nv) {
                const int ar = (int)ring[(size_t)idx[r0 + r] * R.stride + R.act_off[0]];   // actions.long() (DQN.py:114)
                if (j == ar) {
                    const float diff = S.outb[r * S.op + j] - S.y[r0 + r];
                    d = 2.f * diff / (float)B;
                    lossp += diff * diff;
                }
            }
            S.outb[r * S.op + j] = d;
        }
        __syncthreads();
        mlp_bwd(N, 0, nl, theta, grad, S, r0 == 0, false, 0, 0);
    }
    const float loss = block_sum(lossp, S.red) / (float)B;
    __syncthreads();
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t = steps[0] + 1;
    const float gn = adam_net(N.size, theta, D.m + base, D.v + base, grad, target, a.critic_lr, a.adam_eps, a.beta1,
                              a.beta2, 0.f, a.clip_norm, t, a.tau, S.red);
    if (threadIdx.x == 0) {
        steps[0] = t;
        float* st = D.stats + (size_t)p * D.n_agents * ST_COUNT;
        st[ST_CRITIC_LOSS] = loss;
        st[ST_CRITIC_GNORM] = gn;
    }
}

// ------------------------------------------------------------------- DDPG / TD3 / SAC / MADDPG
// One workgroup per (learner, agent).  DDPG_simple.py:137-156, TD3.py:189-233, SAC.py:222-260,
// MADDPG_simple.py:165-186.  MADDPG's target nets are read by every agent's workgroup, so its
// soft updates run in `soft_update_kernel` after this launch; the single-agent algorithms fold
// them into the Adam pass.
__global__ __launch_bounds__(256) void ac_update_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int n = D.n_agents, p = blockIdx.x / n, ag = blockIdx.x - p * n;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[2 * ag];
    const NetDesc& NC = D.net[2 * ag + 1];
    const Lds S = carve(D, smem);
    const int rc = D.rc, B = a.batch;
    const bool sac = (D.algo == ALGO_SAC), maddpg = (D.algo == ALGO_MADDPG);
    const size_t lbase = (size_t)p * D.learner_stride;
    const size_t offA = lbase + D.net_off[2 * ag], offC = lbase + D.net_off[2 * ag + 1];
    float* thA = D.theta + offA;
    float* thC = D.theta + offC;
    float* gA = D.grad + offA;
    float* gC = D.grad + offC;
    float* tgA = D.target + offA;
    float* tgC = D.target + offC;
    const float* ring = D.replay + (size_t)p * D.capacity * R.stride;
    int* idx = D.idx + ((size_t)p * n + ag) * D.batch_max;
    float* noise0 = D.noise + ((size_t)p * n + ag) * 2 * D.batch_max * D.act_max;
    float* noise1 = noise0 + (size_t)D.batch_max * D.act_max;
    const int am = D.act_max;
    const int heads = NC.heads, ql = NC.n_layers / heads;
    const int OT = R.obs_total, AT = R.act_total, kc0 = NC.L[0].k_pad;
    const int Oa = R.obs_dim[ag], Aa = R.act_dim[ag], acol = R.act_off[ag] - R.act_off[0];
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const float invB = 1.f / (float)B;

    const unsigned long long counter = a.rng_counter;
    if (a.device_rng) {
        const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
        draw_indices(idx, reinterpret_cast<int*>(S.y), B, a.size, counter, (unsigned)ag, key);
        for (int e = threadIdx.x; e < B * am; e += kWG) {
            float n0, n1;
            normal2(philox4x32_10(counter, 0x4000u + (unsigned)ag, (unsigned)e, key), n0, n1);
            noise0[e] = n0;
            noise1[e] = n1;
        }
    }
    __syncthreads();

    // ================================ TD targets ================================
    for (int r0 = 0; r0 < B; r0 += rc) {
        const int nv = min(rc, B - r0);
        float lp_next = 0.f;                        // SAC: log pi(a'|s') of row threadIdx.x
        for (int j = 0; j < n; ++j) {
            const NetDesc& NJ = D.net[2 * j];
            const float* tgJ = D.target + lbase + D.net_off[2 * j];
            const int Oj = R.obs_dim[j], Aj = R.act_dim[j], cj = R.act_off[j] - R.act_off[0];
            gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.nobs_off[j], Oj, 0);
            zero_cols(S.xin, S.xp, rc, Oj, NJ.L[0].k_pad);
            __syncthreads();
            mlp_fwd(NJ, 0, NJ.n_layers, tgJ, S, sac ? ACT_NONE : ACT_TANH);
            if (sac) {                              // SAC.py:70-97 on actor_target (SAC.py:227)
                const int r = threadIdx.x;
                if (r < rc) {
                    float lp = 0.f;
                    for (int c = 0; c < Aj; ++c) {
                        const float mean = S.outb[r * S.op + c];
                        const float ls = fminf(fmaxf(tgJ[NJ.extra_off + c], -20.f), 2.f);
                        const float sd = expf(ls);
                        const float eps = (r < nv) ? noise0[(size_t)(r0 + r) * am + c] : 0.f;
                        const float u = mean + sd * eps;
                        const float du = u - mean;
                        lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                        lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                        S.abuf[r * S.ap + cj + c] = tanhf(u);
                    }
                    lp_next = lp;
                }
            } else {
                for (int e = threadIdx.x; e < rc * Aj; e += kWG) {
                    const int r = e / Aj, c = e - r * Aj;
                    float v = S.outb[r * S.op + c];
                    if (a.use_policy_noise && r < nv) {   // TD3.py:196-198
                        float nz = a.policy_noise_scale * (noise0[(size_t)(r0 + r) * am + c] * a.policy_noise);
                        nz = fminf(fmaxf(nz, -a.noise_clip), a.noise_clip);
                        v = fminf(fmaxf(v * a.max_action + nz, -a.max_action), a.max_action) / a.max_action;
                    }
                    S.abuf[r * S.ap + cj + c] = v;
                }
            }
            __syncthreads();
        }
        // centralised target critic on [next_obs_all | a'_all]
        gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.nobs_off[0], OT, 0);
        for (int e = threadIdx.x; e < rc * AT; e += kWG) {
            const int r = e / AT, c = e - r * AT;
            S.xin[r * S.xp + OT + c] = S.abuf[r * S.ap + c];
        }
        zero_cols(S.xin, S.xp, rc, OT + AT, kc0);
        __syncthreads();
        mlp_fwd(NC, 0, ql, tgC, S, ACT_NONE);
        float q = (threadIdx.x < rc) ? S.outb[threadIdx.x * S.op] : 0.f;
        if (heads == 2) {
            __syncthreads();
            mlp_fwd(NC, ql, ql, tgC, S, ACT_NONE);
            if (threadIdx.x < rc) q = fminf(q, S.outb[threadIdx.x * S.op]);
        }
        if (threadIdx.x < nv) {
            const float* rec = ring + (size_t)idx[r0 + threadIdx.x] * R.stride;
            const float rew = rec[R.rew_off + ag], done = rec[R.done_off + ag];
            S.y[r0 + threadIdx.x] = sac ? rew + a.gamma * (1.f - done) * (q + alpha * (-lp_next))
                                        : rew + a.gamma * q * (1.f - done);
        }
        __syncthreads();
    }

    // ================================ critic step ================================
    float lossp = 0.f;
    for (int h = 0; h < heads; ++h) {
        for (int r0 = 0; r0 < B; r0 += rc) {
            const int nv = min(rc, B - r0);
            gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.obs_off[0], OT + AT, 0);
            zero_cols(S.xin, S.xp, rc, OT + AT, kc0);
            __syncthreads();
            mlp_fwd(NC, h * ql, ql, thC, S, ACT_NONE);
            const int npad = NC.L[h * ql + ql - 1].n_pad;
            for (int e = threadIdx.x; e < rc * npad; e += kWG) {
                const int r = e / npad, c = e - r * npad;
                float d = 0.f;
                if (c == 0 && r < nv) {
                    const float diff = S.outb[r * S.op] - S.y[r0 + r];
                    d = 2.f * diff * invB;
                    lossp += diff * diff;
                }
                S.outb[r * S.op + c] = d;
            }
            __syncthreads();
            mlp_bwd(NC, h * ql, ql, thC, gC, S, r0 == 0, false, 0, 0);
        }
    }
    const float closs = block_sum(lossp, S.red) * invB;
    __syncthreads();
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    float* st = D.stats + ((size_t)p * n + ag) * ST_COUNT;
    const int tC = steps[2 * ag + 1] + 1;
    const bool fold_targets = !maddpg && a.do_actor;
    const float gnC = adam_net(NC.size, thC, D.m + offC, D.v + offC, gC, fold_targets ? tgC : nullptr, a.critic_lr,
                               a.adam_eps, a.beta1, a.beta2, a.critic_wd, a.clip_norm, tC, a.tau, S.red);
    __syncthreads();
    if (threadIdx.x == 0) {
        steps[2 * ag + 1] = tC;
        st[ST_CRITIC_LOSS] = closs;
        st[ST_CRITIC_GNORM] = gnC;
    }
    if (!a.do_actor) return;

    // ================================ actor step ================================
    float alossp = 0.f, entp = 0.f, gls = 0.f;      // gls: d loss / d log_std[threadIdx.x] (SAC)
    const int ct0 = (OT + acol) / 16, ct1 = (OT + acol + Aa + 15) / 16;
    const int nq = sac ? heads : 1;                 // SAC: mean of the twins; TD3: Q1 only (TD3.py:227)
    const float dq = sac ? -0.5f * invB : -invB;
    for (int r0 = 0; r0 < B; r0 += rc) {
        const int nv = min(rc, B - r0);
        // -- a = actor(obs)
        gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.obs_off[ag], Oa, 0);
        zero_cols(S.xin, S.xp, rc, Oa, NA.L[0].k_pad);
        __syncthreads();
        mlp_fwd(NA, 0, NA.n_layers, thA, S, sac ? ACT_NONE : ACT_TANH);
        float lp = 0.f;
        if (sac) {
            const int r = threadIdx.x;
            if (r < rc) {
                for (int c = 0; c < Aa; ++c) {
                    const float mean = S.outb[r * S.op + c];
                    const float ls = fminf(fmaxf(thA[NA.extra_off + c], -20.f), 2.f);
                    const float sd = expf(ls);
                    const float eps = (r < nv) ? noise1[(size_t)(r0 + r) * am + c] : 0.f;
                    const float u = mean + sd * eps;
                    const float du = u - mean;
                    lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                    lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                    S.abuf[r * S.ap + c] = tanhf(u);
                }
            }
        } else {
            for (int e = threadIdx.x; e < rc * Aa; e += kWG) {
                const int r = e / Aa, c = e - r * Aa;
                S.abuf[r * S.ap + c] = S.outb[r * S.op + c];
            }
        }
        for (int e = threadIdx.x; e < rc * Aa; e += kWG) {
            const int r = e / Aa, c = e - r * Aa;
            S.dabuf[r * S.ap + c] = 0.f;
        }
        __syncthreads();
        // -- dQ/da through the critic head(s), parameters frozen
        float qsum = 0.f;
        for (int h = 0; h < nq; ++h) {
            gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.obs_off[0], OT + AT, 0);
            zero_cols(S.xin, S.xp, rc, OT + AT, kc0);
            __syncthreads();
            for (int e = threadIdx.x; e < rc * Aa; e += kWG) {
                const int r = e / Aa, c = e - r * Aa;
                S.xin[r * S.xp + OT + acol + c] = S.abuf[r * S.ap + c];
            }
            __syncthreads();
            mlp_fwd(NC, h * ql, ql, thC, S, ACT_NONE);
            if (threadIdx.x < nv) qsum += S.outb[threadIdx.x * S.op];
            __syncthreads();
            const int npad = NC.L[h * ql + ql - 1].n_pad;
            for (int e = threadIdx.x; e < rc * npad; e += kWG) {
                const int r = e / npad, c = e - r * npad;
                S.outb[r * S.op + c] = (c == 0 && r < nv) ? dq : 0.f;
            }
            __syncthreads();
            mlp_bwd(NC, h * ql, ql, thC, nullptr, S, false, true, ct0, ct1);
            for (int e = threadIdx.x; e < rc * Aa; e += kWG) {
                const int r = e / Aa, c = e - r * Aa;
                S.dabuf[r * S.ap + c] += S.xin[r * S.xp + OT + acol + c];
            }
            __syncthreads();
        }
        if (threadIdx.x < nv) {
            if (sac) {
                alossp += -(qsum * 0.5f) - alpha * (-lp);     // (-Q_pi - alpha*entropy), SAC.py:251
                entp += -lp;
            } else {
                alossp += -qsum;
            }
        }
        // -- actor forward again (activations for its backward), head delta, backward
        gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.obs_off[ag], Oa, 0);
        zero_cols(S.xin, S.xp, rc, Oa, NA.L[0].k_pad);
        __syncthreads();
        mlp_fwd(NA, 0, NA.n_layers, thA, S, sac ? ACT_NONE : ACT_TANH);
        const int napad = NA.L[NA.n_layers - 1].n_pad;
        for (int e = threadIdx.x; e < rc * napad; e += kWG) {
            const int r = e / napad, c = e - r * napad;
            float d = 0.f;
            if (r < nv && c < Aa) {
                if (sac) {
                    const float av = S.abuf[r * S.ap + c];
                    d = S.dabuf[r * S.ap + c] * (1.f - av * av) + (alpha * invB) * (2.f * av);
                    const float ls = fminf(fmaxf(thA[NA.extra_off + c], -20.f), 2.f);
                    // per-element contribution to d/d log_std, column-summed below
                    S.dabuf[r * S.ap + c] = d * expf(ls) * noise1[(size_t)(r0 + r) * am + c] - alpha * invB;
                } else {
                    const float av = S.outb[r * S.op + c];      // tanh output
                    d = S.dabuf[r * S.ap + c] * (1.f - av * av);
                }
            } else if (sac && c < Aa) {
                S.dabuf[r * S.ap + c] = 0.f;
            }
            S.outb[r * S.op + c] = d;
        }
        __syncthreads();
        if (sac && threadIdx.x < Aa)
            for (int r = 0; r < rc; ++r) gls += S.dabuf[r * S.ap + threadIdx.x];
        mlp_bwd(NA, 0, NA.n_layers, thA, gA, S, r0 == 0, false, 0, 0);
    }
    if (sac && threadIdx.x < Aa) {
        const float raw = thA[NA.extra_off + threadIdx.x];
        gA[NA.extra_off + threadIdx.x] = (raw >= -20.f && raw <= 2.f) ? gls : 0.f;
    }
    const float aloss = block_sum(alossp, S.red) * invB;
    const float ent_mean = sac ? block_sum(entp, S.red) * invB : 0.f;
    __syncthreads();
    const int tA = steps[2 * ag] + 1;
    const float gnA = adam_net(NA.size, thA, D.m + offA, D.v + offA, gA, maddpg ? nullptr : tgA, a.actor_lr, a.adam_eps,
                               a.beta1, a.beta2, 0.f, a.clip_norm, tA, a.tau, S.red);
    if (threadIdx.x == 0) {
        steps[2 * ag] = tA;
        st[ST_ACTOR_LOSS] = aloss;
        st[ST_ACTOR_GNORM] = gnA;
        if (sac) {
            // Alpha.update_alpha (SAC.py:154-169,257-260): loss = (alpha*(entropy - H_target).detach()).mean()
            float* al = D.alpha + p * 4;
            const float mean_term = ent_mean - a.target_entropy;
            const float g = alpha * mean_term;            // d loss / d log_alpha
            const int t = steps[kMaxNets] + 1;
            float mi = al[1], vi = al[2];
            mi = mi + (g - mi) * (1.f - a.beta1);
            vi = vi * a.beta2 + ((1.f - a.beta2) * g) * g;
            const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
            const float denom = sqrtf(vi) / (float)sqrt(bc2) + 1e-8f;
            al[0] = al[0] - (float)((double)a.alpha_lr / bc1) * (mi / denom);
            al[1] = mi;
            al[2] = vi;
            al[3] = expf(al[0]);
            steps[kMaxNets] = t;
            st[ST_ALPHA_LOSS] = alpha * mean_term;
            st[ST_ALPHA] = al[3];
            st[ST_ENTROPY] = ent_mean;
        }
    }
}

// MADDPG.update_target (MADDPG_simple.py:188-195): every agent's actor then critic.
__global__ __launch_bounds__(256) void soft_update_kernel(const EngineDesc* __restrict__ Dp, float tau) {
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x / D.n_nets, net = blockIdx.x - p * D.n_nets;
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    soft_update_net(D.net[net].size, D.target + off, D.theta + off, tau);
}

}  // namespace frl
