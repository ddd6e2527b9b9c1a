// Replay-sample + batched-update kernels of the off-policy learners (DQN, DDPG, TD3, SAC, MADDPG).
//
// One learn() of every resident learner = a short chain of launches:
//     [draw]  -> grad(critic) -> adam(critic) [-> grad(actor) -> adam(actor)] [-> soft update]
// replacing, per learner, the reference's ~60 (DQN) to ~700 (MADDPG) eager ATen launches per
// learn() and the 5 H2D copies of Buffer.sample (TD3_file/Buffer.py:50-55).
//
// Decomposition (chosen from rocprofv3 counters, profiles/README.md): the batch of one
// (learner, agent) is split into row chunks of `rc` rows; ONE workgroup owns ONE chunk:
// record gather from the HBM ring -> forward passes -> deltas -> backward (all MFMA), its weight
// gradients written ONCE to its partial slab.  The chunks of a learner are placed on the same
// XCD (block -> XCD is id % 8), so the weights they share are served by that XCD's L2 instead
// of being re-fetched per chunk.  `adam_kernel` then sums the slabs in a fixed order
// (deterministic), applies clip_grad_norm_, Adam and the soft target update while streaming
// theta/m/v once.  P learners (independent seeds) x chunks fill the 256 CUs; P = 1 still
// spreads one learner's batch over batch/rc CUs.
//
// This file: index / noise draw, Batch_ObsNorm statistics, reduce + clip + Adam (+ soft update).  The gradient kernels
// live in kernels_dqn.hip, kernels_critic.hip, kernels_actor.hip and kernels_c51.hip (separate translation units).
#include <hip/hip_runtime.h>

#include "device/net.hpp"
#include "device/update_common.hpp"
#include "kernels.h"
#include "device/noisy.hpp"

namespace frl {

// ----------------------------------------------------------------------------------- draw
// Device-side sampling: `batch` distinct rows per (learner, agent) (np.random.choice(size, B,
// replace=False), DQN.py:97) and the N(0,1) draws of TD3.py:197 / SAC.py:227,244.
__global__ __launch_bounds__(256) void draw_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int want_noise) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int n = D.n_agents, p = a.p0 + blockIdx.x / n, ag = blockIdx.x % n, B = a.batch;
    g_i idx = (g_i)(D.idx + ((size_t)p * n + ag) * D.batch_max);
    const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
    // (batches of 257 .. 2048 rows: the hash-table duplicate check; the launch carries its 2 x kDrawTable ints behind the row buffers)
    FRL_LDS int* lb = (FRL_LDS int*)smem;
    // (want_noise bit 1, FRL_DRAW_SCAN=1: the scan of every entry's predecessors whatever the batch — tests hold the two to the same rows)
    draw_indices(idx, lb, B, a.size, a.rng_counter, (unsigned)ag, key, true, (B > kWG && 4 * B <= kDrawTable && !(want_noise & 2)) ? lb + 2 * ((B + 3) & ~3) : nullptr);
    if (want_noise & 1) {
        const int am = D.act_max, NS = D.noise_sets;
        for (int s2 = 0; s2 < NS; s2 += 2) {          // two sets per Philox draw
            g_f noise0 = as_global(D.noise + (((size_t)p * n + ag) * NS + s2) * D.batch_max * am);
            g_f noise1 = noise0 + (size_t)D.batch_max * am;
            for (int e = threadIdx.x; e < B * am; e += kWG) {
                float n0, n1;
                normal2(philox4x32_10(a.rng_counter, 0x4000u + (unsigned)ag + 0x100u * (unsigned)s2, (unsigned)e, key), n0, n1);
                noise0[e] = n0;
                if (s2 + 1 < NS) noise1[e] = n1;
            }
        }
    }
}

// ----------------------------------------------------------------------------- Batch_ObsNorm
// RunningMeanStd_batch_size.update (PPO_file/normalization.py:61-71; SAC.py:215, DDPG.py:191) with
// the batch mean of the sampled observations: n += 1; first call: mean = std = xbar; afterwards a
// Welford step on the batch means.  One workgroup per learner, before the gradient kernels.
__global__ __launch_bounds__(256) void obsnorm_kernel(const EngineDesc* __restrict__ Dp, int batch, int all_rows, int p0) {
    const EngineDesc& D = *Dp;
    const int p = p0 + blockIdx.x, n = D.n_agents, W = D.obsnorm_w;
    const RecordDesc& R = D.rec;
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_f base = as_global(D.obsnorm + (size_t)p * n * n * W);
    // version i = the statistics after updating agent i's sample() (MADDPG.py:189-197: one index draw per updating agent,
    // every agent's running mean/std updated with the batch mean of ITS observations over those rows); n = 1: in place
    for (int i = 0; i < n; ++i) {
        g_ci idx = as_global_i(D.idx + ((size_t)p * n + i) * D.batch_max);
        g_cf prev = base + (size_t)(i == 0 ? n - 1 : i - 1) * n * W;
        g_f cur = base + (size_t)i * n * W;
        for (int e = threadIdx.x; e < n * W; e += kWG) {
            const int j = e / W, c = e - j * W - 1;              // c = -1: the count slot
            const int O = R.obs_dim[j];
            g_cf ps = prev + (size_t)j * W;
            g_f cs = cur + (size_t)j * W;
            const float cnt = ps[0] + 1.f;
            if (c < 0) { if (n > 1) cs[0] = cnt; continue; }
            if (c >= O) continue;
            float s = 0.f;
            for (int r = 0; r < batch; ++r) s += ring[(size_t)(all_rows ? r : idx[r]) * R.stride + R.obs_off[j] + c];
            const float xbar = s / (float)batch;
            if (cnt == 1.f) {
                cs[1 + c] = xbar;
                cs[1 + O + c] = ps[1 + O + c];
                cs[1 + 2 * O + c] = xbar;                      // the reference's first update sets std = x (normalization.py:63-65)
            } else {
                const float old = ps[1 + c];
                const float m2 = old + (xbar - old) / cnt;
                const float S2 = ps[1 + O + c] + (xbar - old) * (xbar - m2);
                cs[1 + c] = m2;
                cs[1 + O + c] = S2;
                cs[1 + 2 * O + c] = sqrtf(S2 / cnt);
            }
        }
        __syncthreads();
        if (n == 1 && threadIdx.x == 0) cur[0] = cur[0] + 1.f;     // in place: the count moves after every column has read it
        __syncthreads();
    }
}

// ------------------------------------------------------------------- reduce + clip + Adam
// Two bandwidth-bound launches over float4 lanes, `G` workgroups per (learner, agent) net:
//   reduce_kernel: g = sum over the row-chunk slabs in a fixed order (deterministic), write g,
//                  per-workgroup sum of squares -> gsq; workgroup 0 advances the Adam step count.
//   adam_kernel  : total norm from gsq -> clip_grad_norm_ coefficient -> torch-order Adam ->
//                  optional soft target update, streaming theta/m/v/target once.  Workgroup 0
//                  also publishes the losses and performs SAC's alpha step (SAC.py:154-169,257-260).
// which = 0: critic / Q-net, 1: actor.


__device__ __forceinline__ int adam_net_index(const EngineDesc& D, int which, int ag) {
    return (D.algo == ALGO_DQN) ? 0 : (which == 0 ? 2 * ag + 1 : 2 * ag);
}

// thread 0 of a unit's first Adam workgroup: losses of the step, SAC's alpha update (SAC.py:154-169,257-260)
__device__ __forceinline__ void adam_publish(const EngineDesc& D, const AdamArgs& a, int p, int ag, float total, int* steps) {
    const int n = D.n_agents;
    const float* pt = D.part + ((size_t)p * n + ag) * D.S * 4;
    float l0 = 0.f, l1 = 0.f;
    for (int k = 0; k < a.ns; ++k) { l0 += pt[4 * k]; l1 += pt[4 * k + 1]; }
    float* st = D.stats + ((size_t)p * n + ag) * ST_COUNT;
    const float invB = 1.f / (float)a.batch;
    st[a.which == 0 ? ST_CRITIC_LOSS : ST_ACTOR_LOSS] = l0 * invB;
    st[a.which == 0 ? ST_CRITIC_GNORM : ST_ACTOR_GNORM] = total;
    if (a.sac_alpha) {
        float* al = D.alpha + p * 4;
        const float alpha = al[3];
        const float ent_mean = l1 * invB;
        const float mean_term = ent_mean - a.target_entropy;
        const float gl = alpha * mean_term;            // d alpha_loss / d log_alpha
        const int ta = steps[kMaxNets] + 1;
        float mi = al[1], vi = al[2];
        mi = mi + (gl - mi) * (1.f - a.beta1);
        vi = vi * a.beta2 + ((1.f - a.beta2) * gl) * gl;
        const double b1 = 1.0 - powi_d((double)a.beta1, ta), b2 = 1.0 - powi_d((double)a.beta2, ta);
        const float denom = sqrtf(vi) / (float)sqrt(b2) + 1e-8f;
        al[0] = al[0] - (float)((double)a.alpha_lr / b1) * (mi / denom);
        al[1] = mi;
        al[2] = vi;
        al[3] = expf(al[0]);
        steps[kMaxNets] = ta;
        st[ST_ALPHA_LOSS] = alpha * mean_term;
        st[ST_ALPHA] = al[3];
        st[ST_ENTROPY] = ent_mean;
    }
}

__global__ __launch_bounds__(256) void reduce_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a) {
    __shared__ float red_s[8];
    lds_f red = (lds_f)red_s;
    const EngineDesc& D = *Dp;
    const int n = D.n_agents, wg = blockIdx.x % a.G;
    const int p = a.p0 + (blockIdx.x / a.G) / n, ag = (blockIdx.x / a.G) % n, unit = p * n + ag;
    const int net = adam_net_index(D, a.which, ag);
    const NetDesc& N = D.net[net];
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    const int n4 = N.size / 4;
    FRL_GLB f32x4* g = (FRL_GLB f32x4*)(D.grad + off);
    const FRL_GLB f32x4* slab = (const FRL_GLB f32x4*)(D.slab + (size_t)p * D.S * D.learner_stride + D.net_off[net]);
    const size_t ls4 = (size_t)D.learner_stride / 4;
    float ss = 0.f;
    const int base = wg * (kWG * kAdamVec);
    // NoisyLinear head (DQN only): the shadow layer's (sigma's) gradient is the head's (mu's: where the effective weights'
    // gradient landed) times the noise of the differentiated forward — set 2, the online net on s.  Derived here from the
    // same slab sums instead of a launch of its own between the two passes.
    const bool noisy = D.noisy && D.algo == ALGO_DQN;
    const LayerDesc& H = N.L[N.n_layers - 1];
    const LayerDesc& SG = N.L[noisy ? N.n_layers : 0];
    const int sw0 = noisy ? SG.w_off / 4 : n4, sw1 = noisy ? sw0 + H.k_pad * H.n_pad / 4 : n4;
    const int sb0 = noisy ? SG.b_off / 4 : n4, sb1 = noisy ? sb0 + H.n_pad / 4 : n4;
    g_cf eps = noisy ? noisy_eps_of(D, H, p, 2) : nullptr;
#pragma unroll
    for (int j = 0; j < kAdamVec; ++j) {
        const int i = base + j * kWG + threadIdx.x;
        if (i < n4) {
            const bool sig_w = i >= sw0 && i < sw1, sig_b = i >= sb0 && i < sb1;
            const int src = sig_w ? H.w_off / 4 + (i - sw0) : (sig_b ? H.b_off / 4 + (i - sb0) : i);
            f32x4 s = slab[src];
            for (int k = 1; k < a.ns; ++k) s += slab[(size_t)k * ls4 + src];
            if (sig_w) {
                const int e0 = 4 * (i - sw0), kk = e0 / H.n_pad, n0 = e0 - kk * H.n_pad;        // Wk[k][n]: four outputs of one input
                s.x *= noisy_eps_w(eps, H, D.noisy_split, kk, n0); s.y *= noisy_eps_w(eps, H, D.noisy_split, kk, n0 + 1);
                s.z *= noisy_eps_w(eps, H, D.noisy_split, kk, n0 + 2); s.w *= noisy_eps_w(eps, H, D.noisy_split, kk, n0 + 3);
            } else if (sig_b) {
                const int n0 = 4 * (i - sb0);
                s.x *= noisy_eps_b(eps, H, D.noisy_split, n0); s.y *= noisy_eps_b(eps, H, D.noisy_split, n0 + 1);
                s.z *= noisy_eps_b(eps, H, D.noisy_split, n0 + 2); s.w *= noisy_eps_b(eps, H, D.noisy_split, n0 + 3);
            }
            g[i] = s;
            ss += s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
        }
    }
    const float tot = block_sum(ss, red);
    if (threadIdx.x == 0) {
        D.gsq[(size_t)unit * D.Gmax + wg] = tot;
        if (wg == 0) D.steps[(size_t)p * (kMaxNets + 1) + net] += 1;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a) {
    const EngineDesc& D = *Dp;
    const int n = D.n_agents, wg = blockIdx.x % a.G;
    const int p = a.p0 + (blockIdx.x / a.G) / n, ag = (blockIdx.x / a.G) % n, unit = p * n + ag;
    const int net = adam_net_index(D, a.which, ag);
    const NetDesc& N = D.net[net];
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    const int n4 = N.size / 4, Gn = (n4 + kWG * kAdamVec - 1) / (kWG * kAdamVec);
    float ss = 0.f;
    for (int k = 0; k < Gn; ++k) ss += D.gsq[(size_t)unit * D.Gmax + k];      // same order in every workgroup
    const float total = sqrtf(ss);
    float coef = 1.f;
    if (a.clip > 0.f) coef = fminf(a.clip / (total + 1e-6f), 1.f);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t = steps[net];                                                   // advanced by reduce_kernel
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    const float step = (float)((double)a.lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
    const FRL_GLB f32x4* g = (const FRL_GLB f32x4*)(D.grad + off);
    FRL_GLB f32x4* th = (FRL_GLB f32x4*)(D.theta + off);
    FRL_GLB f32x4* m = (FRL_GLB f32x4*)(D.m + off);
    FRL_GLB f32x4* v = (FRL_GLB f32x4*)(D.v + off);
    FRL_GLB f32x4* tg = (FRL_GLB f32x4*)(D.target + off);
    const int base = wg * (kWG * kAdamVec);
#pragma unroll
    for (int j = 0; j < kAdamVec; ++j) {
        const int i = base + j * kWG + threadIdx.x;
        if (i < n4) {
            f32x4 gi = g[i] * coef, thi = th[i], mi = m[i], vi = v[i];
            if (a.wd != 0.f) gi += a.wd * thi;
            mi = mi + (gi - mi) * w1;
            vi = vi * a.beta2 + (w2 * gi) * gi;
            f32x4 denom;
            denom.x = sqrtf(vi.x) / bc2s + a.eps; denom.y = sqrtf(vi.y) / bc2s + a.eps;
            denom.z = sqrtf(vi.z) / bc2s + a.eps; denom.w = sqrtf(vi.w) / bc2s + a.eps;
            thi = thi - step * (mi / denom);
            m[i] = mi; v[i] = vi; th[i] = thi;
            if (a.soft) tg[i] = tg[i] * tk + thi * a.tau;
        }
    }
    if (wg == 0 && threadIdx.x == 0) adam_publish(D, a, p, ag, total, steps);
}

// One launch instead of reduce_kernel + adam_kernel when a net's gradient fits in the registers of ONE 1024-thread
// workgroup (<= kFusedVec float4 per thread: every 128-wide net of the reference): the slabs are summed into registers,
// the norm is a block reduction, and the Adam pass takes its gradient from the registers — the reduced gradient is
// never written, the norm partials never leave the workgroup.  11 -> 9 floats of traffic per parameter at 2 slabs.
template <int VEC, bool WIDE>
__device__ __forceinline__ void adam_fused_body(const EngineDesc& D, const AdamArgs& a, float* red) {
    const int n = D.n_agents;
    const int p = a.p0 + blockIdx.x / n, ag = blockIdx.x % n;
    const int net = adam_net_index(D, a.which, ag);
    const NetDesc& N = D.net[net];
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    const int n4 = N.size / 4;
    const FRL_GLB f32x4* slab = (const FRL_GLB f32x4*)(D.slab + (size_t)p * D.S * D.learner_stride + D.net_off[net]);
    const size_t ls4 = (size_t)D.learner_stride / 4;
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t = steps[net] + 1;                             // in flight with the slab loads
    f32x4 g[VEC];
    if constexpr (!WIDE) {
        // slab-major: all of a thread's loads of one slab in flight together (element-major, each element's slab sum was a
        // dependent chain of waits: 57 us for 64 learners); per element the slabs are still added in index order
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int i = j * kFusedThreads + threadIdx.x;
            g[j] = (i < n4) ? slab[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int k = 1; k < a.ns; ++k) {
            const FRL_GLB f32x4* sk = slab + (size_t)k * ls4;
            f32x4 tmp[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int i = j * kFusedThreads + threadIdx.x;
                tmp[j] = (i < n4) ? sk[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) g[j] += tmp[j];
        }
    } else {
        // the wide instance (nets up to 1024 x VEC float4: DQN_with_tricks' Categorical + Noisy head, 68 k parameters with its
        // sigma layer) has no registers for a second set: element-major sums, and — like reduce_kernel — the sigma layer's
        // gradient taken from the head's slab sums times the differentiated forward's noise
        const bool noisy = D.noisy && D.algo == ALGO_DQN;
        const LayerDesc& H = N.L[N.n_layers - 1];
        const LayerDesc& SG = N.L[noisy ? N.n_layers : 0];
        const int sw0 = noisy ? SG.w_off / 4 : n4, sw1 = noisy ? sw0 + H.k_pad * H.n_pad / 4 : n4;
        const int sb0 = noisy ? SG.b_off / 4 : n4, sb1 = noisy ? sb0 + H.n_pad / 4 : n4;
        g_cf eps = noisy ? noisy_eps_of(D, H, p, 2) : nullptr;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int i = j * kFusedThreads + threadIdx.x;
            f32x4 sv = {0.f, 0.f, 0.f, 0.f};
            if (i < n4) {
                const bool sig_w = i >= sw0 && i < sw1, sig_b = i >= sb0 && i < sb1;
                const int src = sig_w ? H.w_off / 4 + (i - sw0) : (sig_b ? H.b_off / 4 + (i - sb0) : i);
                sv = slab[src];
                for (int k = 1; k < a.ns; ++k) sv += slab[(size_t)k * ls4 + src];
                if (sig_w) {
                    const int e0 = 4 * (i - sw0), kk = e0 / H.n_pad, n0 = e0 - kk * H.n_pad;
                    sv.x *= noisy_eps_w(eps, H, D.noisy_split, kk, n0); sv.y *= noisy_eps_w(eps, H, D.noisy_split, kk, n0 + 1);
                    sv.z *= noisy_eps_w(eps, H, D.noisy_split, kk, n0 + 2); sv.w *= noisy_eps_w(eps, H, D.noisy_split, kk, n0 + 3);
                } else if (sig_b) {
                    const int n0 = 4 * (i - sb0);
                    sv.x *= noisy_eps_b(eps, H, D.noisy_split, n0); sv.y *= noisy_eps_b(eps, H, D.noisy_split, n0 + 1);
                    sv.z *= noisy_eps_b(eps, H, D.noisy_split, n0 + 2); sv.w *= noisy_eps_b(eps, H, D.noisy_split, n0 + 3);
                }
            }
            g[j] = sv;
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) ss += g[j].x * g[j].x + g[j].y * g[j].y + g[j].z * g[j].z + g[j].w * g[j].w;
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < kFusedThreads / 64; ++w) tot += red[w];
    const float total = sqrtf(tot);
    float coef = 1.f;
    if (a.clip > 0.f) coef = fminf(a.clip / (total + 1e-6f), 1.f);
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    const float step = (float)((double)a.lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
    FRL_GLB f32x4* th = (FRL_GLB f32x4*)(D.theta + off);
    FRL_GLB f32x4* m = (FRL_GLB f32x4*)(D.m + off);
    FRL_GLB f32x4* v = (FRL_GLB f32x4*)(D.v + off);
    FRL_GLB f32x4* tg = (FRL_GLB f32x4*)(D.target + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int i = j * kFusedThreads + threadIdx.x;
        if (i < n4) {
            f32x4 gi = g[j] * coef, thi = th[i], mi = m[i], vi = v[i];
            if (a.wd != 0.f) gi += a.wd * thi;
            mi = mi + (gi - mi) * w1;
            vi = vi * a.beta2 + (w2 * gi) * gi;
            f32x4 denom;
            denom.x = sqrtf(vi.x) / bc2s + a.eps; denom.y = sqrtf(vi.y) / bc2s + a.eps;
            denom.z = sqrtf(vi.z) / bc2s + a.eps; denom.w = sqrtf(vi.w) / bc2s + a.eps;
            thi = thi - step * (mi / denom);
            m[i] = mi; v[i] = vi; th[i] = thi;
            if (a.soft) tg[i] = tg[i] * tk + thi * a.tau;
        }
    }
    if (threadIdx.x == 0) {
        steps[net] = t;
        adam_publish(D, a, p, ag, total, steps);
    }
}

__global__ __launch_bounds__(kFusedThreads) void adam_fused_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a) {
    __shared__ float red[kFusedThreads / 64];
    adam_fused_body<kFusedVec, false>(*Dp, a, red);
}
__global__ __launch_bounds__(kFusedThreads) void adam_fused_wide_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a) {
    __shared__ float red[kFusedThreads / 64];
    adam_fused_body<kFusedVecWide, true>(*Dp, a, red);
}

// MADDPG.update_target (MADDPG_simple.py:188-195): every agent's actor then critic.
__global__ __launch_bounds__(256) void soft_update_kernel(const EngineDesc* __restrict__ Dp, float tau, int p0) {
    const EngineDesc& D = *Dp;
    const int p = p0 + blockIdx.x / D.n_nets, net = blockIdx.x % D.n_nets;
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    // (grid.y workgroups share a net by stride: one per net streamed config 5's 29 k-float critics at one load in flight per thread, 34 us)
    g_f target = as_global(D.target + off);
    g_cf theta = as_global(D.theta + off);
    const float tk = 1.f - tau;
    for (int i = blockIdx.y * kWG + threadIdx.x; i < D.net[net].size; i += gridDim.y * kWG) target[i] = target[i] * tk + theta[i] * tau;
}

// Fragment-image order -> Wk[k][n] for every weight block of the frag nets of all learners, in all four parameter arrays: what
// frl_obsnorm_enable does to an engine created for the register-chained kernels before it hands it to the row-chunk family
// (those kernels read Wk; Batch_ObsNorm is theirs).  grid = (P, 4 arrays); a learner's block of array y goes through
// scratch[y][p][learner_stride] — `scratch` is FOUR times the engine's reduced-gradient array (frl_obsnorm_enable allocates it;
// D.grad itself is too small).
__global__ __launch_bounds__(256) void relayout_to_wk_kernel(const EngineDesc* __restrict__ Dp, float* scratch) {
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x;
    float* arr = blockIdx.y == 0 ? D.theta : (blockIdx.y == 1 ? D.target : (blockIdx.y == 2 ? D.m : D.v));
    float* sc = scratch + ((size_t)blockIdx.y * D.P + p) * D.learner_stride;
    for (int net = 0; net < D.n_nets; ++net) {
        const NetDesc& N = D.net[net];
        if (!N.frag) continue;
        float* blk = arr + (size_t)p * D.learner_stride + D.net_off[net];
        for (int li = 0; li < N.n_layers + N.n_shadow; ++li) {
            const LayerDesc& L = N.L[li];
            const int cnt = L.n_pad * L.k_pad;
            for (int i = threadIdx.x; i < cnt; i += kWG) sc[i] = blk[L.w_off + i];
            __syncthreads();
            for (int i = threadIdx.x; i < cnt; i += kWG) {
                const int k = i / L.n_pad, n = i - k * L.n_pad;
                blk[L.w_off + i] = sc[weight_index(N, L, n, k)];
            }
            __syncthreads();
        }
    }
}

// One fragment-image net of every learner -> a Wk-layout copy (dst[p][NetDesc::size]; biases / log_std / padding copied as they are):
// select_action on the engines of the K-sliced chained family and of kernels_solow.hip reads it through act_kernel
// (ActArgs::theta_alt).  grid = (P, any number of workgroups per learner): an element's source is found from its own index, so the
// workgroups share a net by stride (one workgroup per learner took 260 us for config 4's 67 k-float actor — more than the one
// learner's whole learn() on kernels_solow.hip, once per env step in its rollout loop)
__global__ __launch_bounds__(256) void frag_to_wk_kernel(const EngineDesc* __restrict__ Dp, int net, int use_target, float* dst) {
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x;
    const NetDesc& N = D.net[net];
    const float* src = (use_target ? D.target : D.theta) + (size_t)p * D.learner_stride + D.net_off[net];
    float* out = dst + (size_t)p * N.size;
    const int nl = N.n_layers + N.n_shadow;
    for (int i = blockIdx.y * kWG + threadIdx.x; i < N.size; i += gridDim.y * kWG) {
        int from = i;
        for (int li = 0; li < nl; ++li) {
            const LayerDesc& L = N.L[li];
            const int j = i - L.w_off;
            if (j >= 0 && j < L.n_pad * L.k_pad) { const int k = j / L.n_pad, n = j - k * L.n_pad; from = L.w_off + weight_index(N, L, n, k); }
        }
        out[i] = src[from];
    }
}

}  // namespace frl