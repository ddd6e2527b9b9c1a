// PPO on-policy update: value pass -> wave-scan GAE (+ advantage normalisation) -> ONE persistent
// launch running all K_epochs x (horizon / minibatch) clipped-surrogate actor and MSE critic
// steps of a learner (PPO_file/PPO_with_tricks.py:290-354).  The reference does the GAE
// recurrence as a sequential host loop (:308-311) after a device->host copy and launches a few
// hundred tiny kernels per minibatch; here the rollout never leaves HBM.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/net.hpp"
#include "device/special.hpp"
#include "device/ppo_timing.hpp"

namespace frl {


// ---- td_delta = r + gamma*(1-done)*V(s') - V(s) for every stored row (PPO_with_tricks.py:304-306)
__global__ __launch_bounds__(256) void ppo_values_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int p = blockIdx.y, r0 = blockIdx.x * D.rc, rc = D.rc, T = a.horizon;
    const int nv = min(rc, T - r0);
    const NetDesc& N = D.net[1];
    const RecordDesc& R = D.rec;
    const Lds S = carve_lds(smem, D.rc, D.hidden, D.lds_kin_pad, D.lds_out_pad, D.lds_batch_pad, D.lds_act_pad, D.lds_hbufs);
    g_cf theta = as_global(D.theta + (size_t)p * D.learner_stride + D.net_off[1]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    const int O = R.obs_dim[0], kpad = N.L[0].k_pad;
    float v0 = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        const int src = pass == 0 ? R.obs_off[0] : R.nobs_off[0];
        for (int e = threadIdx.x; e < rc * kpad; e += kWG) {
            const int r = e / kpad, c = e - r * kpad;
            S.xin[r * S.xp + c] = (r < nv && c < O) ? ring[(size_t)(r0 + r) * R.stride + src + c] : 0.f;
        }
        if (D.obs_norm_on) {
            __syncthreads();
            normalize_cols(S.xin, S.xp, nv, 0, O, as_global(D.obsnorm + (size_t)p * (1 + 3 * O)), O);
        }
        __syncthreads();
        mlp_fwd(N, 0, N.n_layers, theta, S, ACT_NONE);
        if (threadIdx.x < nv) {
            const float v = S.outb[threadIdx.x * S.op];
            if (pass == 0) {
                v0 = v;
            } else {
                g_cf rec = ring + (size_t)(r0 + threadIdx.x) * R.stride;
                const size_t o = (size_t)p * T + r0 + threadIdx.x;
                a.vs[o] = v0;
                a.td[o] = rec[R.rew_off] + a.gamma * (1.f - rec[R.done_off]) * v - v0;
            }
        }
        __syncthreads();
    }
}

// ---- GAE: A_t = delta_t + gamma*lambda*(1-adv_done_t)*A_{t+1}, reverse scan over the horizon.
// Affine maps m_t(x) = delta_t + g_t*x compose associatively; each thread folds a contiguous
// segment, a wave-shuffle suffix scan + 4 wave totals in LDS give every segment its incoming
// value, and the segment is replayed.  One workgroup per sequence.
struct Affine { float a, b; };
__device__ __forceinline__ Affine compose(Affine f, Affine g) { return Affine{f.a * g.a, f.a * g.b + f.b}; }   // f(g(x))

__device__ __forceinline__ void gae_scan(g_cf delta, g_cf ring, int stride, int adv_done_col, int T, float c, g_f adv,
                                         lds_f lds) {
    const int t = threadIdx.x, L = (T + kWG - 1) / kWG;
    const int s0 = min(t * L, T), s1 = min(s0 + L, T);
    Affine f{1.f, 0.f};
    for (int i = s1 - 1; i >= s0; --i) {
        const float g = c * (1.f - ring[(size_t)i * stride + adv_done_col]);
        f = Affine{g * f.a, delta[i] + g * f.b};
    }
    // inclusive suffix scan inside the wave
    const int lane = t & 63, w = t >> 6;
    Affine s = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Affine o{__shfl_down(s.a, off, 64), __shfl_down(s.b, off, 64)};
        if (lane + off < 64) s = compose(s, o);
    }
    if (lane == 0) { lds[2 * w] = s.a; lds[2 * w + 1] = s.b; }
    __syncthreads();
    float right = 0.f;                                    // A just right of this wave's last segment
    for (int ww = kWaves - 1; ww > w; --ww) right = lds[2 * ww] * right + lds[2 * ww + 1];
    const float mine = s.a * right + s.b;                 // A at the start of my segment
    float x = __shfl_down(mine, 1, 64);                   // A at the start of the next segment
    if (lane == 63) x = right;
    for (int i = s1 - 1; i >= s0; --i) {
        const float g = c * (1.f - ring[(size_t)i * stride + adv_done_col]);
        x = delta[i] + g * x;
        adv[i] = x;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void ppo_gae_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a) {
    __shared__ float lds_s[16];
    lds_f lds = (lds_f)lds_s;
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x, T = a.horizon;
    const RecordDesc& R = D.rec;
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_f adv_raw = as_global(a.adv_raw + (size_t)p * T);
    g_f adv = as_global(a.adv + (size_t)p * T);
    g_cf vs = as_global(a.vs + (size_t)p * T);
    g_f vt = as_global(a.vtarget + (size_t)p * T);
    const int adv_done_col = R.extra_off + R.extra - 1;   // last extra column (PPO_file/Buffer.py:282)
    gae_scan(as_global(a.td + (size_t)p * T), ring, R.stride, adv_done_col, T, a.gamma * a.lmbda, adv_raw, lds);
    // v_target = adv + V(s) (:313); optional (adv - mean)/(std + 1e-8), unbiased std (:314-315)
    float s1 = 0.f;
    for (int i = threadIdx.x; i < T; i += kWG) {
        const float x = adv_raw[i];
        vt[i] = x + vs[i];
        s1 += x;
    }
    if (!a.adv_norm) {
        for (int i = threadIdx.x; i < T; i += kWG) adv[i] = adv_raw[i];
        return;
    }
    const float mean = block_sum(s1, lds + 8) / (float)T;
    float s2 = 0.f;
    for (int i = threadIdx.x; i < T; i += kWG) {
        const float d = adv_raw[i] - mean;
        s2 += d * d;
    }
    const float sd = sqrtf(block_sum(s2, lds + 8) / (float)(T - 1));
    for (int i = threadIdx.x; i < T; i += kWG) adv[i] = (adv_raw[i] - mean) / (sd + 1e-8f);
}

// ---- PPO_advance/Buffer.py:480-507 `compute_returns_and_advantage` (stable-baselines3's form): the values were stored
// at rollout time; float64 throughout (the reference scans float64 NumPy arrays with Python-float gamma / lambda), one
// cast to float32 at the end (PPO_2.py:223-224).  Same affine suffix scan as gae_scan, in double.
struct AffineD { double a, b; };
__device__ __forceinline__ AffineD compose(AffineD f, AffineD g) { return AffineD{f.a * g.a, f.a * g.b + f.b}; }

__global__ __launch_bounds__(256) void ppo_gae_sb3_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a) {
    __shared__ double lds_d[8];
    __shared__ float lds_s[16];
    lds_f lds = (lds_f)lds_s;
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x, T = a.horizon, t = threadIdx.x;
    const RecordDesc& R = D.rec;
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_f adv_raw = as_global(a.adv_raw + (size_t)p * T);
    g_f adv = as_global(a.adv + (size_t)p * T);
    g_f vt = as_global(a.vtarget + (size_t)p * T);
    const int ad_col = R.extra_off + R.extra - 1, v_col = ad_col - 1;
    const double gam = a.gamma_d, c = a.gamma_d * a.lmbda_d, last = (double)a.last_value[p];
    auto step = [&](int i, double& g, double& delta) {
        g_cf rec = ring + (size_t)i * R.stride;
        const double nv = (i == T - 1) ? last : (double)rec[R.stride + v_col];
        delta = (double)rec[R.rew_off] + gam * nv * (1.0 - (double)rec[R.done_off]) - (double)rec[v_col];
        g = c * (1.0 - (double)rec[ad_col]);
    };
    const int L = (T + kWG - 1) / kWG, s0 = min(t * L, T), s1 = min(s0 + L, T);
    AffineD f{1.0, 0.0};
    for (int i = s1 - 1; i >= s0; --i) {
        double g, d;
        step(i, g, d);
        f = AffineD{g * f.a, d + g * f.b};
    }
    const int lane = t & 63, w = t >> 6;
    AffineD s = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        AffineD o{__shfl_down(s.a, off, 64), __shfl_down(s.b, off, 64)};
        if (lane + off < 64) s = compose(s, o);
    }
    if (lane == 0) { lds_d[2 * w] = s.a; lds_d[2 * w + 1] = s.b; }
    __syncthreads();
    double right = 0.0;
    for (int ww = kWaves - 1; ww > w; --ww) right = lds_d[2 * ww] * right + lds_d[2 * ww + 1];
    const double mine = s.a * right + s.b;
    double x = __shfl_down(mine, 1, 64);
    if (lane == 63) x = right;
    float s1f = 0.f;
    for (int i = s1 - 1; i >= s0; --i) {
        double g, d;
        step(i, g, d);
        x = d + g * x;
        const float xf = (float)x;
        adv_raw[i] = xf;
        vt[i] = (float)(x + (double)ring[(size_t)i * R.stride + v_col]);
        s1f += xf;
    }
    __syncthreads();
    if (!a.adv_norm) {
        for (int i = t; i < T; i += kWG) adv[i] = adv_raw[i];
        return;
    }
    const float mean = block_sum(s1f, lds + 8) / (float)T;
    float s2 = 0.f;
    for (int i = t; i < T; i += kWG) {
        const float d = adv_raw[i] - mean;
        s2 += d * d;
    }
    const float sd = sqrtf(block_sum(s2, lds + 8) / (float)(T - 1));
    for (int i = t; i < T; i += kWG) adv[i] = (adv_raw[i] - mean) / (sd + 1e-8f);
}

// ---- np.random.permutation(horizon) per (learner, epoch) (PPO_with_tricks.py:320) on the device when the caller does not
// supply the draws: sort (32 random bits, index) pairs — a bitonic sort of one 64-bit word per row in LDS.  (On the host
// this was 13 of the 65 ms of a 256-learner learn(): a Fisher-Yates of P*K*T swaps plus a 21 MB upload.)
__global__ __launch_bounds__(256) void ppo_perm_kernel(int* __restrict__ perm, int T, int Tpad, unsigned long long counter,
                                                        unsigned long long seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int k = blockIdx.x, p = blockIdx.y, K = gridDim.x;
    const unsigned long long key = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(p + 1);
    for (int i = threadIdx.x; i < Tpad; i += kWG) {
        unsigned long long v = ~0ull;                                   // padding sorts to the end
        if (i < T) {
            const Philox4 r = philox4x32_10(counter, 0x7000u + (unsigned)k, (unsigned)(i >> 2), key);
            const unsigned w = (i & 3) == 0 ? r.x : ((i & 3) == 1 ? r.y : ((i & 3) == 2 ? r.z : r.w));
            v = ((unsigned long long)w << 32) | (unsigned)i;
        }
        keys[i] = v;
    }
    __syncthreads();
    for (int size = 2; size <= Tpad; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < Tpad / 2; t += kWG) {
                const int lo = 2 * t - (t & (stride - 1));              // element whose partner is lo + stride
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    int* out = perm + ((size_t)p * K + k) * T;
    for (int i = threadIdx.x; i < T; i += kWG) out[i] = (int)(keys[i] & 0xFFFFFFFFu);
}

// stand-alone K3 entry (frl_gae): adv_done given as a dense array
__global__ __launch_bounds__(256) void gae_dense_kernel(const float* __restrict__ delta, const float* __restrict__ adv_done,
                                                         int T, float c, float* __restrict__ adv) {
    __shared__ float lds_s[16];
    const size_t o = (size_t)blockIdx.x * T;
    gae_scan(as_global(delta + o), as_global(adv_done + o), 1, 0, T, c, as_global(adv + o), (lds_f)lds_s);
}

// ---- all minibatch updates of one learner in one launch
__global__ __launch_bounds__(256) void ppo_update_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    // blockIdx.y: 0 = the actor's steps, 1 = the critic's.  Within learn() the two nets never read each other (the
    // advantages and value targets were fixed by the GAE pass, PPO_with_tricks.py:300-316), so their K_epochs x minibatch
    // step sequences run as two independent persistent workgroups.
    const int p = blockIdx.x, T = a.horizon, mb = a.minibatch, rc = D.rc;
    const bool do_actor = (blockIdx.y == 0), do_critic = (blockIdx.y == 1);
    const NetDesc& NA = D.net[0];
    const NetDesc& NC = D.net[1];
    const RecordDesc& R = D.rec;
    const Lds S = carve_lds(smem, D.rc, D.hidden, D.lds_kin_pad, D.lds_out_pad, D.lds_batch_pad, D.lds_act_pad, D.lds_hbufs);
    const size_t offA = (size_t)p * D.learner_stride + D.net_off[0], offC = (size_t)p * D.learner_stride + D.net_off[1];
    g_f thA = as_global(D.theta + offA);
    g_f thC = as_global(D.theta + offC);
    g_f gA = as_global(D.grad + offA);
    g_f gC = as_global(D.grad + offC);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_cf adv = as_global(a.adv + (size_t)p * T);
    g_cf vt = as_global(a.vtarget + (size_t)p * T);
    const bool discrete = D.n_discrete > 0, beta = D.beta_actor != 0;
    g_cf bn = D.obs_norm_on ? as_global(D.obsnorm + (size_t)p * (1 + 3 * R.obs_dim[0])) : nullptr;
    const int O = R.obs_dim[0], A = discrete ? D.n_discrete : R.act_dim[0], logp_col = R.extra_off;
    const int napad = NA.L[NA.n_layers - 1].n_pad, ncpad = NC.L[NC.n_layers - 1].n_pad;
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    int tA = steps[0], tC = steps[1];
    const int n_mb = (T + mb - 1) / mb;
    float* trace = a.trace + (size_t)p * a.k_epochs * n_mb * 2;
    constexpr float kHalfLog2PiPlusHalf = 1.41893853320467274178f;   // 0.5 + 0.5*log(2*pi)
    constexpr float kLogSqrt2Pi = 0.91893853320467274178f;
    PPO_T0();

    for (int k = 0; k < a.k_epochs; ++k) {
        g_ci perm = as_global_i(a.perm + ((size_t)p * a.k_epochs + k) * T);
        for (int s = 0; s < T; s += mb) {
            const int m = min(mb, T - s);
            const float invm = 1.f / (float)m;
            g_ci idx = perm + s;
            const int j_tr = k * n_mb + s / mb;
            if (do_actor) {
            // ---------------- actor: clipped surrogate + entropy bonus (:324-346)
            float lossp = 0.f, gls = 0.f, ent = 0.f;
            if (!discrete && !beta && threadIdx.x < A) {
                const float ls = fminf(fmaxf(thA[NA.extra_off + threadIdx.x], -20.f), 2.f);
                ent = kHalfLog2PiPlusHalf + ls;
            }
            const float ent_sum = block_sum(ent, S.red);          // same for every row
            for (int r0 = 0; r0 < m; r0 += rc) {
                const int nv = min(rc, m - r0);
                gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.obs_off[0], O, 0);
                zero_cols(S.xin, S.xp, rc, O, NA.L[0].k_pad);
                if (bn) { __syncthreads(); normalize_cols(S.xin, S.xp, nv, 0, O, bn, O); }
                __syncthreads();
                PPO_T(0);
                mlp_fwd(NA, 0, NA.n_layers, thA, S, (discrete || beta) ? ACT_NONE : ACT_TANH);    // mean = tanh(mean_layer(.)) (:99)
                PPO_T(1);
                if (beta) {
                    // Beta(alpha, beta) per action dimension (:325-332): alpha = softplus(z_a) + 1, beta = softplus(z_b) + 1;
                    // log_prob(a) = (alpha-1) ln a + (beta-1) ln(1-a) - ln B; entropy = ln B - (alpha-1) psi(alpha) -
                    // (beta-1) psi(beta) + (alpha+beta-2) psi(alpha+beta), summed over dimensions.  One thread per row.
                    if (threadIdx.x < rc) {
                        const int r = threadIdx.x;
                        if (r < nv) {
                            g_cf rec = ring + (size_t)idx[r0 + r] * R.stride;
                            double lp_now = 0.0, lp_old = 0.0, entr = 0.0;
                            for (int c = 0; c < A; ++c) {
                                const float za = S.outb[r * S.op + c], zb = S.outb[r * S.op + A + c];
                                const double al = (double)(za > 20.f ? za : log1pf(expf(za))) + 1.0;
                                const double be = (double)(zb > 20.f ? zb : log1pf(expf(zb))) + 1.0;
                                const double x = (double)rec[R.act_off[0] + c];
                                const double lB = lbeta_d(al, be);
                                lp_now += (al - 1.0) * log(x) + (be - 1.0) * log1p(-x) - lB;
                                lp_old += (double)rec[logp_col + c];
                                entr += lB - (al - 1.0) * digamma_d(al) - (be - 1.0) * digamma_d(be) + (al + be - 2.0) * digamma_d(al + be);
                            }
                            const float ratio = expf((float)(lp_now - lp_old));
                            const float Ar = adv[idx[r0 + r]];
                            const float s1 = ratio * Ar;
                            const float s2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * Ar;
                            lossp += -fminf(s1, s2) - a.ent_coef * (float)entr;
                            const double coef = (double)((s1 <= s2 ? Ar : 0.f) * (-invm) * ratio), ce = (double)(a.ent_coef * invm);
                            for (int c = 0; c < A; ++c) {
                                const float za = S.outb[r * S.op + c], zb = S.outb[r * S.op + A + c];
                                const double al = (double)(za > 20.f ? za : log1pf(expf(za))) + 1.0;
                                const double be = (double)(zb > 20.f ? zb : log1pf(expf(zb))) + 1.0;
                                const double x = (double)rec[R.act_off[0] + c];
                                const double psi_ab = digamma_d(al + be), tri_ab = trigamma_d(al + be);
                                const double dlp_a = log(x) - digamma_d(al) + psi_ab, dlp_b = log1p(-x) - digamma_d(be) + psi_ab;
                                const double dH_a = -(al - 1.0) * trigamma_d(al) + (al + be - 2.0) * tri_ab;
                                const double dH_b = -(be - 1.0) * trigamma_d(be) + (al + be - 2.0) * tri_ab;
                                const double sa = za > 20.f ? 1.0 : 1.0 / (1.0 + exp(-(double)za));     // d softplus / dz
                                const double sb = zb > 20.f ? 1.0 : 1.0 / (1.0 + exp(-(double)zb));
                                S.abuf[r * S.ap + c] = (float)((coef * dlp_a - ce * dH_a) * sa);
                                S.abuf[r * S.ap + A + c] = (float)((coef * dlp_b - ce * dH_b) * sb);
                            }
                            for (int c = 2 * A; c < napad; ++c) S.abuf[r * S.ap + c] = 0.f;
                        } else {
                            for (int c = 0; c < napad; ++c) S.abuf[r * S.ap + c] = 0.f;
                        }
                        for (int c = 0; c < napad; ++c) S.outb[r * S.op + c] = S.abuf[r * S.ap + c];
                    }
                    __syncthreads();
                } else if (discrete) {
                    // Categorical(probs = softmax(l3)) (:333-336): one thread per row does the softmax,
                    // log-prob of the stored action, entropy, ratio and the logits' delta
                    if (threadIdx.x < rc) {
                        const int r = threadIdx.x;
                        if (r < nv) {
                            g_cf rec = ring + (size_t)idx[r0 + r] * R.stride;
                            float mx = S.outb[r * S.op];
                            for (int c = 1; c < A; ++c) mx = fmaxf(mx, S.outb[r * S.op + c]);
                            float sum = 0.f;
                            for (int c = 0; c < A; ++c) sum += expf(S.outb[r * S.op + c] - mx);
                            float psum = 0.f;
                            for (int c = 0; c < A; ++c) psum += expf(S.outb[r * S.op + c] - mx) / sum;
                            const int ar = (int)rec[R.act_off[0]];
                            const float lse = logf(sum);
                            // log-prob of class c: clamped renormalised probability (probs=) or the log-softmax itself (logits=)
                            auto logp_of = [&](int c, float pc) {
                                return D.cat_logits ? (S.outb[r * S.op + c] - mx) - lse
                                                    : logf(fminf(fmaxf(pc / psum, 1.1920929e-07f), 1.f - 1.1920929e-07f));
                            };
                            float entr = 0.f, lp_now = 0.f;
                            for (int c = 0; c < A; ++c) {
                                const float pc = expf(S.outb[r * S.op + c] - mx) / sum;
                                const float lg = logp_of(c, pc);
                                entr -= lg * pc;
                                if (c == ar) lp_now = lg;
                            }
                            const float ratio = expf(lp_now - rec[logp_col]);
                            const float Ar = adv[idx[r0 + r]];
                            const float s1 = ratio * Ar;
                            const float s2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * Ar;
                            lossp += -fminf(s1, s2) - a.ent_coef * entr;
                            const float coef = (s1 <= s2 ? Ar : 0.f) * (-invm) * ratio;
                            for (int c = 0; c < napad; ++c) {
                                float d = 0.f;
                                if (c < A) {
                                    const float pc = expf(S.outb[r * S.op + c] - mx) / sum;
                                    const float lg = logp_of(c, pc);
                                    d = coef * ((c == ar ? 1.f : 0.f) - pc) + (a.ent_coef * invm) * pc * (lg + entr);
                                }
                                S.abuf[r * S.ap + c] = d;      // staged: outb is still being read by this row
                            }
                        } else {
                            for (int c = 0; c < napad; ++c) S.abuf[r * S.ap + c] = 0.f;
                        }
                        for (int c = 0; c < napad; ++c) S.outb[r * S.op + c] = S.abuf[r * S.ap + c];
                    }
                    __syncthreads();
                } else {
                // per-row ratio and d loss / d sum(logp)
                if (threadIdx.x < rc) {
                    const int r = threadIdx.x;
                    float coef = 0.f;
                    if (r < nv) {
                        g_cf rec = ring + (size_t)idx[r0 + r] * R.stride;
                        float lp_now = 0.f, lp_old = 0.f;
                        for (int c = 0; c < A; ++c) {
                            const float mean = S.outb[r * S.op + c];
                            const float ls = fminf(fmaxf(thA[NA.extra_off + c], -20.f), 2.f);
                            const float sd = expf(ls);
                            const float d = rec[R.act_off[0] + c] - mean;
                            lp_now += -(d * d) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                            lp_old += rec[logp_col + c];
                        }
                        const float ratio = expf(lp_now - lp_old);
                        const float Ar = adv[idx[r0 + r]];
                        const float s1 = ratio * Ar;
                        const float s2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * Ar;
                        lossp += -fminf(s1, s2);
                        coef = (s1 <= s2 ? Ar : 0.f) * (-invm) * ratio;   // d loss / d sum_c logp_now
                    }
                    S.dabuf[r * S.ap] = coef;
                }
                __syncthreads();
                for (int e = threadIdx.x; e < rc * napad; e += kWG) {
                    const int r = e / napad, c = e - r * napad;
                    float d = 0.f, dl = 0.f;
                    if (r < nv && c < A) {
                        const float coef = S.dabuf[r * S.ap];
                        const float mean = S.outb[r * S.op + c];
                        const float ls = fminf(fmaxf(thA[NA.extra_off + c], -20.f), 2.f);
                        const float var = expf(2.f * ls);
                        const float dm = ring[(size_t)idx[r0 + r] * R.stride + R.act_off[0] + c] - mean;
                        d = coef * dm / var * (1.f - mean * mean);        // through mean = tanh(z)
                        dl = coef * (dm * dm / var - 1.f);
                    }
                    S.outb[r * S.op + c] = d;
                    if (c < A) S.abuf[r * S.ap + c] = dl;
                }
                __syncthreads();
                if (threadIdx.x < A)
                    for (int r = 0; r < rc; ++r) gls += S.abuf[r * S.ap + threadIdx.x];
                }
                // one call site for the three policy distributions (every inlined copy of the backward pass is ~20 k
                // instructions of a kernel that walks its whole code once per minibatch step)
                PPO_T(2);
                mlp_bwd(NA, 0, NA.n_layers, thA, gA, S, r0 == 0 ? GS_STORE : GS_ADD, false, 0, 0);
                PPO_T(3);
            }
            if (!discrete && !beta && threadIdx.x < A) {
                const float raw = thA[NA.extra_off + threadIdx.x];
                gA[NA.extra_off + threadIdx.x] = (raw >= -20.f && raw <= 2.f) ? (gls - a.ent_coef) : 0.f;
            }
            const float aloss = block_sum(lossp, S.red) * invm - ((discrete || beta) ? 0.f : a.ent_coef * ent_sum);
            __syncthreads();
            ++tA;
            if (a.optimizer == 1)
                cadamw_net(NA, thA, as_global(D.m + offA), as_global(D.v + offA), gA, a.actor_lr, a.adam_eps, a.beta1, a.beta2, a.clip_norm, tA, S.red);
            else
                adam_net(NA.size, thA, as_global(D.m + offA), as_global(D.v + offA), gA, nullptr, a.actor_lr, a.adam_eps, a.beta1, a.beta2, 0.f,
                         a.clip_norm, tA, 0.f, S.red);
            __syncthreads();
            PPO_T(4);
            if (threadIdx.x == 0) trace[2 * j_tr] = aloss;
            }
            if (!do_critic) continue;
            // ---------------- critic: mse(v_target[idx], V(obs[idx])) (:349-351)
            float closs_p = 0.f;
            for (int r0 = 0; r0 < m; r0 += rc) {
                const int nv = min(rc, m - r0);
                gather_cols(S.xin, S.xp, rc, nv, idx + r0, ring, R.stride, R.obs_off[0], O, 0);
                zero_cols(S.xin, S.xp, rc, O, NC.L[0].k_pad);
                if (bn) { __syncthreads(); normalize_cols(S.xin, S.xp, nv, 0, O, bn, O); }
                __syncthreads();
                PPO_T(0);
                mlp_fwd(NC, 0, NC.n_layers, thC, S, ACT_NONE);
                PPO_T(1);
                for (int e = threadIdx.x; e < rc * ncpad; e += kWG) {
                    const int r = e / ncpad, c = e - r * ncpad;
                    float d = 0.f;
                    if (c == 0 && r < nv) {
                        const float diff = S.outb[r * S.op] - vt[idx[r0 + r]];
                        d = 2.f * diff * invm;
                        closs_p += diff * diff;
                    }
                    S.outb[r * S.op + c] = d;
                }
                __syncthreads();
                PPO_T(2);
                mlp_bwd(NC, 0, NC.n_layers, thC, gC, S, r0 == 0 ? GS_STORE : GS_ADD, false, 0, 0);
                PPO_T(3);
            }
            const float closs = block_sum(closs_p, S.red) * invm;
            __syncthreads();
            ++tC;
            if (a.optimizer == 1)      // one AdamW over both nets: the actor's learning rate (PPO.py:121)
                cadamw_net(NC, thC, as_global(D.m + offC), as_global(D.v + offC), gC, a.actor_lr, a.adam_eps, a.beta1, a.beta2, a.clip_norm, tC, S.red);
            else
                adam_net(NC.size, thC, as_global(D.m + offC), as_global(D.v + offC), gC, nullptr, a.critic_lr, a.adam_eps, a.beta1, a.beta2, 0.f,
                         a.clip_norm, tC, 0.f, S.red);
            __syncthreads();
            PPO_T(4);
            if (threadIdx.x == 0) trace[2 * j_tr + 1] = closs;
        }
    }
    if (threadIdx.x == 0) {
        float* st = D.stats + (size_t)p * D.n_agents * ST_COUNT;
        const int j = a.k_epochs * n_mb - 1;
        if (do_actor) { steps[0] = tA; st[ST_ACTOR_LOSS] = trace[2 * j]; }
        if (do_critic) { steps[1] = tC; st[ST_CRITIC_LOSS] = trace[2 * j + 1]; }
    }
    PPO_TDUMP();
}

}  // namespace frl
