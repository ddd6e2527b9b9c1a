// Actor stage of DDPG / TD3 / SAC for one learner per workgroup, register-chained (device/chain_net.hpp; the counterpart of
// kernels_critic2.hip): a = actor(s) (SAC: the reparameterised tanh-Gaussian sample and its log-prob); Q(s, a) through the
// (already updated, frozen) critic — Q1 for DDPG / TD3, the mean of the twins for SAC; dQ/da; actor backward; clip, Adam and
// the soft update of the actor's target; SAC's alpha step — DDPG_simple.py:151-154, TD3.py:224-233, SAC.py:244-260 — in ONE
// launch, no gradient slabs.
//
// Three passes over the learner's batch, the net a pass needs staged once into the LDS images:
//   A  actor forward (two row tiles per wave)                       -> a[row] in LDS
//   B  critic forward on [s | a] + the dX-only backward chain       -> dQ/da[row] in LDS, sum of Q for the loss
//      (SAC: once per twin head, each staged in turn, the two dQ/da added)
//   C  actor forward AGAIN (its activations are registers of pass A, long gone) + backward with the weight-gradient exchanges
// Recomputing the forward costs 320 of the pass's 928 MFMAs per wave and chunk; keeping both nets' images resident instead
// would leave no LDS for the exchange buffers.  Shape: as kernels_critic2.hip.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain_net.hpp"
#include "device/update_common.hpp"
#include "device/ppo_timing.hpp"

namespace frl {

// NW = waves per workgroup (device/chain_net.hpp; 8 since round 6, 4 = round 2-5's kernel for A/B runs), TA = 16-row tiles per wave
// in passes A and B (nothing is accumulated there)
template <int NW, int TA>
__device__ __forceinline__ void ac_actor_v2_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    constexpr int kRC = 16 * NW, kRowsT = kRC * TA;                    // rows per chunk of pass C / of passes A and B
    constexpr int kNC = kChainBatch / kRC, kNCT = kChainBatch / kRowsT;
    static_assert(kNCT >= 1, "a chunk of passes A / B is at most the whole batch");
    const int p = a.p0 + blockIdx.x;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[0];
    const NetDesc& NC = D.net[1];
    using Net = ChainNetT<NW>;
    Net C;
    C.init(smem);
    const ChainLds& S = C.S;
    const int tid = C.tid, l = C.l, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, O = R.obs_dim[0], A = R.act_dim[0];
    const size_t lbase = (size_t)p * D.learner_stride;
    g_f thA = as_global(D.theta + lbase + D.net_off[0]);
    g_f tgA = as_global(D.target + lbase + D.net_off[0]);
    g_f mA = as_global(D.m + lbase + D.net_off[0]);
    g_f vA = as_global(D.v + lbase + D.net_off[0]);
    g_cf thC = as_global(D.theta + lbase + D.net_off[1]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_ci idx = as_global_i(D.idx + (size_t)p * D.batch_max);
    const float invB = 1.f / (float)B;
    const bool sac = (D.algo == ALGO_SAC);
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const int am = D.act_max;
    g_cf noise1 = as_global(D.noise + ((size_t)p * D.noise_sets + 1) * D.batch_max * am);     // the actor stage's eps (set 1)
    const int nq = sac ? NC.heads : 1;                                 // SAC.py:250: mean of the twins; TD3.py:227: Q1 only
    const float dqv = sac ? -0.5f * invB : -invB;
    const int nchunks = (B + kRC - 1) / kRC, nch2 = (B + kRowsT - 1) / kRowsT;
    lds_f dab = S.eb;                                                  // dQ/da[row][4] at the start of eb: pass B has no exchanges

    // observation columns of this lane's rows: pass A / B in the TA-tile mapping (kRowsT c2 + 16 TA w + 16 t + i16), pass C in the
    // one-tile mapping (kRC c + 16 w + i16); ring addresses once, fields one chunk ahead of their use
    int ridxT[kNC], ridx[kNC];
#pragma unroll
    for (int j4 = 0; j4 < kNC; ++j4) {
        const int rowT = (j4 / TA) * kRowsT + 16 * TA * w + (j4 % TA) * 16 + i16, row = j4 * kRC + 16 * w + i16;
        ridxT[j4] = rowT < B ? idx[rowT] : -1;
        ridx[j4] = row < B ? idx[row] : -1;
    }
    struct RowIn2 { f32x4 x[TA]; };
    auto load_obs2 = [&](int c2) {
        RowIn2 X;
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            X.x[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            int ri = ridxT[t];
#pragma unroll
            for (int cc = 1; cc < kNCT; ++cc) ri = c2 == cc ? ridxT[TA * cc + t] : ri;
            if (ri >= 0) {
                g_cf rec = ring + (size_t)ri * R.stride;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < O) X.x[t][e] = rec[R.obs_off[0] + 4 * q + e];
            }
        }
        return X;
    };
    auto load_obs = [&](int c) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        const int ri = pick(ridx, c);
        if (ri >= 0) {
            g_cf rec = ring + (size_t)ri * R.stride;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < O) x[e] = rec[R.obs_off[0] + 4 * q + e];
        }
        return x;
    };

    // =========================================================== A: a = tanh(actor(s)) -> S.ab  (SAC: a = tanh(mean + std eps), sum of log pi)
    float lpsum = 0.f;
    PPO_T0();
    RowIn2 nxt2 = load_obs2(0);
    // every net's image is fetched (global -> registers) in front of the last pass over the previous net: only the first staging
    // waits for HBM in the open
    typename Net::StageRegs pend = C.stage_fetch((g_cf)thA, 0, NA.extra_n);
    C.stage_commit(pend);
    PPO_T(0);
    for (int c2 = 0; c2 < nch2; ++c2) {
        const RowIn2 cur = nxt2;
        nxt2 = load_obs2(c2 + 1 < nch2 ? c2 + 1 : 0);                  // (after the last chunk: chunk 0 again, for pass B)
        if (c2 + 1 == nch2) pend = C.stage_fetch(thC, 0);
        f32x4 z[TA], h1[TA][kHT], h2[TA][kHT];
        C.template forward_vh<TA>(cur.x, h1, h2, z, A);
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const int row = c2 * kRowsT + 16 * TA * w + 16 * t + i16;
            if (q == 0 && row < kChainBatch) {
                f32x4 an = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r < A && row < B) {
                        if (sac) {                                     // SAC.py:70-97
                            const float ls = fminf(fmaxf(S.ls[r], -20.f), 2.f), sd = expf(ls);
                            const float u = z[t][r] + sd * noise1[(size_t)row * am + r], du = u - z[t][r];
                            lpsum += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                            lpsum -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                            an[r] = tanhf(u);
                        } else {
                            an[r] = tanhf(z[t][r]);
                        }
                    }
                }
                st4(S.ab + row * 4, an);
            }
        }
    }
    // =========================================================== B: Q(s, a) and dQ/da through the frozen critic (TD3.py:227: Q1 only; SAC.py:250: both heads)
    float qsum = 0.f;
    f32x4 nxt;
    PPO_T(1);
    for (int hd = 0; hd < nq; ++hd) {
        C.stage_commit(pend);
        PPO_T(0);
        for (int c2 = 0; c2 < nch2; ++c2) {
            const RowIn2 cur = nxt2;
            if (c2 + 1 < nch2) nxt2 = load_obs2(c2 + 1);
            else if (hd + 1 < nq) nxt2 = load_obs2(0);                 // the next head starts over
            else nxt = load_obs(0);                                    // first chunk of pass C
            f32x4 xb[TA], z[TA], h1[TA][kHT], h2[TA][kHT];
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const int row = c2 * kRowsT + 16 * TA * w + 16 * t + i16;
                xb[t] = cur.x[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * q + e;
                    if (row < B && f >= O && f < O + A) xb[t][e] = S.ab[row * 4 + f - O];
                }
            }
            C.template forward_vh<TA>(xb, h1, h2, z, 1);
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const int row = c2 * kRowsT + 16 * TA * w + 16 * t + i16;
                const bool valid = row < B;
                f32x4 dz = {0.f, 0.f, 0.f, 0.f};
                if (q == 0 && valid) { qsum += z[t][0]; dz[0] = dqv; } // actor_loss = -Q(s, actor(s)).mean() [+ alpha log pi]
                f32x4 d2[kHT], d1[kHT];
                C.delta2_valu(dz, h2[t], d2, 1);
                C.delta1(d2, h1[t], d1);
                const f32x4 dx = C.delta0(d1);                         // d loss / d [s | a] column 4q + r of this row
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 4 * q + r;
                    if (valid && f >= O && f < O + A) dab[row * 4 + f - O] = hd == 0 ? dx[r] : dab[row * 4 + f - O] + dx[r];
                }
            }
        }
        // the next image is fetched AFTER this pass, in the open: held across the two-tile forward + dX chain its 84 registers
        // put the four-wave kernel at 512 VGPRs with 54 spilled (fetch-ahead here: 0.355 ms per launch; this way 470 VGPRs, no scratch, 0.347)
        pend = hd + 1 < nq ? C.stage_fetch(thC, hd + 1) : C.stage_fetch((g_cf)thA, 0, NA.extra_n);
        PPO_T(2);
    }
    // =========================================================== C: actor forward again, delta through tanh, backward into the accumulators
    typename Net::Grad g;
    PPO_T(2);
    C.grad_zero(g);
    C.stage_commit(pend);                                              // (its leading barrier also publishes dab)
    PPO_T(0);
    // dab lives in eb, which the backward's exchanges overwrite (and at eight waves a[row] in ab as well: ex): this lane's four
    // values per chunk into registers first
    f32x4 dqa[kNC], epsa[kNC], ava[NW == 8 ? kNC : 1];                 // (SAC: the rows' eps as well, ahead of the loop)
#pragma unroll
    for (int c = 0; c < kNC; ++c) {
        const int row = c * kRC + 16 * w + i16;
        dqa[c] = (q == 0 && row < B) ? ld4((lds_cf)(dab + row * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NW == 8) ava[c] = (q == 0 && row < B) ? ld4((lds_cf)(S.ab + row * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        epsa[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sac && q == 0 && row < B) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < A) epsa[c][r] = noise1[(size_t)row * am + r];
        }
    }
    float gls[4] = {0.f, 0.f, 0.f, 0.f};                               // d loss / d log_std, this lane's rows
    for (int c = 0; c < nchunks; ++c) {
        const int row = c * kRC + 16 * w + i16;
        f32x4 xb[1] = {nxt}, z[1], h1[1][kHT], h2[1][kHT];
        nxt = load_obs(c + 1 < nchunks ? c + 1 : 0);
        PPO_T(5);
        C.template forward_vh<1>(xb, h1, h2, z, A);
        PPO_T(3);
        const f32x4 dq = pick(dqa, c);
        f32x4 dz = {0.f, 0.f, 0.f, 0.f};
        if (q == 0 && row < B) {
            if (sac) {                                                 // through a = tanh(u), u = mean + exp(log_std) eps, and alpha log pi
                const f32x4 ep = pick(epsa, c);
                f32x4 av4;
                if constexpr (NW == 8) av4 = pick(ava, c); else av4 = ld4((lds_cf)(S.ab + row * 4));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r < A) {
                        const float av = av4[r];
                        const float d = dq[r] * (1.f - av * av) + (alpha * invB) * (2.f * av);
                        const float ls = fminf(fmaxf(S.ls[r], -20.f), 2.f);
                        dz[r] = d;
                        gls[r] += d * expf(ls) * ep[r] - alpha * invB;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < A) { const float av = tanhf(z[0][r]); dz[r] = dq[r] * (1.f - av * av); }
            }
        }
        C.backward(g, xb[0], h1[0], h2[0], dz, A);
        PPO_T(4);
    }
    C.grad_finish(g);
    // =========================================================== clip_grad_norm_, Adam, soft update of the actor's target
    float ss = wave_sum(C.grad_sumsq(g));
    const float qs = wave_sum(qsum), lps = wave_sum(lpsum);
#pragma unroll
    for (int r = 0; r < 4; ++r) gls[r] = wave_sum(gls[r]);
    lds_barrier();
    if (l == 0) {                                                      // red: [0, 8) norm, [8, 16) Q, [16, 24) log pi, [24 + 8 r, ...) log_std gradients, 56 the step count
        S.red[w] = ss; S.red[8 + w] = qs; S.red[16 + w] = lps;
#pragma unroll
        for (int r = 0; r < 4; ++r) S.red[24 + 8 * r + w] = gls[r];
    }
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (tid == 0) S.red[56] = __int_as_float(steps[0]);                // (thread 0 rewrites steps[0] after the update: the count travels with the partials)
    lds_barrier();
    auto red_sum = [&](int base) {
        float v = S.red[base];
#pragma unroll
        for (int i = 1; i < NW; ++i) v += S.red[base + i];
        return v;
    };
    // log_std gradient of component i16 (lanes i16 < A); outside the clamp [-20, 2] the gradient is zero (SAC.py:77)
    float g_extra = 0.f, ss_extra = 0.f;
    if (sac) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < A) {
                const float raw = S.ls[r];
                const float gr = (raw >= -20.f && raw <= 2.f) ? red_sum(24 + 8 * r) : 0.f;
                ss_extra += gr * gr;
                if (i16 == r) g_extra = gr;
            }
        }
    }
    const float total = sqrtf(red_sum(0) + ss_extra);
    const float qtot = red_sum(8);
    const float lptot = red_sum(16);
    const int t = __float_as_int(S.red[56]) + 1;
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    AdamCoef co;
    co.coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    co.step = (float)((double)a.actor_lr / bc1); co.inv_bc2s = 1.f / (float)sqrt(bc2);
    co.w1 = 1.f - a.beta1; co.w2 = 1.f - a.beta2; co.beta2 = a.beta2; co.eps = a.adam_eps; co.wd = 0.f;
    co.tk = 1.f - a.tau; co.tau = a.tau;
    PPO_T(5);
    C.template adam_head<true, 0, true>(g, thA, mA, vA, tgA, co, g_extra, sac ? NA.extra_n : 0);      // (theta from the actor's image, still staged from pass C)
    PPO_T(6);
    PPO_TDUMP();
    if (tid == 0) {
        steps[0] = t;
        float* st = D.stats + (size_t)p * ST_COUNT;
        st[ST_ACTOR_LOSS] = sac ? (-(qtot * 0.5f) + alpha * lptot) * invB : -qtot * invB;   // SAC.py:251: (alpha log pi - Q).mean()
        st[ST_ACTOR_GNORM] = total;
        if (sac) {                                                     // alpha step on the batch's entropy (SAC.py:154-169,257-260)
            float* al = D.alpha + p * 4;
            const float ent_mean = -lptot * invB;
            const float mean_term = ent_mean - a.target_entropy;
            const float gl = alpha * mean_term;                        // d alpha_loss / d log_alpha
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
}

#ifndef FRL_ACTOR8_TA
#define FRL_ACTOR8_TA 1
#endif
__global__ __launch_bounds__(512) void ac_actor_v2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ac_actor_v2_body<8, FRL_ACTOR8_TA>(*Dp, a, smem);
}
__global__ __launch_bounds__(256) void ac_actor_v2w4_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ac_actor_v2_body<4, 2>(*Dp, a, smem);
}

}  // namespace frl
