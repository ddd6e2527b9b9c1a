// Actor stage of DDPG / TD3 / SAC / MADDPG for one (learner, agent) per workgroup on the K-sliced chained design
// (device/chain_wide.hpp; the counterpart of kernels_criticw.hip): a_i = actor_i(s_i) (SAC: the reparameterised tanh-Gaussian sample
// and its log-prob); Q(s, a) through the (already updated, frozen) critic with agent i's stored action replaced by a_i — Q1 for
// DDPG / TD3 / MADDPG, the mean of the twins for SAC; dQ/da_i; actor backward; clip, Adam and (single agent) the soft update of the
// actor's target; SAC's alpha step — DDPG_simple.py:151-154, TD3.py:224-233, SAC.py:244-260, MADDPG_simple.py:182-186.
//
// Three passes, the per-row results of one pass handed to the next through the unit's HBM scratch (L2-resident):
//   A  actor forward: first-layer sweep + layers 2, 3   -> the joint action row [ring actions | a_i], the hidden activations h1, h2
//   B  critic forward on [s | a] + the dX chain down to the action k-blocks of W1 (staged into the union behind the sweep)
//                                                        -> dQ/da_i, sum of Q for the loss   (SAC: both heads, added)
//   C  actor backward from the stored h1, h2 (with a 393-column first layer, re-reading 2 x 128 floats per row is far cheaper
//      than a second sweep), first-layer deltas -> scratch, then the dW1 pass, clip + Adam streamed over the net
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain_wide.hpp"
#include "device/wide_timing.hpp"

namespace frl {

template <int NT3A>
__device__ __forceinline__ void ac_actor_wide_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    const int nag = D.n_agents;
    const int unit = blockIdx.x, p = a.p0 + unit / nag, ag = unit % nag;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[2 * ag];
    const NetDesc& NC = D.net[2 * ag + 1];
    WideNet W;
    W.init(smem);
    const ChainNet& C = W.C;
    const ChainLds& S = C.S;
    const int tid = C.tid, l = C.l, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, OT = R.obs_total, AT = R.act_total, XT = OT + AT, am = D.act_max;
    const int Oi = R.obs_dim[ag], Ai = R.act_dim[ag], aoff = R.act_off[ag] - R.act_off[0];
    const bool sac = (D.algo == ALGO_SAC);
    const size_t lbase = (size_t)p * D.learner_stride;
    const int noffA = D.net_off[2 * ag];
    g_f thA = as_global(D.theta + lbase + noffA);
    g_f tgA = as_global(D.target + lbase + noffA);
    g_f mA = as_global(D.m + lbase + noffA);
    g_f vA = as_global(D.v + lbase + noffA);
    g_f grA = as_global(D.grad + lbase + noffA);
    g_cf thC = as_global(D.theta + lbase + D.net_off[2 * ag + 1]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_ci idx = as_global_i(D.idx + ((size_t)p * nag + ag) * D.batch_max);
    g_cf noise1 = as_global(D.noise + (((size_t)p * nag + ag) * D.noise_sets + 1) * D.batch_max * am);     // the actor stage's eps (set 1)
    WideScratch X;
    X.init(as_global(D.wide_scr + ((size_t)p * nag + ag) * D.wide_unit), D.wide_bm, D.wide_xp, D.wide_op, nag);
    const float invB = 1.f / (float)B;
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const int nq = sac ? NC.heads : 1;                                 // SAC.py:250: mean of the twins; TD3.py:227 / MADDPG: Q1 only
    const float dqv = sac ? -0.5f * invB : -invB;
    const int nsc = (B + 255) / 256, nchunks = (B + 63) / 64;
    const int KB1a = NA.L[0].k_pad >> 4, KB1c = NC.L[0].k_pad >> 4;
    auto row_of = [&](int sc, int t) { return 256 * sc + 64 * t + 16 * w + i16; };
    auto rec_of = [&](int row) { return ring + (size_t)idx[row < B ? row : B - 1] * R.stride; };

    // =========================================================== A: a_i = tanh(actor_i(s_i)) (SAC: tanh(mean + std eps), sum of log pi)
    float lpsum = 0.f;
    WIDE_T0();
    // xrow = the critic's input row [s_all | a_all], a record's first XT columns; pass A overwrites agent i's action columns with
    // a_i (MADDPG_simple.py:183: only agent i's action is recomputed).  Agent i's observation is read in place when it starts on
    // a 16-byte boundary of the record, from a copy otherwise.
    const FRL_LDS int* tab0 = W.stage_idx(idx, B);                     // (the union is free until the first sweep)
    W.copy_cols(X.xrow, X.xp, ring, R.stride, tab0, B, R.obs_off[0], XT);
    const bool direct = ((R.obs_off[ag] - R.obs_off[0]) & 3) == 0;
    if (!direct) W.copy_cols(X.xobs, X.op, ring, R.stride, tab0, B, R.obs_off[ag], Oi);
    __syncthreads();
    auto obs_of = [&](int row) {
        const int rc = row < B ? row : B - 1;
        return direct ? ring + (size_t)idx[rc] * R.stride + R.obs_off[ag] : (g_cf)X.xobs + (size_t)rc * X.op;
    };
    W.stage23((g_cf)thA, NA.L, NT3A, NA.extra_off, NA.extra_n);
    WIDE_T(0);
    for (int sc = 0; sc < nsc; ++sc) {
        g_cf rp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) rp[t] = obs_of(row_of(sc, t));
        f32x4 h1[4][kHT];
        W.l1_sweep<4>(h1, rp, (g_cf)thA + NA.L[0].w_off, KB1a);
        WIDE_T(1);
        // (only the chunks that exist: the scratch holds ceil(batch_max / 64) of them)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (64 * (4 * sc + t) < B) {
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot) st4(X.ah1 + ((size_t)(((4 * sc + t) * 4 + w) * kHT + ot) * 256 + 4 * l), h1[t][ot]);
            }
        static_for<0, 2>([&](auto hc) {
            constexpr int half = decltype(hc)::value;
            f32x4 h2[2][kHT], z[2][NT3A];
            W.l23<2, NT3A, false, 4, 2 * half>(h1, h2, z, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tt = 2 * half + t, row = row_of(sc, tt);
                if (64 * (4 * sc + tt) < B) {
#pragma unroll
                    for (int ot = 0; ot < kHT; ++ot) st4(X.ah2 + ((size_t)(((4 * sc + tt) * 4 + w) * kHT + ot) * 256 + 4 * l), h2[t][ot]);
                }
                if (row < B) {
#pragma unroll
                    for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * o3 + 4 * q + r;
                            if (c < Ai) {
                                const float zr = z[t][o3][r];
                                float av;
                                if (sac) {                             // SAC.py:70-97
                                    const float ls = fminf(fmaxf(S.ls[c], -20.f), 2.f), sd = expf(ls);
                                    const float u = zr + sd * noise1[(size_t)row * am + c], du = u - zr;
                                    lpsum += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                                    lpsum -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                                    av = tanhf(u);
                                } else {
                                    av = tanhf(zr);
                                }
                                X.xrow[(size_t)row * X.xp + OT + aoff + c] = av;
                            }
                        }
                }
            }
        });
        WIDE_T(2);
    }
    __syncthreads();                                                   // a_i in xrow is read by every lane group of a row below

    // =========================================================== B: Q(s, a) and dQ/da_i through the frozen critic
    float qsum = 0.f;
    const int kbA0 = (OT + aoff) >> 4, kbA1 = (OT + aoff + Ai - 1) >> 4, nA = kbA1 - kbA0 + 1;     // the k-blocks that hold agent i's action columns (<= 3)
    for (int hd = 0; hd < nq; ++hd) {
        const LayerDesc* L = NC.L + 3 * hd;
        g_cf w1 = thC + L[0].w_off;
        W.stage23(thC, L, 1, -1, 0);
        WIDE_T(0);
        // two tiles per sweep, one tile at a time down the dX chain: h1 of four tiles next to h2, both deltas and the transposed
        // fragments of a chain does not fit the register file (a four-tile build spilled inside the chains: 4.4x their MFMA time)
        for (int sc2 = 0; sc2 < 2 * nsc; ++sc2) {
            if (128 * sc2 >= B) break;
            g_cf rp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 128 * sc2 + 64 * t + 16 * w + i16;
                rp[t] = (g_cf)X.xrow + (size_t)(row < B ? row : B - 1) * X.xp;
            }
            f32x4 h1[2][kHT];
            W.l1_sweep<2>(h1, rp, w1, KB1c);
            WIDE_T(3);
            // W1's action k-blocks -> the union, tile (ot, j) at (ot * 3 + j) * 256 (the sweep is done with its slices)
            lds_barrier();
            for (int T = w; T < kHT * 3; T += 4) {
                const int ot = T / 3, j = T - 3 * ot;
                if (j < nA) st4(W.u + T * 256 + 4 * l, ld4(w1 + ((size_t)(ot * KB1c + kbA0 + j) * 256 + 4 * l)));
            }
            lds_barrier();
            WIDE_T(4);
            // both tiles down the chain together: every weight fragment (forward and transposed) feeds two MFMAs
            f32x4 h2[2][kHT], z[2][1], d2[2][kHT], d1[2][kHT];
            W.l23<2, 1, true, 2, 0>(h1, h2, z, 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 128 * sc2 + 64 * t + 16 * w + i16;
                f32x4 dz = {0.f, 0.f, 0.f, 0.f};
                if (q == 0 && row < B) { qsum += z[t][0][0]; dz[0] = dqv; }       // actor_loss = -Q(s, actor(s)).mean() [+ alpha log pi]
                C.delta2_valu(dz, h2[t], d2[t], 1);
            }
            W.delta1_t<2, 2, 0>(d2, h1, d1);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < nA) {                                          // dX of k-block kbA0 + j = W1^T d1 (transposed fragment reads)
                    f32x4 dx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ob = 0; ob < kHT; ++ob) {
                        f32x4 wa;
#pragma unroll
                        for (int e = 0; e < 4; ++e) wa[e] = W.u[(ob * 3 + j) * 256 + C.tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
#pragma unroll
                        for (int t = 0; t < 2; ++t) dx[t] = mfma4(dx[t], wa, d1[t][ob]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = 128 * sc2 + 64 * t + 16 * w + i16;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * (kbA0 + j) + 4 * q + r - OT - aoff;          // agent i's action component
                            if (row < B && c >= 0 && c < Ai) {
                                g_f dst = X.dqa + (size_t)row * kWideApitch + c;
                                *dst = hd == 0 ? dx[t][r] : *dst + dx[t][r];
                            }
                        }
                    }
                }
            }
            WIDE_T(5);
        }
    }
    __syncthreads();

    // =========================================================== C: backward from the stored activations
    WideGrad<NT3A> g;
    W.grad_zero(g);
    W.stage23((g_cf)thA, NA.L, NT3A, NA.extra_off, NA.extra_n);
    WIDE_T(0);
    float gls[NT3A][4];                                                // d loss / d log_std of this lane's action components, its rows
#pragma unroll
    for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
        for (int r = 0; r < 4; ++r) gls[o3][r] = 0.f;
    for (int cg = 0; cg < nchunks; ++cg) {
        const int row = 64 * cg + 16 * w + i16;
        const bool valid = row < B;
        f32x4 h1[1][kHT], h2[1][kHT], z[1][NT3A], dz[NT3A];
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            h1[0][ot] = ld4((g_cf)(X.ah1 + ((size_t)((cg * 4 + w) * kHT + ot) * 256 + 4 * l)));
            h2[0][ot] = ld4((g_cf)(X.ah2 + ((size_t)((cg * 4 + w) * kHT + ot) * 256 + 4 * l)));
        }
        W.head_tiles<1, NT3A>(h2, z);
        WIDE_T(6);
#pragma unroll
        for (int o3 = 0; o3 < NT3A; ++o3) {
            dz[o3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * o3 + 4 * q + r;
                if (valid && c < Ai) {
                    const float dq = X.dqa[(size_t)row * kWideApitch + c];
                    if (sac) {                                         // through a = tanh(u), u = mean + exp(log_std) eps, and alpha log pi
                        const float av = X.xrow[(size_t)row * X.xp + OT + aoff + c];
                        const float d = dq * (1.f - av * av) + (alpha * invB) * (2.f * av);
                        const float ls = fminf(fmaxf(S.ls[c], -20.f), 2.f);
                        dz[o3][r] = d;
                        gls[o3][r] += d * expf(ls) * noise1[(size_t)row * am + c] - alpha * invB;
                    } else {
                        const float av = tanhf(z[0][o3][r]);
                        dz[o3][r] = dq * (1.f - av * av);
                    }
                }
            }
        }
        W.backward<NT3A, false>(g, h1[0], h2[0], dz, 0, X.dz1 + (size_t)cg * 8192);
        WIDE_T(7);
    }
    W.grad_finish(g);
    float ss = W.grad_store_23<NT3A>(grA, NA.L, g);
    __syncthreads();
    {
        const FRL_LDS int* tab = W.stage_idx(idx, B);
        ss += W.dw1_grad_any(grA, NA.L, (g_cf)X.dz1, nchunks, B, KB1a, Oi, [&](int row) { return ring + (size_t)tab[row] * R.stride + R.obs_off[ag]; });
    }
    WIDE_T(8);

    // =========================================================== clip_grad_norm_, Adam, soft update of the actor's target; SAC: alpha
    // log_std: sum over this wave's rows (lanes of one lane group), then over the waves through LDS
#pragma unroll
    for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = gls[o3][r];
            v += lane_xor<1>(v); v += lane_xor<2>(v); v += lane_xor<4>(v); v += lane_xor<8>(v);
            gls[o3][r] = v;
        }
    lds_f lsred = W.u;                                                 // [4 waves][32 components]
    lds_barrier();
    if (i16 == 0) {
#pragma unroll
        for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
            for (int r = 0; r < 4; ++r) lsred[w * 32 + 16 * o3 + 4 * q + r] = gls[o3][r];
    }
    lds_barrier();
    float ss_extra = 0.f;
    if (sac && tid < Ai) {                                             // outside the clamp [-20, 2] the gradient is zero (SAC.py:77)
        const float raw = S.ls[tid];
        const float gr = (raw >= -20.f && raw <= 2.f) ? ((lsred[tid] + lsred[32 + tid]) + lsred[64 + tid]) + lsred[96 + tid] : 0.f;
        grA[NA.extra_off + tid] = gr;
        ss_extra = gr * gr;
    }
    ss = wave_sum(ss + ss_extra);
    const float qs = wave_sum(qsum), lps = wave_sum(lpsum);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { S.red[w] = ss; S.red[8 + w] = qs; S.red[12 + w] = lps; }
    if (tid == 0) S.red[32] = __int_as_float(steps[2 * ag]);
    __syncthreads();                                                   // (also: every gradient store of this workgroup has landed)
    const float total = sqrtf(((S.red[0] + S.red[1]) + S.red[2]) + S.red[3]);
    const float qtot = ((S.red[8] + S.red[9]) + S.red[10]) + S.red[11];
    const float lptot = ((S.red[12] + S.red[13]) + S.red[14]) + S.red[15];
    const int tstep = __float_as_int(S.red[32]) + 1;
    const double bc1 = 1.0 - powi_d((double)a.beta1, tstep), bc2 = 1.0 - powi_d((double)a.beta2, tstep);
    AdamCoef co;
    co.coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    co.step = (float)((double)a.actor_lr / bc1); co.inv_bc2s = 1.f / (float)sqrt(bc2);
    co.w1 = 1.f - a.beta1; co.w2 = 1.f - a.beta2; co.beta2 = a.beta2; co.eps = a.adam_eps; co.wd = 0.f;
    co.tk = 1.f - a.tau; co.tau = a.tau;
    WIDE_T(9);
    if (nag == 1) W.adam_stream<true>(thA, mA, vA, tgA, (g_cf)grA, NA.size >> 2, co);
    else W.adam_stream<false>(thA, mA, vA, tgA, (g_cf)grA, NA.size >> 2, co);
    WIDE_T(10);
    WIDE_TDUMP(1);
    if (tid == 0) {
        steps[2 * ag] = tstep;
        float* st = D.stats + ((size_t)p * nag + ag) * ST_COUNT;
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

#define FRL_ACTOR_WIDE(NAME, NT3A)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(const EngineDesc* __restrict__ Dp, LearnArgs a) {                          \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                       \
        ac_actor_wide_body<NT3A>(*Dp, a, smem);                                                                            \
    }
FRL_ACTOR_WIDE(ac_actor_wide_a1_kernel, 1)
FRL_ACTOR_WIDE(ac_actor_wide_a2_kernel, 2)

}  // namespace frl
