// Actor stage of DDPG / TD3 / SAC / MADDPG at HIDDEN 256 for one (learner, agent) per workgroup on device/chain_wide16.hpp (the
// counterpart of kernels_criticx.hip): kernels_actorw.hip's three passes with the activations of a 256-row super-chunk in
// registers from the first layer to the head and back (l1_x, sweep_x) — DDPG_simple.py:151-154, TD3.py:224-233, SAC.py:244-260,
// MADDPG_simple.py:182-186.
//   A  actor forward                                   -> a_i into the critic's input row; h1, h2 (tile-lane) and their ReLU masks -> scratch
//   B  critic forward on [s | a], deltas down to layer 1, dX of agent i's action columns (W1's action k-blocks in LDS) -> dQ/da_i;
//      nothing of the critic's leaves the registers
//   C  actor deltas from dQ/da_i down to layer 1        -> scratch; the weight-gradient passes; clip + Adam streamed over the net
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain_wide16.hpp"

namespace frl {

template <int NT3A>
__device__ __forceinline__ void ac_actor_x_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    const int nag = D.n_agents;
    const int unit = blockIdx.x, p = a.p0 + unit / nag, ag = unit % nag;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[2 * ag];
    const NetDesc& NC = D.net[2 * ag + 1];
    SweepNet N;
    N.init(smem);
    const WideNet& W = N.W;
    const ChainNet& C = W.C;
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
    g_cf noise1 = as_global(D.noise + (((size_t)p * nag + ag) * D.noise_sets + 1) * D.batch_max * am);
    Wide16Scratch X;
    X.init(as_global(D.wide_scr + ((size_t)p * nag + ag) * D.wide_unit), D.wide_bm, D.wide_xp, D.wide_op, nag);
    g_f bits = X.d1t + (size_t)256 * D.wide_bm;                       // the actor's ReLU masks, pass A -> pass C: [super-chunk][wave][lane][16]
    const float invB = 1.f / (float)B;
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const int nq = sac ? NC.heads : 1;
    const float dqv = sac ? -0.5f * invB : -invB;
    const int nsc = (B + 255) / 256, nchunks = (B + 63) / 64;
    const int KB1a = NA.L[0].k_pad >> 4, KB1c = NC.L[0].k_pad >> 4;
    auto row_of = [&](int sc, int t) { return 256 * sc + 64 * t + 16 * w + i16; };
    // =========================================================== A: a_i = tanh(actor_i(s_i)) (SAC: tanh(mean + std eps), sum of log pi)
    float lpsum = 0.f;
    const FRL_LDS int* tab0 = W.stage_idx(idx, B);
    W.copy_cols(X.xrow, X.xp, ring, R.stride, tab0, B, R.obs_off[0], XT);
    const bool direct = ((R.obs_off[ag] - R.obs_off[0]) & 3) == 0;
    if (!direct) W.copy_cols(X.xobs, X.op, ring, R.stride, tab0, B, R.obs_off[ag], Oi);
    __syncthreads();
    N.stage3((g_cf)thA, NA.L, NT3A, NA.extra_off, NA.extra_n);
    for (int sc = 0; sc < nsc; ++sc) {
        g_cf px[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = row_of(sc, t), rc = row < B ? row : B - 1;
            px[t] = (direct ? ring + (size_t)idx[rc] * R.stride + R.obs_off[ag] : (g_cf)X.xobs + (size_t)rc * X.op) + 4 * q;
        }
        f32x4 XR[2][4][8];
        unsigned m1[8], m2[8] = {};
        N.l1_x(XR, px, (g_cf)thA + NA.L[0].w_off, KB1a);
        const SweepNet::Slice0 f0 = N.sweep_fetch0<false>((g_cf)thA + NA.L[1].w_off);
        N.mask_bits(XR, m1);
        N.store_x(X.h1t, sc, XR);
        f32x4 z[4][NT3A];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o3 = 0; o3 < NT3A; ++o3) z[t][o3] = ld4((lds_cf)(N.b3 + 16 * o3 + 4 * q));
        N.sweep_x<false>(XR, (g_cf)thA + NA.L[1].w_off, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {
            N.mask_push(m2, N.relu_pair(acc));
            N.store_pair(X.h2t, sc, s, acc);
            N.head_tiles_pair<NT3A>(acc, s, z);
        }, f0);
        {
            g_f bp = bits + ((size_t)(sc * 4 + w) * 64 + l) * 16;
            st4(bp, f32x4{__uint_as_float(m1[0]), __uint_as_float(m1[1]), __uint_as_float(m1[2]), __uint_as_float(m1[3])});
            st4(bp + 4, f32x4{__uint_as_float(m1[4]), __uint_as_float(m1[5]), __uint_as_float(m1[6]), __uint_as_float(m1[7])});
            st4(bp + 8, f32x4{__uint_as_float(m2[0]), __uint_as_float(m2[1]), __uint_as_float(m2[2]), __uint_as_float(m2[3])});
            st4(bp + 12, f32x4{__uint_as_float(m2[4]), __uint_as_float(m2[5]), __uint_as_float(m2[6]), __uint_as_float(m2[7])});
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = row_of(sc, t);
            if (row < B) {
#pragma unroll
                for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * o3 + 4 * q + r;
                        if (c < Ai) {
                            const float zr = z[t][o3][r];
                            float av;
                            if (sac) {                                 // SAC.py:70-97
                                const float lsc = fminf(fmaxf(N.ls[c], -20.f), 2.f), sd = expf(lsc);
                                const float u = zr + sd * noise1[(size_t)row * am + c], du = u - zr;
                                lpsum += -(du * du) / (2.f * sd * sd) - lsc - kLogSqrt2Pi;
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
    }
    __syncthreads();                                                   // a_i in xrow is read by every lane group of a row below

    // =========================================================== B: Q(s, a) and dQ/da_i through the frozen critic
    float qsum = 0.f;
    const int kbA0 = (OT + aoff) >> 4, kbA1 = (OT + aoff + Ai - 1) >> 4, nA = kbA1 - kbA0 + 1;     // the k-blocks of agent i's action columns (<= 3)
    for (int hd = 0; hd < nq; ++hd) {
        const LayerDesc* L = NC.L + 3 * hd;
        g_cf w1 = thC + L[0].w_off, w2 = thC + L[1].w_off;
        N.stage3(thC, L, 1, -1, 0);
        // W1's action k-blocks -> LDS, tile (ot, j) at (ot * 3 + j) * 256
        for (int T = w; T < kHT2 * 3; T += 4) {
            const int ot = T / 3, j = T - 3 * ot;
            if (j < nA) st4(N.w1a + T * 256 + 4 * l, ld4(w1 + ((size_t)(ot * KB1c + kbA0 + j) * 256 + 4 * l)));
        }
        lds_barrier();
        for (int sc = 0; sc < nsc; ++sc) {
            g_cf px[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
                px[t] = (g_cf)X.xrow + (size_t)(row < B ? row : B - 1) * X.xp + 4 * q;
            }
            f32x4 XR[2][4][8];
            unsigned m1[8], m2[8] = {};
            N.l1_x(XR, px, w1, KB1c);
            const SweepNet::Slice0 f0 = N.sweep_fetch0<false>(w2);
            N.mask_bits(XR, m1);
            float zp[4] = {0.f, 0.f, 0.f, 0.f};
            N.sweep_x<false>(XR, w2, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {
                N.mask_push(m2, N.relu_pair(acc));
                N.head_valu_pair(acc, s, zp);
            }, f0);
            const SweepNet::Slice0 f1 = N.sweep_fetch0<true>(w2);
            float dzv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
                float qv = zp[t];
                qv += lane_xor<16>(qv);
                qv += lane_xor<32>(qv);
                qv += N.b3[0];
                dzv[t] = row < B ? dqv : 0.f;                          // actor_loss = -Q(s, actor(s)).mean() [+ alpha log pi]
                if (q == 0 && row < B) qsum += qv;
            }
            N.delta2_x_valu(XR, m2, dzv);
            f32x4 dx[4][3];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 3; ++j) dx[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            N.sweep_x<true>(XR, w2, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {      // d1 = (W2^T d2) o relu'(h1), pair by pair, straight into the dX MFMAs
                N.mask_pair(acc, N.mask_next(m1));
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j < nA) {                                      // dX of k-block kbA0 + j += W1^T d1 over this pair's output tiles
#pragma unroll
                        for (int o = 0; o < 2; ++o) {
                            f32x4 wa;
#pragma unroll
                            for (int e = 0; e < 4; ++e) wa[e] = N.w1a[((2 * s + o) * 3 + j) * 256 + C.tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
#pragma unroll
                            for (int t = 0; t < 4; ++t) dx[t][j] = mfma4(dx[t][j], wa, acc[o][t]);
                        }
                    }
                }
            }, f1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * (kbA0 + j) + 4 * q + r - OT - aoff;
                        if (j < nA && row < B && c >= 0 && c < Ai) {
                            g_f dst = X.dqa + (size_t)row * kWideApitch + c;
                            *dst = hd == 0 ? dx[t][j][r] : *dst + dx[t][j][r];
                        }
                    }
            }
        }
    }
    __syncthreads();

    // =========================================================== C: the actor's deltas from dQ/da_i down to layer 1
    N.stage3((g_cf)thA, NA.L, NT3A, NA.extra_off, NA.extra_n);
    float gls[NT3A][4];
#pragma unroll
    for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
        for (int r = 0; r < 4; ++r) gls[o3][r] = 0.f;
    for (int sc = 0; sc < nsc; ++sc) {
        f32x4 XR[2][4][8];
        unsigned m1[8], m2[8];
        {
            g_cf bp = (g_cf)bits + ((size_t)(sc * 4 + w) * 64 + l) * 16;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f32x4 a1 = ld4(bp + 4 * k), a2 = ld4(bp + 8 + 4 * k);
#pragma unroll
                for (int r = 0; r < 4; ++r) { m1[4 * k + r] = __float_as_uint(a1[r]); m2[4 * k + r] = __float_as_uint(a2[r]); }
            }
        }
        const SweepNet::Slice0 f1 = N.sweep_fetch0<true>((g_cf)thA + NA.L[1].w_off);
        f32x4 dz[4][NT3A];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = row_of(sc, t);
            const bool valid = row < B;
#pragma unroll
            for (int o3 = 0; o3 < NT3A; ++o3) {
                dz[t][o3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * o3 + 4 * q + r;
                    if (valid && c < Ai) {
                        const float dq = X.dqa[(size_t)row * kWideApitch + c];
                        const float av = X.xrow[(size_t)row * X.xp + OT + aoff + c];      // a_i (pass A)
                        if (sac) {                                     // through a = tanh(u), u = mean + exp(log_std) eps, and alpha log pi
                            const float d = dq * (1.f - av * av) + (alpha * invB) * (2.f * av);
                            const float lsc = fminf(fmaxf(N.ls[c], -20.f), 2.f);
                            dz[t][o3][r] = d;
                            gls[o3][r] += d * expf(lsc) * noise1[(size_t)row * am + c] - alpha * invB;
                        } else {
                            dz[t][o3][r] = dq * (1.f - av * av);
                        }
                    }
                }
                st4(N.tl(X.dzt, 4 * sc + t, NT3A) + o3 * 256, dz[t][o3]);
            }
        }
        N.delta2_x_tiles<NT3A>(XR, m2, dz);
        N.store_x(X.d2t, sc, XR);
        N.sweep_x<true>(XR, (g_cf)thA + NA.L[1].w_off, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {
            N.mask_pair(acc, N.mask_next(m1));
            N.store_pair(X.d1t, sc, s, acc);
        }, f1);
    }
    __syncthreads();
    float ss = N.dw2(grA + NA.L[1].w_off, grA + NA.L[1].b_off, (g_cf)X.h1t, (g_cf)X.d2t, nchunks, B);
    ss += N.dw3<NT3A>(grA + NA.L[2].w_off, grA + NA.L[2].b_off, (g_cf)X.h2t, (g_cf)X.dzt, nchunks, B);
    {
        const FRL_LDS int* tab = W.stage_idx(idx, B);
        ss += N.dw1(grA + NA.L[0].w_off, grA + NA.L[0].b_off, KB1a, Oi, [&](int row) { return ring + (size_t)tab[row] * R.stride + R.obs_off[ag]; }, (g_cf)X.d1t, nchunks, B);
    }

    // =========================================================== clip_grad_norm_, Adam, soft update of the actor's target; SAC: alpha
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
        const float raw = N.ls[tid];
        const float gr = (raw >= -20.f && raw <= 2.f) ? ((lsred[tid] + lsred[32 + tid]) + lsred[64 + tid]) + lsred[96 + tid] : 0.f;
        grA[NA.extra_off + tid] = gr;
        ss_extra = gr * gr;
    }
    ss = wave_sum(ss + ss_extra);
    const float qs = wave_sum(qsum), lps = wave_sum(lpsum);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { N.red[w] = ss; N.red[8 + w] = qs; N.red[12 + w] = lps; }
    if (tid == 0) N.red[32] = __int_as_float(steps[2 * ag]);
    __syncthreads();
    const float total = sqrtf(((N.red[0] + N.red[1]) + N.red[2]) + N.red[3]);
    const float qtot = ((N.red[8] + N.red[9]) + N.red[10]) + N.red[11];
    const float lptot = ((N.red[12] + N.red[13]) + N.red[14]) + N.red[15];
    const int tstep = __float_as_int(N.red[32]) + 1;
    const double bc1 = 1.0 - powi_d((double)a.beta1, tstep), bc2 = 1.0 - powi_d((double)a.beta2, tstep);
    AdamCoef co;
    co.coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    co.step = (float)((double)a.actor_lr / bc1); co.inv_bc2s = 1.f / (float)sqrt(bc2);
    co.w1 = 1.f - a.beta1; co.w2 = 1.f - a.beta2; co.beta2 = a.beta2; co.eps = a.adam_eps; co.wd = 0.f;
    co.tk = 1.f - a.tau; co.tau = a.tau;
    if (nag == 1) W.adam_stream<true>(thA, mA, vA, tgA, (g_cf)grA, NA.size >> 2, co);
    else W.adam_stream<false>(thA, mA, vA, tgA, (g_cf)grA, NA.size >> 2, co);
    if (tid == 0) {
        steps[2 * ag] = tstep;
        float* st = D.stats + ((size_t)p * nag + ag) * ST_COUNT;
        st[ST_ACTOR_LOSS] = sac ? (-(qtot * 0.5f) + alpha * lptot) * invB : -qtot * invB;   // SAC.py:251: (alpha log pi - Q).mean()
        st[ST_ACTOR_GNORM] = total;
        if (sac) {                                                     // alpha step on the batch's entropy (SAC.py:154-169,257-260)
            float* al = D.alpha + p * 4;
            const float ent_mean = -lptot * invB;
            const float mean_term = ent_mean - a.target_entropy;
            const float gl = alpha * mean_term;
            const int ta = steps[kMaxNets] + 1;
            float mi = al[1], vi = al[2];
            mi = mi + (gl - mi) * (1.f - a.beta1);
            vi = vi * a.beta2 + ((1.f - a.beta2) * gl) * gl;
            const double b1c = 1.0 - powi_d((double)a.beta1, ta), b2c = 1.0 - powi_d((double)a.beta2, ta);
            const float denom = sqrtf(vi) / (float)sqrt(b2c) + 1e-8f;
            al[0] = al[0] - (float)((double)a.alpha_lr / b1c) * (mi / denom);
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

#define FRL_ACTOR_X(NAME, NT3A)                                                                                             \
    __global__ __launch_bounds__(256) void NAME(const EngineDesc* __restrict__ Dp, LearnArgs a) {                          \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                       \
        ac_actor_x_body<NT3A>(*Dp, a, smem);                                                                               \
    }
FRL_ACTOR_X(ac_actor_x_a1_kernel, 1)
FRL_ACTOR_X(ac_actor_x_a2_kernel, 2)

}  // namespace frl
