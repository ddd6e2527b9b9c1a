// Actor stage of DDPG / TD3 / SAC / MADDPG at HIDDEN 256 for one (learner, agent) per workgroup: kernels_actorw.hip's three passes on
// device/chain_wide16.hpp (every layer a sweep over 32 KB slices of its image) — DDPG_simple.py:151-154, TD3.py:224-233,
// SAC.py:244-260, MADDPG_simple.py:182-186.
//   A  actor forward (two tiles per wave)            -> a_i into the critic's input row, h1 / h2 -> scratch
//   B  critic forward on [s | a] + the dX chain      -> dQ/da_i (the transposed sweep of W2, then W1's action k-blocks from the union)
//   C  actor backward from the stored activations    -> head gradient in registers, h1 rows / delta images -> scratch; dW2 and dW1
//      passes; clip + Adam streamed over the net
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
    WideNet16 N16;
    N16.init(smem);
    const WideNet& W = N16.W;
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
    const float invB = 1.f / (float)B;
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const int nq = sac ? NC.heads : 1;
    const float dqv = sac ? -0.5f * invB : -invB;
    const int nchunks = (B + 63) / 64, npair = (B + 127) / 128;
    const int KB1a = NA.L[0].k_pad >> 4, KB1c = NC.L[0].k_pad >> 4;
    auto row2 = [&](int pr, int t) { return 128 * pr + 64 * t + 16 * w + i16; };

    // =========================================================== A: a_i = tanh(actor_i(s_i)) (SAC: tanh(mean + std eps), sum of log pi)
    float lpsum = 0.f;
    const FRL_LDS int* tab0 = W.stage_idx(idx, B);
    W.copy_cols(X.xrow, X.xp, ring, R.stride, tab0, B, R.obs_off[0], XT);
    const bool direct = ((R.obs_off[ag] - R.obs_off[0]) & 3) == 0;
    if (!direct) W.copy_cols(X.xobs, X.op, ring, R.stride, tab0, B, R.obs_off[ag], Oi);
    __syncthreads();
    auto obs_of = [&](int row) {
        const int rc = row < B ? row : B - 1;
        return direct ? ring + (size_t)idx[rc] * R.stride + R.obs_off[ag] : (g_cf)X.xobs + (size_t)rc * X.op;
    };
    N16.stage3((g_cf)thA, NA.L, NT3A, NA.extra_off, NA.extra_n);
    for (int pr = 0; pr < npair; ++pr) {
        g_cf rp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) rp[t] = obs_of(row2(pr, t));
        f32x4 h1[2][kHT2], h2[2][kHT2], z[2][NT3A];
        N16.layer1<2>(h1, rp, (g_cf)thA + NA.L[0].w_off, KB1a);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT2; ++ot) st4(X.ah1 + ((size_t)(((2 * pr + t) * 4 + w) * kHT2 + ot) * 256 + 4 * l), h1[t][ot]);
        N16.sweep_regs<2>(h2, h1, (g_cf)thA + NA.L[1].w_off);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT2; ++ot) st4(X.ah2 + ((size_t)(((2 * pr + t) * 4 + w) * kHT2 + ot) * 256 + 4 * l), h2[t][ot]);
        N16.head_tiles<2, NT3A>(h2, z);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = row2(pr, t);
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
                                const float lsc = fminf(fmaxf(N16.ls[c], -20.f), 2.f), sd = expf(lsc);
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
        N16.stage3(thC, L, 1, -1, 0);
        for (int pr = 0; pr < npair; ++pr) {
            g_cf rp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { const int row = row2(pr, t); rp[t] = (g_cf)X.xrow + (size_t)(row < B ? row : B - 1) * X.xp; }
            f32x4 h1[2][kHT2], d1[2][kHT2];
            N16.layer1<2>(h1, rp, w1, KB1c);
            {
                f32x4 h2[2][kHT2], z[2], d2[2][kHT2];
                N16.sweep_regs<2>(h2, h1, w2);
                N16.head_valu<2>(h2, z, 1);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = row2(pr, t);
                    f32x4 dz = {0.f, 0.f, 0.f, 0.f};
                    if (q == 0 && row < B) { qsum += z[t][0]; dz[0] = dqv; }       // actor_loss = -Q(s, actor(s)).mean() [+ alpha log pi]
                    N16.delta2_valu(dz, h2[t], d2[t], 1);
                }
                N16.sweep_t<2>(d1, d2, w2);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int it = 0; it < kHT2; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d1[t][it][r] = h1[t][it][r] > 0.f ? d1[t][it][r] : 0.f;
            // W1's action k-blocks -> the union, tile (ot, j) at (ot * 3 + j) * 256 (48 tiles; the sweeps are done with their slices)
            lds_barrier();
            for (int T = w; T < kHT2 * 3; T += 4) {
                const int ot = T / 3, j = T - 3 * ot;
                if (j < nA) st4(W.u + T * 256 + 4 * l, ld4(w1 + ((size_t)(ot * KB1c + kbA0 + j) * 256 + 4 * l)));
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < nA) {                                          // dX of k-block kbA0 + j = W1^T d1 (transposed fragment reads)
                    f32x4 dx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ob = 0; ob < kHT2; ++ob) {
                        f32x4 wa;
#pragma unroll
                        for (int e = 0; e < 4; ++e) wa[e] = W.u[(ob * 3 + j) * 256 + C.tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
#pragma unroll
                        for (int t = 0; t < 2; ++t) dx[t] = mfma4(dx[t], wa, d1[t][ob]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = row2(pr, t);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * (kbA0 + j) + 4 * q + r - OT - aoff;
                            if (row < B && c >= 0 && c < Ai) {
                                g_f dst = X.dqa + (size_t)row * kWideApitch + c;
                                *dst = hd == 0 ? dx[t][r] : *dst + dx[t][r];
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // =========================================================== C: backward from the stored activations
    Wide16Grad<NT3A> g;
    N16.grad_zero(g);
    N16.stage3((g_cf)thA, NA.L, NT3A, NA.extra_off, NA.extra_n);
    float gls[NT3A][4];
#pragma unroll
    for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
        for (int r = 0; r < 4; ++r) gls[o3][r] = 0.f;
    for (int pr = 0; pr < npair; ++pr) {
        g_cf h1row[2];
        f32x4 h2[2][kHT2], z[2][NT3A], dz[2][NT3A];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int cg = 2 * pr + t;
            h1row[t] = (g_cf)X.h1s + (size_t)row2(pr, t) * 256;
            // h1 back from pass A (tile order) and out again row-major: the dW2 pass reads it transposed, backward_pair its ReLU mask
#pragma unroll
            for (int ot = 0; ot < kHT2; ++ot) {
                const f32x4 hv = ld4((g_cf)(X.ah1 + ((size_t)((cg * 4 + w) * kHT2 + ot) * 256 + 4 * l)));
                st4((g_f)h1row[t] + 16 * ot + 4 * q, hv);
                h2[t][ot] = ld4((g_cf)(X.ah2 + ((size_t)((cg * 4 + w) * kHT2 + ot) * 256 + 4 * l)));
            }
        }
        N16.head_tiles<2, NT3A>(h2, z);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = row2(pr, t);
            const bool valid = row < B;
#pragma unroll
            for (int o3 = 0; o3 < NT3A; ++o3) {
                dz[t][o3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * o3 + 4 * q + r;
                    if (valid && c < Ai) {
                        const float dq = X.dqa[(size_t)row * kWideApitch + c];
                        if (sac) {
                            const float av = X.xrow[(size_t)row * X.xp + OT + aoff + c];
                            const float d = dq * (1.f - av * av) + (alpha * invB) * (2.f * av);
                            const float lsc = fminf(fmaxf(N16.ls[c], -20.f), 2.f);
                            dz[t][o3][r] = d;
                            gls[o3][r] += d * expf(lsc) * noise1[(size_t)row * am + c] - alpha * invB;
                        } else {
                            const float av = tanhf(z[t][o3][r]);
                            dz[t][o3][r] = dq * (1.f - av * av);
                        }
                    }
                }
            }
        }
        N16.backward_pair<NT3A, false>(g, h2, dz, 0, (g_cf)thA + NA.L[1].w_off, h1row, X.d2i + (size_t)(2 * pr) * 16384, X.dz1 + (size_t)(2 * pr) * 16384);
    }
    N16.grad_finish(g);
    float ss = N16.grad_store_3<NT3A>(grA, NA.L, g);
    __syncthreads();
    ss += N16.dw_grad<8>(grA + NA.L[1].w_off, (g_cf)X.d2i, nchunks, B, kHT2, 256, [&](int row) { return (g_cf)X.h1s + (size_t)row * 256; });
    {
        const FRL_LDS int* tab = W.stage_idx(idx, B);
        auto rowf = [&](int row) { return ring + (size_t)tab[row] * R.stride + R.obs_off[ag]; };
        if (KB1a <= 2) ss += N16.dw_grad<1>(grA + NA.L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1a, Oi, rowf);
        else if (KB1a <= 6) ss += N16.dw_grad<3>(grA + NA.L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1a, Oi, rowf);
        else if (KB1a <= 14) ss += N16.dw_grad<7>(grA + NA.L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1a, Oi, rowf);
        else ss += N16.dw_grad<kWideMaxKT>(grA + NA.L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1a, Oi, rowf);
    }

    // =========================================================== clip_grad_norm_, Adam, soft update of the actor's target; SAC: alpha
#pragma unroll
    for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = gls[o3][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
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
        const float raw = N16.ls[tid];
        const float gr = (raw >= -20.f && raw <= 2.f) ? ((lsred[tid] + lsred[32 + tid]) + lsred[64 + tid]) + lsred[96 + tid] : 0.f;
        grA[NA.extra_off + tid] = gr;
        ss_extra = gr * gr;
    }
    ss = wave_sum(ss + ss_extra);
    const float qs = wave_sum(qsum), lps = wave_sum(lpsum);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { N16.red[w] = ss; N16.red[8 + w] = qs; N16.red[12 + w] = lps; }
    if (tid == 0) N16.red[32] = __int_as_float(steps[2 * ag]);
    __syncthreads();
    const float total = sqrtf(((N16.red[0] + N16.red[1]) + N16.red[2]) + N16.red[3]);
    const float qtot = ((N16.red[8] + N16.red[9]) + N16.red[10]) + N16.red[11];
    const float lptot = ((N16.red[12] + N16.red[13]) + N16.red[14]) + N16.red[15];
    const int tstep = __float_as_int(N16.red[32]) + 1;
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
