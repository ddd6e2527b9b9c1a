// Critic stage of DDPG / TD3 / SAC / MADDPG at HIDDEN 256 for one (learner, agent) per workgroup: kernels_criticw.hip's structure on
// device/chain_wide16.hpp, where every layer is a sweep over 32 KB slices of its image — DDPG_simple.py:139-149, TD3.py:193-213,
// 235-244, SAC.py:226-238, MADDPG_simple.py:165-180 with the hidden width the reference hard-codes (TD3.py:30) doubled to north_star's
// 256.  Target passes carry two 16-row tiles per wave (128 registers of activations per layer), the differentiated pass one;
// the weight gradients of the first two layers are contracted in passes of their own from what the chunk loop left in scratch.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain_wide16.hpp"
#include "device/wide_timing.hpp"

namespace frl {

template <int NH, int NT3A>
__device__ __forceinline__ void ac_critic_x_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    const int nag = D.n_agents;
    const int unit = blockIdx.x, p = a.p0 + unit / nag, ag = unit % nag;
    const RecordDesc& R = D.rec;
    const NetDesc& NC = D.net[2 * ag + 1];
    WideNet16 N16;
    N16.init(smem);
    const WideNet& W = N16.W;
    const ChainNet& C = W.C;
    const int tid = C.tid, l = C.l, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, OT = R.obs_total, AT = R.act_total, XT = OT + AT, am = D.act_max;
    const bool sac = (D.algo == ALGO_SAC);
    const size_t lbase = (size_t)p * D.learner_stride;
    const int noff = D.net_off[2 * ag + 1];
    g_cf tgC = as_global(D.target + lbase + noff);
    g_f thC = as_global(D.theta + lbase + noff);
    g_f tgCw = as_global(D.target + lbase + noff);
    g_f mC = as_global(D.m + lbase + noff);
    g_f vC = as_global(D.v + lbase + noff);
    g_f grC = as_global(D.grad + lbase + noff);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_ci idx = as_global_i(D.idx + ((size_t)p * nag + ag) * D.batch_max);
    g_cf noise_u = as_global(D.noise + ((size_t)p * nag + ag) * D.noise_sets * D.batch_max * am);
    Wide16Scratch X;
    X.init(as_global(D.wide_scr + ((size_t)p * nag + ag) * D.wide_unit), D.wide_bm, D.wide_xp, D.wide_op, nag);
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const float invB = 1.f / (float)B;
    const int nchunks = (B + 63) / 64, npair = (B + 127) / 128;
    const int KB1c = NC.L[0].k_pad >> 4;
    auto row2 = [&](int pr, int t) { return 128 * pr + 64 * t + 16 * w + i16; };       // tile t of wave w in 128-row pair pr = chunk 2 pr + t
    auto rec_of = [&](int row) { return ring + (size_t)idx[row < B ? row : B - 1] * R.stride; };
    auto xrow_of = [&](int row) { return (g_cf)X.xrow + (size_t)(row < B ? row : B - 1) * X.xp; };

    // =========================================================== a'_j = actor_target_j(s'_j) for every agent j -> xrow = [s'_all | a'_all]
    WIDE_T0();
    const FRL_LDS int* tab0 = W.stage_idx(idx, B);
    W.copy_cols(X.xrow, X.xp, ring, R.stride, tab0, B, R.nobs_off[0], OT);
    for (int j = 0; j < nag; ++j)
        if ((R.obs_off[j] - R.obs_off[0]) & 3) W.copy_cols(X.xobs + (size_t)j * D.wide_bm * X.op, X.op, ring, R.stride, tab0, B, R.nobs_off[j], R.obs_dim[j]);
    __syncthreads();
    for (int j = 0; j < nag; ++j) {
        const NetDesc& NA = D.net[2 * j];
        g_cf tgA = as_global(D.target + lbase + D.net_off[2 * j]);
        const int Aj = R.act_dim[j], aoff = R.act_off[j] - R.act_off[0], ooff = R.obs_off[j] - R.obs_off[0];
        const int KB1a = NA.L[0].k_pad >> 4;
        g_cf nz = noise_u + (size_t)(nag > 1 ? j : 0) * D.batch_max * am;
        const bool direct = (ooff & 3) == 0;
        N16.stage3(tgA, NA.L, NT3A, NA.extra_off, NA.extra_n);
        WIDE_T(0);
        for (int pr = 0; pr < npair; ++pr) {
            g_cf rp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = row2(pr, t), rc = row < B ? row : B - 1;
                rp[t] = direct ? (g_cf)X.xrow + (size_t)rc * X.xp + ooff : (g_cf)X.xobs + ((size_t)j * D.wide_bm + rc) * X.op;
            }
            f32x4 h1[2][kHT2], h2[2][kHT2], z[2][NT3A];
            N16.layer1<2>(h1, rp, tgA + NA.L[0].w_off, KB1a);
            WIDE_T(1);
            N16.sweep_regs<2>(h2, h1, tgA + NA.L[1].w_off);
            WIDE_T(2);
            N16.head_tiles<2, NT3A>(h2, z);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = row2(pr, t);
                const bool valid = row < B;
                float lp = 0.f;
#pragma unroll
                for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * o3 + 4 * q + r;
                        if (valid && c < Aj) {
                            const float zr = z[t][o3][r];
                            float av;
                            if (sac) {                                 // SAC.py:70-97 on actor_target (SAC.py:227)
                                const float lsc = fminf(fmaxf(N16.ls[c], -20.f), 2.f), sd = expf(lsc);
                                const float u = zr + sd * nz[(size_t)row * am + c], du = u - zr;
                                lp += -(du * du) / (2.f * sd * sd) - lsc - kLogSqrt2Pi;
                                lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                                av = tanhf(u);
                            } else {
                                av = tanhf(zr);
                                if (a.use_policy_noise) {              // TD3.py:196-198
                                    float n1 = a.policy_noise_scale * (nz[(size_t)row * am + c] * a.policy_noise);
                                    n1 = fminf(fmaxf(n1, -a.noise_clip), a.noise_clip);
                                    av = fminf(fmaxf(av * a.max_action + n1, -a.max_action), a.max_action) / a.max_action;
                                }
                            }
                            X.xrow[(size_t)row * X.xp + OT + aoff + c] = av;
                        }
                    }
                lp += __shfl_xor(lp, 16, 64);
                lp += __shfl_xor(lp, 32, 64);
                if (valid && q == 0) X.lpn[row] = lp;
            }
            WIDE_T(3);
        }
    }
    __syncthreads();                                                   // a' in xrow is read by every lane group of a row below

    // =========================================================== y = r + gamma (1 - d) min_h Q_target_h(s', a')  (SAC: - alpha log pi)
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        const LayerDesc* L = NC.L + 3 * hd;
        N16.stage3(tgC, L, 1, -1, 0);
        WIDE_T(0);
        for (int pr = 0; pr < npair; ++pr) {
            g_cf rp[2], recp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { const int row = row2(pr, t); recp[t] = rec_of(row); rp[t] = xrow_of(row); }
            f32x4 h1[2][kHT2], h2[2][kHT2], z[2];
            N16.layer1<2>(h1, rp, tgC + L[0].w_off, KB1c);
            WIDE_T(1);
            N16.sweep_regs<2>(h2, h1, tgC + L[1].w_off);
            WIDE_T(2);
            N16.head_valu<2>(h2, z, 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = row2(pr, t);
                if (q == 0 && row < B) {
                    float qv = z[t][0];
                    if (hd == 1) qv = fminf(X.q1[row], qv);
                    if (hd == NH - 1) {
                        const float rew = recp[t][R.rew_off + ag], done = recp[t][R.done_off + ag];
                        X.yb[row] = sac ? rew + a.gamma * (1.f - done) * (qv + alpha * (-X.lpn[row])) : rew + a.gamma * qv * (1.f - done);
                    } else {
                        X.q1[row] = qv;
                    }
                }
            }
            WIDE_T(3);
        }
    }

    // =========================================================== critic heads: forward, TD delta, backward; gradient passes; -> grad
    float lossp = 0.f, ss = 0.f;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        const LayerDesc* L = NC.L + 3 * hd;
        Wide16Grad<1> g;
        N16.grad_zero(g);
        N16.stage3((g_cf)thC, L, 1, -1, 0);
        WIDE_T(0);
        // pairs of chunks: tile t of pair pr = chunk 2 pr + t; one forward and one transposed sweep of W2 per pair
        for (int pr = 0; pr < npair; ++pr) {
            g_cf rp[2], h1row[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = row2(pr, t);
                rp[t] = rec_of(row) + R.obs_off[0];                    // (a record's [obs | act] columns are contiguous from its start)
                h1row[t] = (g_cf)X.h1s + (size_t)row * 256;
            }
            f32x4 h2[2][kHT2], z[2], dz[2][1];
            {
                f32x4 h1[2][kHT2];
                N16.layer1<2>(h1, rp, (g_cf)thC + L[0].w_off, KB1c);
                WIDE_T(4);
                // h1 of this lane's rows, row-major: the dW2 pass reads it transposed, backward_pair re-reads it for the ReLU mask
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ot = 0; ot < kHT2; ++ot) st4((g_f)h1row[t] + 16 * ot + 4 * q, h1[t][ot]);
                N16.sweep_regs<2>(h2, h1, (g_cf)thC + L[1].w_off);
            }
            WIDE_T(5);
            N16.head_valu<2>(h2, z, 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = row2(pr, t);
                dz[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (q == 0 && row < B) {                               // loss(Q_h(s, a), y): F.mse_loss, or the Huber option
                    float lrow, grow;
                    td_loss_row(a, z[t][0] - X.yb[row], lrow, grow);
                    dz[t][0][0] = grow * invB;
                    lossp += lrow;
                }
            }
            N16.backward_pair<1, true>(g, h2, dz, 1, (g_cf)thC + L[1].w_off, h1row, X.d2i + (size_t)(2 * pr) * 16384, X.dz1 + (size_t)(2 * pr) * 16384);
            WIDE_T(6);
        }
        N16.grad_finish(g);
        ss += N16.grad_store_3<1>(grC, L, g);
        __syncthreads();                                               // every wave's h1 rows and delta images are in scratch
        WIDE_T(7);
        ss += N16.dw_grad<8>(grC + L[1].w_off, (g_cf)X.d2i, nchunks, B, kHT2, 256, [&](int row) { return (g_cf)X.h1s + (size_t)row * 256; });
        WIDE_T(8);
        {
            const FRL_LDS int* tab = W.stage_idx(idx, B);
            auto rowf = [&](int row) { return ring + (size_t)tab[row] * R.stride + R.obs_off[0]; };
            if (KB1c <= 2) ss += N16.dw_grad<1>(grC + L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1c, XT, rowf);
            else if (KB1c <= 6) ss += N16.dw_grad<3>(grC + L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1c, XT, rowf);
            else if (KB1c <= 14) ss += N16.dw_grad<7>(grC + L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1c, XT, rowf);
            else ss += N16.dw_grad<kWideMaxKT>(grC + L[0].w_off, (g_cf)X.dz1, nchunks, B, KB1c, XT, rowf);
        }
        __syncthreads();                                               // the scratch images are free for the next head
        WIDE_T(9);
    }

    // =========================================================== clip_grad_norm_ over the whole critic net, Adam, soft update
    ss = wave_sum(ss);
    const float lsum = wave_sum(lossp);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { N16.red[w] = ss; N16.red[8 + w] = lsum; }
    if (tid == 0) N16.red[16] = __int_as_float(steps[2 * ag + 1]);
    __syncthreads();
    const float total = sqrtf(((N16.red[0] + N16.red[1]) + N16.red[2]) + N16.red[3]);
    const float loss = ((N16.red[8] + N16.red[9]) + N16.red[10]) + N16.red[11];
    const int tstep = __float_as_int(N16.red[16]) + 1;
    const double bc1 = 1.0 - powi_d((double)a.beta1, tstep), bc2 = 1.0 - powi_d((double)a.beta2, tstep);
    AdamCoef co;
    co.coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    co.step = (float)((double)a.critic_lr / bc1); co.inv_bc2s = 1.f / (float)sqrt(bc2);
    co.w1 = 1.f - a.beta1; co.w2 = 1.f - a.beta2; co.beta2 = a.beta2; co.eps = a.adam_eps; co.wd = a.critic_wd;
    co.tk = 1.f - a.tau; co.tau = a.tau;
    WIDE_T(10);
    if (nag == 1 && a.do_actor != 0) W.adam_stream<true>(thC, mC, vC, tgCw, (g_cf)grC, NC.size >> 2, co);
    else W.adam_stream<false>(thC, mC, vC, tgCw, (g_cf)grC, NC.size >> 2, co);
    WIDE_T(11);
    WIDE_TDUMP(0);
    if (tid == 0) {
        steps[2 * ag + 1] = tstep;
        float* st = D.stats + ((size_t)p * nag + ag) * ST_COUNT;
        st[ST_CRITIC_LOSS] = loss * invB;
        st[ST_CRITIC_GNORM] = total;
    }
}

#define FRL_CRITIC_X(NAME, NH, NT3A)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(const EngineDesc* __restrict__ Dp, LearnArgs a) {                          \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                       \
        ac_critic_x_body<NH, NT3A>(*Dp, a, smem);                                                                          \
    }
FRL_CRITIC_X(ac_critic_x_h1a1_kernel, 1, 1)
FRL_CRITIC_X(ac_critic_x_h1a2_kernel, 1, 2)
FRL_CRITIC_X(ac_critic_x_h2a1_kernel, 2, 1)
FRL_CRITIC_X(ac_critic_x_h2a2_kernel, 2, 2)

}  // namespace frl
