// Critic stage of DDPG / TD3 / SAC / MADDPG at HIDDEN 256 for one (learner, agent) per workgroup on device/chain_wide16.hpp: the
// activations of a 256-row super-chunk (four 16-row tiles per wave) stay in registers from the first layer to the head and back
// down (l1_x -> sweep_x -> deltas -> sweep_x<TR>), the weight images stream through LDS 32 KB at a time, and only what the
// weight-gradient passes need (h1, h2, d2, d1, tile-lane order) goes to the unit's scratch — DDPG_simple.py:139-149,
// TD3.py:193-213,235-244, SAC.py:226-238, MADDPG_simple.py:165-180 with the hidden width the reference hard-codes (TD3.py:30)
// doubled to north_star's 256.
//
// Row mapping: tile t of wave w in super-chunk sc holds rows 256 sc + 64 t + 16 w + i16 (tile t = 64-row chunk 4 sc + t).
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
    SweepNet N;
    N.init(smem);
    const WideNet& W = N.W;
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
    const int nsc = (B + 255) / 256, nchunks = (B + 63) / 64;
    const int KB1c = NC.L[0].k_pad >> 4;
    auto row_of = [&](int sc, int t) { return 256 * sc + 64 * t + 16 * w + i16; };
    auto rec_of = [&](int row) { return ring + (size_t)idx[row < B ? row : B - 1] * R.stride; };
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
        N.stage3(tgA, NA.L, NT3A, NA.extra_off, NA.extra_n);
        WIDE_T(0);
        for (int sc = 0; sc < nsc; ++sc) {
            g_cf px[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t), rc = row < B ? row : B - 1;
                px[t] = (direct ? (g_cf)X.xrow + (size_t)rc * X.xp + ooff : (g_cf)X.xobs + ((size_t)j * D.wide_bm + rc) * X.op) + 4 * q;
            }
            f32x4 XR[2][4][8];
            N.l1_x(XR, px, tgA + NA.L[0].w_off, KB1a);
            WIDE_T(1);
            f32x4 z[4][NT3A];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int o3 = 0; o3 < NT3A; ++o3) z[t][o3] = ld4((lds_cf)(N.b3 + 16 * o3 + 4 * q));
            N.sweep_x<false>(XR, tgA + NA.L[1].w_off, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {
                N.relu_pair(acc);
                N.head_tiles_pair<NT3A>(acc, s, z);
            });
            WIDE_T(2);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
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
                                const float lsc = fminf(fmaxf(N.ls[c], -20.f), 2.f), sd = expf(lsc);
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
                lp += lane_xor<16>(lp);
                lp += lane_xor<32>(lp);
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
        N.stage3(tgC, L, 1, -1, 0);
        WIDE_T(0);
        for (int sc = 0; sc < nsc; ++sc) {
            g_cf px[4], recp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
                recp[t] = rec_of(row);
                px[t] = (g_cf)X.xrow + (size_t)(row < B ? row : B - 1) * X.xp + 4 * q;
            }
            f32x4 XR[2][4][8];
            N.l1_x(XR, px, tgC + L[0].w_off, KB1c);
            WIDE_T(1);
            float zp[4] = {0.f, 0.f, 0.f, 0.f};
            N.sweep_x<false>(XR, tgC + L[1].w_off, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {
                N.relu_pair(acc);
                N.head_valu_pair(acc, s, zp);
            });
            WIDE_T(2);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
                float qv = zp[t];
                qv += lane_xor<16>(qv);
                qv += lane_xor<32>(qv);
                qv += N.b3[0];
                if (q == 0 && row < B) {
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

    // =========================================================== critic heads: forward, TD delta, deltas down to layer 1; gradient passes
    float lossp = 0.f, ss = 0.f;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        const LayerDesc* L = NC.L + 3 * hd;
        g_cf w2 = (g_cf)thC + L[1].w_off;
        N.stage3((g_cf)thC, L, 1, -1, 0);
        WIDE_T(0);
        for (int sc = 0; sc < nsc; ++sc) {
            g_cf px[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) px[t] = rec_of(row_of(sc, t)) + R.obs_off[0] + 4 * q;   // (a record's [obs | act] columns are contiguous from its start)
            f32x4 XR[2][4][8];
            unsigned m1[8], m2[8] = {};
            N.l1_x(XR, px, (g_cf)thC + L[0].w_off, KB1c);
            const SweepNet::Slice0 f0 = N.sweep_fetch0<false>(w2);      // W2's first slice, in flight under the mask words and h1's stores
            N.mask_bits(XR, m1);
            N.store_x(X.h1t, sc, XR);
            WIDE_T(4);
            float zp[4] = {0.f, 0.f, 0.f, 0.f};
            N.sweep_x<false>(XR, w2, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {
                N.mask_push(m2, N.relu_pair(acc));
                N.store_pair(X.h2t, sc, s, acc);
                N.head_valu_pair(acc, s, zp);
            }, f0);
            WIDE_T(5);
            const SweepNet::Slice0 f1 = N.sweep_fetch0<true>(w2);       // W2^T's first slice, in flight under the delta arithmetic
            float dzv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
                float qv = zp[t];
                qv += lane_xor<16>(qv);
                qv += lane_xor<32>(qv);
                qv += N.b3[0];
                dzv[t] = 0.f;                                          // loss(Q_h(s, a), y): F.mse_loss, or the Huber option
                if (row < B) {
                    float lrow, grow;
                    td_loss_row(a, qv - X.yb[row], lrow, grow);
                    dzv[t] = grow * invB;
                    if (q == 0) lossp += lrow;
                }
                f32x4 dzt = {0.f, 0.f, 0.f, 0.f};
                if (q == 0) dzt[0] = dzv[t];                           // D layout of the head tile: output 0 on lane group 0, register 0
                st4(N.tl(X.dzt, 4 * sc + t, 1), dzt);
            }
            N.delta2_x_valu(XR, m2, dzv);                              // d2 = (W3^T dz) o relu'(h2): the B operand of the transposed sweep
            N.store_x(X.d2t, sc, XR);
            WIDE_T(6);
            N.sweep_x<true>(XR, w2, (lds_cf)N.b2, [&](int s, f32x4 (&acc)[2][4]) {       // d1 = (W2^T d2) o relu'(h1) -> d1t
                N.mask_pair(acc, N.mask_next(m1));
                N.store_pair(X.d1t, sc, s, acc);
            }, f1);
            WIDE_T(7);
        }
        __syncthreads();                                               // every wave's activations and deltas are in scratch
        ss += N.dw2(grC + L[1].w_off, grC + L[1].b_off, (g_cf)X.h1t, (g_cf)X.d2t, nchunks, B);
        WIDE_T(8);
        ss += N.dw3<1>(grC + L[2].w_off, grC + L[2].b_off, (g_cf)X.h2t, (g_cf)X.dzt, nchunks, B);
        {
            const FRL_LDS int* tab = W.stage_idx(idx, B);
            ss += N.dw1(grC + L[0].w_off, grC + L[0].b_off, KB1c, XT, [&](int row) { return ring + (size_t)tab[row] * R.stride + R.obs_off[0]; }, (g_cf)X.d1t, nchunks, B);
        }
        __syncthreads();                                               // the scratch tensors are free for the next head
        WIDE_T(9);
    }

    // =========================================================== clip_grad_norm_ over the whole critic net, Adam, soft update
    ss = wave_sum(ss);
    const float lsum = wave_sum(lossp);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { N.red[w] = ss; N.red[8 + w] = lsum; }
    if (tid == 0) N.red[16] = __int_as_float(steps[2 * ag + 1]);
    __syncthreads();
    const float total = sqrtf(((N.red[0] + N.red[1]) + N.red[2]) + N.red[3]);
    const float loss = ((N.red[8] + N.red[9]) + N.red[10]) + N.red[11];
    const int tstep = __float_as_int(N.red[16]) + 1;
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
