// Critic stage of DDPG / TD3 / SAC / MADDPG for one (learner, agent) per workgroup on the K-sliced chained design
// (device/chain_wide.hpp): wide first layers (SAC at Humanoid-v4's dims: 393 input columns; MADDPG's centralised critics: 69),
// heads of up to 32 outputs, batches of any number of 256-row super-chunks — DDPG_simple.py:139-149, TD3.py:193-213,235-244,
// SAC.py:226-238, MADDPG_simple.py:165-180 — in ONE launch: target actions of every agent, TD target through the target critic(s),
// critic forward / backward, clip, Adam and (single agent) the soft update of the critic's target.
//
// Row mapping everywhere: tile t of wave w in super-chunk sc holds rows 256 sc + 64 t + 16 w + i16 — tile t IS 64-row chunk t of the
// super-chunk, so the first layer of four chunks runs as one sweep over the K-slices of W1 and the chunk loop behind it walks the tiles.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain_wide.hpp"
#include "device/wide_timing.hpp"

namespace frl {

// NH critic heads (2: TD3 / SAC twins), NT3A head tiles of the actors (act_dim <= 16 -> 1, <= 32 -> 2)
template <int NH, int NT3A>
__device__ __forceinline__ void ac_critic_wide_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    const int nag = D.n_agents;
    const int unit = blockIdx.x, p = a.p0 + unit / nag, ag = unit % nag;
    const RecordDesc& R = D.rec;
    const NetDesc& NC = D.net[2 * ag + 1];
    WideNet W;
    W.init(smem);
    const ChainNet& C = W.C;
    const ChainLds& S = C.S;
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
    g_cf noise_u = as_global(D.noise + ((size_t)p * nag + ag) * D.noise_sets * D.batch_max * am);      // this unit's sets
    WideScratch X;
    X.init(as_global(D.wide_scr + ((size_t)p * nag + ag) * D.wide_unit), D.wide_bm, D.wide_xp, D.wide_op, nag);
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const float invB = 1.f / (float)B;
    const int nsc = (B + 255) / 256, nchunks = (B + 63) / 64;
    const int KB1c = NC.L[0].k_pad >> 4;
    auto row_of = [&](int sc, int t) { return 256 * sc + 64 * t + 16 * w + i16; };
    auto rec_of = [&](int row) { return ring + (size_t)idx[row < B ? row : B - 1] * R.stride; };
    auto xrow_of = [&](int row) { return (g_cf)X.xrow + (size_t)(row < B ? row : B - 1) * X.xp; };

    // =========================================================== a'_j = actor_target_j(s'_j) for every agent j -> xrow = [s'_all | a'_all] (SAC: + log pi)
    WIDE_T0();
    const FRL_LDS int* tab0 = W.stage_idx(idx, B);                     // (the union is free until the first sweep)
    W.copy_cols(X.xrow, X.xp, ring, R.stride, tab0, B, R.nobs_off[0], OT);        // s' of every agent: contiguous in the record
    for (int j = 0; j < nag; ++j)                                      // ... and the agents whose columns start off a 16-byte boundary there
        if ((R.obs_off[j] - R.obs_off[0]) & 3) W.copy_cols(X.xobs + (size_t)j * D.wide_bm * X.op, X.op, ring, R.stride, tab0, B, R.nobs_off[j], R.obs_dim[j]);
    __syncthreads();
    for (int j = 0; j < nag; ++j) {
        const NetDesc& NA = D.net[2 * j];
        g_cf tgA = as_global(D.target + lbase + D.net_off[2 * j]);
        const int Oj = R.obs_dim[j], Aj = R.act_dim[j], aoff = R.act_off[j] - R.act_off[0], ooff = R.obs_off[j] - R.obs_off[0];
        const int KB1a = NA.L[0].k_pad >> 4;
        // MATD3's per-agent smoothing noise is set j of the updating agent; single agent: set 0 (TD3 policy noise / SAC eps')
        g_cf nz = noise_u + (size_t)(nag > 1 ? j : 0) * D.batch_max * am;
        // agent j's columns of xrow, or — when they do not start on a 16-byte boundary there — a copy of them
        const bool direct = (ooff & 3) == 0;

        W.stage23(tgA, NA.L, NT3A, NA.extra_off, NA.extra_n);
        WIDE_T(0);
        for (int sc = 0; sc < nsc; ++sc) {
            g_cf rp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t), rc = row < B ? row : B - 1;
                rp[t] = direct ? (g_cf)X.xrow + (size_t)rc * X.xp + ooff : (g_cf)X.xobs + ((size_t)j * D.wide_bm + rc) * X.op;
            }
            f32x4 h1[4][kHT];
            W.l1_sweep<4>(h1, rp, tgA + NA.L[0].w_off, KB1a);
            WIDE_T(1);
            static_for<0, 2>([&](auto hc) {
                constexpr int half = decltype(hc)::value;
                f32x4 h2[2][kHT], z[2][NT3A];
                W.l23<2, NT3A, false, 4, 2 * half>(h1, h2, z, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = row_of(sc, 2 * half + t);
                    const bool valid = row < B;
                    float lp = 0.f;
                    f32x4 an[NT3A];
#pragma unroll
                    for (int o3 = 0; o3 < NT3A; ++o3) {
                        an[o3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * o3 + 4 * q + r;
                            if (valid && c < Aj) {
                                const float zr = z[t][o3][r];
                                if (sac) {                             // SAC.py:70-97 on actor_target (SAC.py:227)
                                    const float ls = fminf(fmaxf(S.ls[c], -20.f), 2.f), sd = expf(ls);
                                    const float u = zr + sd * nz[(size_t)row * am + c], du = u - zr;
                                    lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                                    lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                                    an[o3][r] = tanhf(u);
                                } else {
                                    float v = tanhf(zr);
                                    if (a.use_policy_noise) {          // TD3.py:196-198
                                        float n1 = a.policy_noise_scale * (nz[(size_t)row * am + c] * a.policy_noise);
                                        n1 = fminf(fmaxf(n1, -a.noise_clip), a.noise_clip);
                                        v = fminf(fmaxf(v * a.max_action + n1, -a.max_action), a.max_action) / a.max_action;
                                    }
                                    an[o3][r] = v;
                                }
                            }
                        }
                    }
                    lp += lane_xor<16>(lp);
                    lp += lane_xor<32>(lp);
                    if (valid) {
                        // agent j's columns of the joint target action (the agents' blocks need not be 16-byte aligned: scalar stores)
#pragma unroll
                        for (int o3 = 0; o3 < NT3A; ++o3)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 16 * o3 + 4 * q + r;
                                if (c < Aj) X.xrow[(size_t)row * X.xp + OT + aoff + c] = an[o3][r];
                            }
                        if (q == 0) X.lpn[row] = lp;
                    }
                }
            });
            WIDE_T(2);
        }
    }
    __syncthreads();                                                   // a' in xrow is read by every lane group of a row below

    // =========================================================== y = r + gamma (1 - d) min_h Q_target_h(s', a')  (SAC: - alpha log pi)
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        const LayerDesc* L = NC.L + 3 * hd;
        W.stage23(tgC, L, 1, -1, 0);
        WIDE_T(0);
        for (int sc = 0; sc < nsc; ++sc) {
            g_cf rp[4], recp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = row_of(sc, t);
                recp[t] = rec_of(row);
                rp[t] = xrow_of(row);
            }
            f32x4 h1[4][kHT];
            W.l1_sweep<4>(h1, rp, tgC + L[0].w_off, KB1c);
            WIDE_T(3);
            static_for<0, 2>([&](auto hc) {
                constexpr int half = decltype(hc)::value;
                f32x4 h2[2][kHT], z[2][1];
                W.l23<2, 1, true, 4, 2 * half>(h1, h2, z, 1);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = row_of(sc, 2 * half + t);
                    if (q == 0 && row < B) {
                        float qv = z[t][0][0];
                        if (hd == 1) qv = fminf(X.q1[row], qv);
                        if (hd == NH - 1) {
                            g_cf rec = recp[2 * half + t];
                            const float rew = rec[R.rew_off + ag], done = rec[R.done_off + ag];
                            X.yb[row] = sac ? rew + a.gamma * (1.f - done) * (qv + alpha * (-X.lpn[row])) : rew + a.gamma * qv * (1.f - done);
                        } else {
                            X.q1[row] = qv;
                        }
                    }
                }
            });
            WIDE_T(4);
        }
    }

    // =========================================================== critic heads: forward, TD delta, backward; first-layer deltas -> X.dz1;
    // dW1 pass; every gradient -> grad
    float lossp = 0.f, ss = 0.f;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        const LayerDesc* L = NC.L + 3 * hd;
        WideGrad<1> g;
        W.grad_zero(g);
        W.stage23((g_cf)thC, L, 1, -1, 0);
        WIDE_T(0);
        // two tiles (chunks) per sweep here: next to the backward's live state (accumulators, h2, both deltas, the exchange
        // fragments) a third and fourth tile of first-layer activations do not fit in the register file
        for (int sc2 = 0; sc2 < 2 * nsc; ++sc2) {
            if (128 * sc2 >= B) break;
            g_cf rp[2];                                                // (a record's [obs | act] columns are contiguous from its start: build_record)
#pragma unroll
            for (int t = 0; t < 2; ++t) rp[t] = rec_of(128 * sc2 + 64 * t + 16 * w + i16) + R.obs_off[0];
            f32x4 h1[2][kHT];
            W.l1_sweep<2>(h1, rp, (g_cf)thC + L[0].w_off, KB1c);
            WIDE_T(5);
            for (int c = 0; c < 2; ++c) {
                const int cg = 2 * sc2 + c;                            // 64-row chunk of the batch
                if (64 * cg < B) {                                     // (uniform: the chunk exists)
                    const int row = 64 * cg + 16 * w + i16;
                    const bool valid = row < B;
                    f32x4 h2[1][kHT], z[1][1];
                    W.l23<1, 1, true, 2, 0>(h1, h2, z, 1);
                    WIDE_T(6);
                    f32x4 dz[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
                    if (q == 0 && valid) {                             // loss(Q_h(s, a), y): F.mse_loss, or the Huber option
                        float lrow, grow;
                        td_loss_row(a, z[0][0][0] - X.yb[row], lrow, grow);
                        dz[0][0] = grow * invB;
                        lossp += lrow;
                    }
                    W.backward<1, true>(g, h1[0], h2[0], dz, 1, X.dz1 + (size_t)cg * 8192);
                    WIDE_T(7);
                }
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot) h1[0][ot] = h1[1][ot];
            }
        }
        W.grad_finish(g);
        ss += W.grad_store_23<1>(grC, L, g);                           // (before the dW1 pass: its 208 accumulators want the registers)
        __syncthreads();                                               // every wave's deltas of the last chunk are in X.dz1
        {
            const FRL_LDS int* tab = W.stage_idx(idx, B);
            ss += W.dw1_grad_any(grC, L, (g_cf)X.dz1, nchunks, B, KB1c, XT, [&](int row) { return ring + (size_t)tab[row] * R.stride + R.obs_off[0]; });
        }
        WIDE_T(8);
        __syncthreads();                                               // X.dz1 is free for the next head
    }

    // =========================================================== clip_grad_norm_ over the whole critic net, Adam, soft update
    ss = wave_sum(ss);
    const float lsum = wave_sum(lossp);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { S.red[w] = ss; S.red[8 + w] = lsum; }
    if (tid == 0) S.red[16] = __int_as_float(steps[2 * ag + 1]);
    __syncthreads();                                                   // (also: every gradient store of this workgroup has landed)
    const float total = sqrtf(((S.red[0] + S.red[1]) + S.red[2]) + S.red[3]);
    const float loss = ((S.red[8] + S.red[9]) + S.red[10]) + S.red[11];
    const int tstep = __float_as_int(S.red[16]) + 1;
    const double bc1 = 1.0 - powi_d((double)a.beta1, tstep), bc2 = 1.0 - powi_d((double)a.beta2, tstep);
    AdamCoef co;
    co.coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    co.step = (float)((double)a.critic_lr / bc1); co.inv_bc2s = 1.f / (float)sqrt(bc2);
    co.w1 = 1.f - a.beta1; co.w2 = 1.f - a.beta2; co.beta2 = a.beta2; co.eps = a.adam_eps; co.wd = a.critic_wd;
    co.tk = 1.f - a.tau; co.tau = a.tau;
    WIDE_T(9);
    // single agent: the target moves here (TD3: with the delayed policy step, TD3.py:224-233); MADDPG: soft_update_kernel afterwards
    // (every agent's workgroups read every target actor)
    if (nag == 1 && a.do_actor != 0) W.adam_stream<true>(thC, mC, vC, tgCw, (g_cf)grC, NC.size >> 2, co);
    else W.adam_stream<false>(thC, mC, vC, tgCw, (g_cf)grC, NC.size >> 2, co);
    WIDE_T(10);
    WIDE_TDUMP(0);
    if (tid == 0) {
        steps[2 * ag + 1] = tstep;
        float* st = D.stats + ((size_t)p * nag + ag) * ST_COUNT;
        st[ST_CRITIC_LOSS] = loss * invB;                              // (the sum over the heads of their means, as the reference adds them)
        st[ST_CRITIC_GNORM] = total;
    }
}

#define FRL_CRITIC_WIDE(NAME, NH, NT3A)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(const EngineDesc* __restrict__ Dp, LearnArgs a) {                          \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                       \
        ac_critic_wide_body<NH, NT3A>(*Dp, a, smem);                                                                       \
    }
FRL_CRITIC_WIDE(ac_critic_wide_h1a1_kernel, 1, 1)
FRL_CRITIC_WIDE(ac_critic_wide_h1a2_kernel, 1, 2)
FRL_CRITIC_WIDE(ac_critic_wide_h2a1_kernel, 2, 1)
FRL_CRITIC_WIDE(ac_critic_wide_h2a2_kernel, 2, 2)

}  // namespace frl
