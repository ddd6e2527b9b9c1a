// Critic stage of DDPG / TD3 / SAC for one learner per workgroup, register-chained (see device/chain.hpp, kernels_ppo2.hip):
// TD target with the target nets, twin / single critic forward, TD delta, backward, clip, Adam and the soft target update
// — DDPG_simple.py:139-149, TD3.py:193-213,235-244, SAC.py:226-238 — in ONE launch, no gradient slabs.
//
// ac_critic_kernel (kernels_critic.hip) splits a learner's batch over row-chunk workgroups that read every weight from L2,
// keep activations in LDS behind ~45 barrier phases per chunk and write partial gradients to HBM slabs for a separate
// reduce + Adam launch (0.51 of the fp32 MFMA peak, 2.5x the algorithmic HBM bytes).  Here ONE workgroup owns the learner's
// whole batch: each net's weights are staged once into a fragment-ordered LDS image and used for all its rows (four 64-row
// chunks at batch 256, every wave carrying 16 rows through the MLP in registers), the weight gradients of BOTH heads stay in
// the owner lanes' accumulators across the chunks, and clip + Adam + soft update run from those registers against
// theta / m / v / target in global memory — read and written exactly once per update.
//
// Shape: single agent, hidden 128 (ReLU), obs_dim + act_dim <= 16, act_dim <= 4, batch <= 256, no Batch_ObsNorm; populations
// of >= 128 learners (one workgroup per CU needs that many to fill the chip).  Everything else runs ac_critic_kernel.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain_net.hpp"
#include "device/update_common.hpp"
#include "device/ppo_timing.hpp"

#if defined(FRL_BWD_TIMING)
#undef PPO_UDUMP
#define PPO_UDUMP() do {} while (0)      // (row 1 of the clock array belongs to the backward's stamps in this build)
#endif

namespace frl {

// TT = 16-row tiles per wave in the target passes (4: one 256-row chunk, every weight fragment read from LDS feeds 16 MFMAs;
// 2: 128-row chunks, for batches of <= 128 rows)
// Developer knobs.  Measured in round 4: four tiles per wave (one 256-row chunk, 16 MFMAs per fragment read) put 256 more
// registers of activations next to the accumulators' AGPR half — 309 spilled VGPRs with or without the fetch-ahead staging; the
// two-tile build has none.
#ifndef FRL_CRITIC2_TT
#define FRL_CRITIC2_TT 2
#endif
#ifndef FRL_CRITIC2_AHEAD
#define FRL_CRITIC2_AHEAD 1     // the next net's image fetched in front of a target pass's last forward (1) or behind it (0)
#endif
// NW = waves per workgroup (device/chain_net.hpp): 8 since round 6 — 512 threads, every wave inside 256 registers, two waves
// per SIMD; a chunk is 16 NW rows (the target passes: TT times that).  NW = 4 is round 2-5's kernel (one wave per SIMD at
// 458-472 registers), kept as ac_critic_v2w4_* for same-box A/B runs (FRL_CHAIN_WAVES=4).
template <bool TWIN, int TT, int NW, bool NVEC>
__device__ __forceinline__ void ac_critic_v2_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    constexpr int kRC = 16 * NW;                                       // rows per chunk of the training pass
    constexpr int kRowsT = kRC * TT;                                   // rows per target-pass chunk
    constexpr int kNC = kChainBatch / kRC;                             // chunks of the largest batch = 16-row tiles per wave
    constexpr int kNCT = kChainBatch / kRowsT;
    static_assert(kNCT >= 1, "a target-pass chunk is at most the whole batch");
    constexpr int NH = TWIN ? 2 : 1;
    const int p = a.p0 + blockIdx.x;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[0];
    const NetDesc& NC = D.net[1];
    using Net = ChainNetT<NW>;
    Net C;
    C.init(smem);
    const ChainLds& S = C.S;
    const int tid = C.tid, l = C.l, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, O = R.obs_dim[0], A = R.act_dim[0], am = D.act_max;
    const bool sac = (D.algo == ALGO_SAC);
    const size_t lbase = (size_t)p * D.learner_stride;
    g_cf tgA = as_global(D.target + lbase + D.net_off[0]);
    g_cf tgC = as_global(D.target + lbase + D.net_off[1]);
    g_f thC = as_global(D.theta + lbase + D.net_off[1]);
    g_f tgCw = as_global(D.target + lbase + D.net_off[1]);
    g_f mC = as_global(D.m + lbase + D.net_off[1]);
    g_f vC = as_global(D.v + lbase + D.net_off[1]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_ci idx = as_global_i(D.idx + (size_t)p * D.batch_max);
    g_cf noise0 = as_global(D.noise + (size_t)p * D.noise_sets * D.batch_max * am);
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const float invB = 1.f / (float)B;
    const int nchunks = (B + kRC - 1) / kRC;

    // ---- this lane's rows (one per chunk) are the same in every pass: their ring addresses once, up front; a chunk's
    // record fields are loaded one chunk ahead of their use (nobody else hides that latency)
    // Every load below is UNCONDITIONAL — rows past the batch read the batch's last row, columns past the input read its last
    // column — because hipcc's s_waitcnt counts only loads that are always issued: one load under an `if` turns every wait of
    // the loop into vmcnt(0), i.e. into a wait for the chunk that was just prefetched (4.6 k cycles per target chunk pass).
    // What the clamped lanes loaded is dropped where it is USED, a chunk later (zero_pad), not where it is loaded.
    int ridx[kNC];
#pragma unroll
    for (int c = 0; c < kNC; ++c) {
        const int row = c * kRC + 16 * w + i16;
        ridx[c] = idx[row < B ? row : B - 1];
    }
    // [s | a] is one contiguous, 16-byte aligned slice of the record (chained_shape checks it): this lane's four columns
    // 4q .. 4q + 3 of it are ONE dwordx4 — a wave-load of 16 rows x 64 B instead of four dword gathers at ~65 cycles of the
    // CU's texture path each.  Columns past the input hold whatever follows in the record: zero_pad drops them by select.
    const int xoff = R.obs_off[0] + 4 * q;
    struct RowIn { f32x4 x; };
    auto load_row = [&](int c) {                                       // [s | a] of this lane's row of chunk c
        RowIn X;
        X.x = ld4(ring + (size_t)pick(ridx, c) * R.stride + xoff);
        return X;
    };
    auto zero_pad = [&](const f32x4& x, int width) {                   // columns >= width of a loaded slice are padding: exact zeros
        f32x4 r;                                                       // (the weight gradients of the padded columns must stay zero)
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = 4 * q + e < width ? x[e] : 0.f;
        return r;
    };
    // ---- The target passes carry TT 16-row tiles per wave (16 NW TT rows per chunk): nothing is differentiated through them, so
    // the registers the gradient accumulators need later hold more tiles now, and every weight fragment read from LDS feeds
    // 4 TT MFMAs.  Row of (chunk c2, tile t) on this lane: kRowsT c2 + 16 TT w + 16 t + i16; j4 = TT c2 + t.
    int ridxT[kNC];
#pragma unroll
    for (int j4 = 0; j4 < kNC; ++j4) {
        const int row = (j4 / TT) * kRowsT + 16 * TT * w + (j4 % TT) * 16 + i16;
        ridxT[j4] = idx[row < B ? row : B - 1];
    }
    // ... of s' (the columns past obs_dim: clamped to the last one in the dword form, dropped by zero_pad in both).  NVEC: the
    // record layout has next_obs 16-byte aligned and (reward, done) as one aligned pair — one dwordx4 + one dwordx2 per tile
    // instead of six dword gathers
    int colo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) colo[e] = R.nobs_off[0] + (NVEC ? 4 * q : (4 * q + e < O ? 4 * q + e : O - 1));
    struct RowIn2 { f32x4 x[TT]; float rew[TT], done[TT]; };
    auto tile_of = [&](const int (&v)[kNC], int c2, int t) {           // entry TT c2 + t (t is a constant of an unrolled loop)
        int r = v[t];
#pragma unroll
        for (int cc = 1; cc < kNCT; ++cc) r = c2 == cc ? v[TT * cc + t] : r;
        return r;
    };
    auto load_row2 = [&](bool want_rd, int c2) {                       // s' (obs columns) [+ reward / done] of the chunk's tiles
        RowIn2 X;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            g_cf rec = ring + (size_t)tile_of(ridxT, c2, t) * R.stride;
            if constexpr (NVEC) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                X.x[t] = ld4(rec + colo[0]);
                const f32x2 rd = *reinterpret_cast<const FRL_GLB f32x2*>(rec + R.rew_off);
                X.rew[t] = rd[0]; X.done[t] = rd[1];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) X.x[t][e] = rec[colo[e]];
                X.rew[t] = rec[R.rew_off]; X.done[t] = rec[R.done_off];
            }
        }
        return X;
    };
    const int nch2 = (B + kRowsT - 1) / kRowsT;

    // Developer knob (FRL_STAGGER): every workgroup runs the same ~600 k cycles and ends in the one phase that streams HBM
    // (theta / m / v / target, ~1 MB per learner), so the 256 CUs reach it together and share the chip's bandwidth (111 k
    // cycles per learner against 61 k for a CU on its own).  Four start phases spread the bursts (72 k) but the delayed
    // groups finish later by as much: no net gain (profiles/README.md), so the default is 0.
    if (a.stagger > 0 && (int)blockIdx.x < a.stagger_wgs) {            // (the first round's workgroups only: later ones inherit the spread)
        const int group = (blockIdx.x >> 3) & (a.stagger_groups - 1);
        for (int i = 0; i < group * a.stagger; ++i) __builtin_amdgcn_s_sleep(32);       // 32 x 64 cycles each
    }
    // =========================================================== a' = actor_target(s') for the whole batch -> S.ab (SAC: + log pi)
    PPO_T0();
    RowIn2 nxt2 = load_row2(false, 0);
    // the target actor's noise of this lane's rows (lane group 0 finalises the rows), loaded ahead of the pass: inside the epilogue
    // every load was an exposed HBM round trip (~2 k cycles, four per 128-row chunk)
    f32x4 nzr[kNC];
#pragma unroll
    for (int j4 = 0; j4 < kNC; ++j4) {
        nzr[j4] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = (j4 / TT) * kRowsT + 16 * TT * w + (j4 % TT) * 16 + i16;
        if (q == 0 && row < B && (sac || a.use_policy_noise)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < A) nzr[j4][r] = noise0[(size_t)row * am + r];
        }
    }
    // every net's image is FETCHED (global -> registers) in front of the last pass over the previous net and committed to LDS
    // when that pass is done: of the five stagings of a critic update only the first waits for HBM in the open
    typename Net::StageRegs pend = C.stage_fetch(tgA, 0, NA.extra_n);
    C.stage_commit(pend);
    PPO_T(0);
    PPO_U0();
    // (the last chunk of every pass is a peeled copy of the loop body: the next net's fetch rides on it, and a fetch under
    // `if (c2 + 1 == nch2)` inside the loop is a conditional load like any other)
    auto target_actor_chunk = [&](int c2, auto last_c) {
        constexpr bool last = decltype(last_c)::value;
        const RowIn2 cur = nxt2;
        nxt2 = load_row2(true, last ? 0 : c2 + 1);                     // (after the last chunk: chunk 0 of the target-critic pass)
        if constexpr (last && FRL_CRITIC2_AHEAD) pend = C.stage_fetch(tgC, 0);
        PPO_U(0);
        f32x4 xo[TT], z[TT], h1[TT][kHT], h2[TT][kHT];
#pragma unroll
        for (int t = 0; t < TT; ++t) xo[t] = zero_pad(cur.x[t], O);
        C.template forward_vh<TT>(xo, h1, h2, z, A);                    // (the actor head's act_dim <= 4 outputs as dot products)
        PPO_U(1);
        if constexpr (last && !FRL_CRITIC2_AHEAD) pend = C.stage_fetch(tgC, 0);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int row = c2 * kRowsT + 16 * TT * w + 16 * t + i16;
            const bool valid = row < B;
            if (q == 0 && row < kChainBatch) {                         // act_dim <= 4: the head's outputs sit on lane group 0
                f32x4 an = {0.f, 0.f, 0.f, 0.f};
                float lp = 0.f;
                f32x4 nr = nzr[t];
#pragma unroll
                for (int cc = 1; cc < kNCT; ++cc) nr = c2 == cc ? nzr[TT * cc + t] : nr;
                if (valid) {
                    if (sac) {                                         // SAC.py:70-97 on actor_target (SAC.py:227)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (r < A) {
                                const float ls = fminf(fmaxf(S.ls[r], -20.f), 2.f), sd = expf(ls);
                                const float u = z[t][r] + sd * nr[r], du = u - z[t][r];
                                lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                                lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                                an[r] = tanhf(u);
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (r < A) {
                                float v = tanhf(z[t][r]);
                                if (a.use_policy_noise) {              // TD3.py:196-198
                                    float nz = a.policy_noise_scale * (nr[r] * a.policy_noise);
                                    nz = fminf(fmaxf(nz, -a.noise_clip), a.noise_clip);
                                    v = fminf(fmaxf(v * a.max_action + nz, -a.max_action), a.max_action) / a.max_action;
                                }
                                an[r] = v;
                            }
                        }
                    }
                }
                st4(S.ab + row * 4, an);
                S.lpn[row] = lp;
            }
        }
        PPO_U(2);
    };
    for (int c2 = 0; c2 + 1 < nch2; ++c2) target_actor_chunk(c2, IC<0>{});
    target_actor_chunk(nch2 - 1, IC<1>{});
    PPO_T(1);
    // =========================================================== y = r + gamma (1 - d) min_h Q_target_h(s', a')  (SAC: - alpha log pi)
    RowIn nxt;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        PPO_UR();
        C.stage_commit(pend);
        PPO_U(3);
        auto target_critic_chunk = [&](int c2, auto last_c) {
            constexpr bool last = decltype(last_c)::value;
            const RowIn2 cur = nxt2;
            if constexpr (!last) nxt2 = load_row2(true, c2 + 1);
            else if (hd + 1 < NH) nxt2 = load_row2(true, 0);           // (hd is a constant of the unrolled head loop)
            else nxt = load_row(0);                                    // first chunk of the critic pass: [s | a], training-pass mapping
            if constexpr (last && FRL_CRITIC2_AHEAD) pend = hd + 1 < NH ? C.stage_fetch(tgC, hd + 1) : C.stage_fetch((g_cf)thC, 0);
            f32x4 xb[TT], z[TT], h1[TT][kHT], h2[TT][kHT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int row = c2 * kRowsT + 16 * TT * w + 16 * t + i16;
                xb[t] = zero_pad(cur.x[t], O);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * q + e;
                    if (row < B && f >= O && f < O + A) xb[t][e] = S.ab[row * 4 + f - O];      // a' from the target-actor pass
                }
            }
            PPO_U(4);
            C.template forward_vh<TT>(xb, h1, h2, z, 1);
            PPO_U(5);
            if constexpr (last && !FRL_CRITIC2_AHEAD) pend = hd + 1 < NH ? C.stage_fetch(tgC, hd + 1) : C.stage_fetch((g_cf)thC, 0);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int row = c2 * kRowsT + 16 * TT * w + 16 * t + i16;
                if (q == 0 && row < B) {
                    float qv = z[t][0];
                    if (hd == 1) qv = fminf(S.q1[row], qv);
                    if (hd == NH - 1) {
                        const float rew = cur.rew[t], done = cur.done[t];
                        S.yb[row] = sac ? rew + a.gamma * (1.f - done) * (qv + alpha * (-S.lpn[row])) : rew + a.gamma * qv * (1.f - done);
                    } else {
                        S.q1[row] = qv;
                    }
                }
            }
            PPO_U(6);
        };
        for (int c2 = 0; c2 + 1 < nch2; ++c2) target_critic_chunk(c2, IC<0>{});
        target_critic_chunk(nch2 - 1, IC<1>{});
    }
    PPO_UDUMP();
    PPO_T(2);
    // =========================================================== critic heads: forward, TD delta, backward into the owners' accumulators
    typename Net::Grad G[NH];
    float lossp = 0.f;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        typename Net::Grad& g = G[hd];
        C.grad_zero(g);
        PPO_T(3);
        C.stage_commit(pend);
        PPO_T(0);
        auto critic_chunk = [&](int c, auto last_c) {
            constexpr bool last = decltype(last_c)::value;
            const int row = c * kRC + 16 * w + i16;
            const bool valid = row < B;
            const RowIn cur = nxt;
            nxt = load_row(last ? 0 : c + 1);                          // (after the last chunk: the second head re-reads chunk 0)
            if constexpr (last && NW == 4) {
                if (hd + 1 < NH) pend = C.stage_fetch((g_cf)thC, hd + 1);
            }
            f32x4 xb[1] = {zero_pad(cur.x, O + A)}, z[1], h1[1][kHT], h2[1][kHT];
            PPO_T(4);
            C.template forward_vh<1>(xb, h1, h2, z, 1);
            PPO_T(5);
            f32x4 dz = {0.f, 0.f, 0.f, 0.f};
            if (q == 0 && valid) {                                     // loss(Q_h(s, a), y): F.mse_loss, or the Huber option
                float lrow, grow;
                td_loss_row(a, z[0][0] - S.yb[row], lrow, grow);
                dz[0] = grow * invB;
                lossp += lrow;
            }
            C.backward(g, xb[0], h1[0], h2[0], dz, 1, [&]() {         // (eight waves: the next head's image fetched where few registers are live)
                if constexpr (last && NW == 8) {
                    if (hd + 1 < NH) pend = C.stage_fetch((g_cf)thC, hd + 1);
                }
            });
            PPO_T(6);
        };
        for (int c = 0; c + 1 < nchunks; ++c) critic_chunk(c, IC<0>{});
        critic_chunk(nchunks - 1, IC<1>{});
        C.grad_finish(g);
    }

    PPO_T(3);
    // =========================================================== clip_grad_norm_ over the whole critic net, Adam, soft update
    float ss = 0.f;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) ss += C.grad_sumsq(G[hd]);
    ss = wave_sum(ss);
    const float lsum = wave_sum(lossp);
    lds_barrier();
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { S.red[w] = ss; S.red[8 + w] = lsum; }
    // the step count travels through LDS with the norm partials: thread 0 rewrites steps[1] at the end of the update with no
    // barrier in between, so a wave must not read it from global memory on its own schedule
    if (tid == 0) S.red[16] = __int_as_float(steps[1]);
    lds_barrier();
    float tot2 = S.red[0], loss = S.red[8];
#pragma unroll
    for (int i = 1; i < NW; ++i) { tot2 += S.red[i]; loss += S.red[8 + i]; }
    const float total = sqrtf(tot2);
    const int t = __float_as_int(S.red[16]) + 1;
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    AdamCoef co;
    co.coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    co.step = (float)((double)a.critic_lr / bc1); co.inv_bc2s = 1.f / (float)sqrt(bc2);
    co.w1 = 1.f - a.beta1; co.w2 = 1.f - a.beta2; co.beta2 = a.beta2; co.eps = a.adam_eps; co.wd = a.critic_wd;
    co.tk = 1.f - a.tau; co.tau = a.tau;
#if !(FRL_ABL & 1)
    if (a.do_actor != 0) {                                             // TD3: targets move with the delayed policy step (TD3.py:224-233)
        static_for<0, NH>([&](auto hd) { C.template adam_head<true, decltype(hd)::value, decltype(hd)::value == NH - 1>(G[decltype(hd)::value], thC, mC, vC, tgCw, co); });
    } else {
        static_for<0, NH>([&](auto hd) { C.template adam_head<false, decltype(hd)::value, decltype(hd)::value == NH - 1>(G[decltype(hd)::value], thC, mC, vC, tgCw, co); });
    }
#else
    if (co.coef == 123.f) static_for<0, NH>([&](auto hd) { S.red[40 + decltype(hd)::value] = C.grad_sumsq(G[decltype(hd)::value]) * co.step; });
#endif
    PPO_T(7);
    PPO_TDUMP();
    if (tid == 0) {
        steps[1] = t;
        float* st = D.stats + (size_t)p * ST_COUNT;
        st[ST_CRITIC_LOSS] = loss * invB;
        st[ST_CRITIC_GNORM] = total;
    }
}

#ifndef FRL_CRITIC8_TT
#define FRL_CRITIC8_TT 1        // 16-row tiles per wave in the eight-wave kernels' target passes (2 holds 128 activation registers next to the fetched image: spills)
#endif
// _nv: records whose next_obs slice is 16-byte aligned and whose (reward, done) pair 8-byte (critic2_nvec() in frl_api.hip)
#define FRL_CRITIC2_KERNEL(name, threads, twin, tt, nw, nvec)                                              \
    __global__ __launch_bounds__(threads) void name(const EngineDesc* __restrict__ Dp, LearnArgs a) {       \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                       \
        ac_critic_v2_body<twin, tt, nw, nvec>(*Dp, a, smem);                                               \
    }
FRL_CRITIC2_KERNEL(ac_critic_v2_twin_kernel, 512, true, FRL_CRITIC8_TT, 8, false)
FRL_CRITIC2_KERNEL(ac_critic_v2_single_kernel, 512, false, FRL_CRITIC8_TT, 8, false)
FRL_CRITIC2_KERNEL(ac_critic_v2_twin_nv_kernel, 512, true, FRL_CRITIC8_TT, 8, true)
FRL_CRITIC2_KERNEL(ac_critic_v2_single_nv_kernel, 512, false, FRL_CRITIC8_TT, 8, true)
FRL_CRITIC2_KERNEL(ac_critic_v2w4_twin_kernel, 256, true, FRL_CRITIC2_TT, 4, false)
FRL_CRITIC2_KERNEL(ac_critic_v2w4_single_kernel, 256, false, FRL_CRITIC2_TT, 4, false)

}  // namespace frl
