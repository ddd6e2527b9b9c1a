// Critic stage of DDPG / TD3 / SAC, PERSISTENT and pipelined across learners (round 3; kernels_critic2.hip is the
// one-learner-per-workgroup form it grew out of and remains the reference implementation of the arithmetic).
//
// What round 2's profile showed (profiles/r02/critic2_timing.txt, re-read in round 3 with the staging of the target critics
// counted where it belongs): of a learner's ~680 k cycles, 75 k are the five weight stagings and 117 k the clip + Adam + soft
// update — the two phases that move HBM bytes — and neither overlaps anything: one workgroup owns a CU (156 KB of LDS, 512
// registers), all 256 workgroups run the same phase at the same time, so each of those phases is a chip-wide HBM burst
// (256 x 84 KB per staging, 256 x ~1 MB per update: ~5.5 TB/s while it lasts, nothing in between).  With the parameters in
// fragment-image order in HBM (NetDesc::frag) the update needs no LDS and no barrier, so it can run anywhere a wave has
// issue slots to spare.  Here a workgroup walks through its learners (grid = min(learners, CUs)) and learner k's update is
// issued, one accumulator tile per k-block, inside the MFMA chains of learner k + 1's TARGET passes (ChainNet::forward's
// background hook): its HBM stream is spread over ~190 k cycles of matrix work instead of a 117 k cycle burst.  Only the
// last learner of a workgroup pays for its update in the open.
//
// DDPG_simple.py:139-149, TD3.py:193-213,235-244, SAC.py:226-238.  Shape: as kernels_critic2.hip.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain_net.hpp"
#include "device/update_common.hpp"
#include "device/ppo_timing.hpp"

namespace frl {

// everything of ONE learner the critic stage touches
struct CriticLearner {
    g_cf tgA, tgC;
    g_f thC, tgCw, mC, vC;
    g_cf ring, noise0;
    g_ci idx;
    int* steps;
    float* stats;
    float alpha;
};

__device__ __forceinline__ CriticLearner critic_learner(const EngineDesc& D, int p) {
    CriticLearner L;
    const size_t lbase = (size_t)p * D.learner_stride;
    L.tgA = as_global(D.target + lbase + D.net_off[0]);
    L.tgC = as_global(D.target + lbase + D.net_off[1]);
    L.thC = as_global(D.theta + lbase + D.net_off[1]);
    L.tgCw = as_global(D.target + lbase + D.net_off[1]);
    L.mC = as_global(D.m + lbase + D.net_off[1]);
    L.vC = as_global(D.v + lbase + D.net_off[1]);
    L.ring = as_global(D.replay + (size_t)p * D.capacity * D.rec.stride);
    L.idx = as_global_i(D.idx + (size_t)p * D.batch_max);
    L.noise0 = as_global(D.noise + (size_t)p * D.noise_sets * D.batch_max * D.act_max);
    L.steps = D.steps + (size_t)p * (kMaxNets + 1);
    L.stats = D.stats + (size_t)p * ST_COUNT;
    L.alpha = (D.algo == ALGO_SAC) ? D.alpha[p * 4 + 3] : 0.f;
    return L;
}

// The previous learner's clip + Adam + soft update as ChainNet::forward's background task: one accumulator tile per slot,
// software-pipelined over three slots — slot S stores the results of tile S - 2 and issues the four 16-byte loads of tile S
// (pre, right behind the block's vmcnt(0): see ChainNet::forward), and does the arithmetic of tile S - 1 (post, in the MFMAs'
// shadow), whose loads have had a whole k-block to land.  NH * 20 tiles in all — per head 16 of the 128 x 128 layer, 2 of the
// first layer, 2 of the head layer; the biases follow in finish().
template <int NH, bool SOFT, int STRIDE>
struct AdamBackground {
    // STRIDE: a tile every STRIDE-th slot.  Packed into consecutive slots the update asks for 16 KB per CU and ~1 k cycles — twice
    // what HBM delivers with all 256 CUs in step — and the MFMA chains wait at every drain; spread over the whole target phase it
    // stays under the chip's bandwidth.
    static constexpr bool kPipelined = true;
    static constexpr int kUnits = NH * 20;
    template <int S> static constexpr bool has_load() { return S % STRIDE == 0 && S / STRIDE < kUnits; }
    template <int S> static constexpr bool has_store() { return S % STRIDE == 0 && S / STRIDE >= 2 && S / STRIDE < kUnits + 2; }
    static constexpr int kSlotsNeeded = (kUnits + 1) * STRIDE + 1;
    const ChainNet& C;
    const HeadGrad (&G)[NH];
    g_f th, mA, vA, tg;                                                // the critic NET's blocks of the previous learner
    AdamBuf B;                                                         // ... as buffer resources
    AdamCoef co;
    ChainNet::AdamIn in, nxt, res;

    template <int S>
    __device__ __forceinline__ void pre() {
        if constexpr (S % STRIDE == 0) {
            constexpr int T = S / STRIDE;
            if constexpr (T >= 2 && T < kUnits + 2) {
                constexpr int U = T - 2;
                C.template adam_store<SOFT, U % 20, (U / 20) * kHeadFloats * 4>(B, res);
            }
            if constexpr (T < kUnits) nxt = C.template adam_load<SOFT, T % 20, (T / 20) * kHeadFloats * 4>(B);
        }
    }
    template <int S>
    __device__ __forceinline__ void post() {
        if constexpr (S % STRIDE == 0) {
            constexpr int T = S / STRIDE;
            if constexpr (T >= 1 && T < kUnits + 1) {
                constexpr int U = T - 1;
                res = C.template adam_compute<SOFT>(co, ChainNet::unit_grad<U % 20>(G[U / 20]), in);
            }
            if constexpr (T < kUnits) in = nxt;
        }
    }
    __device__ __forceinline__ void finish() {
#pragma unroll
        for (int hd = 0; hd < NH; ++hd)
            C.template adam_biases<SOFT>(G[hd], th + hd * kHeadFloats, mA + hd * kHeadFloats, vA + hd * kHeadFloats, tg + hd * kHeadFloats, co, 0.f, 0);
    }
};

__device__ __forceinline__ float pin_vgpr(float uniform) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));
    return v;
}

struct RowIn { f32x4 x; };
struct RowNext { f32x4 x; float rew, done; };

// this lane's row of 64-row chunk c: 64 c + 16 w + i16
__device__ __forceinline__ int pick4(const int (&v)[4], int c) { return c == 0 ? v[0] : (c == 1 ? v[1] : (c == 2 ? v[2] : v[3])); }
// [s | a] of the row
__device__ __forceinline__ RowIn load_row(const ChainNet& C, const RecordDesc& R, g_cf ring, const int (&ridx)[4], int c) {
    RowIn X;
    X.x = f32x4{0.f, 0.f, 0.f, 0.f};
    const int O = R.obs_dim[0], A = R.act_dim[0];
    const int ri = pick4(ridx, c);
    if (ri >= 0) {
        g_cf rec = ring + (size_t)ri * R.stride;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = 4 * C.q + e;
            if (f < O + A) X.x[e] = rec[f < O ? R.obs_off[0] + f : R.act_off[0] + f - O];
        }
    }
    return X;
}
// s' (obs columns) [+ reward / done] of the row
__device__ __forceinline__ RowNext load_next(const ChainNet& C, const RecordDesc& R, g_cf ring, const int (&ridx)[4], bool want_rd, int c) {
    RowNext X;
    X.x = f32x4{0.f, 0.f, 0.f, 0.f}; X.rew = 0.f; X.done = 0.f;
    const int O = R.obs_dim[0];
    const int ri = pick4(ridx, c);
    if (ri >= 0) {
        g_cf rec = ring + (size_t)ri * R.stride;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * C.q + e < O) X.x[e] = rec[R.nobs_off[0] + 4 * C.q + e];
        if (want_rd) { X.rew = rec[R.rew_off]; X.done = rec[R.done_off]; }
    }
    return X;
}

// ---- targets of one learner: a' = actor_target(s') [SAC: + log pi], y = r + gamma (1 - d) min_h Q_target_h(s', a') -> S.yb.
// NCH 64-row chunks (compile time: the background task's slots are template arguments), one 16-row tile per wave — round 2
// carried two tiles per wave here, which measured SLOWER per MFMA than the one-tile critic forward (74 % against 86 % of the
// issue rate) and would not leave the registers the previous learner's gradients need now.  8 slots per forward,
// (1 + NH) * NCH forwards.  Returns the first [s | a] row of the critic pass.
template <int NH, int NCH, class BG>
__device__ __forceinline__ RowIn target_phase(const ChainNet& C, const EngineDesc& D, const LearnArgs& a, const CriticLearner& L,
                                              const int (&ridx)[4], BG& bg PPO_TPARAMS) {
    const ChainLds& S = C.S;
    const RecordDesc& R = D.rec;
    const int w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, O = R.obs_dim[0], A = R.act_dim[0], am = D.act_max;
    const bool sac = (D.algo == ALGO_SAC);
    const bool noisy = sac || a.use_policy_noise;
    RowNext nxt2 = load_next(C, R, L.ring, ridx, false, 0);
    RowIn nxt;
    nxt.x = f32x4{0.f, 0.f, 0.f, 0.f};
    // the target actor's noise of a row (lane group 0 finalises the rows) is loaded one chunk ahead of its epilogue
    auto load_noise = [&](int c) {
        f32x4 n = {0.f, 0.f, 0.f, 0.f};
        const int row = c * 64 + 16 * w + i16;
        if (q == 0 && row < B && noisy) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < A) n[r] = L.noise0[(size_t)row * am + r];
        }
        return n;
    };
    f32x4 nz_next = load_noise(0);
    PPO_T(7);
    C.stage(L.tgA, 0, D.net[0].extra_n);
    PPO_T(0);
    static_for<0, NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const RowNext cur = nxt2;
        const f32x4 nr = nz_next;
        nxt2 = load_next(C, R, L.ring, ridx, true, c + 1 < NCH ? c + 1 : 0);         // (after the last chunk: chunk 0 of the target-critic pass)
        if constexpr (c + 1 < NCH) nz_next = load_noise(c + 1);
        f32x4 xb[1] = {cur.x}, z[1], h1[1][kHT], h2[1][kHT];
        C.template forward<1, 8 * c, BG, true>(xb, h1, h2, z, bg, A);
        const int row = c * 64 + 16 * w + i16;
        if (q == 0) {                                                  // act_dim <= 4: the head's outputs sit on lane group 0
            f32x4 an = {0.f, 0.f, 0.f, 0.f};
            float lp = 0.f;
            if (row < B) {
                if (sac) {                                             // SAC.py:70-97 on actor_target (SAC.py:227)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r < A) {
                            const float ls = fminf(fmaxf(S.ls[r], -20.f), 2.f), sd = expf(ls);
                            const float u = z[0][r] + sd * nr[r], du = u - z[0][r];
                            lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                            lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                            an[r] = tanhf(u);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r < A) {
                            float v = tanhf(z[0][r]);
                            if (a.use_policy_noise) {                  // TD3.py:196-198
                                float nzv = a.policy_noise_scale * (nr[r] * a.policy_noise);
                                nzv = fminf(fmaxf(nzv, -a.noise_clip), a.noise_clip);
                                v = fminf(fmaxf(v * a.max_action + nzv, -a.max_action), a.max_action) / a.max_action;
                            }
                            an[r] = v;
                        }
                    }
                }
            }
            st4(S.ab + row * 4, an);
            S.lpn[row] = lp;
        }
    });
    PPO_T(1);
    static_for<0, NH>([&](auto hdc) {
        constexpr int hd = decltype(hdc)::value;
        PPO_T(2);
        C.stage(L.tgC, hd);
        PPO_T(0);
        static_for<0, NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const RowNext cur = nxt2;
            if constexpr (c + 1 < NCH) nxt2 = load_next(C, R, L.ring, ridx, true, c + 1);
            else if constexpr (hd + 1 < NH) nxt2 = load_next(C, R, L.ring, ridx, true, 0);
            else nxt = load_row(C, R, L.ring, ridx, 0);                // first chunk of the critic pass: [s | a]
            const int row = c * 64 + 16 * w + i16;
            f32x4 xb[1] = {cur.x}, z[1], h1[1][kHT], h2[1][kHT];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 4 * q + e;
                if (row < B && f >= O && f < O + A) xb[0][e] = S.ab[row * 4 + f - O];          // a' from the target-actor pass
            }
            C.template forward<1, 8 * ((1 + hd) * NCH + c), BG, true>(xb, h1, h2, z, bg, 1);
            if (q == 0 && row < B) {
                float qv = z[0][0];
                if (hd == 1) qv = fminf(S.q1[row], qv);
                if (hd == NH - 1) S.yb[row] = sac ? cur.rew + a.gamma * (1.f - cur.done) * (qv + L.alpha * (-S.lpn[row])) : cur.rew + a.gamma * qv * (1.f - cur.done);
                else S.q1[row] = qv;
            }
        });
    });
    PPO_T(2);
    return nxt;
}

// SOFT: this launch also moves the critic's target (TD3: with the delayed policy step only, TD3.py:224-233; DDPG / SAC: always)
template <bool TWIN, int NCH, bool SOFT>
__device__ __forceinline__ void ac_critic_v3_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    constexpr int NH = TWIN ? 2 : 1;
    constexpr int kSlots = 8 * (1 + NH) * NCH;                         // background slots of one target phase
    constexpr int STRIDE = (kSlots - 1) / (NH * 20 + 1) > 0 ? (kSlots - 1) / (NH * 20 + 1) : 1;
    static_assert(kSlots >= (NH * 20 + 1) * STRIDE + 1, "not enough background slots for the update's tiles");
    const RecordDesc& R = D.rec;
    ChainNet C;
    C.init(smem);
    const ChainLds& S = C.S;
    const int tid = C.tid, l = C.l, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch;
    const float invB = 1.f / (float)B;
    const int nchunks = (B + 63) / 64;

    HeadGrad G[NH];
    AdamCoef co;
    co.coef = 1.f; co.step = 0.f; co.inv_bc2s = 1.f; co.w1 = 1.f - a.beta1; co.w2 = 1.f - a.beta2; co.beta2 = a.beta2; co.eps = a.adam_eps;
    co.wd = a.critic_wd; co.tk = 1.f - a.tau; co.tau = a.tau;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) C.grad_zero(G[hd]);
    CriticLearner prev;
    bool have_prev = false;
    PPO_T0();
    for (int p = a.p0 + (int)blockIdx.x; p < a.p0 + a.p_count; p += (int)gridDim.x) {
        const CriticLearner L = critic_learner(D, p);
        int ridx[4];                                                   // this lane's rows are the same in every pass: their ring addresses once
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = c * 64 + 16 * w + i16;
            ridx[c] = row < B ? L.idx[row] : -1;
        }
        // =========================================================== targets (with the previous learner's update in their shadow)
        RowIn nxt;
        if (!have_prev) {
            NoBackground nb;
            nxt = target_phase<NH, NCH>(C, D, a, L, ridx, nb PPO_TARGS);
        } else {
            // The update's coefficients as per-lane values the compiler cannot re-create: they are uniform, VALU instructions
            // take one scalar operand each, so hipcc keeps VGPR copies of them — and (measured, round 3: single-critic build)
            // re-materialises those copies inside the lane-group-0-only epilogue of a target pass, after which three quarters
            // of every tile of the background update used garbage.  An opaque v_mov at full exec pins them.
            AdamCoef cv;
            cv.coef = pin_vgpr(co.coef); cv.step = pin_vgpr(co.step); cv.inv_bc2s = pin_vgpr(co.inv_bc2s); cv.w1 = pin_vgpr(co.w1);
            cv.w2 = pin_vgpr(co.w2); cv.beta2 = pin_vgpr(co.beta2); cv.eps = pin_vgpr(co.eps); cv.wd = pin_vgpr(co.wd);
            cv.tk = pin_vgpr(co.tk); cv.tau = pin_vgpr(co.tau);
            AdamBackground<NH, SOFT, STRIDE> bg{C, G, prev.thC, prev.mC, prev.vC, prev.tgCw, adam_buf(prev.thC, prev.mC, prev.vC, prev.tgCw), cv, {}, {}, {}};
            nxt = target_phase<NH, NCH>(C, D, a, L, ridx, bg PPO_TARGS);
            bg.finish();
        }
        // =========================================================== critic heads: forward, TD delta, backward into the owners' accumulators
        float lossp = 0.f;
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {
            HeadGrad& g = G[hd];
            C.grad_zero(g);
            PPO_T(3);
            C.stage(L.thC, hd);
            PPO_T(0);
            for (int c = 0; c < nchunks; ++c) {
                const int row = c * 64 + 16 * w + i16;
                const bool valid = row < B;
                const RowIn cur = nxt;
                nxt = load_row(C, R, L.ring, ridx, c + 1 < nchunks ? c + 1 : 0);       // (after the last chunk: the second head re-reads chunk 0)
                f32x4 xb[1] = {cur.x}, z[1], h1[1][kHT], h2[1][kHT];
                PPO_T(4);
                C.template forward_vh<1>(xb, h1, h2, z, 1);
                PPO_T(5);
                f32x4 dz = {0.f, 0.f, 0.f, 0.f};
                if (q == 0 && valid) {                                 // loss(Q_h(s, a), y): F.mse_loss, or the Huber option
                    float lrow, grow;
                    td_loss_row(a, z[0][0] - S.yb[row], lrow, grow);
                    dz[0] = grow * invB;
                    lossp += lrow;
                }
                C.backward(g, xb[0], h1[0], h2[0], dz, 1);
                PPO_T(6);
            }
            C.grad_finish(g);
        }
        PPO_T(3);
        // =========================================================== clip_grad_norm_ over the whole critic net -> the update's coefficients
        float ss = 0.f;
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) ss += C.grad_sumsq(G[hd]);
        ss = wave_sum(ss);
        const float lsum = wave_sum(lossp);
        lds_barrier();
        if (l == 0) { S.red[w] = ss; S.red[8 + w] = lsum; }
        if (tid == 0) S.red[16] = __int_as_float(L.steps[1]);          // the step count through LDS: thread 0 writes it back below
        lds_barrier();
        const float total = sqrtf(((S.red[0] + S.red[1]) + S.red[2]) + S.red[3]);
        const float loss = ((S.red[8] + S.red[9]) + S.red[10]) + S.red[11];
        const int t = __float_as_int(S.red[16]) + 1;
        const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
        co.coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
        co.step = (float)((double)a.critic_lr / bc1); co.inv_bc2s = 1.f / (float)sqrt(bc2);
        if (tid == 0) {
            L.steps[1] = t;
            L.stats[ST_CRITIC_LOSS] = loss * invB;
            L.stats[ST_CRITIC_GNORM] = total;
        }
        prev = L;
        have_prev = true;
        PPO_T(7);
    }
    // =========================================================== the last learner's update, in the open
    if (have_prev) {
        static_for<0, NH>([&](auto hd) { C.template adam_head<SOFT, decltype(hd)::value>(G[decltype(hd)::value], prev.thC, prev.mC, prev.vC, prev.tgCw, co); });
    }
    PPO_T(7);
    PPO_TDUMP();
}

// NCH = 64-row chunks of the batch the target phase is compiled for: 4 (batch <= 256) or 2 (batch <= 128)
#define FRL_CRITIC3(name, TWIN, NCH, SOFT)                                                                        \
    __global__ __launch_bounds__(256) void name(const EngineDesc* __restrict__ Dp, LearnArgs a) {                   \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                \
        ac_critic_v3_body<TWIN, NCH, SOFT>(*Dp, a, smem);                                                           \
    }
FRL_CRITIC3(ac_critic_v3_twin_soft_kernel, true, 4, true)
FRL_CRITIC3(ac_critic_v3_twin_hold_kernel, true, 4, false)
#ifndef FRL_CRITIC3_ONLY_TWIN
FRL_CRITIC3(ac_critic_v3_twin_b128_soft_kernel, true, 2, true)
FRL_CRITIC3(ac_critic_v3_twin_b128_hold_kernel, true, 2, false)
FRL_CRITIC3(ac_critic_v3_single_soft_kernel, false, 4, true)
FRL_CRITIC3(ac_critic_v3_single_hold_kernel, false, 4, false)
FRL_CRITIC3(ac_critic_v3_single_b128_soft_kernel, false, 2, true)
FRL_CRITIC3(ac_critic_v3_single_b128_hold_kernel, false, 2, false)
#endif
#undef FRL_CRITIC3

}  // namespace frl
