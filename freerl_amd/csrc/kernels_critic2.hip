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
#include "device/chain.hpp"
#include "device/update_common.hpp"
#include "device/ppo_timing.hpp"

namespace frl {

namespace {

constexpr int kCr2Batch = 256;

struct Cr2Lds {
    lds_f w1, w2, w3, b1, b2, b3, ls, ea, eb, ab, yb, q1, lpn, red;
};

__device__ __forceinline__ Cr2Lds cr2_carve(float* smem) {
    Cr2Lds S;
    lds_f p = (lds_f)smem;
    S.w1 = p; p += kHT * 256;
    S.w2 = p; p += kHT * kHT * 256;
    S.w3 = p; p += kHT * 256;
    S.ea = p; p += kHT * 4 * 256;
    S.eb = p; p += kHT * 4 * 256;
    S.b1 = p; p += kHid;
    S.b2 = p; p += kHid;
    S.b3 = p; p += 16;
    S.ls = p; p += 16;
    S.ab = p; p += kCr2Batch * 4;
    S.yb = p; p += kCr2Batch;
    S.q1 = p; p += kCr2Batch;
    S.lpn = p; p += kCr2Batch;
    S.red = p; p += 64;
    return S;
}

// the weight-gradient accumulators one lane owns for one 3-layer head (MFMA D layout, out = 16*ot + 4q + r, in = 16*kt + i16):
// layer 2: ot in {2w, 2w+1} x kt 0..7; layer 1 (one 16-wide input block): ot in {2w, 2w+1}; head: kt in {2w, 2w+1}
struct HeadGrad {
    f32x4 g2[2][kHT], g1[2], g3[2];
    float gb1[2], gb2[2], gb3;
};

}  // namespace

template <bool TWIN>
__device__ __forceinline__ void ac_critic_v2_body(const EngineDesc& D, const LearnArgs& a, float* smem) {
    constexpr int NH = TWIN ? 2 : 1;
    const int p = a.p0 + blockIdx.x;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[0];
    const NetDesc& NC = D.net[1];
    const Cr2Lds S = cr2_carve(smem);
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), i16 = l & 15, q = l >> 4;
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
    const int nchunks = (B + 63) / 64;
    const int fslot = (q * 16 + (i16 ^ q)) << 2;                                  // forward / exchange fragment read (16 B)
    const int tslot = (((i16 >> 2) * 16) << 2) + (i16 & 3);                       // transposed read / owner write: + ((f ^ (i16 >> 2)) << 2)

    // ---- one net's three layers -> LDS images (fragment order).  Engine layout: Wk[k][n] (n contiguous), then b[n_pad].
    auto stage = [&](g_cf th, const NetDesc& N, int l0) {
        const LayerDesc &L1 = N.L[l0], &L2 = N.L[l0 + 1], &L3 = N.L[l0 + 2];
        lds_barrier();                                             // every wave is done with the previous images
        // a 16-byte LDS slot holds W[out = n][in = 4*k4 .. 4*k4 + 3]: four rows of Wk[in][out] at column n — lanes walk n, so
        // every load is a coalesced row segment; all loads of an image are issued before its stores
        {
            const int n = tid & 127, half = tid >> 7;              // layer 2: 128 columns x 32 k-quads, 16 quads per thread
            f32x4 t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k4 = 2 * j + half;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[j][e] = th[L2.w_off + (4 * k4 + e) * kHid + n];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k4 = 2 * j + half, kb = k4 >> 2, qq = k4 & 3;
                st4(S.w2 + ((n >> 4) * kHT + kb) * 256 + ((qq * 16 + ((n & 15) ^ qq)) << 2), t[j]);
            }
            f32x4 u1, u3[2];                                       // layer 1: 128 columns x 4 k-quads; head: 16 columns x 32 k-quads
#pragma unroll
            for (int e = 0; e < 4; ++e) u1[e] = th[L1.w_off + (4 * (tid >> 7) + e) * kHid + n];
            f32x4 u1b;
#pragma unroll
            for (int e = 0; e < 4; ++e) u1b[e] = th[L1.w_off + (4 * (2 + (tid >> 7)) + e) * kHid + n];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k4 = (tid >> 4) + 16 * j;
#pragma unroll
                for (int e = 0; e < 4; ++e) u3[j][e] = th[L3.w_off + (4 * k4 + e) * 16 + (tid & 15)];
            }
            { const int qq = tid >> 7; st4(S.w1 + (n >> 4) * 256 + ((qq * 16 + ((n & 15) ^ qq)) << 2), u1); }
            { const int qq = 2 + (tid >> 7); st4(S.w1 + (n >> 4) * 256 + ((qq * 16 + ((n & 15) ^ qq)) << 2), u1b); }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k4 = (tid >> 4) + 16 * j, kb = k4 >> 2, qq = k4 & 3;
                st4(S.w3 + kb * 256 + ((qq * 16 + ((tid & 15) ^ qq)) << 2), u3[j]);
            }
        }
        if (tid < kHid) { S.b1[tid] = th[L1.b_off + tid]; S.b2[tid] = th[L2.b_off + tid]; }
        if (tid < 16) {
            S.b3[tid] = th[L3.b_off + tid];
            S.ls[tid] = (N.extra_n > 0 && tid < N.extra_n) ? th[N.extra_off + tid] : 0.f;
        }
        lds_barrier();
    };
    // ---- the chained forward of one wave's 16 rows: x (B operand of layer 1) -> h1, h2 (kept for the backward) -> head tile
    auto forward = [&](const f32x4& xb, f32x4 (&h1)[kHT], f32x4 (&h2)[kHT]) {
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            f32x4 acc = mfma4(ld4((lds_cf)(S.b1 + ot * 16 + 4 * q)), ld4((lds_cf)(S.w1 + ot * 256 + fslot)), xb);
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[ot][r] = fmaxf(acc[r], 0.f);
        }
        // the 32 MFMAs of one output tile are ONE dependent accumulator chain: walk the eight tiles' chains side by side (k-block
        // outer, k-step middle, tile inner) so that consecutive MFMAs never wait for each other's result
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) h2[ot] = ld4((lds_cf)(S.b2 + ot * 16 + 4 * q));
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {
            f32x4 wf[kHT];
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot) wf[ot] = ld4((lds_cf)(S.w2 + (ot * kHT + kb) * 256 + fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot) h2[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], h1[kb][e], h2[ot], 0, 0, 0);
        }
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[ot][r] = fmaxf(h2[ot][r], 0.f);
        f32x4 z = ld4((lds_cf)(S.b3 + 4 * q));
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) z = mfma4(z, ld4((lds_cf)(S.w3 + kb * 256 + fslot)), h2[kb]);
        return z;
    };

    // ---- this lane's rows (one per 64-row chunk) are the same in every pass: their ring addresses once, up front; a chunk's
    // record fields are loaded one chunk ahead of their use (one workgroup per CU: nobody else hides that latency)
    int ridx[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int row = c * 64 + 16 * w + i16;
        ridx[c] = row < B ? idx[row] : -1;
    }
    struct RowIn { f32x4 x; float rew, done; };
    // kind 0: s' (obs columns only); 1: s' + reward / done; 2: [s | a]
    auto load_row = [&](int kind, int c) {
        RowIn X;
        X.x = f32x4{0.f, 0.f, 0.f, 0.f}; X.rew = 0.f; X.done = 0.f;
        const int ri = c == 0 ? ridx[0] : (c == 1 ? ridx[1] : (c == 2 ? ridx[2] : ridx[3]));
        if (ri >= 0) {
            g_cf rec = ring + (size_t)ri * R.stride;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 4 * q + e;
                if (kind == 2) { if (f < O + A) X.x[e] = rec[f < O ? R.obs_off[0] + f : R.act_off[0] + f - O]; }
                else if (f < O) X.x[e] = rec[R.nobs_off[0] + f];
            }
            if (kind == 1) { X.rew = rec[R.rew_off]; X.done = rec[R.done_off]; }
        }
        return X;
    };

    // Developer knob (FRL_STAGGER): every workgroup runs the same ~700 k cycles and ends in the one phase that streams HBM
    // (theta / m / v / target, ~1 MB per learner), so the 256 CUs reach it together and share the chip's bandwidth (111 k
    // cycles per learner against 61 k for a CU on its own).  Four start phases spread the bursts (72 k) but the delayed
    // groups finish later by as much: no net gain at two learners per CU (profiles/README.md), so the default is 0.
    if (a.stagger > 0) {
        const int group = (blockIdx.x >> 3) & 3;
        for (int i = 0; i < group * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);      // 127 x 64 cycles each
    }
    // ---- The target passes carry TWO 16-row tiles per wave (128 rows per chunk): nothing is differentiated through them, so the
    // registers the gradient accumulators need later hold a second tile now, and every weight fragment read from LDS feeds
    // eight MFMAs on two independent accumulator chains.  Row of (chunk c2, tile t) on this lane: 128 c2 + 32 w + 16 t + i16.
    int ridxT[4];
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
        const int row = (j4 >> 1) * 128 + 32 * w + (j4 & 1) * 16 + i16;
        ridxT[j4] = row < B ? idx[row] : -1;
    }
    struct RowIn2 { f32x4 x[2]; float rew[2], done[2]; };
    auto load_row2 = [&](bool want_rd, int c2) {                       // s' (obs columns) [+ reward / done] of both tiles
        RowIn2 X;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            X.x[t] = f32x4{0.f, 0.f, 0.f, 0.f}; X.rew[t] = 0.f; X.done[t] = 0.f;
            const int ri = c2 == 0 ? ridxT[t] : ridxT[2 + t];
            if (ri >= 0) {
                g_cf rec = ring + (size_t)ri * R.stride;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < O) X.x[t][e] = rec[R.nobs_off[0] + 4 * q + e];
                if (want_rd) { X.rew[t] = rec[R.rew_off]; X.done[t] = rec[R.done_off]; }
            }
        }
        return X;
    };
    auto forward2 = [&](const f32x4 (&xb)[2], f32x4 (&z)[2]) {
        f32x4 h1[2][kHT], h2[2][kHT];
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            const f32x4 wf = ld4((lds_cf)(S.w1 + ot * 256 + fslot)), bb = ld4((lds_cf)(S.b1 + ot * 16 + 4 * q));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 acc = mfma4(bb, wf, xb[t]);
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[t][ot][r] = fmaxf(acc[r], 0.f);
            }
        }
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) h2[0][ot] = h2[1][ot] = ld4((lds_cf)(S.b2 + ot * 16 + 4 * q));
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {                             // sixteen accumulator chains side by side (see forward)
            f32x4 wf[kHT];
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot) wf[ot] = ld4((lds_cf)(S.w2 + (ot * kHT + kb) * 256 + fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                    for (int t = 0; t < 2; ++t) h2[t][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], h1[t][kb][e], h2[t][ot], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[t][ot][r] = fmaxf(h2[t][ot][r], 0.f);
        z[0] = z[1] = ld4((lds_cf)(S.b3 + 4 * q));
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {
            const f32x4 wf = ld4((lds_cf)(S.w3 + kb * 256 + fslot));
#pragma unroll
            for (int t = 0; t < 2; ++t) z[t] = mfma4(z[t], wf, h2[t][kb]);
        }
    };
    const int nch2 = (B + 127) / 128;

    // =========================================================== a' = actor_target(s') for the whole batch -> S.ab (SAC: + log pi)
    PPO_T0();
    RowIn2 nxt2 = load_row2(false, 0);
    stage(tgA, NA, 0);
    PPO_T(0);
    for (int c2 = 0; c2 < nch2; ++c2) {
        const RowIn2 cur = nxt2;
        nxt2 = load_row2(true, c2 + 1 < nch2 ? c2 + 1 : 0);            // (after the last chunk: chunk 0 of the target-critic pass)
        f32x4 z[2];
        forward2(cur.x, z);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = c2 * 128 + 32 * w + 16 * t + i16;
            const bool valid = row < B;
            if (q == 0 && row < kCr2Batch) {                           // act_dim <= 4: the head's outputs sit on lane group 0
                f32x4 an = {0.f, 0.f, 0.f, 0.f};
                float lp = 0.f;
                if (valid) {
                    if (sac) {                                         // SAC.py:70-97 on actor_target (SAC.py:227)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (r < A) {
                                const float ls = fminf(fmaxf(S.ls[r], -20.f), 2.f), sd = expf(ls);
                                const float u = z[t][r] + sd * noise0[(size_t)row * am + r], du = u - z[t][r];
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
                                    float nz = a.policy_noise_scale * (noise0[(size_t)row * am + r] * a.policy_noise);
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
    }
    PPO_T(1);
    // =========================================================== y = r + gamma (1 - d) min_h Q_target_h(s', a')  (SAC: - alpha log pi)
    RowIn nxt;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        stage(tgC, NC, 3 * hd);
        for (int c2 = 0; c2 < nch2; ++c2) {
            const RowIn2 cur = nxt2;
            if (c2 + 1 < nch2) nxt2 = load_row2(true, c2 + 1);
            else if (hd + 1 < NH) nxt2 = load_row2(true, 0);
            else nxt = load_row(2, 0);                                 // first chunk of the critic pass: [s | a], 64-row mapping
            f32x4 xb[2], z[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = c2 * 128 + 32 * w + 16 * t + i16;
                xb[t] = cur.x[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * q + e;
                    if (row < B && f >= O && f < O + A) xb[t][e] = S.ab[row * 4 + f - O];      // a' from the target-actor pass
                }
            }
            forward2(xb, z);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = c2 * 128 + 32 * w + 16 * t + i16;
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
        }
    }
    PPO_T(2);
    // =========================================================== critic heads: forward, TD delta, backward into the owners' accumulators
    HeadGrad G[NH];
    float lossp = 0.f;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        HeadGrad& g = G[hd];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.g1[x] = f32x4{0.f, 0.f, 0.f, 0.f}; g.g3[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.gb1[x] = 0.f; g.gb2[x] = 0.f;
        }
        g.gb3 = 0.f;
        PPO_T(3);
        stage(as_global((const float*)(D.theta + lbase + D.net_off[1])), NC, 3 * hd);
        PPO_T(0);
        for (int c = 0; c < nchunks; ++c) {
            const int row = c * 64 + 16 * w + i16;
            const bool valid = row < B;
            const RowIn cur = nxt;
            nxt = load_row(2, c + 1 < nchunks ? c + 1 : 0);            // (after the last chunk: the second head re-reads chunk 0)
            const f32x4 xb = cur.x;
            f32x4 h1[kHT], h2[kHT];
            PPO_T(4);
            const f32x4 z = forward(xb, h1, h2);
            PPO_T(5);
            f32x4 dz = {0.f, 0.f, 0.f, 0.f};
            if (q == 0 && valid) {                                     // loss(Q_h(s, a), y): F.mse_loss, or the Huber option
                float lrow, grow;
                td_loss_row(a, z[0] - S.yb[row], lrow, grow);
                dz[0] = grow * invB;
                lossp += lrow;
            }
            // ---- exchange 1: H2 and dz -> head gradient (this wave's in-tiles 2w, 2w+1)
            lds_barrier();
            auto put_tile = [&](lds_f E, int ft, const f32x4& t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) E[(ft * 4 + w) * 256 + tslot + (((4 * q + r) ^ (i16 >> 2)) << 2)] = t[r];
            };
            auto get_frag = [&](lds_cf E, int ft, int bb) { return ld4(E + (ft * 4 + bb) * 256 + fslot); };
#pragma unroll
            for (int ft = 0; ft < kHT; ++ft) put_tile(S.ea, ft, h2[ft]);
            put_tile(S.eb, 0, dz);
            lds_barrier();
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const f32x4 af = get_frag(S.eb, 0, bb);
                if (w == 0) g.gb3 += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
                for (int x = 0; x < 2; ++x) g.g3[x] = mfma4(g.g3[x], af, get_frag(S.ea, 2 * w + x, bb));
            }
            f32x4 d2[kHT];                                             // dH2 = W3^T dz through the ReLU
#pragma unroll
            for (int it = 0; it < kHT; ++it) {
                f32x4 wa;
#pragma unroll
                for (int e = 0; e < 4; ++e) wa[e] = S.w3[it * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
                const f32x4 acc = mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, wa, dz);
#pragma unroll
                for (int r = 0; r < 4; ++r) d2[it][r] = h2[it][r] > 0.f ? acc[r] : 0.f;
            }
            lds_barrier();
            // ---- exchange 2: H1 and dz2 -> layer-2 gradient
#pragma unroll
            for (int ft = 0; ft < kHT; ++ft) { put_tile(S.ea, ft, h1[ft]); put_tile(S.eb, ft, d2[ft]); }
            lds_barrier();
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                f32x4 af[2], bf[kHT];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    af[x] = get_frag(S.eb, 2 * w + x, bb);
                    g.gb2[x] += (af[x][0] + af[x][1]) + (af[x][2] + af[x][3]);
                }
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt) bf[kt] = get_frag(S.ea, kt, bb);
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = mfma4(g.g2[x][kt], af[x], bf[kt]);
            }
            f32x4 d1[kHT];                                             // dH1 = W2^T dz2 through the ReLU; eight chains side by side
#pragma unroll
            for (int it = 0; it < kHT; ++it) d1[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ob = 0; ob < kHT; ++ob) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float wa[kHT];
#pragma unroll
                    for (int it = 0; it < kHT; ++it) wa[it] = S.w2[(ob * kHT + it) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
#pragma unroll
                    for (int it = 0; it < kHT; ++it) d1[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[it], d2[ob][e], d1[it], 0, 0, 0);
                }
            }
#pragma unroll
            for (int it = 0; it < kHT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[it][r] = h1[it][r] > 0.f ? d1[it][r] : 0.f;
            lds_barrier();
            // ---- exchange 3: X and dz1 -> layer-1 gradient
            put_tile(S.ea, 0, xb);
#pragma unroll
            for (int ft = 0; ft < kHT; ++ft) put_tile(S.eb, ft, d1[ft]);
            lds_barrier();
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const f32x4 bf = get_frag(S.ea, 0, bb);
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const f32x4 af = get_frag(S.eb, 2 * w + x, bb);
                    g.gb1[x] += (af[0] + af[1]) + (af[2] + af[3]);
                    g.g1[x] = mfma4(g.g1[x], af, bf);
                }
            }
            PPO_T(6);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            g.gb1[x] += __shfl_xor(g.gb1[x], 16, 64); g.gb1[x] += __shfl_xor(g.gb1[x], 32, 64);
            g.gb2[x] += __shfl_xor(g.gb2[x], 16, 64); g.gb2[x] += __shfl_xor(g.gb2[x], 32, 64);
        }
        g.gb3 += __shfl_xor(g.gb3, 16, 64); g.gb3 += __shfl_xor(g.gb3, 32, 64);
    }

    PPO_T(3);
    // =========================================================== clip_grad_norm_ over the whole critic net, Adam, soft update
    float ss = 0.f;
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        const HeadGrad& g = G[hd];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt)
                ss += (g.g2[x][kt][0] * g.g2[x][kt][0] + g.g2[x][kt][1] * g.g2[x][kt][1]) + (g.g2[x][kt][2] * g.g2[x][kt][2] + g.g2[x][kt][3] * g.g2[x][kt][3]);
            ss += (g.g1[x][0] * g.g1[x][0] + g.g1[x][1] * g.g1[x][1]) + (g.g1[x][2] * g.g1[x][2] + g.g1[x][3] * g.g1[x][3]);
            ss += (g.g3[x][0] * g.g3[x][0] + g.g3[x][1] * g.g3[x][1]) + (g.g3[x][2] * g.g3[x][2] + g.g3[x][3] * g.g3[x][3]);
            if (q == 0) ss += g.gb1[x] * g.gb1[x] + g.gb2[x] * g.gb2[x];
        }
        if (w == 0 && q == 0) ss += g.gb3 * g.gb3;
    }
    ss = wave_sum(ss);
    const float lsum = wave_sum(lossp);
    lds_barrier();
    if (l == 0) { S.red[w] = ss; S.red[8 + w] = lsum; }
    lds_barrier();
    const float total = sqrtf(((S.red[0] + S.red[1]) + S.red[2]) + S.red[3]);
    const float loss = ((S.red[8] + S.red[9]) + S.red[10]) + S.red[11];
    float coef = 1.f;
    if (a.clip_norm > 0.f) coef = fminf(a.clip_norm / (total + 1e-6f), 1.f);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t = steps[1] + 1;
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    const float step = (float)((double)a.critic_lr / bc1), inv_bc2s = 1.f / (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
    const bool soft = a.do_actor != 0;                                 // TD3: targets move with the delayed policy step (TD3.py:224-233)
    // Clip + Adam + soft update run over the parameter arrays LINEARLY (thread t takes the float4s t, t + 256, ...: every
    // wave-instruction moves 1 KB of contiguous theta / m / v / target), as the streaming Adam kernels do.  The owner lanes'
    // accumulators are in MFMA layout — 16 input rows x 64 bytes per instruction if they went to global memory directly
    // (measured: 187 k cycles per learner in this phase) — so they are transposed through the free exchange buffers first:
    // row-major Wk[in][out] images, 16-byte slots XOR-swizzled with the row so that the owners' ds_write_b128 and the
    // linear ds_read_b128 are both conflict-free.
    lds_f GB = S.ea;                                                   // ea and eb are adjacent: 16384 floats
    // loads of a batch first, then the updates and stores: the compiler cannot prove the four arrays distinct and waits for
    // every store before the next load, so a load -> store -> load chain per float4 is one HBM round trip each (138 k cycles
    // per learner for 42 of them)
    struct AdamIn { f32x4 th, mm, vv, tg; };
    auto adam_load = [&](int o) {
        AdamIn X;
        X.th = ld4((g_cf)(thC + o)); X.mm = ld4((g_cf)(mC + o)); X.vv = ld4((g_cf)(vC + o));
        X.tg = soft ? ld4((g_cf)(tgCw + o)) : f32x4{0.f, 0.f, 0.f, 0.f};
        return X;
    };
    auto adam4 = [&](int o, const f32x4& gr, const AdamIn& in) {       // one float4 of the net at offset o, gradient gr
        f32x4 th = in.th, mm = in.mm, vv = in.vv, tg = in.tg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gi = gr[r] * coef;
            if (a.critic_wd != 0.f) gi += a.critic_wd * th[r];
            float m1 = mm[r], v1 = vv[r];
            th[r] = adam_elem(th[r], gi, m1, v1, w1, w2, a.beta2, inv_bc2s, a.adam_eps, step);
            mm[r] = m1; vv[r] = v1;
            tg[r] = tg[r] * tk + th[r] * a.tau;
        }
        st4(thC + o, th); st4(mC + o, mm); st4(vC + o, vv);
        if (soft) st4(tgCw + o, tg);
    };
#pragma unroll
    for (int hd = 0; hd < NH; ++hd) {
        const HeadGrad& g = G[hd];
        const LayerDesc &L1 = NC.L[3 * hd], &L2 = NC.L[3 * hd + 1], &L3 = NC.L[3 * hd + 2];
        // ---- round A: the 128 x 128 layer
        lds_barrier();
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) {
                const int row = kt * 16 + i16, slot = (2 * w + x) * 4 + q;
                st4(GB + row * kHid + ((slot ^ (row & 7)) << 2), g.g2[x][kt]);
            }
        lds_barrier();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            AdamIn in[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) in[jj] = adam_load(L2.w_off + 4 * (tid + 256 * (8 * half + jj)));
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int f = tid + 256 * (8 * half + jj), row = f >> 5, slot = f & 31;
                adam4(L2.w_off + 4 * f, ld4((lds_cf)(GB + row * kHid + ((slot ^ (row & 7)) << 2))), in[jj]);
            }
        }
        // ---- round B: first layer (16 x 128), head (128 x 16), biases
        lds_barrier();
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int slot = (2 * w + x) * 4 + q;
            st4(GB + i16 * kHid + ((slot ^ (i16 & 7)) << 2), g.g1[x]);
            const int r3 = (2 * w + x) * 16 + i16;
            st4(GB + 2048 + r3 * 16 + ((q ^ ((r3 >> 1) & 3)) << 2), g.g3[x]);
            if (q == 0) { GB[4096 + (2 * w + x) * 16 + i16] = g.gb1[x]; GB[4224 + (2 * w + x) * 16 + i16] = g.gb2[x]; }
        }
        if (w == 0 && q == 0) GB[4352 + i16] = g.gb3;
        lds_barrier();
        {
            int o[5], ga[5];                                           // global offset / LDS address of this thread's five float4s (-1: none)
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
                const int f = tid + 256 * jj;
                if (f < 512) { const int row = f >> 5, slot = f & 31; o[jj] = L1.w_off + 4 * f; ga[jj] = row * kHid + ((slot ^ (row & 7)) << 2); }
                else if (f < 1024) { const int ff = f - 512, row = ff >> 2, slot = ff & 3; o[jj] = L3.w_off + 4 * ff; ga[jj] = 2048 + row * 16 + ((slot ^ ((row >> 1) & 3)) << 2); }
                else if (f < 1056) { o[jj] = L1.b_off + 4 * (f - 1024); ga[jj] = 4096 + 4 * (f - 1024); }
                else if (f < 1088) { o[jj] = L2.b_off + 4 * (f - 1056); ga[jj] = 4224 + 4 * (f - 1056); }
                else if (f < 1092) { o[jj] = L3.b_off + 4 * (f - 1088); ga[jj] = 4352 + 4 * (f - 1088); }
                else { o[jj] = -1; ga[jj] = 0; }
            }
            AdamIn in[5];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) in[jj] = adam_load(o[jj] >= 0 ? o[jj] : 0);
#pragma unroll
            for (int jj = 0; jj < 5; ++jj)
                if (o[jj] >= 0) adam4(o[jj], ld4((lds_cf)(GB + ga[jj])), in[jj]);
        }
    }
    PPO_T(7);
    PPO_TDUMP();
    if (tid == 0) {
        steps[1] = t;
        float* st = D.stats + (size_t)p * ST_COUNT;
        st[ST_CRITIC_LOSS] = loss * invB;
        st[ST_CRITIC_GNORM] = total;
    }
}

__global__ __launch_bounds__(256) void ac_critic_v2_twin_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ac_critic_v2_body<true>(*Dp, a, smem);
}
__global__ __launch_bounds__(256) void ac_critic_v2_single_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ac_critic_v2_body<false>(*Dp, a, smem);
}

}  // namespace frl
