// Actor gradient kernel of DDPG / TD3 / SAC / MADDPG / MATD3 (see kernels_update.hip for the launch chain).
#include <hip/hip_runtime.h>

#include "device/net.hpp"
#include "device/update_common.hpp"
#include "kernels.h"

namespace frl {

// -------------------------------------------------------- DDPG / TD3 / SAC / MADDPG: actor
// a = actor(s); Q(s, a) through the (already updated, frozen) critic; dQ/da; actor backward.
// DDPG_simple.py:151-154, TD3.py:224-231, SAC.py:244-252, MADDPG_simple.py:178-183.
__global__ __launch_bounds__(256, FRL_GRAD_WGS) void ac_actor_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const UnitSlice us = unit_slice(ns);
    const int n = D.n_agents;
    if (us.unit >= a.p_count * n) return;
    const int p = a.p0 + us.unit / n, ag = us.unit % n, sl = us.slice;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[2 * ag];
    const NetDesc& NC = D.net[2 * ag + 1];
    const Lds S = carve(D, smem);
    const int rc = D.rc, B = a.batch;
    const ChunkRange cr = chunk_range(D, B, sl);
    const bool sac = (D.algo == ALGO_SAC);
    const size_t lbase = (size_t)p * D.learner_stride;
    g_cf thA = as_global(D.theta + lbase + D.net_off[2 * ag]);
    g_cf thC = as_global(D.theta + lbase + D.net_off[2 * ag + 1]);
    g_f slab = as_global(D.slab + ((size_t)p * D.S + sl) * D.learner_stride + D.net_off[2 * ag]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    const int am = D.act_max;
    const int heads = NC.heads, ql = NC.n_layers / heads;
    const int OT = R.obs_total, AT = R.act_total, kc0 = NC.L[0].k_pad;
    const int Oa = R.obs_dim[ag], Aa = R.act_dim[ag], acol = R.act_off[ag] - R.act_off[0];
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const float invB = 1.f / (float)B;
    const int ct0 = (OT + acol) / 16, ct1 = (OT + acol + Aa + 15) / 16;
    const int nq = sac ? heads : 1;                 // SAC: mean of the twins (SAC.py:250); TD3: Q1 only (TD3.py:227)
    const bool direct = (n == 1) && kc0 <= NA.L[0].k_pad;   // the action can be written into the critic's input row in place
    const float dq = sac ? -0.5f * invB : -invB;
    g_cf bn = D.obs_norm_on ? as_global(D.obsnorm + ((size_t)p * n + ag) * n * D.obsnorm_w) : nullptr;
    g_cf bn_own = bn ? bn + (size_t)ag * D.obsnorm_w : nullptr;
    auto normalize_joint = [&](int nvalid) {
        for (int j = 0; j < n; ++j)
            normalize_cols(S.xin, S.xp, nvalid, R.obs_off[j] - R.obs_off[0], R.obs_dim[j], bn + (size_t)j * D.obsnorm_w, R.obs_dim[j]);
    };

    FRL_PHASE_INIT(S);
    float alossp = 0.f, entp = 0.f;
    for (int ck = cr.c0; ck < cr.c1; ++ck) {       // the row chunks of this workgroup, their gradients summed in its slab
    const bool first = (ck == cr.c0);
    const int gs = first ? (D.cps > 1 ? GS_STORE : GS_STREAM) : GS_ADD;
    const int r0 = ck * rc, nv = min(rc, B - r0);
    g_ci idx = as_global_i(D.idx + ((size_t)p * n + ag) * D.batch_max + r0);
    g_cf noise1 = as_global(D.noise + (((size_t)p * n + ag) * D.noise_sets + 1) * D.batch_max * am + (size_t)r0 * am);
    if (!first) lds_barrier();
    // -- a = actor(obs)
    gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.obs_off[ag], Oa, 0);
    zero_cols(S.xin, S.xp, rc, Oa, NA.L[0].k_pad);
    if (bn) { lds_barrier(); normalize_cols(S.xin, S.xp, nv, 0, Oa, bn_own, Oa); }
    FRL_PHASE(S);
    // park the actor's hidden activations in HBM: the critic pass below reuses h1 / h2, the actor's backward needs them
    // again, and a second actor forward cost 16 % of this kernel (tools/phase_timing.py actor)
    const int spill_n4 = 2 * rc * S.hp / 4;                 // h1 and h2 are adjacent in LDS
    FRL_GLB f32x4* spill = (FRL_GLB f32x4*)(D.act_spill + (((size_t)p * n + ag) * D.S + sl) * 2 * rc * S.hp);
    float lp = 0.f;
    // the action of row r in the finalize phase of the actor; single agent: straight into the critic's input row (xin[:, 0:O)
    // still holds the normalised obs, the columns past the action are zero), which saves the [s|a] gather of the first head
    mlp_fwd_rows(NA, 0, NA.n_layers, thA, S, sac ? ACT_NONE : ACT_TANH, [&](int r) {
        for (int c = 0; c < Aa; ++c) {
            float av = S.outb[r * S.op + c];
            if (sac) {
                const float ls = fminf(fmaxf(thA[NA.extra_off + c], -20.f), 2.f);
                const float sd = expf(ls);
                const float eps = (r < nv) ? noise1[(size_t)r * am + c] : 0.f;
                const float u = av + sd * eps;
                const float du = u - av;
                lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                av = tanhf(u);
            }
            S.abuf[r * S.ap + c] = av;
            S.dabuf[r * S.ap + c] = 0.f;
            if (direct) S.xin[r * S.xp + OT + c] = av;
        }
    }, [&]() {
        for (int i = threadIdx.x; i < spill_n4; i += kWG) spill[i] = ld4((lds_cf)(S.h1 + 4 * i));
    });
    // -- dQ/da through the critic head(s)
    float qsum = 0.f;
    for (int h = 0; h < nq; ++h) {
        if (!(direct && h == 0)) {
            gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.obs_off[0], OT + AT, 0);
            zero_cols(S.xin, S.xp, rc, OT + AT, kc0);
            if (bn) { lds_barrier(); normalize_joint(nv); }
            FRL_PHASE(S);
            for (int e = threadIdx.x; e < rc * Aa; e += kWG) {
                const int r = e / Aa, c = e - r * Aa;
                S.xin[r * S.xp + OT + acol + c] = S.abuf[r * S.ap + c];
            }
            FRL_PHASE(S);
        }
        const int npad = NC.L[h * ql + ql - 1].n_pad;
        mlp_fwd_rows(NC, h * ql, ql, thC, S, ACT_NONE, [&](int r) {      // Q of row r -> loss sum; its delta for the dX-only backward
            lds_f o = S.outb + r * S.op;
            if (r < nv) qsum += o[0];
            o[0] = (r < nv) ? dq : 0.f;
            for (int c = 1; c < npad; ++c) o[c] = 0.f;
        });
        mlp_bwd(NC, h * ql, ql, thC, nullptr, S, GS_ADD, true, ct0, ct1);
        for (int e = threadIdx.x; e < rc * Aa; e += kWG) {
            const int r = e / Aa, c = e - r * Aa;
            S.dabuf[r * S.ap + c] += S.xin[r * S.xp + OT + acol + c];
        }
        FRL_PHASE(S);
    }
    if (threadIdx.x < nv) {
        if (sac) {
            alossp += -(qsum * 0.5f) - alpha * (-lp);     // (-Q_pi - alpha*entropy), SAC.py:251
            entp += -lp;
        } else {
            alossp += -qsum;
        }
    }
    // -- the actor's activations back from HBM (same thread, same addresses as the spill), its input back in xin
    for (int i = threadIdx.x; i < spill_n4; i += kWG) st4(S.h1 + 4 * i, spill[i]);
    gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.obs_off[ag], Oa, 0);    // (the critic's dX1 landed on xin)
    zero_cols(S.xin, S.xp, rc, Oa, NA.L[0].k_pad);
    if (bn) { lds_barrier(); normalize_cols(S.xin, S.xp, nv, 0, Oa, bn_own, Oa); }
    // the head delta in the same phase: it reads dabuf / abuf and writes outb, none of which the reload above touches
    const int napad = NA.L[NA.n_layers - 1].n_pad;
    for (int e = threadIdx.x; e < rc * napad; e += kWG) {
        const int r = e / napad, c = e - r * napad;
        float d = 0.f;
        if (r < nv && c < Aa) {
            if (sac) {
                const float av = S.abuf[r * S.ap + c];
                d = S.dabuf[r * S.ap + c] * (1.f - av * av) + (alpha * invB) * (2.f * av);
                const float ls = fminf(fmaxf(thA[NA.extra_off + c], -20.f), 2.f);
                S.dabuf[r * S.ap + c] = d * expf(ls) * noise1[(size_t)r * am + c] - alpha * invB;   // d/d log_std
            } else {
                const float av = S.abuf[r * S.ap + c];      // the actor's tanh output (kept from the forward)
                d = S.dabuf[r * S.ap + c] * (1.f - av * av);
            }
        } else if (sac && c < Aa) {
            S.dabuf[r * S.ap + c] = 0.f;
        }
        S.outb[r * S.op + c] = d;
    }
    FRL_PHASE(S);
    if (sac && threadIdx.x < Aa) {
        float gls = 0.f;
        for (int r = 0; r < rc; ++r) gls += S.dabuf[r * S.ap + threadIdx.x];
        const float raw = thA[NA.extra_off + threadIdx.x];
        const float gl = (raw >= -20.f && raw <= 2.f) ? gls : 0.f;
        slab[NA.extra_off + threadIdx.x] = first ? gl : slab[NA.extra_off + threadIdx.x] + gl;
    }
    mlp_bwd(NA, 0, NA.n_layers, thA, slab, S, gs, false, 0, 0);
    }
    FRL_PHASE_DUMP(S, 1);
    const float la = block_sum(alossp, S.red);
    const float le = sac ? block_sum(entp, S.red) : 0.f;
    if (threadIdx.x == 0) {
        float* pt = D.part + (((size_t)p * n + ag) * D.S + sl) * 4;
        pt[0] = la;
        pt[1] = le;
    }
}
}  // namespace frl
