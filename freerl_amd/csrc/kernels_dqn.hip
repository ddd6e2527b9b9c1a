// DQN gradient kernel (see kernels_update.hip for the launch chain and the decomposition).
#include <hip/hip_runtime.h>

#include "device/net.hpp"
#include "device/update_common.hpp"
#include "kernels.h"

namespace frl {

// ------------------------------------------------------------------------------------- DQN
// DQN.learn (DQN_file/DQN.py:104-118) for one row chunk: y = r + gamma*max_a Q_t(s',a)*(1-d);
// delta = 2(Q(s)[a] - y)/B on the taken action; backward -> partial slab.
__global__ __launch_bounds__(256, FRL_GRAD_WGS) void dqn_grad_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const UnitSlice us = unit_slice(ns);
    if (us.unit >= a.p_count) return;
    const int p = a.p0 + us.unit, sl = us.slice;
    const NetDesc& N = D.net[0];
    const RecordDesc& R = D.rec;
    const Lds S = carve(D, smem);
    const int rc = D.rc, B = a.batch, nl = N.n_layers;
    const ChunkRange cr = chunk_range(D, B, sl);
    const size_t base = (size_t)p * D.learner_stride + D.net_off[0];
    // noisy head: the three forwards read the effective parameter sets frl_learn has materialised (kernels_noisy.hip)
    g_cf eff = D.noisy ? as_global(D.theta_eff + (size_t)p * 3 * D.learner_stride + D.net_off[0]) : nullptr;
    g_cf theta_next = D.noisy ? eff : as_global(D.theta + base);                              // online net on s' (Double)
    g_cf target = D.noisy ? eff + D.learner_stride : as_global(D.target + base);              // target net on s'
    g_cf theta = D.noisy ? eff + 2 * (size_t)D.learner_stride : as_global(D.theta + base);    // online net on s (differentiated)
    g_f slab = as_global(D.slab + ((size_t)p * D.S + sl) * D.learner_stride + D.net_off[0]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    const int O = R.obs_dim[0], nA = D.n_discrete, npad = N.L[nl - 1].n_pad, k0pad = N.L[0].k_pad;
    const bool duel = D.dueling != 0;
    // Q(s, j) of row r out of the head in outb: plain, or Dueling's V + A_j - mean(A) with the head laid out [V ; A]
    auto q_of = [&](int r, int j, float mean) { lds_cf o = S.outb + r * S.op; return duel ? (o[0] + o[1 + j]) - mean : o[j]; };
    auto a_mean = [&](int r) {
        float m = 0.f;
        if (duel) { for (int j = 0; j < nA; ++j) m += S.outb[r * S.op + 1 + j]; m /= (float)nA; }
        return m;
    };

    // use_isw == 1: the reference's arithmetic — `(is_weight * td_error**2).mean()` multiplies a [B] by a [B,1] tensor
    // (DQN_with_tricks.py:277-278), i.e. mean(w) * mean(td^2): every row carries the MEAN weight.  2: per-row weights.
    float wbar = 1.f;
    if (a.use_isw == 1) {
        float ws = 0.f;
        for (int i = threadIdx.x; i < B; i += kWG) ws += D.isw[(size_t)p * D.batch_max + i];
        wbar = block_sum(ws, S.red) / (float)B;
    }
    float lossp = 0.f;
    for (int ck = cr.c0; ck < cr.c1; ++ck) {       // the row chunks of this workgroup, their gradients summed in its slab
    const bool first = (ck == cr.c0);
    const int gs = first ? (D.cps > 1 ? GS_STORE : GS_STREAM) : GS_ADD;
    const int r0 = ck * rc, nv = min(rc, B - r0);
    g_ci idx = as_global_i(D.idx + (size_t)p * D.n_agents * D.batch_max + r0);
    if (!first) lds_barrier();
    gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.nobs_off[0], O, 0);
    zero_cols(S.xin, S.xp, rc, O, k0pad);
    lds_barrier();
    if (a.double_dqn) {              // the online net picks the action, the target net values it (DQN_with_tricks.py:263-265)
        mlp_fwd(N, 0, nl, theta_next, S, ACT_NONE);
        for (int r = threadIdx.x; r < nv; r += kWG) {
            const float mean = a_mean(r);
            int best = 0;
            float mx = q_of(r, 0, mean);
            for (int j = 1; j < nA; ++j) {
                const float v = q_of(r, j, mean);
                if (v > mx) { mx = v; best = j; }     // first maximum, like argmax
            }
            S.abuf[r * S.ap] = (float)best;
        }
        lds_barrier();
    }
    mlp_fwd(N, 0, nl, target, S, ACT_NONE);
    for (int r = threadIdx.x; r < nv; r += kWG) {
        const float mean = a_mean(r);
        float mx;
        if (a.double_dqn) mx = q_of(r, (int)S.abuf[r * S.ap], mean);
        else {
            mx = q_of(r, 0, mean);
            for (int j = 1; j < nA; ++j) mx = fmaxf(mx, q_of(r, j, mean));
        }
        g_cf rec = ring + (size_t)idx[r] * R.stride;
        S.y[r] = rec[R.rew_off] + a.gamma * mx * (1.f - rec[R.done_off]);
    }
    lds_barrier();
    gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.obs_off[0], O, 0);
    zero_cols(S.xin, S.xp, rc, O, k0pad);
    lds_barrier();
    mlp_fwd(N, 0, nl, theta, S, ACT_NONE);
    g_cf isw = as_global(D.isw + (size_t)p * D.batch_max + r0);
    g_f tde = as_global(D.td_err + (size_t)p * D.batch_max + r0);
    // head delta, one thread per row: d = 2 w (Q(s,a) - y) / B on the taken action; through Dueling's recombination
    // dV = d, dA_j = d (delta_ja - 1/nA)
    for (int r = threadIdx.x; r < rc; r += kWG) {
        float d = 0.f;
        int ar = 0;
        if (r < nv) {
            ar = (int)ring[(size_t)idx[r] * R.stride + R.act_off[0]];             // actions.long() (DQN.py:114)
            const float diff = q_of(r, ar, a_mean(r)) - S.y[r];
            const float w = a.use_isw == 2 ? isw[r] : wbar;
            float lrow, grow;
            td_loss_row(a, diff, lrow, grow);
            d = w * grow / (float)B;
            lossp += w * lrow;
            tde[r] = diff;
        }
        lds_f o = S.outb + r * S.op;
        for (int j = 0; j < npad; ++j) {
            float v = 0.f;
            if (duel) { if (j == 0) v = d; else if (j <= nA) v = d * ((j - 1 == ar ? 1.f : 0.f) - 1.f / (float)nA); }
            else if (j == ar) v = d;
            o[j] = v;
        }
    }
    lds_barrier();
    mlp_bwd(N, 0, nl, theta, slab, S, gs, false, 0, 0);
    }
    const float ls = block_sum(lossp, S.red);
    if (threadIdx.x == 0) D.part[((size_t)p * D.n_agents * D.S + sl) * 4] = ls;
}
}  // namespace frl
