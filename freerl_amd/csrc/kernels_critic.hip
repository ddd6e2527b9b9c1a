// Critic gradient kernel of DDPG / TD3 / SAC / MADDPG / MATD3 (see kernels_update.hip for the launch chain).
#include <hip/hip_runtime.h>

#include "device/net.hpp"
#include "device/update_common.hpp"
#include "kernels.h"

namespace frl {

// ------------------------------------------------------- DDPG / TD3 / SAC / MADDPG: critic
// TD target with the target nets, twin/single critic forward, MSE delta, backward -> slab.
// DDPG_simple.py:139-149, TD3.py:193-213, SAC.py:226-238, MADDPG_simple.py:169-176.
__global__ __launch_bounds__(256, FRL_GRAD_WGS) void ac_critic_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const UnitSlice us = unit_slice(ns);
    const int n = D.n_agents;
    if (us.unit >= a.p_count * n) return;
    const int p = a.p0 + us.unit / n, ag = us.unit % n, sl = us.slice;
    const RecordDesc& R = D.rec;
    const NetDesc& NC = D.net[2 * ag + 1];
    const Lds S = carve(D, smem);
    const int rc = D.rc, B = a.batch;
    const ChunkRange cr = chunk_range(D, B, sl);
    const bool sac = (D.algo == ALGO_SAC);
    const size_t lbase = (size_t)p * D.learner_stride;
    g_cf thC = as_global(D.theta + lbase + D.net_off[2 * ag + 1]);
    g_cf tgC = as_global(D.target + lbase + D.net_off[2 * ag + 1]);
    g_f slab = as_global(D.slab + ((size_t)p * D.S + sl) * D.learner_stride + D.net_off[2 * ag + 1]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    const int am = D.act_max;
    const int heads = NC.heads, ql = NC.n_layers / heads;
    const int OT = R.obs_total, AT = R.act_total, kc0 = NC.L[0].k_pad;
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const float invB = 1.f / (float)B;
    const bool direct = (n == 1) && kc0 <= D.net[0].L[0].k_pad;      // a' can be written into the critic input in place
    // Batch_ObsNorm statistics as of THIS agent's sample() (version ag), one block of obsnorm_w per agent
    g_cf bn = D.obs_norm_on ? as_global(D.obsnorm + ((size_t)p * n + ag) * n * D.obsnorm_w) : nullptr;
    auto normalize_joint = [&](int nvalid) {      // every agent's segment of a joint [obs_0 | obs_1 | ...] block in xin[:, 0:OT)
        for (int j = 0; j < n; ++j)
            normalize_cols(S.xin, S.xp, nvalid, R.obs_off[j] - R.obs_off[0], R.obs_dim[j], bn + (size_t)j * D.obsnorm_w, R.obs_dim[j]);
    };
    FRL_PHASE_INIT(S);
#ifdef FRL_EXP_TOUCH      // experiment: pull every weight line this kernel will read into L2 up front
    {
        float acc_t = 0.f;
        g_cf spans[3] = {as_global(D.target + lbase + D.net_off[2 * ag]), tgC, thC};
        const int sizes[3] = {D.net[2 * ag].size, NC.size, NC.size};
        for (int sidx = 0; sidx < 3; ++sidx)
            for (int o = threadIdx.x * 32; o < sizes[sidx]; o += kWG * 32) acc_t += spans[sidx][o];
        if (acc_t == 123456.789f) S.red[0] = acc_t;
    }
#endif

    float lossp = 0.f;
    for (int ck = cr.c0; ck < cr.c1; ++ck) {       // the row chunks of this workgroup, their gradients summed in its slab
    const bool first = (ck == cr.c0);
    const int gs = first ? (D.cps > 1 ? GS_STORE : GS_STREAM) : GS_ADD;
    const int r0 = ck * rc, nv = min(rc, B - r0);
    g_ci idx = as_global_i(D.idx + ((size_t)p * n + ag) * D.batch_max + r0);
    g_cf noise_u = as_global(D.noise + ((size_t)p * n + ag) * D.noise_sets * D.batch_max * am + (size_t)r0 * am);
    if (!first) lds_barrier();
    // ---- a' = actor_target_j(s'_j) for every agent j (MADDPG_simple.py:155; n = 1 otherwise)
    float lp_next = 0.f;                            // SAC: log pi(a'|s') of row threadIdx.x
    for (int j = 0; j < n; ++j) {
        const NetDesc& NJ = D.net[2 * j];
        g_cf tgJ = as_global(D.target + lbase + D.net_off[2 * j]);
        const int Oj = R.obs_dim[j], Aj = R.act_dim[j], cj = R.act_off[j] - R.act_off[0];
        g_cf noise0 = noise_u + (size_t)j * D.batch_max * am;     // set j (n = 1: set 0); MATD3_simple.py:199-201
        gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.nobs_off[j], Oj, 0);
        zero_cols(S.xin, S.xp, rc, Oj, NJ.L[0].k_pad);
        if (bn) { lds_barrier(); normalize_cols(S.xin, S.xp, nv, 0, Oj, bn + (size_t)j * D.obsnorm_w, Oj); }
        FRL_PHASE(S);
        // a'_j per row in the finalize phase of the target actor; single agent: straight into the critic's input row
        // (xin[:, 0:O) still holds the normalised next_obs, the columns past the action are already zero)
        mlp_fwd_rows(NJ, 0, NJ.n_layers, tgJ, S, sac ? ACT_NONE : ACT_TANH, [&](int r) {
            if (sac) {                              // SAC.py:70-97 on actor_target (SAC.py:227)
                float lp = 0.f;
                for (int c = 0; c < Aj; ++c) {
                    const float mean = S.outb[r * S.op + c];
                    const float ls = fminf(fmaxf(tgJ[NJ.extra_off + c], -20.f), 2.f);
                    const float sd = expf(ls);
                    const float eps = (r < nv) ? noise0[(size_t)r * am + c] : 0.f;
                    const float u = mean + sd * eps;
                    const float du = u - mean;
                    lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                    lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                    S.abuf[r * S.ap + cj + c] = tanhf(u);
                }
                lp_next = lp;
            } else {
                for (int c = 0; c < Aj; ++c) {
                    float v = S.outb[r * S.op + c];
                    if (a.use_policy_noise && r < nv) {   // TD3.py:196-198
                        float nz = a.policy_noise_scale * (noise0[(size_t)r * am + c] * a.policy_noise);
                        nz = fminf(fmaxf(nz, -a.noise_clip), a.noise_clip);
                        v = fminf(fmaxf(v * a.max_action + nz, -a.max_action), a.max_action) / a.max_action;
                    }
                    S.abuf[r * S.ap + cj + c] = v;
                }
            }
            if (direct)
                for (int c = 0; c < Aj; ++c) S.xin[r * S.xp + OT + c] = S.abuf[r * S.ap + c];
        });
    }
    // ---- centralised target critic on [next_obs_all | a'_all]
    // single agent: xin[:, 0:O) still holds the (normalised) next_obs the target actor has just read
    if (!direct) {
        if (n > 1) gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.nobs_off[0], OT, 0);
        for (int e = threadIdx.x; e < rc * AT; e += kWG) {
            const int r = e / AT, c = e - r * AT;
            S.xin[r * S.xp + OT + c] = S.abuf[r * S.ap + c];
        }
        zero_cols(S.xin, S.xp, rc, OT + AT, kc0);
        if (bn && n > 1) { lds_barrier(); normalize_joint(nv); }
        FRL_PHASE(S);
    }
    float q = 0.f;
    if (twin_target_fusable(NC)) {
        twin_target_fwd(NC, tgC, S);
        if (threadIdx.x < rc) q = fminf(twin_target_q(NC, tgC, S, threadIdx.x, 0), twin_target_q(NC, tgC, S, threadIdx.x, 1));
    } else {
        mlp_fwd(NC, 0, ql, tgC, S, ACT_NONE);
        if (threadIdx.x < rc) q = S.outb[threadIdx.x * S.op];
        if (heads == 2) {
            FRL_PHASE(S);
            mlp_fwd(NC, ql, ql, tgC, S, ACT_NONE);
            if (threadIdx.x < rc) q = fminf(q, S.outb[threadIdx.x * S.op]);
        }
    }
    if (threadIdx.x < nv) {
        g_cf rec = ring + (size_t)idx[threadIdx.x] * R.stride;
        const float rew = rec[R.rew_off + ag], done = rec[R.done_off + ag];
        S.y[threadIdx.x] = sac ? rew + a.gamma * (1.f - done) * (q + alpha * (-lp_next))
                               : rew + a.gamma * q * (1.f - done);
    }
    FRL_PHASE(S);

    // ---- critic heads: forward, MSE delta, backward
    for (int h = 0; h < heads; ++h) {
        if (h == 0) {           // the second head reads the same [obs | act] rows: nothing in between writes xin
            gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.obs_off[0], OT + AT, 0);
            zero_cols(S.xin, S.xp, rc, OT + AT, kc0);
            if (bn) { lds_barrier(); normalize_joint(nv); }
        }
        FRL_PHASE(S);
        const int npad = NC.L[h * ql + ql - 1].n_pad;
        mlp_fwd_rows(NC, h * ql, ql, thC, S, ACT_NONE, [&](int r) {     // MSE delta of row r in the finalize phase
            lds_f o = S.outb + r * S.op;
            float d = 0.f;
            if (r < nv) {
                float lrow, grow;
                td_loss_row(a, o[0] - S.y[r], lrow, grow);
                d = grow * invB;
                lossp += lrow;
            }
            o[0] = d;
            for (int c = 1; c < npad; ++c) o[c] = 0.f;
        });
        mlp_bwd(NC, h * ql, ql, thC, slab, S, gs, false, 0, 0);
    }
    }
    FRL_PHASE_DUMP(S, 0);
    const float ls = block_sum(lossp, S.red);
    if (threadIdx.x == 0) D.part[(((size_t)p * n + ag) * D.S + sl) * 4] = ls;
}
}  // namespace frl
