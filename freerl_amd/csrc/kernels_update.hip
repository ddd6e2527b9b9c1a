// Replay-sample + batched-update kernels of the off-policy learners (DQN, DDPG, TD3, SAC, MADDPG).
//
// One learn() of every resident learner = a short chain of launches:
//     [draw]  -> grad(critic) -> adam(critic) [-> grad(actor) -> adam(actor)] [-> soft update]
// replacing, per learner, the reference's ~60 (DQN) to ~700 (MADDPG) eager ATen launches per
// learn() and the 5 H2D copies of Buffer.sample (TD3_file/Buffer.py:50-55).
//
// Decomposition (chosen from rocprofv3 counters, profiles/README.md): the batch of one
// (learner, agent) is split into row chunks of `rc` rows; ONE workgroup owns ONE chunk:
// record gather from the HBM ring -> forward passes -> deltas -> backward (all MFMA), its weight
// gradients written ONCE to its partial slab.  The chunks of a learner are placed on the same
// XCD (block -> XCD is id % 8), so the weights they share are served by that XCD's L2 instead
// of being re-fetched per chunk.  `adam_kernel` then sums the slabs in a fixed order
// (deterministic), applies clip_grad_norm_, Adam and the soft target update while streaming
// theta/m/v once.  P learners (independent seeds) x chunks fill the 256 CUs; P = 1 still
// spreads one learner's batch over batch/rc CUs.
#include <hip/hip_runtime.h>

#include "device/net.hpp"

#ifndef FRL_GRAD_WGS
#define FRL_GRAD_WGS 2      // gradient-kernel workgroups per CU the register budget is sized for (frl_create picks rc to match)
#endif

namespace frl {

namespace {

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;
constexpr float kLog2 = 0.69314718055994530942f;

__device__ __forceinline__ float softplus_t(float x) {     // F.softplus (beta 1, threshold 20)
    return x > 20.f ? x : log1pf(expf(x));
}

__device__ __forceinline__ Lds carve(const EngineDesc& D, float* smem) {
    return carve_lds(smem, D.rc, D.hidden, D.lds_kin_pad, D.lds_out_pad, D.lds_batch_pad, D.lds_act_pad);
}

// block id -> (unit, slice): the `ns` row chunks of unit u = learner*n_agents + agent sit on
// blocks {8*ns*g + x + 8s}, i.e. all on XCD x and adjacent in dispatch order.
struct UnitSlice { int unit, slice; };
__device__ __forceinline__ UnitSlice unit_slice(int ns) {
    const int group = 8 * ns, g = blockIdx.x / group, l = blockIdx.x - g * group;
    return UnitSlice{g * 8 + (l & 7), l >> 3};
}

// A workgroup owns `cps` consecutive row chunks of its unit (frl_create picks cps so that one round of workgroups fills
// the chip): chunk 0 stores its weight gradients in the workgroup's slab, the others add to them — one slab per
// workgroup instead of one per chunk for reduce_kernel to stream.
struct ChunkRange { int c0, c1; };
__device__ __forceinline__ ChunkRange chunk_range(const EngineDesc& D, int batch, int slab) {
    const int nchunks = (batch + D.rc - 1) / D.rc, c0 = slab * D.cps;
    return ChunkRange{c0, min(c0 + D.cps, nchunks)};
}

}  // namespace

// ----------------------------------------------------------------------------------- draw
// Device-side sampling: `batch` distinct rows per (learner, agent) (np.random.choice(size, B,
// replace=False), DQN.py:97) and the N(0,1) draws of TD3.py:197 / SAC.py:227,244.
__global__ __launch_bounds__(256) void draw_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int want_noise) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int n = D.n_agents, p = a.p0 + blockIdx.x / n, ag = blockIdx.x % n, B = a.batch;
    g_i idx = (g_i)(D.idx + ((size_t)p * n + ag) * D.batch_max);
    const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
    draw_indices(idx, (FRL_LDS int*)smem, B, a.size, a.rng_counter, (unsigned)ag, key);
    if (want_noise) {
        const int am = D.act_max, NS = D.noise_sets;
        for (int s2 = 0; s2 < NS; s2 += 2) {          // two sets per Philox draw
            g_f noise0 = as_global(D.noise + (((size_t)p * n + ag) * NS + s2) * D.batch_max * am);
            g_f noise1 = noise0 + (size_t)D.batch_max * am;
            for (int e = threadIdx.x; e < B * am; e += kWG) {
                float n0, n1;
                normal2(philox4x32_10(a.rng_counter, 0x4000u + (unsigned)ag + 0x100u * (unsigned)s2, (unsigned)e, key), n0, n1);
                noise0[e] = n0;
                if (s2 + 1 < NS) noise1[e] = n1;
            }
        }
    }
}

// ----------------------------------------------------------------------------- Batch_ObsNorm
// RunningMeanStd_batch_size.update (PPO_file/normalization.py:61-71; SAC.py:215, DDPG.py:191) with
// the batch mean of the sampled observations: n += 1; first call: mean = std = xbar; afterwards a
// Welford step on the batch means.  One workgroup per learner, before the gradient kernels.
__global__ __launch_bounds__(256) void obsnorm_kernel(const EngineDesc* __restrict__ Dp, int batch, int all_rows, int p0) {
    const EngineDesc& D = *Dp;
    const int p = p0 + blockIdx.x, n = D.n_agents, W = D.obsnorm_w;
    const RecordDesc& R = D.rec;
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_f base = as_global(D.obsnorm + (size_t)p * n * n * W);
    // version i = the statistics after updating agent i's sample() (MADDPG.py:189-197: one index draw per updating agent,
    // every agent's running mean/std updated with the batch mean of ITS observations over those rows); n = 1: in place
    for (int i = 0; i < n; ++i) {
        g_ci idx = as_global_i(D.idx + ((size_t)p * n + i) * D.batch_max);
        g_cf prev = base + (size_t)(i == 0 ? n - 1 : i - 1) * n * W;
        g_f cur = base + (size_t)i * n * W;
        for (int e = threadIdx.x; e < n * W; e += kWG) {
            const int j = e / W, c = e - j * W - 1;              // c = -1: the count slot
            const int O = R.obs_dim[j];
            g_cf ps = prev + (size_t)j * W;
            g_f cs = cur + (size_t)j * W;
            const float cnt = ps[0] + 1.f;
            if (c < 0) { if (n > 1) cs[0] = cnt; continue; }
            if (c >= O) continue;
            float s = 0.f;
            for (int r = 0; r < batch; ++r) s += ring[(size_t)(all_rows ? r : idx[r]) * R.stride + R.obs_off[j] + c];
            const float xbar = s / (float)batch;
            if (cnt == 1.f) {
                cs[1 + c] = xbar;
                cs[1 + O + c] = ps[1 + O + c];
                cs[1 + 2 * O + c] = xbar;                      // the reference's first update sets std = x (normalization.py:63-65)
            } else {
                const float old = ps[1 + c];
                const float m2 = old + (xbar - old) / cnt;
                const float S2 = ps[1 + O + c] + (xbar - old) * (xbar - m2);
                cs[1 + c] = m2;
                cs[1 + O + c] = S2;
                cs[1 + 2 * O + c] = sqrtf(S2 / cnt);
            }
        }
        __syncthreads();
        if (n == 1 && threadIdx.x == 0) cur[0] = cur[0] + 1.f;     // in place: the count moves after every column has read it
        __syncthreads();
    }
}

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
            d = 2.f * w * diff / (float)B;
            lossp += w * diff * diff;
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
                const float diff = o[0] - S.y[r];
                d = 2.f * diff * invB;
                lossp += diff * diff;
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

// ------------------------------------------------------------------- reduce + clip + Adam
// Two bandwidth-bound launches over float4 lanes, `G` workgroups per (learner, agent) net:
//   reduce_kernel: g = sum over the row-chunk slabs in a fixed order (deterministic), write g,
//                  per-workgroup sum of squares -> gsq; workgroup 0 advances the Adam step count.
//   adam_kernel  : total norm from gsq -> clip_grad_norm_ coefficient -> torch-order Adam ->
//                  optional soft target update, streaming theta/m/v/target once.  Workgroup 0
//                  also publishes the losses and performs SAC's alpha step (SAC.py:154-169,257-260).
// which = 0: critic / Q-net, 1: actor.
struct AdamArgs {
    int which, ns, batch, soft, sac_alpha, G, p0;
    float lr, eps, beta1, beta2, wd, clip, tau, alpha_lr, target_entropy;
};

constexpr int kAdamVec = 8;      // float4 per thread per workgroup

__device__ __forceinline__ int adam_net_index(const EngineDesc& D, int which, int ag) {
    return (D.algo == ALGO_DQN) ? 0 : (which == 0 ? 2 * ag + 1 : 2 * ag);
}

// thread 0 of a unit's first Adam workgroup: losses of the step, SAC's alpha update (SAC.py:154-169,257-260)
__device__ __forceinline__ void adam_publish(const EngineDesc& D, const AdamArgs& a, int p, int ag, float total, int* steps) {
    const int n = D.n_agents;
    const float* pt = D.part + ((size_t)p * n + ag) * D.S * 4;
    float l0 = 0.f, l1 = 0.f;
    for (int k = 0; k < a.ns; ++k) { l0 += pt[4 * k]; l1 += pt[4 * k + 1]; }
    float* st = D.stats + ((size_t)p * n + ag) * ST_COUNT;
    const float invB = 1.f / (float)a.batch;
    st[a.which == 0 ? ST_CRITIC_LOSS : ST_ACTOR_LOSS] = l0 * invB;
    st[a.which == 0 ? ST_CRITIC_GNORM : ST_ACTOR_GNORM] = total;
    if (a.sac_alpha) {
        float* al = D.alpha + p * 4;
        const float alpha = al[3];
        const float ent_mean = l1 * invB;
        const float mean_term = ent_mean - a.target_entropy;
        const float gl = alpha * mean_term;            // d alpha_loss / d log_alpha
        const int ta = steps[kMaxNets] + 1;
        float mi = al[1], vi = al[2];
        mi = mi + (gl - mi) * (1.f - a.beta1);
        vi = vi * a.beta2 + ((1.f - a.beta2) * gl) * gl;
        const double b1 = 1.0 - powi_d((double)a.beta1, ta), b2 = 1.0 - powi_d((double)a.beta2, ta);
        const float denom = sqrtf(vi) / (float)sqrt(b2) + 1e-8f;
        al[0] = al[0] - (float)((double)a.alpha_lr / b1) * (mi / denom);
        al[1] = mi;
        al[2] = vi;
        al[3] = expf(al[0]);
        steps[kMaxNets] = ta;
        st[ST_ALPHA_LOSS] = alpha * mean_term;
        st[ST_ALPHA] = al[3];
        st[ST_ENTROPY] = ent_mean;
    }
}

__global__ __launch_bounds__(256) void reduce_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a) {
    __shared__ float red_s[8];
    lds_f red = (lds_f)red_s;
    const EngineDesc& D = *Dp;
    const int n = D.n_agents, wg = blockIdx.x % a.G;
    const int p = a.p0 + (blockIdx.x / a.G) / n, ag = (blockIdx.x / a.G) % n, unit = p * n + ag;
    const int net = adam_net_index(D, a.which, ag);
    const NetDesc& N = D.net[net];
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    const int n4 = N.size / 4;
    FRL_GLB f32x4* g = (FRL_GLB f32x4*)(D.grad + off);
    const FRL_GLB f32x4* slab = (const FRL_GLB f32x4*)(D.slab + (size_t)p * D.S * D.learner_stride + D.net_off[net]);
    const size_t ls4 = (size_t)D.learner_stride / 4;
    float ss = 0.f;
    const int base = wg * (kWG * kAdamVec);
#pragma unroll
    for (int j = 0; j < kAdamVec; ++j) {
        const int i = base + j * kWG + threadIdx.x;
        if (i < n4) {
            f32x4 s = slab[i];
            for (int k = 1; k < a.ns; ++k) s += slab[(size_t)k * ls4 + i];
            g[i] = s;
            ss += s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
        }
    }
    const float tot = block_sum(ss, red);
    if (threadIdx.x == 0) {
        D.gsq[(size_t)unit * D.Gmax + wg] = tot;
        if (wg == 0) D.steps[(size_t)p * (kMaxNets + 1) + net] += 1;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a) {
    const EngineDesc& D = *Dp;
    const int n = D.n_agents, wg = blockIdx.x % a.G;
    const int p = a.p0 + (blockIdx.x / a.G) / n, ag = (blockIdx.x / a.G) % n, unit = p * n + ag;
    const int net = adam_net_index(D, a.which, ag);
    const NetDesc& N = D.net[net];
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    const int n4 = N.size / 4, Gn = (n4 + kWG * kAdamVec - 1) / (kWG * kAdamVec);
    float ss = 0.f;
    for (int k = 0; k < Gn; ++k) ss += D.gsq[(size_t)unit * D.Gmax + k];      // same order in every workgroup
    const float total = sqrtf(ss);
    float coef = 1.f;
    if (a.clip > 0.f) coef = fminf(a.clip / (total + 1e-6f), 1.f);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t = steps[net];                                                   // advanced by reduce_kernel
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    const float step = (float)((double)a.lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
    const FRL_GLB f32x4* g = (const FRL_GLB f32x4*)(D.grad + off);
    FRL_GLB f32x4* th = (FRL_GLB f32x4*)(D.theta + off);
    FRL_GLB f32x4* m = (FRL_GLB f32x4*)(D.m + off);
    FRL_GLB f32x4* v = (FRL_GLB f32x4*)(D.v + off);
    FRL_GLB f32x4* tg = (FRL_GLB f32x4*)(D.target + off);
    const int base = wg * (kWG * kAdamVec);
#pragma unroll
    for (int j = 0; j < kAdamVec; ++j) {
        const int i = base + j * kWG + threadIdx.x;
        if (i < n4) {
            f32x4 gi = g[i] * coef, thi = th[i], mi = m[i], vi = v[i];
            if (a.wd != 0.f) gi += a.wd * thi;
            mi = mi + (gi - mi) * w1;
            vi = vi * a.beta2 + (w2 * gi) * gi;
            f32x4 denom;
            denom.x = sqrtf(vi.x) / bc2s + a.eps; denom.y = sqrtf(vi.y) / bc2s + a.eps;
            denom.z = sqrtf(vi.z) / bc2s + a.eps; denom.w = sqrtf(vi.w) / bc2s + a.eps;
            thi = thi - step * (mi / denom);
            m[i] = mi; v[i] = vi; th[i] = thi;
            if (a.soft) tg[i] = tg[i] * tk + thi * a.tau;
        }
    }
    if (wg == 0 && threadIdx.x == 0) adam_publish(D, a, p, ag, total, steps);
}

// One launch instead of reduce_kernel + adam_kernel when a net's gradient fits in the registers of ONE 1024-thread
// workgroup (<= kFusedVec float4 per thread: every 128-wide net of the reference): the slabs are summed into registers,
// the norm is a block reduction, and the Adam pass takes its gradient from the registers — the reduced gradient is
// never written, the norm partials never leave the workgroup.  11 -> 9 floats of traffic per parameter at 2 slabs.
constexpr int kFusedThreads = 1024, kFusedVec = 12;
__global__ __launch_bounds__(kFusedThreads) void adam_fused_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a) {
    __shared__ float red[kFusedThreads / 64];
    const EngineDesc& D = *Dp;
    const int n = D.n_agents;
    const int p = a.p0 + blockIdx.x / n, ag = blockIdx.x % n;
    const int net = adam_net_index(D, a.which, ag);
    const NetDesc& N = D.net[net];
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    const int n4 = N.size / 4;
    const FRL_GLB f32x4* slab = (const FRL_GLB f32x4*)(D.slab + (size_t)p * D.S * D.learner_stride + D.net_off[net]);
    const size_t ls4 = (size_t)D.learner_stride / 4;
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t = steps[net] + 1;                             // in flight with the slab loads
    // slab-major: all of a thread's loads of one slab in flight together (element-major, each element's slab sum was a
    // dependent chain of waits: 57 us for 64 learners); per element the slabs are still added in index order
    f32x4 g[kFusedVec];
#pragma unroll
    for (int j = 0; j < kFusedVec; ++j) {
        const int i = j * kFusedThreads + threadIdx.x;
        g[j] = (i < n4) ? slab[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int k = 1; k < a.ns; ++k) {
        const FRL_GLB f32x4* sk = slab + (size_t)k * ls4;
        f32x4 tmp[kFusedVec];
#pragma unroll
        for (int j = 0; j < kFusedVec; ++j) {
            const int i = j * kFusedThreads + threadIdx.x;
            tmp[j] = (i < n4) ? sk[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < kFusedVec; ++j) g[j] += tmp[j];
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < kFusedVec; ++j) ss += g[j].x * g[j].x + g[j].y * g[j].y + g[j].z * g[j].z + g[j].w * g[j].w;
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < kFusedThreads / 64; ++w) tot += red[w];
    const float total = sqrtf(tot);
    float coef = 1.f;
    if (a.clip > 0.f) coef = fminf(a.clip / (total + 1e-6f), 1.f);
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    const float step = (float)((double)a.lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
    FRL_GLB f32x4* th = (FRL_GLB f32x4*)(D.theta + off);
    FRL_GLB f32x4* m = (FRL_GLB f32x4*)(D.m + off);
    FRL_GLB f32x4* v = (FRL_GLB f32x4*)(D.v + off);
    FRL_GLB f32x4* tg = (FRL_GLB f32x4*)(D.target + off);
#pragma unroll
    for (int j = 0; j < kFusedVec; ++j) {
        const int i = j * kFusedThreads + threadIdx.x;
        if (i < n4) {
            f32x4 gi = g[j] * coef, thi = th[i], mi = m[i], vi = v[i];
            if (a.wd != 0.f) gi += a.wd * thi;
            mi = mi + (gi - mi) * w1;
            vi = vi * a.beta2 + (w2 * gi) * gi;
            f32x4 denom;
            denom.x = sqrtf(vi.x) / bc2s + a.eps; denom.y = sqrtf(vi.y) / bc2s + a.eps;
            denom.z = sqrtf(vi.z) / bc2s + a.eps; denom.w = sqrtf(vi.w) / bc2s + a.eps;
            thi = thi - step * (mi / denom);
            m[i] = mi; v[i] = vi; th[i] = thi;
            if (a.soft) tg[i] = tg[i] * tk + thi * a.tau;
        }
    }
    if (threadIdx.x == 0) {
        steps[net] = t;
        adam_publish(D, a, p, ag, total, steps);
    }
}

// MADDPG.update_target (MADDPG_simple.py:188-195): every agent's actor then critic.
__global__ __launch_bounds__(256) void soft_update_kernel(const EngineDesc* __restrict__ Dp, float tau, int p0) {
    const EngineDesc& D = *Dp;
    const int p = p0 + blockIdx.x / D.n_nets, net = blockIdx.x % D.n_nets;
    const size_t off = (size_t)p * D.learner_stride + D.net_off[net];
    soft_update_net(D.net[net].size, as_global(D.target + off), as_global(D.theta + off), tau);
}

}  // namespace frl
