// DDPG / TD3 / SAC / MADDPG / MATD3 update of a SINGLE learner (or a handful) with a WIDE first layer on sixteen workgroups per
// (learner, agent) unit — 64 for MADDPG's batches of 1024 — (device/solo_wide.hpp): the shapes of BASELINE.json's config 4 (SAC at
// Humanoid-v4's dims, 376 + 17 input columns, 17 actions), of the MuJoCo-sized TD3 / DDPG runs and of config 5 (MADDPG_simple on
// simple_spread: three agents, centralised critics on 54 + 15 columns), one `learn()` per env step as the reference drives them
// (SAC.py:519-576, TD3.py:403-450, MADDPG_simple.py:165-186).
// The critic stage — target actions of every agent, TD target with the target critic(s), critic forward / backward, clip, Adam,
// (single agent) soft update: DDPG_simple.py:139-149, TD3.py:193-213,235-244, SAC.py:226-238, MADDPG_simple.py:165-180 — and the actor
// stage — a_i = actor_i(s_i), Q_i(s, a) through the updated critic, dQ/da_i, actor backward, clip, Adam, soft update, SAC's alpha step:
// DDPG_simple.py:151-154, TD3.py:224-233, SAC.py:244-260, MADDPG_simple.py:182-186 — one launch each; a single agent's batch rows are
// drawn inside the critic launch and its noise sets regenerated where they are used (kernels_solo.hip's way; the same bits as
// draw_kernel's EngineDesc::idx / noise, which are read instead when the caller uploaded them, and for multi-agent engines, whose
// draw_kernel launch stays).  Same arithmetic per row as kernels_criticw.hip / kernels_actorw.hip; the decomposition is
// kernels_solo.hip's.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/solo_wide.hpp"
#include "device/rng.hpp"

namespace frl {

namespace {

// the policy head's row rule on this lane's outputs z[t][r] = component 16 t + 4 q + r of its row: a = tanh(.) [TARGET, TD3 / MATD3:
// + clipped smoothing noise, TD3.py:196-198; SAC: the tanh-Gaussian sample and the row's log-prob, SAC.py:70-97].  Returns the row's
// log pi (summed over the lanes of the row); lsv: log_std as stored (the actor stage's backward needs it)
template <int NT3, bool TARGET>
__device__ __forceinline__ float policy_rows(const SoloWNet& N, const LearnArgs& a, bool sac, int A, const f32x4 (&z)[NT3], const f32x4 (&nz)[NT3],
                                             f32x4 (&an)[NT3], f32x4 (&lsv)[NT3]) {
    const int q = N.C.q;
    float lp = 0.f;
#pragma unroll
    for (int t = 0; t < NT3; ++t) {
        an[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        lsv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * t + 4 * q + r;
            if (c < A) {
                const float zr = z[t][r];
                if (sac) {
                    lsv[t][r] = N.C.S.ls[c];
                    const float ls = fminf(fmaxf(lsv[t][r], -20.f), 2.f), sd = expf(ls);
                    const float u = zr + sd * nz[t][r], du = u - zr;
                    lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                    lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                    an[t][r] = tanhf(u);
                } else {
                    float v = tanhf(zr);
                    if (TARGET && a.use_policy_noise) {
                        float n1 = a.policy_noise_scale * (nz[t][r] * a.policy_noise);
                        n1 = fminf(fmaxf(n1, -a.noise_clip), a.noise_clip);
                        v = fminf(fmaxf(v * a.max_action + n1, -a.max_action), a.max_action) / a.max_action;
                    }
                    an[t][r] = v;
                }
            }
        }
    }
    lp += lane_xor<16>(lp);
    lp += lane_xor<32>(lp);
    return lp;
}

}  // namespace

// NT3 = head tiles of the actors (act_dim <= 16 -> 1, <= 32 -> 2)
// MULTI: MADDPG / MATD3 (a compile-time 1 for the single-agent kernels: their loop over the agents' target actors is straight-line code)
// T: row tiles per workgroup — 1.  T = 2 (populations of 17 .. 32 single-agent units on eight workgroups each, kernels_solo.hip's _w8
// form) was built and measured, SAC at 376 / 17, us per learn(): 17 / 24 / 32 learners 384 / 457 / 536 against the row-chunk chain's
// 435 / ~480 / 530 — every CU holds a workgroup that streams W1 five times per tile and writes two 560 KB slabs, and the chip's
// memory side is what the 256 of them share — so it is not instantiated (as a RUNTIME tile loop hipcc hoisted the tile-invariant
// address arithmetic of the whole body in front of it: 512 registers, 250-410 spilled; unrolled: 20-56 spilled)
// FUSED: the critic half of a policy step in ONE launch (solow_step_*): a workgroup with a row tile flags its slab and goes on to the
// policy's forward, which needs nothing of the update; the unit's HELPERS alone sum the slabs and step the critic meanwhile
template <bool TWIN, int NT3, bool MULTI, int T, bool FUSED>
__device__ __forceinline__ void solow_critic_body(const EngineDesc& D, const LearnArgs& a, const SoloArgs& s, float* smem) {
    constexpr int NH = TWIN ? 2 : 1;
    const int Wt = s.update_wgs, NT = s.tiles, Wc = s.row_wgs;   // the unit's Wc workgroups with row tiles (NT / Wc each, walked one after the other), then its helpers (the update only)
    const int nag = MULTI ? D.n_agents : 1;
    if (MULTI && (int)blockIdx.x >= a.p_count * nag * Wt) {
        // a spare workgroup per unit behind the units' own in the grid (MADDPG without smoothing noise): the rows agent `ag` of learner `p`
        // samples in the NEXT call — draw_kernel's draw (stream = the agent, the duplicate table in this workgroup's LDS), a launch
        // ahead; the host remembers what they were drawn for and skips draw_kernel when the next call asks for exactly that
        const int us = a.p0 * nag + (int)blockIdx.x - a.p_count * nag * Wt, ps = us / nag, as_ = us - ps * nag;
        int* out = s.pre_write + (size_t)(us - a.p0 * nag) * (8 + D.batch_max);
        FRL_LDS int* lb = (FRL_LDS int*)smem;
        draw_indices((g_i)(out + 8), lb, a.batch, a.size, s.pre_counter, (unsigned)as_, D.seed + 0x9E3779B97F4A7C15ull * (ps + 1), true,
                     (a.batch > kWG && 4 * a.batch <= kDrawTable) ? lb + 2 * ((a.batch + 3) & ~3) : nullptr);
        return;
    }
    const int unit = a.p0 * nag + blockIdx.x / Wt, p = unit / nag, ag = unit - p * nag, b = blockIdx.x % Wt;
    const RecordDesc& R = D.rec;
    const NetDesc& NC = D.net[2 * ag + 1];
    SoloWNet N;
    N.init(smem);
    const ChainNet& C = N.C;
    const int tid = C.tid, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, OT = R.obs_total, AT = R.act_total, am = D.act_max;
    const int nb = (B + 15) / 16;
    const bool sac = (D.algo == ALGO_SAC), inline_draw = a.device_rng && nag == 1;
    const size_t lbase = (size_t)p * D.learner_stride;
    const int noff = D.net_off[2 * ag + 1];
    g_f thC = as_global(D.theta + lbase + noff);
    g_f tgC = as_global(D.target + lbase + noff);
    g_f mC = as_global(D.m + lbase + noff);
    g_f vC = as_global(D.v + lbase + noff);
    g_f grC = as_global(D.grad + lbase + noff);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    float* part = s.part + ((size_t)unit * Wt) * kSoloPart;
    const float invB = 1.f / (float)B;
    const int KB1c = NC.L[0].k_pad >> 4;
    const int ka0 = OT >> 4, nka = KB1c - ka0;                             // the k-tiles of the critic's first layer that hold action columns (<= 3)
    const int t_new = steps[2 * ag + 1] + 1;           // read by every workgroup before the hand-over; rewritten behind the mailboxes
    SOLO_T0();

    if (b == Wc && s.pre_write && inline_draw) {
        // the learner's first helper has nothing to do until the hand-over: the rows of the NEXT call (the Philox counter the next
        // frl_learn will take, the current ring size), where nobody waits for them — kernels_solo.hip's spare workgroup
        int* out = s.pre_write + (size_t)(p - a.p0) * kSoloPre;
        draw_indices((g_i)(out + 8), (FRL_LDS int*)N.ea, B, a.size, s.pre_counter, 0u, D.seed + 0x9E3779B97F4A7C15ull * (p + 1), true);
        if (tid == 0) { out[0] = (int)(unsigned)s.pre_counter; out[1] = (int)(unsigned)(s.pre_counter >> 32); out[2] = a.size; out[3] = B; }
    }
    // this workgroup's row tiles b, b + Wc, ..: one after the other, a slab per TILE
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
        const int bt = b + Wc * tt;
        if (bt >= nb || b >= Wc) break;
        if (tt > 0) { lds_barrier(); __builtin_amdgcn_sched_barrier(0); }      // (every wave is done with the tile in front: its rows, action table and exchanges)
        g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
        g_ci idx = as_global_i(D.idx + (size_t)unit * D.batch_max);
        g_cf noise_u = as_global(D.noise + (size_t)unit * D.noise_sets * D.batch_max * am);      // this unit's sets
        g_f slab = as_global(s.slab + ((size_t)unit * NT + bt) * s.slab_stride);
        const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
        const int row = 16 * bt + i16, rc = row < B ? row : B - 1;
        float lossp = 0.f;
        const bool valid = row < B;
        const NetDesc& NA0 = D.net[0];
        g_cf tgA0 = as_global(D.target + lbase + D.net_off[0]);
        SoloWNet::Stage pend = N.stage_fetch(tgA0, NA0.L, NT3, NA0.extra_off, NA0.extra_n);
        SoloWNet::Pre pre = N.pre_fetch(tgA0 + NA0.L[0].w_off, NA0.L[0].k_pad >> 4), pren;
        const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
        int ri;
        // (the next call's rows may have been drawn by the previous launch's first helper workgroup: kernels_solo.hip's tag check)
        const int* tagp = (s.pre_read && inline_draw) ? s.pre_read + (size_t)(p - a.p0) * kSoloPre : nullptr;
        const int ri_pre = tagp ? tagp[8 + rc] : 0;
        const bool pre_ok = tagp && tagp[0] == (int)(unsigned)a.rng_counter && tagp[1] == (int)(unsigned)(a.rng_counter >> 32) && tagp[2] == a.size && tagp[3] == B;
        if (pre_ok) {
            ri = ri_pre;
            if (w == 0 && q == 0 && valid) D.idx[(size_t)p * D.batch_max + row] = ri;       // (the actor stage and frl_last_indices read them)
        } else if (inline_draw) {
            // draw_kernel's work, here: every workgroup of the learner draws the SAME `batch` distinct rows (same Philox key / counter,
            // rejection in its own LDS: ~3 us against a 13 us launch in front of this one) and keeps its tile's; they all write the same
            // values to D.idx (the actor stage and frl_last_indices read them)
            FRL_LDS int* lidx = (FRL_LDS int*)N.ea;
            draw_indices((g_i)(D.idx + (size_t)p * D.batch_max), lidx, B, a.size, a.rng_counter, 0u, key, false);
            ri = lidx[rc];
        } else if (MULTI && s.pre_read) {                  // the previous launch's spare workgroup drew this unit's rows (the host checked what for)
            ri = s.pre_read[(size_t)(unit - a.p0 * nag) * (8 + D.batch_max) + 8 + rc];
            if (w == 0 && q == 0 && valid) D.idx[(size_t)unit * D.batch_max + row] = ri;     // (the actor stage and frl_last_indices read them)
        } else {
            ri = idx[rc];
        }
        g_cf rec = ring + (size_t)ri * R.stride;
        // s'_0 (zero behind its columns; a single agent: out to the critic's k-tiles, whose observation part it also is)
        SoloWNet::XRegs xr = N.x_fetch(SoloWX{rec, R.nobs_off[0], R.obs_dim[0], R.stride}, nag == 1 ? KB1c : NA0.L[0].k_pad >> 4);
        const float rew = rec[R.rew_off + ag], done = rec[R.done_off + ag];
        // ---- a'_j = actor_target_j(s'_j) for every agent j -> ar = the joint target action of the tile's rows [SAC: the tanh-Gaussian
        // sample and its log-prob, SAC.py:70-97,227; TD3 / MATD3: smoothing noise, TD3.py:196-198]
        f32x4 h1o[2], h2o[2], h2f[kHT];
        float lp = 0.f;
        for (int j = 0; j < nag; ++j) {
            const NetDesc& NA = D.net[2 * j];
            g_cf tgA = as_global(D.target + lbase + D.net_off[2 * j]);
            const int Oj = R.obs_dim[j], Aj = R.act_dim[j], aoff = R.act_off[j] - R.act_off[0], KB1a = NA.L[0].k_pad >> 4;
            if (j > 0) {
                pend = N.stage_fetch(tgA, NA.L, NT3, NA.extra_off, NA.extra_n);
                pre = N.pre_fetch(tgA + NA.L[0].w_off, KB1a);
                xr = N.x_fetch(SoloWX{rec, R.nobs_off[j], Oj, R.stride}, KB1a);
            }
            // MATD3's per-agent smoothing noise is set j of the updating agent; single agent: set 0 (TD3 policy noise / SAC eps')
            g_cf noise0 = noise_u + (size_t)(nag > 1 ? j : 0) * D.batch_max * am;
            f32x4 nz[NT3];
            const bool want_nz = sac || a.use_policy_noise;
#pragma unroll
            for (int t = 0; t < NT3; ++t) {
                nz[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (inline_draw) {
                    // wave w regenerates component 16 t + 4 q + w of its lanes' rows (Philox + Box-Muller: ~250 instructions each, and every
                    // wave holds the same rows); the other three come through LDS behind the image's barriers
                    const int c = 16 * t + 4 * q + w;
                    float v = 0.f;
                    if (c < Aj && want_nz) { float n0, n1; normal2(philox4x32_10(a.rng_counter, 0x4000u, (unsigned)(rc * am + c), key), n0, n1); v = n0; }      // = draw_kernel's set 0
                    N.tz[i16 * 32 + c] = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * t + 4 * q + r;
                        if (c < Aj && want_nz) nz[t][r] = noise0[(unsigned)(rc * am + c)];
                    }
                }
            }
            N.x_commit(xr, nag == 1 ? KB1c : KB1a);    // (every wave is behind the first-layer reads of the pass in front: its two barriers)
            N.stage_commit(pend);
            if (inline_draw) {
#pragma unroll
                for (int t = 0; t < NT3; ++t) nz[t] = ld4((lds_cf)(N.tz + i16 * 32 + 16 * t + 4 * q));
            }
            if (j == 0) SOLO_T(0);
            if (j == nag - 1) {
                pend = N.stage_fetch((g_cf)tgC, NC.L, 1, -1, 0);
                pren = N.pre_fetch((g_cf)tgC + NC.L[0].w_off, KB1c);
                // single agent: [s | a], the record's first O + A columns, for the training passes; multi-agent: s' of ALL agents first
                xr = nag == 1 ? N.x_fetch(SoloWX{rec, R.obs_off[0], OT + AT, R.stride}, KB1c) : N.x_fetch(SoloWX{rec, R.nobs_off[0], OT, R.stride}, KB1c);
            }
            N.forward<false>(tgA + NA.L[0].w_off, KB1a, KB1a, pre, h1o, h2o, h2f);
            f32x4 z[NT3], an[NT3], lsv[NT3];
            N.head_tiles<NT3>(h2f, z);
            lp = policy_rows<NT3, true>(N, a, sac, Aj, z, nz, an, lsv);
            if (w == 0) {                                  // the tile's action rows (every wave holds the same values)
#pragma unroll
                for (int t = 0; t < NT3; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * t + 4 * q + r;
                        if (c < Aj) N.ar[i16 * 32 + aoff + c] = an[t][r];
                    }
            }
        }
        if (nag > 1) {                                     // the critic's observation columns: s' of every agent (contiguous in the record)
            N.x_commit(xr, KB1c);
            xr = N.x_fetch(SoloWX{rec, R.obs_off[0], OT + AT, R.stride}, KB1c);      // ... and [s | a] for the training passes
        }
        lds_barrier();
        N.xa_compose(OT, AT, ka0, nka);                    // [s' | a'] of the action k-tiles; read behind the next commit's barriers
        SOLO_T(1);
        // ---- y = r + gamma (1 - d) min_h Q_target_h(s', a')   (SAC: - alpha log pi)
        float qmin = 0.f;
        {
#pragma unroll
            for (int hd = 0; hd < NH; ++hd) {
                N.stage_commit(pend);
                pre = pren;
                pend = hd + 1 < NH ? N.stage_fetch((g_cf)tgC, NC.L + 3 * (hd + 1), 1, -1, 0) : N.stage_fetch((g_cf)thC, NC.L, 1, -1, 0);
                pren = N.pre_fetch(hd + 1 < NH ? (g_cf)tgC + NC.L[3 * (hd + 1)].w_off : (g_cf)thC + NC.L[0].w_off, KB1c);
                N.forward<false>((g_cf)tgC + NC.L[3 * hd].w_off, KB1c, ka0, pre, h1o, h2o, h2f);
                const float qv = N.head_q(h2f);
                qmin = hd == 0 ? qv : fminf(qmin, qv);
            }
        }
        SOLO_T(2);
        const float y = sac ? rew + a.gamma * (1.f - done) * (qmin + alpha * (-lp)) : rew + a.gamma * qmin * (1.f - done);
        // ---- the critic's heads: forward, TD delta, backward -> this workgroup's slab, on [s | a] (every wave is behind the
        // first-layer reads of the last target pass: its two barriers)
        N.x_commit(xr, KB1c);
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {
            const LayerDesc* L = NC.L + 3 * hd;
            N.stage_commit(pend);
            pre = pren;
            if (hd + 1 < NH) { pend = N.stage_fetch((g_cf)thC, NC.L + 3 * (hd + 1), 1, -1, 0); pren = N.pre_fetch((g_cf)thC + NC.L[3 * (hd + 1)].w_off, KB1c); }
            N.forward<true>((g_cf)thC + L[0].w_off, KB1c, KB1c, pre, h1o, h2o, h2f);
            const float z = N.head_q(h2f);
            float dz = 0.f;
            if (valid) {
                float lrow, grow;
                td_loss_row(a, z - y, lrow, grow);
                dz = grow * invB;
                lossp += lrow;
            }
            f32x4 d2o[2], d1o[2];
            N.head_bwd_q<true>(slab, L, dz, h2o, d2o);
            N.hidden_bwd<true>(slab, L, KB1c, d2o, h1o, d1o);
        }
        lossp = SoloNet::rows_sum(lossp);
        if (tid == 0) part[bt * kSoloPart + 0] = lossp;
    }
    SOLO_T(3);
    int ub = b, uw = Wt;                               // this workgroup's place among the ones that share the update
    if constexpr (FUSED) {
        if (b < Wc) { solow_publish(s.bar + (size_t)unit * NT + b, s.bar_base + kSoloWG); return; }
        solow_wait_flags(s.bar + (size_t)unit * NT, Wc, s.bar_base + kSoloWG, s.err);
        ub = b - Wc; uw = Wt - Wc;
    } else {
        solow_grid_sync(s.bar + (size_t)unit * NT, b, Wc, s.bar_base + kSoloWG, s.err);
    }
    SOLO_T(4);
    SoloUpdate u;
    u.th = thC; u.mm = mC; u.vv = vC; u.tg = tgC; u.size = NC.size; u.lr = a.critic_lr; u.wd = a.critic_wd;
    // single agent: the target moves here (TD3: with the delayed policy step, TD3.py:224-233); MADDPG: soft_update_kernel afterwards
    // (every agent's workgroups read every target actor)
    u.soft = (nag == 1 && a.do_actor != 0) ? 1 : 0;
    u.t_new = t_new;
    const float total = solow_update(s, a, u, grC, unit, part, ub, nb, NT, uw, N.red, N.ea, s.bar_base + kSoloWG SOLO_TARG);
    SOLO_T(7);
    if (ub == 0 && tid == 0) {
        float loss = 0.f;
        for (int k = 0; k < nb; ++k) loss += __hip_atomic_load(part + k * kSoloPart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        steps[2 * ag + 1] = t_new;
        float* sts = D.stats + (size_t)unit * ST_COUNT;
        sts[ST_CRITIC_LOSS] = loss * invB;
        sts[ST_CRITIC_GNORM] = total;
    }
    // (the loss partials are read: the row-tile workgroups may overwrite them with the actor stage's once they see this flag)
    if constexpr (FUSED) solow_publish(s.bar2 + (size_t)unit * 64 + ub, s.bar_base + kSoloWG);
}

#define FRL_SOLOW_CRITIC(name, twin, nt3, multi, tiles)                                                                            \
    __global__ __launch_bounds__(256) void name(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s) {                        \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                                \
        solow_critic_body<twin, nt3, multi, tiles, false>(*Dp, a, s, smem);                                                                \
    }
FRL_SOLOW_CRITIC(solow_critic_h1a1_kernel, false, 1, false, 1)
FRL_SOLOW_CRITIC(solow_critic_h1a2_kernel, false, 2, false, 1)
FRL_SOLOW_CRITIC(solow_critic_h2a1_kernel, true, 1, false, 1)
FRL_SOLOW_CRITIC(solow_critic_h2a2_kernel, true, 2, false, 1)
FRL_SOLOW_CRITIC(solow_critic_ma_h1a1_kernel, false, 1, true, 1)
FRL_SOLOW_CRITIC(solow_critic_ma_h1a2_kernel, false, 2, true, 1)
FRL_SOLOW_CRITIC(solow_critic_ma_h2a1_kernel, true, 1, true, 1)
FRL_SOLOW_CRITIC(solow_critic_ma_h2a2_kernel, true, 2, true, 1)


// ------------------------------------------------------------------------------------------------------------- actor stage
template <int NT3, bool MULTI, int T, bool FUSED>
__device__ __forceinline__ void solow_actor_body(const EngineDesc& D, const LearnArgs& a, const SoloArgs& s, float* smem) {
    const int Wt = s.update_wgs, NT = s.tiles, Wc = s.row_wgs;
    const int nag = MULTI ? D.n_agents : 1;
    const int unit = a.p0 * nag + blockIdx.x / Wt, p = unit / nag, ag = unit - p * nag, b = blockIdx.x % Wt;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[2 * ag];
    const NetDesc& NC = D.net[2 * ag + 1];
    SoloWNet N;
    N.init(smem);
    const ChainNet& C = N.C;
    const int tid = C.tid, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, OT = R.obs_total, AT = R.act_total, Oi = R.obs_dim[ag], Ai = R.act_dim[ag], am = D.act_max;
    const int aoff = R.act_off[ag] - R.act_off[0];
    const int nb = (B + 15) / 16;
    const bool sac = (D.algo == ALGO_SAC), inline_draw = a.device_rng && nag == 1;
    const size_t lbase = (size_t)p * D.learner_stride;
    g_f thA = as_global(D.theta + lbase + D.net_off[2 * ag]);
    g_f tgA = as_global(D.target + lbase + D.net_off[2 * ag]);
    g_f mA = as_global(D.m + lbase + D.net_off[2 * ag]);
    g_f vA = as_global(D.v + lbase + D.net_off[2 * ag]);
    g_f grA = as_global(D.grad + lbase + D.net_off[2 * ag]);
    g_cf thC = as_global(D.theta + lbase + D.net_off[2 * ag + 1]);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t_new = steps[2 * ag] + 1;
    float* part = s.part + ((size_t)unit * Wt) * kSoloPart;
    const float invB = 1.f / (float)B;
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const int nq = sac ? NC.heads : 1;                                     // SAC.py:250: mean of the twins; TD3.py:227: Q1 only
    const int KB1a = NA.L[0].k_pad >> 4, KB1c = NC.L[0].k_pad >> 4;
    // the k-tiles of the critic's first layer from agent i's first action column on are composed in xa (the batch's joint action with
    // a_i = actor_i(s_i) in it: MADDPG_simple.py:183); dQ/da_i walks back through the ones that hold agent i's columns (<= 3)
    const int ca = OT + aoff, ka0 = ca >> 4, nkx = KB1c - ka0, nkd = ((ca + Ai - 1) >> 4) - ka0 + 1;
    // single agent: s is the head of the critic's rows [s | a], one image serves both nets; multi-agent: agent i's own rows sit behind them
    const int xb = nag == 1 ? 0 : kSoloWActorBase;
    SOLO_T0();

#pragma unroll
    for (int tt = 0; tt < T; ++tt) {                       // this workgroup's row tiles, one after the other; a slab per TILE
        const int bt = b + Wc * tt;
        if (bt >= nb || b >= Wc) break;
        if (tt > 0) { lds_barrier(); __builtin_amdgcn_sched_barrier(0); }
        float qrow = 0.f, lp = 0.f;
        g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
        g_ci idx = as_global_i(D.idx + (size_t)unit * D.batch_max);
        g_cf noise1 = as_global(D.noise + ((size_t)unit * D.noise_sets + 1) * D.batch_max * am);      // the actor stage's eps (set 1; SAC)
        g_f slab = as_global(s.slab + ((size_t)unit * NT + bt) * s.slab_stride);
        const int row = 16 * bt + i16, rc = row < B ? row : B - 1;
        const bool valid = row < B;
        SoloWNet::Stage pend = N.stage_fetch((g_cf)thA, NA.L, NT3, NA.extra_off, NA.extra_n);
        SoloWNet::Pre pre = N.pre_fetch((g_cf)thA + NA.L[0].w_off, KB1a), pren;
        g_cf rec = ring + (size_t)idx[rc] * R.stride;
        // single agent: s, zero behind its columns, out to the critic's k-tiles; multi-agent: s_i, and the joint [s | a] of the batch
        const SoloWNet::XRegs xr = N.x_fetch(SoloWX{rec, R.obs_off[ag], Oi, R.stride}, nag == 1 ? KB1c : KB1a);
        SoloWNet::XRegs xj = xr;
        if (nag > 1) xj = N.x_fetch(SoloWX{rec, R.obs_off[0], OT + AT, R.stride}, KB1c);
        f32x4 ep[NT3];
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            ep[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (inline_draw) {                             // (one component per wave, the rest through LDS: the critic stage's nz)
                const int c = 16 * t + 4 * q + w;
                float v = 0.f;
                if (sac && c < Ai) { float n0, n1; normal2(philox4x32_10(a.rng_counter, 0x4000u, (unsigned)(rc * am + c), D.seed + 0x9E3779B97F4A7C15ull * (p + 1)), n0, n1); v = n1; }   // = draw_kernel's set 1
                N.tz[i16 * 32 + c] = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * t + 4 * q + r;
                    if (sac && c < Ai) ep[t][r] = noise1[(unsigned)(rc * am + c)];
                }
            }
        }
        N.x_commit(xr, nag == 1 ? KB1c : KB1a, xb);
        if (nag > 1) N.x_commit(xj, KB1c);
        N.stage_commit(pend);
        if (inline_draw) {
#pragma unroll
            for (int t = 0; t < NT3; ++t) ep[t] = ld4((lds_cf)(N.tz + i16 * 32 + 16 * t + 4 * q));
        }
        SOLO_T(0);
        SoloWNet::DxRegs dxr;
        if constexpr (!FUSED) {                            // (a pass ahead: the critic is final when this launch starts)
            pend = N.stage_fetch(thC, NC.L, 1, -1, 0);
            pren = N.pre_fetch(thC + NC.L[0].w_off, KB1c);
            dxr = N.input_bwd_fetch(thC + NC.L[0].w_off, KB1c, ka0, nkd);
        }
        // ---- A: a = tanh(actor(s))   (SAC: a = tanh(mean + std eps) and the row's log pi, SAC.py:70-97)
        f32x4 ah1[2], ah2[2], h2f[kHT], an[NT3], lsv[NT3];
        float lpr = 0.f;
        {
            N.forward<true>((g_cf)thA + NA.L[0].w_off, KB1a, KB1a, pre, ah1, ah2, h2f, xb);   // (th1 / xs keep the actor's h1 and s for pass C: pass B leaves them alone)
            SOLO_T(9);
            f32x4 za[NT3];
            N.head_tiles<NT3>(h2f, za);
            SOLO_T(10);
            lpr = policy_rows<NT3, false>(N, a, sac, Ai, za, ep, an, lsv);
            SOLO_T(11);
            if (w == 0) {
#pragma unroll
                for (int t = 0; t < NT3; ++t) st4(N.ar + i16 * 32 + 16 * t + 4 * q, an[t]);
            }
            lds_barrier();
            N.xa_compose(ca, Ai, ka0, nkx);                // the action k-tiles with a_i in them; read behind the next commit's barriers
        }
        if constexpr (FUSED) {
            // the unit's helpers have stepped the critic meanwhile (its slab sum, clip, Adam, soft update): nothing of it is read before here
            solow_wait_flags(s.bar2 + (size_t)unit * 64, Wt - Wc, s.bar_base + kSoloWG, s.err);
            pend = N.stage_fetch(thC, NC.L, 1, -1, 0);
            pren = N.pre_fetch(thC + NC.L[0].w_off, KB1c);
            dxr = N.input_bwd_fetch(thC + NC.L[0].w_off, KB1c, ka0, nkd);
        }
        SOLO_T(1);
#ifdef FRL_SOLO_TIMING
        if (tid == 0) {
            part[b * kSoloPart + 8 + 8] = (float)(((unsigned)N.red[120] - (unsigned)(solo_t0_ & 0xFFFFFFull)) & 0xFFFFFFu);   // pass A: its first layer done
            part[b * kSoloPart + 8 + 12] = (float)(((unsigned)N.red[121] - (unsigned)(solo_t0_ & 0xFFFFFFull)) & 0xFFFFFFu);  // ... entered
            part[b * kSoloPart + 8 + 13] = (float)(((unsigned)N.red[122] - (unsigned)(solo_t0_ & 0xFFFFFFull)) & 0xFFFFFFu);  // ... first batch swept
        }
#endif
        // ---- B: Q(s, a) and dQ/da through the frozen (already stepped) critic
        const float dqv = sac ? -0.5f * invB : -invB;
        f32x4 dq[NT3];                                                     // d loss / d a[16 t + 4 q + r] of this lane's row
#pragma unroll
        for (int t = 0; t < NT3; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            for (int hd = 0; hd < nq; ++hd) {
                const LayerDesc* L = NC.L + 3 * hd;
                N.stage_commit(pend);
                pre = pren;
                pend = hd + 1 < nq ? N.stage_fetch(thC, NC.L + 3 * (hd + 1), 1, -1, 0) : N.stage_fetch((g_cf)thA, NA.L, NT3, NA.extra_off, NA.extra_n);
                SoloWNet::DxRegs dxn = dxr;
                if (hd + 1 < nq) { dxn = N.input_bwd_fetch(thC + NC.L[3 * (hd + 1)].w_off, KB1c, ka0, nkd); pren = N.pre_fetch(thC + NC.L[3 * (hd + 1)].w_off, KB1c); }
                f32x4 h1o[2], h2o[2], d2o[2], d1o[2];
                N.forward<false>(thC + L[0].w_off, KB1c, ka0, pre, h1o, h2o, h2f);
                const float z = N.head_q(h2f);
                if (valid) qrow += z;
                N.head_bwd_q<false>(nullptr, L, valid ? dqv : 0.f, h2o, d2o);
                N.hidden_bwd<false>(nullptr, L, KB1c, d2o, h1o, d1o);
                N.input_bwd(dxr, nkd, d1o);
#pragma unroll
                for (int t = 0; t < NT3; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * t + 4 * q + r;
                        if (c < Ai) dq[t][r] += N.dxa[i16 * 48 + (ca + c) - 16 * ka0];
                    }
                dxr = dxn;
            }
        }
        SOLO_T(2);
        // ---- C: through a = tanh(.) into the actor; its activations are pass A's (own tiles in registers, h1 / s transposed in LDS)
        N.stage_commit(pend);
        f32x4 dz[NT3], gls[NT3];
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            dz[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            gls[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * t + 4 * q + r;
                if (valid && c < Ai) {
                    const float av = an[t][r];
                    if (sac) {                                             // through u = mean + exp(log_std) eps, and alpha log pi
                        const float d = dq[t][r] * (1.f - av * av) + (alpha * invB) * (2.f * av);
                        const float ls = fminf(fmaxf(lsv[t][r], -20.f), 2.f);
                        dz[t][r] = d;
                        gls[t][r] = d * expf(ls) * ep[t][r] - alpha * invB;
                    } else {
                        dz[t][r] = dq[t][r] * (1.f - av * av);
                    }
                }
            }
        }
        f32x4 d2o[2], d1o[2];
        N.head_bwd_a<NT3>(slab, NA.L, dz, ah2, d2o);
        N.hidden_bwd<true>(slab, NA.L, KB1a, d2o, ah1, d1o, xb);
        // log_std's gradient of this row tile (zero outside the clamp [-20, 2], SAC.py:77) behind the net's layers
        if (sac) {
#pragma unroll
            for (int t = 0; t < NT3; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * t + 4 * q + r;
                    const float sgl = SoloNet::rows_sum(gls[t][r]);
                    if (w == 0 && i16 == 0 && c < Ai) slab[NA.extra_off + c] = (lsv[t][r] >= -20.f && lsv[t][r] <= 2.f) ? sgl : 0.f;
                }
        }
        lp = valid ? lpr : 0.f;
        qrow = SoloNet::rows_sum(qrow);
        lp = SoloNet::rows_sum(lp);
        if (tid == 0) { part[bt * kSoloPart + 0] = qrow; part[bt * kSoloPart + 1] = lp; }
    }
    SOLO_T(3);
    const unsigned epoch = s.bar_base + kSoloWG + (FUSED ? 1u : 0u);      // (the fused step's second hand-over: its own epoch for flags and mailboxes)
    solow_grid_sync(s.bar + (size_t)unit * NT, b, Wc, epoch, s.err);
    SOLO_T(4);
    SoloUpdate u;
    // (the actor's target moves here also for MADDPG: nothing in this launch reads a target net)
    u.th = thA; u.mm = mA; u.vv = vA; u.tg = tgA; u.size = NA.size; u.lr = a.actor_lr; u.wd = 0.f; u.soft = 1; u.t_new = t_new;
    const float total = solow_update(s, a, u, grA, unit, part, b, nb, NT, Wt, N.red, N.ea, epoch SOLO_TARG);
    if (MULTI) {
        // MADDPG_simple.py:188-190 / MATD3_simple.py:245-246: every target follows its net once all agents are updated — the critic's here,
        // a share per workgroup of the unit (soft_update_kernel's arithmetic; its launch, one workgroup per net, was 34 us of config 5's learn())
        g_cf thCw = as_global(D.theta + lbase + D.net_off[2 * ag + 1]);
        g_f tgCw = as_global(D.target + lbase + D.net_off[2 * ag + 1]);
        const int n4 = NC.size >> 2, per = (n4 + Wt - 1) / Wt, i1 = min(n4, (b + 1) * per);
        const float tk = 1.f - a.tau;
        for (int i = b * per + tid; i < i1; i += kWG) st4(tgCw + 4 * (size_t)i, ld4((g_cf)(tgCw + 4 * (size_t)i)) * tk + ld4(thCw + 4 * (size_t)i) * a.tau);
    }
    SOLO_T(7);
    if (b == 0 && tid == 0) {
        float qtot = 0.f, lptot = 0.f;
        for (int k = 0; k < nb; ++k) {
            qtot += __hip_atomic_load(part + k * kSoloPart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lptot += __hip_atomic_load(part + k * kSoloPart + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        steps[2 * ag] = t_new;
        float* sts = D.stats + (size_t)unit * ST_COUNT;
        sts[ST_ACTOR_LOSS] = sac ? (-(qtot * 0.5f) + alpha * lptot) * invB : -qtot * invB;   // SAC.py:251: (alpha log pi - Q).mean()
        sts[ST_ACTOR_GNORM] = total;
        if (sac) {                                                         // alpha step on the batch's entropy (SAC.py:154-169,257-260)
            float* al = D.alpha + p * 4;
            const float ent_mean = -lptot * invB;
            const float mean_term = ent_mean - a.target_entropy;
            const float gl = alpha * mean_term;                            // d alpha_loss / d log_alpha
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
            sts[ST_ALPHA_LOSS] = alpha * mean_term;
            sts[ST_ALPHA] = al[3];
            sts[ST_ENTROPY] = ent_mean;
        }
    }
}

#define FRL_SOLOW_ACTOR(name, nt3, multi, tiles)                                                                                     \
    __global__ __launch_bounds__(256) void name(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s) {                        \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                                \
        solow_actor_body<nt3, multi, tiles, false>(*Dp, a, s, smem);                                                                \
    }
FRL_SOLOW_ACTOR(solow_actor_a1_kernel, 1, false, 1)
FRL_SOLOW_ACTOR(solow_actor_a2_kernel, 2, false, 1)
FRL_SOLOW_ACTOR(solow_actor_ma_a1_kernel, 1, true, 1)
FRL_SOLOW_ACTOR(solow_actor_ma_a2_kernel, 2, true, 1)


// ------------------------------------------------------------------------------------------------------------- a policy step in one launch
// Single-agent engines with helper workgroups (up to eight learners): critic stage and actor stage in ONE launch, the critic's update
// (slab sum, clip, Adam, soft update — on the unit's helpers alone) under the policy's forward on the workgroups with row tiles.
#define FRL_SOLOW_STEP(name, twin, nt3)                                                                                            \
    __global__ __launch_bounds__(256) void name(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s) {                        \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                                \
        solow_critic_body<twin, nt3, false, 1, true>(*Dp, a, s, smem);                                                              \
        __syncthreads();                                                                                                            \
        solow_actor_body<nt3, false, 1, true>(*Dp, a, s, smem);                                                                     \
    }
FRL_SOLOW_STEP(solow_step_h1a1_kernel, false, 1)
FRL_SOLOW_STEP(solow_step_h1a2_kernel, false, 2)
FRL_SOLOW_STEP(solow_step_h2a1_kernel, true, 1)
FRL_SOLOW_STEP(solow_step_h2a2_kernel, true, 2)

}  // namespace frl
