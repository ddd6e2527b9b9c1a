// DDPG / TD3 / SAC update of a SINGLE learner (or a handful) on sixteen workgroups per learner (device/solo.hpp): the critic stage
// — TD target with the target nets, critic forward / backward, clip, Adam, soft update: DDPG_simple.py:139-149, TD3.py:193-213,235-244,
// SAC.py:226-238 — and the actor stage — a = actor(s), Q(s, a) through the updated critic, dQ/da, actor backward, clip, Adam, soft
// update, SAC's alpha step: DDPG_simple.py:151-154, TD3.py:224-233, SAC.py:244-260 — one launch each.  Same arithmetic per row as
// kernels_critic2.hip / kernels_actor2.hip (whose row rules are restated here on the "every lane of the row" layout this
// decomposition produces); different decomposition: rows over workgroups, output features over waves, the gradient met by a
// deterministic slab sum behind a grid barrier.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/solo.hpp"
#include "device/rng.hpp"
#include "device/act_common.hpp"

namespace frl {

namespace {

__device__ __forceinline__ float pick4(const f32x4& v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }

// the fields of this lane's row that every pass needs
struct SoloRow {
    g_cf rec;               // the row's record in the ring (the batch's last row for lanes past the batch)
    bool valid;
    int row;
};

// input columns 4q .. 4q + 3 of the row: observation columns [0, O) from obs0 (obs or next_obs), then A action columns from act4
__device__ __forceinline__ f32x4 critic_input(const SoloNet& N, const f32x4& ob, const f32x4& act4, int O, int A) {
    f32x4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int f = 4 * N.C.q + e;
        x[e] = f < O ? ob[e] : (f < O + A ? pick4(act4, f - O) : 0.f);
    }
    return x;
}

// Buffer.add of the vector step (replay_commit_kernel's ring writes, without its obs_cur update: that is the tail's): a 16-lane group
// per env of learner p.  Ends with a draining barrier: the gathers behind the draw may read these rows.
__device__ __forceinline__ void solo_step_head(const EngineDesc& D, const SoloStepArgs& st, int p) {
    const CommitArgs& c = st.c;
    const RecordDesc& R = D.rec;
    const int lane = threadIdx.x & 15;
    for (int j = threadIdx.x >> 4; j < c.E; j += kWG / 16) {
        const size_t i = (size_t)p * c.E + j;
        g_f r = as_global(D.replay + ((size_t)p * D.capacity + c.row[i]) * R.stride);
        const unsigned char fl = c.flags[i];
        for (int k = lane; k < c.O; k += 16) {
            r[R.obs_off[0] + k] = c.obs_cur[i * c.O + k];
            r[R.nobs_off[0] + k] = c.next_obs[i * c.O + k];
        }
        for (int k = lane; k < c.aout; k += 16) r[R.act_off[0] + k] = c.store_act[i * c.aout + k];
        if (lane == 0) { r[R.rew_off] = c.reward[i]; r[R.done_off] = (fl & 1) ? 1.f : 0.f; }
    }
    __syncthreads();
}

// The step's tail, workgroup 0 of learner p, behind whatever made the actor's parameters final: obs_cur <- obs_next, then
// select_action + exploration on it (act_frag_body re-carves the workgroup's LDS: everything else of this launch is done), then
// the hand-over to the host once every learner of the launch is through.
__device__ __forceinline__ void solo_step_tail(const EngineDesc& D, const LearnArgs& a, const SoloStepArgs& st, float* smem, int p) {
    const CommitArgs& c = st.c;
    __syncthreads();
    for (int e = threadIdx.x; e < c.E * c.O; e += kWG) c.obs_cur[(size_t)p * c.E * c.O + e] = c.obs_next[(size_t)p * c.E * c.O + e];
    __syncthreads();
    if (!st.act) return;
    for (int r0 = 0; r0 < c.E; r0 += 64) {
        act_frag_body(D, st.act_args, smem, p, r0);
        __syncthreads();
    }
    if (st.done_flag) {
        // this learner's actions are out (sync_stores: every wave's stores acknowledged); the last learner to get here flags
        // the host — lane 0's system-scope release + an explicit wait in front of the flag
        sync_stores();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (a.p_count == 1 || atomicAdd(st.ticket, 1) == a.p_count - 1) {     // (one learner: no ticket round trip)
                if (a.p_count > 1) { *st.ticket = 0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_store(st.done_flag, st.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// a folded step's workgroup on its way out: its parameter slices, flags and mailboxes are written — the next step's pre-armed critic
// launch (other stream) counts the workgroups of this step before it reads any of them
__device__ __forceinline__ void solo_leave(const SoloStepArgs& st) {
    if (!st.dev_cnt) return;
    sync_stores();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(st.dev_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

// W = workgroups per learner: 16 (one 16-row tile each: up to 16 learners) or 8 (two tiles each, walked one after the other, every tile's
// gradient in a slab of its own: populations of 17 .. 32 learners, whose 8 P workgroups still are all resident).  Measured, TD3 us per learn():
// 17 / 24 / 32 learners 113 / 124 / 140 against the row-chunk kernels' 151 / 155 / 160; W = 4 (33 .. 64 learners, four tiles each) was built
// too: 255 / 275 / 319 us at 33 / 48 / 64 against 160-190 — every further tile costs ~37 us (five image stagings, the target critics'
// fragment fetch and the row fetch in the open again), so it is not instantiated
template <bool TWIN, int W>
__device__ __forceinline__ void solo_critic_body(const EngineDesc& D, const LearnArgs& a, const SoloArgs& s, const SoloStepArgs& st, float* smem) {
    constexpr int NH = TWIN ? 2 : 1, kT = kSoloWG / W;                 // kT: row tiles of a 256-row batch per workgroup
    if ((int)blockIdx.x >= a.p_count * W) {
        // a spare workgroup (one per learner, behind the learners' own in the grid: it lands on a CU they leave idle and is gone
        // long before they are): the rows of the NEXT call, where nobody waits for them
        const int pp = a.p0 + (int)blockIdx.x - a.p_count * W;
        int* out = s.pre_write + (size_t)(pp - a.p0) * kSoloPre;
        FRL_LDS int* lidx = (FRL_LDS int*)smem;
        draw_indices((g_i)(out + 8), lidx, a.batch, a.size, s.pre_counter, 0u, D.seed + 0x9E3779B97F4A7C15ull * (pp + 1), true);
        if (threadIdx.x == 0) { out[0] = (int)(unsigned)s.pre_counter; out[1] = (int)(unsigned)(s.pre_counter >> 32); out[2] = a.size; out[3] = a.batch; }
        return;
    }
    const int p = a.p0 + blockIdx.x / W, b = blockIdx.x % W;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[0];
    const NetDesc& NC = D.net[1];
    SoloNet N;
    N.init(smem);
    const ChainNet& C = N.C;
    const int tid = C.tid, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, O = R.obs_dim[0], A = R.act_dim[0], am = D.act_max;
    const int nb = (B + 15) / 16;
    const bool sac = (D.algo == ALGO_SAC);
    const size_t lbase = (size_t)p * D.learner_stride;
    g_cf tgA = as_global(D.target + lbase + D.net_off[0]);
    g_f thC = as_global(D.theta + lbase + D.net_off[1]);
    g_f tgC = as_global(D.target + lbase + D.net_off[1]);
    g_f mC = as_global(D.m + lbase + D.net_off[1]);
    g_f vC = as_global(D.v + lbase + D.net_off[1]);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    float* part = s.part + ((size_t)p * kSoloWG) * kSoloPart;
    const float invB = 1.f / (float)B;
    SOLO_T0();
    bool drawn_early = false;
    int ri_t[kT];                                      // the ring rows of this lane's row in each of the workgroup's tiles (tile b + W t)
    auto tile_rows = [&](auto from) {                  // ri_t[t] = from(row of tile t), rows past the batch clamped to its last
#pragma unroll
        for (int t = 0; t < kT; ++t) { const int rr = 16 * (b + W * t) + i16; ri_t[t] = from(rr < B ? rr : B - 1); }
    };
    if (st.go_flag) {
        // pre-armed (frl_rollout): enqueued a vector step ahead on the pool's second stream — the previous step's launches may still be
        // running.  The batch's rows depend on the launch's arguments only: drawn now (LDS only: the actor launch in front may still be
        // reading D.idx).  Then: every workgroup of the previous step has left (dev_cnt: the nets are final), and the host's doorbell —
        // the step's block is there (2 s each, then give up: a launch that returns here has touched nothing; the actor launch queued
        // behind it sees the same word)
        if (b < nb && a.device_rng) {
            FRL_LDS int* lidx = (FRL_LDS int*)N.ea;
            draw_indices((g_i) nullptr, lidx, B, a.size, a.rng_counter, 0u, D.seed + 0x9E3779B97F4A7C15ull * (p + 1), false);
            tile_rows([&](int rr) { return lidx[rr]; });
            drawn_early = true;
        }
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            int c = __hip_atomic_load(st.dev_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (c - st.dev_wait < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ull) break;
                c = __hip_atomic_load(st.dev_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int v = -1;
            if (c - st.dev_wait >= 0) {
                v = __hip_atomic_load(st.go_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                while (v != st.go_value && v != -1) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t0 > 400000000ull) { v = -1; break; }
                    v = __hip_atomic_load(st.go_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");              // (relaxed polls, then ONE lane's invalidate for the CU, in front of the barrier)
            N.red[100] = __int_as_float(v);
        }
        __syncthreads();                               // (also: every wave has read its row out of ea)
        if (__float_as_int(N.red[100]) != st.go_value) return;
        if (drawn_early && q == 0 && w == 0) {
#pragma unroll
            for (int t = 0; t < kT; ++t) { const int rr = 16 * (b + W * t) + i16; if (rr < B) D.idx[(size_t)p * D.batch_max + rr] = ri_t[t]; }
        }
    }
    const int t_new = steps[1] + 1;                    // read by every workgroup before the first grid barrier; rewritten behind the second
    if (st.head) solo_step_head(D, st, p);             // (every workgroup, also the ones without rows: they all pass the same barriers)

    float lossp = 0.f;
    if (b < nb) {
        g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
        g_ci idx = as_global_i(D.idx + (size_t)p * D.batch_max);
        g_cf noise0 = as_global(D.noise + (size_t)p * D.noise_sets * D.batch_max * am);
        const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
        // the first image travels while the indices are drawn and the row's fields fetched
        ChainNet::StageRegs pend;
        pend = C.stage_fetch(tgA, 0, NA.extra_n);
        const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
        // (the rows are fetched next to the tag, not behind it: one round trip)
        const int* pre = s.pre_read ? s.pre_read + (size_t)(p - a.p0) * kSoloPre : nullptr;
        int ri_pre[kT];
#pragma unroll
        for (int t = 0; t < kT; ++t) { const int rr = 16 * (b + W * t) + i16; ri_pre[t] = pre ? pre[8 + (rr < B ? rr : B - 1)] : 0; }
        const bool pre_ok = pre && a.device_rng && !drawn_early && pre[0] == (int)(unsigned)a.rng_counter && pre[1] == (int)(unsigned)(a.rng_counter >> 32) &&
                            pre[2] == a.size && pre[3] == B;
        if (drawn_early) {
        } else if (pre_ok) {
            // the previous launch's spare workgroup drew this call's rows (same counter, ring size and batch: the same bits as the
            // draw below); this workgroup's tiles go to D.idx for the actor stage and frl_last_indices
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                const int rr = 16 * (b + W * t) + i16;
                ri_t[t] = ri_pre[t];
                if (w == 0 && q == 0 && rr < B) D.idx[(size_t)p * D.batch_max + rr] = ri_t[t];
            }
            SOLO_T(8);
        } else if (a.device_rng) {
            // draw_kernel's work, here: every workgroup of the learner draws the SAME `batch` distinct rows (same Philox key / counter,
            // rejection in its own LDS: ~3 us, against a 10 us launch in front of this one) and keeps its tile's; they all write the
            // same values to D.idx (the actor stage and frl_last_indices read them)
            FRL_LDS int* lidx = (FRL_LDS int*)N.ea;
            SOLO_T(10);
            draw_indices((g_i)(D.idx + (size_t)p * D.batch_max), lidx, B, a.size, a.rng_counter, 0u, key, false);
            tile_rows([&](int rr) { return lidx[rr]; });
            SOLO_T(8);
        } else {
            tile_rows([&](int rr) { return idx[rr]; });
        }
#pragma unroll
        for (int t = 0; t < kT; ++t) {                 // this workgroup's row tiles, one after the other (W = 16: one); a slab per TILE
        const int bt = b + W * t;
        if (bt >= nb) break;
        g_f slab = as_global(s.slab + ((size_t)p * kSoloWG + bt) * s.slab_stride);
        if (t > 0) {
            __builtin_amdgcn_sched_barrier(0);         // (a tile starts where the one in front ends: with the next tile's image and target-head fragments
            pend = C.stage_fetch(tgA, 0, NA.extra_n);  //  hoisted over the backward in front, the twin kernel was 512 registers and 37 spilled)
        }
        const int row = 16 * bt + i16;
        const bool valid = row < B;
        const int ri = ri_t[t];
        g_cf rec = ring + (size_t)ri * R.stride;
        f32x4 sn, so, ac = {0.f, 0.f, 0.f, 0.f}, nz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = 4 * q + e, fc = f < O ? f : O - 1;
            sn[e] = rec[R.nobs_off[0] + fc]; so[e] = rec[R.obs_off[0] + fc];
            if (f >= O) { sn[e] = 0.f; so[e] = 0.f; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < A) {
                ac[r] = rec[R.act_off[0] + r];
                if (sac || a.use_policy_noise) {
                    const unsigned e1 = (unsigned)((valid ? row : B - 1) * am + r);
                    if (a.device_rng) { float n0, n1; normal2(philox4x32_10(a.rng_counter, 0x4000u, e1, key), n0, n1); nz[r] = n0; }      // = draw_kernel's set 0
                    else nz[r] = noise0[e1];
                }
            }
        }
        const float rew = rec[R.rew_off], done = rec[R.done_off];
        SOLO_T(9);
        C.stage_commit(pend);
        SOLO_T(0);
        // the target critics are forward-only: their fragments go straight from the block into registers (SoloNet::forward_g), both
        // heads in one pass — in flight under the target actor's pass
        SoloNet::Frag<1> FC[NH];
        if constexpr (kT == 1) {
#pragma unroll
            for (int hd = 0; hd < NH; ++hd) FC[hd] = N.frag_fetch<1>((g_cf)tgC + hd * kHeadFloats, 1, 0);
        }
        // ---- a' = actor_target(s') [SAC: the tanh-Gaussian sample and its log-prob, SAC.py:70-97,227; TD3: smoothing noise, TD3.py:196-198]
        f32x4 h1o[2], h2o[2], h2f[kHT], z, an = {0.f, 0.f, 0.f, 0.f};
        float lp = 0.f;
        N.forward<false>(sn, h1o, h2o, h2f, z, A);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < A) {
                if (sac) {
                    const float ls = fminf(fmaxf(C.S.ls[r], -20.f), 2.f), sd = expf(ls);
                    const float u = z[r] + sd * nz[r], du = u - z[r];
                    lp += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                    lp -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                    an[r] = tanhf(u);
                } else {
                    float v = tanhf(z[r]);
                    if (a.use_policy_noise) {
                        float n1 = a.policy_noise_scale * (nz[r] * a.policy_noise);
                        n1 = fminf(fmaxf(n1, -a.noise_clip), a.noise_clip);
                        v = fminf(fmaxf(v * a.max_action + n1, -a.max_action), a.max_action) / a.max_action;
                    }
                    an[r] = v;
                }
            }
        }
        SOLO_T(1);
        if constexpr (kT > 1) {                            // (two tiles per workgroup: the fragments of two target heads next to the actor's pass, unrolled
#pragma unroll                                             //  over both tiles, were 512 registers and 37 spilled — fetched behind it here)
            for (int hd = 0; hd < NH; ++hd) FC[hd] = N.frag_fetch<1>((g_cf)tgC + hd * kHeadFloats, 1, 0);
        }
        // ---- y = r + gamma (1 - d) min_h Q_target_h(s', a')   (SAC: - alpha log pi): the twin heads in one pass; the online critic's
        // first image travels under it
        pend = C.stage_fetch((g_cf)thC, 0);
        float qmin;
        {
            f32x4 xc[NH], zc[NH];
#pragma unroll
            for (int hd = 0; hd < NH; ++hd) xc[hd] = critic_input(N, sn, an, O, A);
            N.forward_g<NH, 1>(FC, xc, zc, 1);
            qmin = zc[0][0];
            if constexpr (NH == 2) qmin = fminf(qmin, zc[1][0]);
        }
        SOLO_T(2);
        const float y = sac ? rew + a.gamma * (1.f - done) * (qmin + alpha * (-lp)) : rew + a.gamma * qmin * (1.f - done);
        // ---- the critic's heads: forward, TD delta, backward -> this workgroup's slab
        const f32x4 xin = critic_input(N, so, ac, O, A);
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {
            C.stage_commit(pend);
            if (hd + 1 < NH) pend = C.stage_fetch((g_cf)thC, hd + 1);
            N.forward<true>(xin, h1o, h2o, h2f, z, 1);
            f32x4 dz = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                float lrow, grow;
                td_loss_row(a, z[0] - y, lrow, grow);
                dz[0] = grow * invB;
                lossp += lrow;
            }
            f32x4 d2o[2], d1o[2];
            g_f hs = slab + hd * kHeadFloats;
            N.head_bwd<true>(hs, dz, h2o, d2o, 1);
            N.hidden_bwd<true>(hs, d2o, h1o, d1o);
        }
        }   // tiles
        lossp = SoloNet::rows_sum(lossp);
        if (tid == 0) part[b * kSoloPart + 0] = lossp;
    }
    SOLO_T(3);
    solo_grid_sync(s.bar + (size_t)p * kSoloWG, b, s.bar_base + kSoloWG, s.err, W);
    SOLO_T(4);
    SoloUpdate u;
    u.th = thC; u.mm = mC; u.vv = vC; u.tg = tgC; u.size = NC.size; u.lr = a.critic_lr; u.wd = a.critic_wd;
    u.soft = a.do_actor != 0 ? 1 : 0;                                     // TD3: the targets move with the delayed policy step (TD3.py:224-233)
    u.t_new = t_new;
    const float total = solo_update<W>(s, a, u, p, b, nb, N.red, s.bar_base + kSoloWG SOLO_TARG);
    SOLO_T(7);
    if (b == 0 && tid == 0) {
        float loss = 0.f;
        for (int k = 0; k < min(nb, W); ++k) loss += __hip_atomic_load(part + k * kSoloPart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        steps[1] = t_new;
        float* sts = D.stats + (size_t)p * ST_COUNT;
        sts[ST_CRITIC_LOSS] = loss * invB;
        sts[ST_CRITIC_GNORM] = total;
    }
    // the rollout step's tail when no actor stage follows (the actor is unchanged: nothing to wait for; every workgroup's reads of
    // obs_cur / store_act in the head lie in front of the slab hand-over above)
    if (st.tail && b == 0) solo_step_tail(D, a, st, smem, p);
    solo_leave(st);
}

#define FRL_SOLO_CRITIC(name, twin, wgs)                                                                                          \
    __global__ __launch_bounds__(256) void name(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st) {       \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                                \
        solo_critic_body<twin, wgs>(*Dp, a, s, st, smem);                                                                           \
    }
FRL_SOLO_CRITIC(solo_critic_twin_kernel, true, 16)
FRL_SOLO_CRITIC(solo_critic_single_kernel, false, 16)
FRL_SOLO_CRITIC(solo_critic_twin_w8_kernel, true, 8)
FRL_SOLO_CRITIC(solo_critic_single_w8_kernel, false, 8)

// ------------------------------------------------------------------------------------------------------------- actor stage
template <int W>
__device__ __forceinline__ void solo_actor_body(const EngineDesc& D, const LearnArgs& a, const SoloArgs& s, const SoloStepArgs& st, float* smem) {
    constexpr int kT = kSoloWG / W;
    const int p = a.p0 + blockIdx.x / W, b = blockIdx.x % W;
    const RecordDesc& R = D.rec;
    const NetDesc& NA = D.net[0];
    const NetDesc& NC = D.net[1];
    SoloNet N;
    N.init(smem);
    const ChainNet& C = N.C;
    const int tid = C.tid, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, O = R.obs_dim[0], A = R.act_dim[0], am = D.act_max;
    const int nb = (B + 15) / 16;
    const bool sac = (D.algo == ALGO_SAC);
    const size_t lbase = (size_t)p * D.learner_stride;
    g_f thA = as_global(D.theta + lbase + D.net_off[0]);
    g_f tgA = as_global(D.target + lbase + D.net_off[0]);
    g_f mA = as_global(D.m + lbase + D.net_off[0]);
    g_f vA = as_global(D.v + lbase + D.net_off[0]);
    g_cf thC = as_global(D.theta + lbase + D.net_off[1]);
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    const int t_new = steps[0] + 1;
    float* part = s.part + ((size_t)p * kSoloWG) * kSoloPart;
    const float invB = 1.f / (float)B;
    const float alpha = sac ? D.alpha[p * 4 + 3] : 0.f;
    const int nq = sac ? NC.heads : 1;                                     // SAC.py:250: mean of the twins; TD3.py:227: Q1 only
    SOLO_T0();
    if (st.go_flag) {                                                      // (pre-armed step: the critic launch in front of this one waited for the doorbell; -1 = given up)
        if (tid == 0) N.red[100] = __int_as_float(__hip_atomic_load(st.go_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        __syncthreads();
        if (__float_as_int(N.red[100]) != st.go_value) return;
    }

    float qrow = 0.f, lp = 0.f;
    if (b < nb) {
        g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
        g_ci idx = as_global_i(D.idx + (size_t)p * D.batch_max);
        g_cf noise1 = as_global(D.noise + ((size_t)p * D.noise_sets + 1) * D.batch_max * am);      // the actor stage's eps (set 1)
#pragma unroll
        for (int t = 0; t < kT; ++t) {                 // this workgroup's row tiles, one after the other (W = 16: one); a slab per TILE
        const int bt = b + W * t;
        if (bt >= nb) break;
        g_f slab = as_global(s.slab + ((size_t)p * kSoloWG + bt) * s.slab_stride);
        const int row = 16 * bt + i16;
        const bool valid = row < B;
        g_cf rec = ring + (size_t)idx[valid ? row : B - 1] * R.stride;
        ChainNet::StageRegs pend = C.stage_fetch((g_cf)thA, 0, NA.extra_n);
        f32x4 so, ep = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = 4 * q + e, fc = f < O ? f : O - 1;
            so[e] = rec[R.obs_off[0] + fc];
            if (f >= O) so[e] = 0.f;
        }
        if (sac) {
            const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < A) {
                    const unsigned e1 = (unsigned)((valid ? row : B - 1) * am + r);
                    if (a.device_rng) { float n0, n1; normal2(philox4x32_10(a.rng_counter, 0x4000u, e1, key), n0, n1); ep[r] = n1; }      // = draw_kernel's set 1
                    else ep[r] = noise1[e1];
                }
            }
        }
        C.stage_commit(pend);
        SOLO_T(0);
        pend = C.stage_fetch(thC, 0);
        // ---- A: a = tanh(actor(s))   (SAC: a = tanh(mean + std eps) and the row's log pi, SAC.py:70-97)
        f32x4 ah1[2], ah2[2], h2f[kHT], za, an = {0.f, 0.f, 0.f, 0.f}, lsv = {0.f, 0.f, 0.f, 0.f};
        float lpr = 0.f;
        N.forward<true>(so, ah1, ah2, h2f, za, A);                         // (th1 / tx keep the actor's h1 and s for pass C: pass B leaves them alone)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < A) {
                if (sac) {
                    lsv[r] = C.S.ls[r];
                    const float ls = fminf(fmaxf(lsv[r], -20.f), 2.f), sd = expf(ls);
                    const float u = za[r] + sd * ep[r], du = u - za[r];
                    lpr += -(du * du) / (2.f * sd * sd) - ls - kLogSqrt2Pi;
                    lpr -= 2.f * (kLog2 - u - softplus_t(-2.f * u));
                    an[r] = tanhf(u);
                } else {
                    an[r] = tanhf(za[r]);
                }
            }
        }
        SOLO_T(1);
        // ---- B: Q(s, a) and dQ/da through the frozen (already stepped) critic
        const f32x4 xin = critic_input(N, so, an, O, A);
        const float dqv = sac ? -0.5f * invB : -invB;
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};                                   // d loss / d a[r] of this lane's row
        for (int hd = 0; hd < nq; ++hd) {
            C.stage_commit(pend);
            pend = hd + 1 < nq ? C.stage_fetch(thC, hd + 1) : C.stage_fetch((g_cf)thA, 0, NA.extra_n);
            f32x4 h1o[2], h2o[2], z, d2o[2], d1o[2];
            N.forward<false>(xin, h1o, h2o, h2f, z, 1);
            f32x4 dz = {0.f, 0.f, 0.f, 0.f};
            if (valid) { qrow += z[0]; dz[0] = dqv; }
            N.head_bwd<false>(nullptr, dz, h2o, d2o, 1);
            N.hidden_bwd<false>(nullptr, d2o, h1o, d1o);
            const f32x4 dx = N.input_bwd(d1o);                             // column 4q + r of [s | a]; a's columns sit on lanes q = (O + j) / 4
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < A) {
                    const int f = O + r;
                    dq[r] += __shfl(pick4(dx, f & 3), (f >> 2) * 16 + i16, 64);
                }
            }
        }
        SOLO_T(2);
        // ---- C: through a = tanh(.) into the actor; its activations are pass A's (own tiles in registers, h1 / s transposed in LDS)
        C.stage_commit(pend);
        f32x4 dz = {0.f, 0.f, 0.f, 0.f};
        float gls[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < A) {
                    const float av = an[r];
                    if (sac) {                                             // through u = mean + exp(log_std) eps, and alpha log pi
                        const float d = dq[r] * (1.f - av * av) + (alpha * invB) * (2.f * av);
                        const float ls = fminf(fmaxf(lsv[r], -20.f), 2.f);
                        dz[r] = d;
                        gls[r] = d * expf(ls) * ep[r] - alpha * invB;
                    } else {
                        dz[r] = dq[r] * (1.f - av * av);
                    }
                }
            }
        }
        f32x4 d2o[2], d1o[2];
        // head_bwd needs h2's own tiles and the head image: both the actor's again
        N.head_bwd<true>(slab, dz, ah2, d2o, A);
        N.hidden_bwd<true>(slab, d2o, ah1, d1o);
        // log_std's gradient of this row tile (zero outside the clamp [-20, 2], SAC.py:77) behind the head block; Q and log-pi sums
        float gl = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sgl = SoloNet::rows_sum(gls[r]);
            if (i16 == r) gl = (sac && r < A && lsv[r] >= -20.f && lsv[r] <= 2.f) ? sgl : 0.f;
        }
        if (w == 0 && q == 0) slab[kHeadFloats + i16] = gl;
        lp += valid ? lpr : 0.f;
        }   // tiles
        qrow = SoloNet::rows_sum(qrow);
        lp = SoloNet::rows_sum(lp);
        if (tid == 0) { part[b * kSoloPart + 0] = qrow; part[b * kSoloPart + 1] = lp; }
    }
    SOLO_T(3);
    solo_grid_sync(s.bar + (size_t)p * kSoloWG, b, s.bar_base + kSoloWG, s.err, W);
    SOLO_T(4);
    SoloUpdate u;
    u.th = thA; u.mm = mA; u.vv = vA; u.tg = tgA; u.size = NA.size; u.lr = a.actor_lr; u.wd = 0.f; u.soft = 1; u.t_new = t_new;
    const float total = solo_update<W>(s, a, u, p, b, nb, N.red, s.bar_base + kSoloWG SOLO_TARG);
    SOLO_T(7);
    if (b == 0 && tid == 0) {
        float qtot = 0.f, lptot = 0.f;
        for (int k = 0; k < min(nb, W); ++k) {
            qtot += __hip_atomic_load(part + k * kSoloPart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lptot += __hip_atomic_load(part + k * kSoloPart + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        steps[0] = t_new;
        float* sts = D.stats + (size_t)p * ST_COUNT;
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
    // the rollout step's tail: the next select_action reads the WHOLE stepped actor, sixteen workgroups' slices of it — a second
    // flag hand-over (its own flag words; same epoch) in front of workgroup 0's act
    if (st.tail) {
        solo_grid_sync(st.bar2 + (size_t)p * kSoloWG, b, s.bar_base + kSoloWG, s.err, W);
        if (b == 0) solo_step_tail(D, a, st, smem, p);
    }
    solo_leave(st);
}

#define FRL_SOLO_ACTOR(name, wgs)                                                                                                    \
    __global__ __launch_bounds__(256) void name(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st) {       \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                                \
        solo_actor_body<wgs>(*Dp, a, s, st, smem);                                                                                  \
    }
FRL_SOLO_ACTOR(solo_actor_kernel, 16)
FRL_SOLO_ACTOR(solo_actor_w8_kernel, 8)

}  // namespace frl
