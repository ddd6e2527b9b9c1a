// DQN.learn (DQN_file/DQN.py:97-118; Double: DQN_with_tricks.py:263-265) for one learner in ONE launch, register-chained
// (device/chain.hpp): the index draw, y = r + gamma max_a Q_t(s', a) (1 - d), the TD loss on the taken action, backward, clip,
// Adam and the soft target update.  The reference's Q-net is obs -> 128 -> n_actions (DQN.py:32-45): 1.7 k parameters and
// 1.5 k MACs per row, so the three-launch chain of the row-chunk path (draw_kernel -> dqn_grad_kernel -> adam_fused_kernel,
// gradient slabs in between) is launch- and latency-bound at every population size.  Here both nets sit in LDS as
// fragment-ordered images (16 KB), every wave carries 16 rows through them in registers, the weight gradients live in the
// owner lanes' accumulators across the batch, and Adam runs from those registers.
//
// `a.p_count` learners x `a.dqn_split` workgroups: a learner's 64-row chunks are dealt round-robin to its workgroups; with
// more than one, each writes its partial gradient to the learner's slab and the last one to arrive (atomic ticket) adds the
// partials in workgroup order and applies the update — the single learner's latency path (one chunk per workgroup).
//
// Shape: plain or Dueling head ([V ; A], DQN_with_tricks.py:60-79), Double target, PER's importance weights (:276-278); hidden
// 128 (ReLU), obs_dim <= 16, head rows <= 16, batch <= 256.  Noisy and Categorical heads run dqn_grad_kernel / c51_grad_kernel.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain.hpp"
#include "device/update_common.hpp"
#include "device/ppo_timing.hpp"

namespace frl {

namespace {

struct Dqn2Lds {
    lds_f w1[2], w2[2], b1[2], b2[2];      // [0] online, [1] target
    lds_f ea, eb, red;
    lds_f onx;                             // [64][16] obs_next of the learner's first 64 envs (frl_rollout's folded step)
    FRL_LDS int* lidx;
};

struct Dqn2 {
    Dqn2Lds S;
    int tid, l, w, i16, q, fslot, tslot;

    __device__ __forceinline__ void init(float* smem) {
        lds_f p = (lds_f)smem;
        for (int k = 0; k < 2; ++k) { S.w1[k] = p; p += kHT * 256; S.w2[k] = p; p += kHT * 256; }
        S.ea = p; p += kHT * 4 * 256;
        S.eb = p; p += 4 * 256;
        for (int k = 0; k < 2; ++k) { S.b1[k] = p; p += kHid; S.b2[k] = p; p += 16; }
        S.red = p; p += 64;
        S.onx = p; p += 64 * 16;
        S.lidx = (FRL_LDS int*)p; p += 2 * kDqn2Batch;
        tid = threadIdx.x; l = tid & 63; w = __builtin_amdgcn_readfirstlane(tid >> 6); i16 = l & 15; q = l >> 4;
        fslot = (q * 16 + (i16 ^ q)) << 2;
        tslot = (((i16 >> 2) * 16) << 2) + (i16 & 3);
    }

    // engine layout Wk[k][n] (n contiguous), then b[n_pad] -> fragment-ordered images (as ChainNet::stage, two layers); the
    // loads and the LDS stores are separate calls so that a kernel can keep other work between them
    struct StageRegs { f32x4 u1, u1b, u3[2]; float b1, b2; };
    __device__ __forceinline__ StageRegs stage_load(g_cf th, const LayerDesc& L1, const LayerDesc& L2) const {
        StageRegs X;
        const int n = tid & 127;
#pragma unroll
        for (int e = 0; e < 4; ++e) X.u1[e] = th[L1.w_off + (4 * (tid >> 7) + e) * kHid + n];
#pragma unroll
        for (int e = 0; e < 4; ++e) X.u1b[e] = th[L1.w_off + (4 * (2 + (tid >> 7)) + e) * kHid + n];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k4 = (tid >> 4) + 16 * j;
#pragma unroll
            for (int e = 0; e < 4; ++e) X.u3[j][e] = th[L2.w_off + (4 * k4 + e) * 16 + (tid & 15)];
        }
        X.b1 = tid < kHid ? th[L1.b_off + tid] : 0.f;
        X.b2 = tid < 16 ? th[L2.b_off + tid] : 0.f;
        return X;
    }
    __device__ __forceinline__ void stage_store(int k, const StageRegs& X) const {
        const int n = tid & 127;
        { const int qq = tid >> 7; st4(S.w1[k] + (n >> 4) * 256 + ((qq * 16 + ((n & 15) ^ qq)) << 2), X.u1); }
        { const int qq = 2 + (tid >> 7); st4(S.w1[k] + (n >> 4) * 256 + ((qq * 16 + ((n & 15) ^ qq)) << 2), X.u1b); }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k4 = (tid >> 4) + 16 * j, kb = k4 >> 2, qq = k4 & 3;
            st4(S.w2[k] + kb * 256 + ((qq * 16 + ((tid & 15) ^ qq)) << 2), X.u3[j]);
        }
        if (tid < kHid) S.b1[k][tid] = X.b1;
        if (tid < 16) S.b2[k][tid] = X.b2;
    }

    // h1 = relu(W1 x + b1), z = W2 h1 + b2 for this wave's 16 rows (x: B operand, columns 4q + e of row i16)
    __device__ __forceinline__ void forward(int k, const f32x4& xb, f32x4 (&h1)[kHT], f32x4& z) const {
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            const f32x4 wf = ld4((lds_cf)(S.w1[k] + ot * 256 + fslot)), bb = ld4((lds_cf)(S.b1[k] + ot * 16 + 4 * q));
            const f32x4 acc = mfma4(bb, wf, xb);
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[ot][r] = fmaxf(acc[r], 0.f);
        }
        // the head's 32 MFMAs as two accumulator chains (even / odd k-blocks), added at the end
        f32x4 z0 = ld4((lds_cf)(S.b2[k] + 4 * q)), z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < kHT; kb += 2) {
            const f32x4 wa = ld4((lds_cf)(S.w2[k] + kb * 256 + fslot)), wb = ld4((lds_cf)(S.w2[k] + (kb + 1) * 256 + fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], h1[kb][e], z0, 0, 0, 0);
                z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[e], h1[kb + 1][e], z1, 0, 0, 0);
            }
        }
        z = z0 + z1;
    }
    // forward without keeping h1 (target / action-selection passes)
    __device__ __forceinline__ f32x4 forward_z(int k, const f32x4& xb) const {
        f32x4 h1[kHT], z;
        forward(k, xb, h1, z);
        return z;
    }

    __device__ __forceinline__ void put_tile(lds_f E, int ft, const f32x4& t) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) E[(ft * 4 + w) * 256 + tslot + (((4 * q + r) ^ (i16 >> 2)) << 2)] = t[r];
    }
    __device__ __forceinline__ f32x4 get_frag(lds_cf E, int ft, int bb) const { return ld4(E + (ft * 4 + bb) * 256 + fslot); }
};

// the weight-gradient accumulators of one lane (MFMA D layout): layer 1 out = 16 (2w + x) + 4q + r, in = i16;
// head out = 4q + r, in = 16 (2w + x) + i16; bias partials per lane group until the end
struct Dqn2Grad { f32x4 g1[2], g2[2]; float gb1[2], gb2; };

}  // namespace

__global__ __launch_bounds__(256, 2) void dqn_fused_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, DqnStepArgs s) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int nsp = a.dqn_split, unit = blockIdx.x / nsp, sp = blockIdx.x - unit * nsp;
    const int p = a.p0 + unit;
    const NetDesc& N = D.net[0];
    const RecordDesc& R = D.rec;
    const LayerDesc &L1 = N.L[0], &L2 = N.L[1];
    Dqn2 C;
    C.init(smem);
    const Dqn2Lds& S = C.S;
    const int tid = C.tid, l = C.l, w = C.w, i16 = C.i16, q = C.q;
    const int B = a.batch, O = R.obs_dim[0], nA = D.n_discrete;
    const size_t base = (size_t)p * D.learner_stride + D.net_off[0];
    g_f th = as_global(D.theta + base);
    g_f tg = as_global(D.target + base);
    g_f mA = as_global(D.m + base);
    g_f vA = as_global(D.v + base);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_i idx = (g_i)(D.idx + (size_t)p * D.batch_max);
    const float invB = 1.f / (float)B;
    const int nchunks = (B + 63) / 64;
    const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(p + 1);

    // ---- everything that does not depend on the index draw is issued first: both nets' weights, and add() of this vector
    // step (Buffer.add, DQN_file/Buffer.py:28-38): a 16-lane group per env of the learner.  Every workgroup of the learner
    // writes the same rows (it samples from them next, and its own stores are the ones it sees); obs_next is kept in LDS for
    // the select_action at the end.
    PPO_T0();
    Dqn2::StageRegs so, stg;
    if (s.go_flag) {
        // pre-armed, on its own stream behind nothing: the previous step's launch may still be running.  The batch's rows depend on the
        // launch's arguments only (ring size after this step's add, Philox counter): drawn now, into LDS.  Then the previous launch's
        // device word (its parameters are final: release there, acquire here), the images, and the host's doorbell (2 s each, then give up)
        if (a.device_rng) draw_indices((g_i) nullptr, S.lidx, B, a.size, a.rng_counter, 0u, key);
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            int v = __hip_atomic_load(s.dev_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (v - s.dev_wait < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ull) { if (s.err) __hip_atomic_store(s.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                v = __hip_atomic_load(s.dev_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (relaxed polls, then ONE lane's invalidate for the CU, in front of the barrier)
            S.red[21] = __int_as_float(v - s.dev_wait < 0 ? 0 : 1);
        }
        __syncthreads();
        if (__float_as_int(S.red[21]) == 0) return;
        so = C.stage_load(th, L1, L2); stg = C.stage_load(tg, L1, L2);
        C.stage_store(0, so);
        C.stage_store(1, stg);
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            int v = __hip_atomic_load(s.go_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            while (v != s.go_value && v != -1) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > 200000000ull) { v = -1; if (s.err) __hip_atomic_store(s.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                v = __hip_atomic_load(s.go_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            S.red[20] = __int_as_float(v);
        }
        __syncthreads();
        if (__float_as_int(S.red[20]) != s.go_value) return;
        if (a.device_rng && tid < B) idx[tid] = S.lidx[tid];            // (frl_last_indices; B <= 256 on this kernel)
    } else {
        so = C.stage_load(th, L1, L2); stg = C.stage_load(tg, L1, L2);
    }
    if (s.commit) {
        g_f wring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
        const int lane = tid & 15;
        for (int j = tid >> 4; j < s.E; j += kWG / 16) {
            const size_t i = (size_t)p * s.E + j;
            const int row = s.row[i];
            float oc = 0.f, no = 0.f, on = 0.f, sa = 0.f, rw = 0.f;
            unsigned char fl = 0;
            if (lane < s.O) { oc = s.obs_cur[i * s.O + lane]; no = s.next_obs[i * s.O + lane]; on = s.obs_next[i * s.O + lane]; }
            if (lane == 0) { sa = s.store_act[i]; rw = s.reward[i]; fl = s.flags[i]; }
            g_f r = wring + (size_t)row * R.stride;
            if (lane < s.O) { r[R.obs_off[0] + lane] = oc; r[R.nobs_off[0] + lane] = no; }
            if (lane == 0) { r[R.act_off[0]] = sa; r[R.rew_off] = rw; r[R.done_off] = (fl & 1) ? 1.f : 0.f; }
            if (j < 64) S.onx[j * 16 + lane] = on;
        }
    }
    if (!s.go_flag) {
        C.stage_store(0, so);
        C.stage_store(1, stg);
    }
    PPO_T(1);
    // ---- sample(): the batch's row indices (every workgroup of the learner draws the same ones).  The barriers inside wait
    // for the stores above: the gathers below may read the rows just added.
    if (a.device_rng && s.go_flag) {
        __syncthreads();                                               // (drawn in front of the doorbell; the rows just added are stored)
    } else if (a.device_rng) {
        draw_indices(idx, S.lidx, B, a.size, a.rng_counter, 0u, key);
    } else {
        for (int i = tid; i < B; i += kWG) S.lidx[i] = idx[i];
        __syncthreads();
    }
    PPO_T(0);

    struct RowIn { f32x4 xs, xn; float act, rew, done; };
    auto load_row = [&](int c) {
        RowIn X;
        X.xs = f32x4{0.f, 0.f, 0.f, 0.f}; X.xn = X.xs; X.act = 0.f; X.rew = 0.f; X.done = 0.f;
        const int row = c * 64 + 16 * w + i16;
        if (row < B) {
            g_cf rec = ring + (size_t)S.lidx[row] * R.stride;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < O) { X.xs[e] = rec[R.obs_off[0] + 4 * q + e]; X.xn[e] = rec[R.nobs_off[0] + 4 * q + e]; }
            X.act = rec[R.act_off[0]]; X.rew = rec[R.rew_off]; X.done = rec[R.done_off];
        }
        return X;
    };
    // Head tile -> Q values in place: plain head: as is (action j on output j); Dueling (DQN_with_tricks.py:60-79, head rows
    // [V ; A_0..A_nA-1]): Q_j = (V + A_j) - mean_a A on output 1 + j.  o0 = output of action 0.
    const bool duel = D.dueling != 0;
    const int o0 = duel ? 1 : 0;
    auto to_q = [&](f32x4 z) {
        if (duel) {
            float sa = 0.f, v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * q + r;
                if (o >= 1 && o <= nA) sa += z[r];
                if (o == 0) v = z[r];
            }
            sa += lane_xor<16>(sa); sa += lane_xor<32>(sa);
            v += lane_xor<16>(v); v += lane_xor<32>(v);
            const float mean = sa / (float)nA;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = (v + z[r]) - mean;
        }
        return z;
    };
    // value / first index of the row's maximum over the nA actions of a Q tile (outputs 4q + r of row i16)
    auto row_max = [&](const f32x4& z, float& mx, int& best) {
        mx = -INFINITY; best = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * q + r - o0;
            if (j >= 0 && j < nA && z[r] > mx) { mx = z[r]; best = j; }
        }
#pragma unroll
        for (int s = 16; s < 64; s <<= 1) {
            const float om = lane_xor(mx, s);
            const int ob = lane_xor(best, s);
            if (om > mx || (om == mx && ob < best)) { mx = om; best = ob; }
        }
    };
    auto row_pick = [&](const f32x4& z, int j) {      // Q of action j of the row (j uniform over the row's four lanes)
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r == o0 + j) v = z[r];
        v += lane_xor<16>(v);
        v += lane_xor<32>(v);
        return v;
    };

    Dqn2Grad g;
#pragma unroll
    for (int x = 0; x < 2; ++x) { g.g1[x] = f32x4{0.f, 0.f, 0.f, 0.f}; g.g2[x] = g.g1[x]; g.gb1[x] = 0.f; }
    g.gb2 = 0.f;
    float lossp = 0.f;
    g_f tde = as_global(D.td_err + (size_t)p * D.batch_max);
    // PER (DQN_with_tricks.py:276-278): use_isw == 1: `(is_weight * td_error**2).mean()` multiplies a [B] by a [B,1] tensor,
    // i.e. every row carries the MEAN weight; 2: per-row weights
    g_cf isw = as_global(D.isw + (size_t)p * D.batch_max);
    float wbar = 1.f;
    if (a.use_isw == 1) {
        float ws = 0.f;
        for (int i = tid; i < B; i += kWG) ws += isw[i];
        wbar = block_sum(ws, S.red) / (float)B;
    }
    RowIn nxt = load_row(sp);
    for (int c = sp; c < nchunks; c += nsp) {
        const RowIn cur = nxt;
        if (c + nsp < nchunks) nxt = load_row(c + nsp);
        PPO_T(2);
        const int row = c * 64 + 16 * w + i16;
        const bool valid = row < B;
        // ---- y = r + gamma max_a Q_target(s', a) (1 - d); Double: the online net picks a
        float mx; int best;
        const f32x4 zt = to_q(C.forward_z(1, cur.xn));
        if (a.double_dqn) {
            const f32x4 zo = to_q(C.forward_z(0, cur.xn));
            row_max(zo, mx, best);
            mx = row_pick(zt, best);
        } else {
            row_max(zt, mx, best);
        }
        const float y = cur.rew + a.gamma * mx * (1.f - cur.done);
        // ---- Q(s, a), TD loss, head delta (d loss / d Q on the taken action)
        f32x4 h1[kHT], z;
        C.forward(0, cur.xs, h1, z);
        const int at = (int)cur.act;                                   // actions.long() (DQN.py:114)
        const float diff = row_pick(to_q(z), at) - y;
        float lrow, grow;
        td_loss_row(a, diff, lrow, grow);
        const float wr = a.use_isw == 2 ? (valid ? isw[row] : 0.f) : wbar;
        f32x4 dz = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            const float d = wr * grow * invB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * q + r;
                if (!duel) { if (o == at) dz[r] = d; }
                else if (o == 0) dz[r] = d;                                              // dV
                else if (o <= nA) dz[r] = d * ((o - 1 == at ? 1.f : 0.f) - 1.f / (float)nA);   // dA_j through V + A_j - mean A
            }
            if (q == 0) { lossp += wr * lrow; tde[row] = diff; }
        }
        PPO_T(3);
        // ---- backward: head gradient (H1 x dz over the chunk's 64 rows), dH1, layer-1 gradient (X x dH1)
        lds_barrier();                                                 // the previous chunk's readers of ea / eb are done
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) C.put_tile(S.ea, ft, h1[ft]);
        C.put_tile(S.eb, 0, dz);
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 af = C.get_frag(S.eb, 0, bb);
            if (w == 0) g.gb2 += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
            for (int x = 0; x < 2; ++x) g.g2[x] = mfma4(g.g2[x], af, C.get_frag(S.ea, 2 * w + x, bb));
        }
        f32x4 d1[kHT];
#pragma unroll
        for (int it = 0; it < kHT; ++it) {
            f32x4 wa;
#pragma unroll
            for (int e = 0; e < 4; ++e) wa[e] = S.w2[0][it * 256 + C.tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
            const f32x4 acc = mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, wa, dz);
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[it][r] = h1[it][r] > 0.f ? acc[r] : 0.f;
        }
        lds_barrier();
        C.put_tile(S.eb, 0, cur.xs);
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) C.put_tile(S.ea, ft, d1[ft]);
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 bf = C.get_frag(S.eb, 0, bb);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const f32x4 af = C.get_frag(S.ea, 2 * w + x, bb);
                g.gb1[x] += (af[0] + af[1]) + (af[2] + af[3]);
                g.g1[x] = mfma4(g.g1[x], af, bf);
            }
        }
        PPO_T(4);
    }
    // bias partials of the four lane groups (rows 4q .. 4q + 3 of every 16-row block) added up
#pragma unroll
    for (int x = 0; x < 2; ++x) { g.gb1[x] += lane_xor<16>(g.gb1[x]); g.gb1[x] += lane_xor<32>(g.gb1[x]); }
    g.gb2 += lane_xor<16>(g.gb2); g.gb2 += lane_xor<32>(g.gb2);
    float lsum = wave_sum(lossp);

    // ---- several workgroups per learner: partials to the slab; the last to arrive adds them in workgroup order
    const int o1[2] = {L1.w_off + i16 * kHid + 16 * (2 * w) + 4 * q, L1.w_off + i16 * kHid + 16 * (2 * w + 1) + 4 * q};
    const int o2[2] = {L2.w_off + (16 * (2 * w) + i16) * 16 + 4 * q, L2.w_off + (16 * (2 * w + 1) + i16) * 16 + 4 * q};
    const int ob1[2] = {L1.b_off + 16 * (2 * w) + i16, L1.b_off + 16 * (2 * w + 1) + i16};
    const int ob2 = L2.b_off + i16;
    // the update's loads of theta / m / v / target now: in flight under the partial-gradient exchange and the reductions (and
    // all of them before the first store — the compiler cannot prove the four arrays distinct and would otherwise wait for
    // each float4's stores before the next one's loads: six dependent round trips instead of one)
    struct In4 { f32x4 th, m, v, tg; };
    auto load4 = [&](int o) { return In4{ld4((g_cf)(th + o)), ld4((g_cf)(mA + o)), ld4((g_cf)(vA + o)), ld4((g_cf)(tg + o))}; };
    const In4 i1[2] = {load4(o1[0]), load4(o1[1])}, i2[2] = {load4(o2[0]), load4(o2[1])};
    const bool own_b2 = (w == 0 && q == 0);
    float bt[3] = {0.f, 0.f, 0.f}, bm[3] = {0.f, 0.f, 0.f}, bv[3] = {0.f, 0.f, 0.f}, bg[3] = {0.f, 0.f, 0.f};
    const int ob[3] = {ob1[0], ob1[1], ob2};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < 2 ? q == 0 : own_b2) { bt[k] = th[ob[k]]; bm[k] = mA[ob[k]]; bv[k] = vA[ob[k]]; bg[k] = tg[ob[k]]; }
    if (nsp > 1) {
        g_f slab = as_global(D.slab + ((size_t)p * D.S + sp) * D.learner_stride + D.net_off[0]);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            st4(slab + o1[x], g.g1[x]); st4(slab + o2[x], g.g2[x]);
            if (q == 0) slab[ob1[x]] = g.gb1[x];
        }
        if (w == 0 && q == 0) slab[ob2] = g.gb2;
        lds_barrier();
        if (l == 0) S.red[w] = lsum;
        lds_barrier();
        if (tid == 0) D.part[((size_t)p * D.S + sp) * 4] = ((S.red[0] + S.red[1]) + S.red[2]) + S.red[3];
        // publish: every wave's stores acknowledged (sync_stores: each drains its own), then ONE lane's agent-scope release, an explicit wait (hipcc may
        // drop the fence's own) and the ticket; the last arriver: one lane's acquire for the CU.  (All 256 threads running
        // __threadfence() on both sides wrote the L2 back and invalidated it once per thread.)
        sync_stores();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            S.lidx[0] = __hip_atomic_fetch_add(D.ticket + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (S.lidx[0] == nsp - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (S.lidx[0] != nsp - 1) return;
        if (tid == 0) D.ticket[p] = 0;                                 // ready for the next launch
        lsum = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x) { g.g1[x] = f32x4{0.f, 0.f, 0.f, 0.f}; g.g2[x] = g.g1[x]; g.gb1[x] = 0.f; }
        g.gb2 = 0.f;
        // every partial in flight at once (nsp <= 4: dqn_split_for), then added in workgroup order: a runtime loop with the adds inside
        // was one dependent round trip to the memory side per partial — 16 k of this launch's 54 k cycles (tools/dqn2_timing.py 1)
        {
            f32x4 s1[4][2], s2[4][2];
            float sb1[4][2], sb2[4], sl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kc = k < nsp ? k : nsp - 1;
                g_cf sk = as_global(D.slab + ((size_t)p * D.S + kc) * D.learner_stride + D.net_off[0]);
#pragma unroll
                for (int x = 0; x < 2; ++x) { s1[k][x] = ld4(sk + o1[x]); s2[k][x] = ld4(sk + o2[x]); sb1[k][x] = sk[ob1[x]]; }
                sb2[k] = sk[ob2];
                sl[k] = D.part[((size_t)p * D.S + kc) * 4];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < nsp) {
#pragma unroll
                    for (int x = 0; x < 2; ++x) { g.g1[x] += s1[k][x]; g.g2[x] += s2[k][x]; g.gb1[x] += sb1[k][x]; }
                    g.gb2 += sb2[k];
                    if (tid == 0) lsum += sl[k];
                }
            }
        }
        if (l != 0) lsum = 0.f;
        if (w != 0) lsum = 0.f;
    }

    // ---- clip_grad_norm_, Adam (torch's single-tensor order), soft update, losses
    PPO_T(5);
    float ss = 0.f;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        ss += (g.g1[x][0] * g.g1[x][0] + g.g1[x][1] * g.g1[x][1]) + (g.g1[x][2] * g.g1[x][2] + g.g1[x][3] * g.g1[x][3]);
        ss += (g.g2[x][0] * g.g2[x][0] + g.g2[x][1] * g.g2[x][1]) + (g.g2[x][2] * g.g2[x][2] + g.g2[x][3] * g.g2[x][3]);
        if (q == 0) ss += g.gb1[x] * g.gb1[x];
    }
    if (w == 0 && q == 0) ss += g.gb2 * g.gb2;
    ss = wave_sum(ss);
    lds_barrier();
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    if (l == 0) { S.red[w] = ss; S.red[8 + w] = lsum; }
    // the step count reaches every wave through LDS: thread 0 stores the incremented count at the end of the update with no
    // barrier in between, so a late wave reading it from global memory could see the NEXT step's bias corrections
    if (tid == 0) S.red[16] = __int_as_float(steps[0]);
    lds_barrier();
    const float total = sqrtf(((S.red[0] + S.red[1]) + S.red[2]) + S.red[3]);
    const float loss = ((S.red[8] + S.red[9]) + S.red[10]) + S.red[11];
    const int t = __float_as_int(S.red[16]) + 1;
    const float coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    const double bc1 = 1.0 - powi_d((double)a.beta1, t), bc2 = 1.0 - powi_d((double)a.beta2, t);
    const float step = (float)((double)a.critic_lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
    auto adam1 = [&](float gi, float& thi, float& mi, float& vi, float& tgi) {
        gi *= coef;
        if (a.critic_wd != 0.f) gi += a.critic_wd * thi;
        mi = mi + (gi - mi) * w1;
        vi = vi * a.beta2 + (w2 * gi) * gi;
        thi = thi - step * (mi / (sqrtf(vi) / bc2s + a.adam_eps));
        tgi = tgi * tk + thi * a.tau;
    };
    // (s.act: the updated weights also go into the online net's LDS image — this lane's four values sit in four 16-byte
    // slots of its fragment tile, the owner-write pattern of device/chain.hpp)
    auto adam4 = [&](int o, const f32x4& gr, In4 in, lds_f img) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t1 = in.th[r], m1 = in.m[r], v1 = in.v[r], g1 = in.tg[r];
            adam1(gr[r], t1, m1, v1, g1);
            in.th[r] = t1; in.m[r] = m1; in.v[r] = v1; in.tg[r] = g1;
            if (s.act) img[C.tslot + (((4 * q + r) ^ (i16 >> 2)) << 2)] = t1;
        }
        st4(th + o, in.th); st4(mA + o, in.m); st4(vA + o, in.v); st4(tg + o, in.tg);
    };
#pragma unroll
    for (int x = 0; x < 2; ++x) { adam4(o1[x], g.g1[x], i1[x], S.w1[0] + (2 * w + x) * 256); adam4(o2[x], g.g2[x], i2[x], S.w2[0] + (2 * w + x) * 256); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < 2 ? q == 0 : own_b2) {
            adam1(k < 2 ? g.gb1[k] : g.gb2, bt[k], bm[k], bv[k], bg[k]);
            th[ob[k]] = bt[k]; mA[ob[k]] = bm[k]; vA[ob[k]] = bv[k]; tg[ob[k]] = bg[k];
            if (s.act) { if (k < 2) S.b1[0][16 * (2 * w + k) + i16] = bt[k]; else S.b2[0][i16] = bt[k]; }
        }
    }
    if (tid == 0) {
        steps[0] = t;
        float* st = D.stats + (size_t)p * ST_COUNT;
        st[ST_CRITIC_LOSS] = loss * invB;
        st[ST_CRITIC_GNORM] = total;
    }
    PPO_T(6);
    // ---- select_action on what the policy sees next (DQN.py:70-84: argmax_a Q(s, a)) + epsilon-greedy (:307-310), act_kernel's draws
    if (s.act) {
        lds_barrier();
        for (int c = 0; c * 64 < s.E; ++c) {
            const int j = c * 64 + 16 * w + i16;
            f32x4 xb = {0.f, 0.f, 0.f, 0.f};
            if (j < s.E) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < s.O) xb[e] = c == 0 ? S.onx[j * 16 + 4 * q + e] : s.obs_next[((size_t)p * s.E + j) * s.O + 4 * q + e];
            }
            float mx; int best;
            row_max(to_q(C.forward_z(0, xb)), mx, best);
            if (j < s.E && q == 0) {
                const Philox4 u = philox4x32_10(s.act_counter, 0x9000u, (unsigned)j, key);
                if (u01(u.z) <= s.epsilon) best = (int)uniform_index(u, (unsigned)nA);
                s.act_out[(size_t)p * s.E + j] = (float)best;
                s.env_out[(size_t)p * s.E + j] = (float)best;
            }
        }
    }
    // the observation the next vector step starts from (every workgroup of the learner has read obs_cur before its ticket)
    if (s.commit) {
        float* oc = const_cast<float*>(s.obs_cur);
        for (int k = tid; k < s.E * s.O; k += kWG) {
            const int j = k / s.O, col = k - j * s.O;
            oc[(size_t)p * s.E * s.O + k] = j < 64 ? S.onx[j * 16 + col] : s.obs_next[(size_t)p * s.E * s.O + k];
        }
    }
    if (s.act && s.done_flag) {
        sync_stores();                                                 // every wave's env_out stores have been issued and acknowledged
        if (tid == 0) {
            // this learner's actions are out; the last learner to get here flags the host: system-scope release + an explicit wait
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (one learner: it is the last one — no ticket round trip)
            if (a.p_count == 1 || atomicAdd(D.ticket + D.P, 1) == a.p_count - 1) {
                if (a.p_count > 1) { D.ticket[D.P] = 0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_store(s.done_flag, s.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (the host first: it is the one waited for)
                if (s.dev_done) __hip_atomic_store(s.dev_done, s.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    PPO_T(7);
    PPO_TDUMP();
}

}  // namespace frl
