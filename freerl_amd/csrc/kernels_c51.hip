// Categorical DQN (C51) update for one row chunk — DQN_with_tricks.py:82-158 (Categorical net, projection_dist) and
// :248-260 (cross-entropy loss, PER error).  Head rows: [a * atoms + i] (plain) or [V_i ; atoms + a * atoms + i] (Dueling:
// logits = V + A - mean_a A).  Forwards and backward are the shared MFMA layer code; the distribution arithmetic between
// them is vector work, one thread per (row, action) pair or per row.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/c51.hpp"
#include "device/net.hpp"

namespace frl {

__global__ __launch_bounds__(256, FRL_GRAD_WGS) void c51_grad_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EngineDesc& D = *Dp;
    const int group = 8 * ns, gq = blockIdx.x / group, lq = blockIdx.x - gq * group;
    const int unit = gq * 8 + (lq & 7), sl = lq >> 3;
    if (unit >= a.p_count) return;
    const int p = a.p0 + unit;
    const NetDesc& N = D.net[0];
    const RecordDesc& R = D.rec;
    const Lds S = carve_lds(smem, D.rc, D.hidden, D.lds_kin_pad, D.lds_out_pad, D.lds_batch_pad, D.lds_act_pad, D.lds_hbufs);
    const int rc = D.rc, B = a.batch, nl = N.n_layers;
    const int nchunks = (B + rc - 1) / rc, ck0 = sl * D.cps, ck1 = min(ck0 + D.cps, nchunks);
    const size_t base = (size_t)p * D.learner_stride + D.net_off[0];
    g_cf eff = D.noisy ? as_global(D.theta_eff + (size_t)p * 3 * D.learner_stride + D.net_off[0]) : nullptr;
    g_cf theta_next = D.noisy ? eff : as_global(D.theta + base);
    g_cf target = D.noisy ? eff + D.learner_stride : as_global(D.target + base);
    g_cf theta = D.noisy ? eff + 2 * (size_t)D.learner_stride : as_global(D.theta + base);
    g_f slab = as_global(D.slab + ((size_t)p * D.S + sl) * D.learner_stride + D.net_off[0]);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    const int O = R.obs_dim[0], nA = D.n_discrete, atoms = D.c51_atoms, npad = N.L[nl - 1].n_pad, k0pad = N.L[0].k_pad;
    const bool duel = D.dueling != 0;
    const float vmin = D.c51_vmin, vmax = D.c51_vmax, dz = (vmax - vmin) / (float)(atoms - 1);
    lds_f m = S.abuf;                    // [rc][ap] projected target distribution
    lds_f qb = S.dabuf;                  // [rc][ap] q values / scratch
    const int ap = S.ap;
    float lossp = 0.f;
    FRL_PHASE_INIT(S);
    for (int ck = ck0; ck < ck1; ++ck) {            // the row chunks of this workgroup, their gradients summed in its slab
    const bool first = (ck == ck0);
    const int gs = first ? (D.cps > 1 ? GS_STORE : GS_STREAM) : GS_ADD;
    const int r0 = ck * rc, nv = min(rc, B - r0);
    g_ci idx = as_global_i(D.idx + (size_t)p * D.n_agents * D.batch_max + r0);
    if (!first) FRL_PHASE(S);

    // q[r][a] of the logits in outb -> qb, then argmax into S.y (one thread per (row, action), then per row)
    auto pick_action = [&]() {
        const int lb = c51_combine(S.outb, S.op, nv, nA, atoms, duel);
        // two adjacent lanes per (row, action), half the support each (one thread per pair left half the workgroup idle for the
        // longest vector phase of the chunk; a whole wave per pair was measured 2x slower: its reductions are ds_bpermute round
        // trips).  Only the argmax over actions is taken from these values: hardware exp.
        const int half = (atoms + 1) / 2, npair = nv * nA;
        for (int e0 = 0; e0 < 2 * npair; e0 += kWG) {
            const int e = e0 + threadIdx.x, pr = min(e >> 1, npair - 1), hf = e & 1;
            const int r = pr / nA, act = pr - r * nA;
            lds_cf lg = S.outb + r * S.op + lb + act * atoms;
            const int i0 = hf * half, i1 = hf ? atoms : half;
            float mx = lg[i0];
            for (int i = i0 + 1; i < i1; ++i) mx = fmaxf(mx, lg[i]);
            mx = fmaxf(mx, lane_xor<1>(mx));
            float sum = 0.f, zsum = 0.f;
            for (int i = i0; i < i1; ++i) {
                const float ex = __expf(lg[i] - mx);
                sum += ex;
                zsum += ex * (vmin + dz * (float)i);
            }
            sum += lane_xor<1>(sum);
            zsum += lane_xor<1>(zsum);
            if (e < 2 * npair && hf == 0) qb[r * ap + act] = zsum / sum;
        }
        FRL_PHASE(S);
        for (int r = threadIdx.x; r < nv; r += kWG) {
            int best = 0;
            for (int jj = 1; jj < nA; ++jj) if (qb[r * ap + jj] > qb[r * ap + best]) best = jj;
            S.y[r] = (float)best;
        }
        FRL_PHASE(S);
        return lb;
    };
    // ---- next action: argmax_a q(s', a) by the online net (Double, :141-143) or by the target net itself (:145)
    gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.nobs_off[0], O, 0);
    zero_cols(S.xin, S.xp, rc, O, k0pad);
    FRL_PHASE(S);
    if (a.double_dqn) {
        mlp_fwd(N, 0, nl, theta_next, S, ACT_NONE);
        pick_action();
    }
    mlp_fwd(N, 0, nl, target, S, ACT_NONE);
    int lb = a.double_dqn ? c51_combine(S.outb, S.op, nv, nA, atoms, duel) : pick_action();
    // ---- projection of the target distribution (projection_dist :147-158).  qb row = next_dist (one wave per row), then
    constexpr int kRows = 8;                           // rows per wave and pass (rc <= 32 rows: one pass)
    for (int rb = 0; rb < rc; rb += kRows * kWaves) {
        lds_cf lg[kRows];
        float pl[kRows], ql[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            const int r = min(rb + wave_id() + kWaves * k, nv - 1);         // rows past nv recompute a valid one, unused
            lg[k] = S.outb + r * S.op + lb + (int)S.y[r] * atoms;
        }
        c51_softmax_wave_n<kRows>(lg, atoms, vmin, dz, pl, ql);
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            const int r = rb + wave_id() + kWaves * k;
            if (r < nv && lane_id() < atoms) qb[r * ap + lane_id()] = pl[k];
        }
    }
    FRL_PHASE(S);
    // one thread per row walks the support: every lower-bin index_add_ in atom order, then every upper-bin one (:155-156).
    // Measured and not taken (profiles/README.md): one thread per (row, target bin) scanning all sources (51x the arithmetic);
    // per-source tables + one thread per (row, bin) finding its sources' run by bisection (28 k -> 35-65 k cycles: dependent LDS
    // reads); tables + a single register walk per row, 8 rows per wave (35-40 k: a dependent read per run for the previous
    // run's upper parts, divergent lanes).
    for (int r = threadIdx.x; r < nv; r += kWG) {
        lds_cf ndl = qb + r * ap;
        g_cf rec = ring + (size_t)idx[r] * R.stride;
        const float rew = rec[R.rew_off], done = rec[R.done_off];
        lds_f mr = m + r * ap;
        for (int i = 0; i < atoms; ++i) mr[i] = 0.f;
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < atoms; ++i) {
                const float z = vmin + dz * (float)i;
                const float tz = fminf(fmaxf(rew + a.gamma * z * (1.f - done), vmin), vmax);
                const float b = (tz - vmin) / dz;
                const float lf = floorf(b), uf = ceilf(b);
                const int l = (int)lf, u = (int)uf;
                if (pass == 0) mr[l] += (uf + (l == u ? 1.f : 0.f) - b) * ndl[i];
                else mr[u] += (b - lf) * ndl[i];
            }
    }
    FRL_PHASE(S);
    // ---- current distribution of the taken action, cross-entropy against m, head delta
    gather_cols(S.xin, S.xp, rc, nv, idx, ring, R.stride, R.obs_off[0], O, 0);
    zero_cols(S.xin, S.xp, rc, O, k0pad);
    FRL_PHASE(S);
    mlp_fwd(N, 0, nl, theta, S, ACT_NONE);
    lb = c51_combine(S.outb, S.op, nv, nA, atoms, duel);
    g_cf isw = as_global(D.isw + (size_t)p * D.batch_max + r0);
    g_f tde = as_global(D.td_err + (size_t)p * D.batch_max + r0);
    for (int rb = 0; rb < rc; rb += kRows * kWaves) {          // one wave per row (lane = atom), kRows rows interleaved
        const int l = lane_id();
        lds_cf lg[kRows];
        int at[kRows];
        float pi[kRows], ql[kRows], ce[kRows], gi[kRows], gp[kRows], w[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            const int r = min(rb + wave_id() + kWaves * k, nv - 1);
            at[k] = (int)ring[(size_t)idx[r] * R.stride + R.act_off[0]];
            lg[k] = S.outb + r * S.op + lb + at[k] * atoms;
            w[k] = a.use_isw ? isw[r] : 1.f;                    // `is_weight.reshape(-1,1)`: per-row weights here (:256)
        }
        c51_softmax_wave_n<kRows>(lg, atoms, vmin, dz, pi, ql);
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            const int r = min(rb + wave_id() + kWaves * k, nv - 1);
            const float mi = l < atoms ? m[r * ap + l] : 0.f;
            const bool inside = pi[k] > 1e-5f && pi[k] < 1.f - 1e-5f;
            ce[k] = l < atoms ? mi * logf(fminf(fmaxf(pi[k], 1e-5f), 1.f - 1e-5f)) : 0.f;
            gi[k] = (l < atoms && inside) ? -(mi * w[k] / (float)B) / pi[k] : 0.f;      // d loss / d p_i
            gp[k] = gi[k] * pi[k];
        }
        wave_sum_n<kRows>(ce);
        wave_sum_n<kRows>(gp);
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            const int r = rb + wave_id() + kWaves * k;
            if (r < nv) {
                if (l < atoms) qb[r * ap + l] = pi[k] * (gi[k] - gp[k]);     // softmax backward: d loss / d logit_i
                if (l == 0) {
                    lossp += -w[k] * ce[k];
                    tde[r] = ce[k];                                          // `error` of :255 (PER priorities use |error|)
                    S.y[r] = (float)at[k];
                }
            }
        }
    }
    FRL_PHASE(S);
    // head delta from the per-atom logit deltas in qb, all threads: plain head: the taken action's block; Dueling:
    // dV_i = d_i, dA_b,i = d_i (delta_b,at - 1/nA)
    for (int j = threadIdx.x; j < npad; j += kWG) {             // a head column per thread: its (action, atom) once, then down the rows
        const int jj = duel ? j - atoms : j;
        const int b = jj >= 0 ? jj / atoms : -1, i = jj >= 0 ? jj - b * atoms : j;
        const bool live = duel ? (j < atoms + nA * atoms) : (j < nA * atoms);
        for (int r = 0; r < rc; ++r) {
            float v = 0.f;
            if (r < nv && live) {
                const int at = (int)S.y[r];
                const float d = qb[r * ap + i];
                if (!duel) v = (b == at) ? d : 0.f;
                else if (b < 0) v = d;                                                   // dV_i
                else v = d * ((b == at ? 1.f : 0.f) - 1.f / (float)nA);                 // dA_b,i
            }
            S.outb[r * S.op + j] = v;
        }
    }
    FRL_PHASE(S);
    mlp_bwd(N, 0, nl, theta, slab, S, gs, false, 0, 0);
    }
    FRL_PHASE_DUMP(S, 2);
    const float ls = block_sum(lossp, S.red);
    if (threadIdx.x == 0) D.part[((size_t)p * D.n_agents * D.S + sl) * 4] = ls;
}

}  // namespace frl
