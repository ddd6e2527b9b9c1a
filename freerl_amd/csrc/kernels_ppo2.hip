// PPO minibatch updates with the parameters ON CHIP for the whole learn() (PPO_file/PPO_with_tricks.py:318-351).
//
// ppo_update_kernel (kernels_ppo.hip) streams theta / m / v / grad through global memory on every one of the K_epochs x
// n_minibatch steps and synchronises its four waves around every layer (tools/ppo_timing.py: 166 k cycles per step for
// ~32 k cycles of MFMA work, 27 % of them clip + Adam, 12 % per-row surrogate math on one thread per row).  This kernel is
// the same arithmetic for the reference's standard shape (three layers, hidden 128, obs_dim <= 32, head <= 16 outputs,
// torch Adam) laid out for one workgroup that never leaves the CU:
//
//   * theta lives in LDS in MFMA-fragment order, Adam's m and v and the step's gradient in REGISTERS of the lane that owns
//     the element; nothing but the gathered rows is read from global memory between the first and the last step.
//   * every wave carries 16 rows of the minibatch through the whole forward / backward chain in registers: in the
//     transposed formulation Z[out][row] = W[out][in] H[in][row] the 16x16 output tile of one layer IS the B operand of the
//     next (v_mfma_f32_16x16x4_f32: D[4q+r][lane&15] -> B[4q+e][lane&15]), so activations are never stored, and no barrier
//     separates the layers.  The per-row surrogate / value delta is computed where the head tile lands: 4 outputs per lane.
//   * weight gradients contract over the rows, which live on different waves: activations and deltas of a layer are
//     exchanged once through LDS (fragment order again) and wave w accumulates the gradient of ITS quarter of every weight
//     matrix over all 64 rows — the quarter whose m, v it holds — so Adam runs straight from the accumulators.
//   * LDS images are swizzled (16-byte slot (q, f) at q*16 + (f ^ q)) so that the forward's ds_read_b128 fragments, the
//     backward's transposed ds_read_b32 fragments and the owners' ds_write_b32 are all bank-conflict free.
//
// One workgroup per (learner, net): blockIdx.y = 0 actor (Gaussian or Categorical head), 1 critic.  Other shapes / the Beta
// actor / the cautious AdamW stay on ppo_update_kernel.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/chain.hpp"
#include "device/ppo_timing.hpp"

#ifndef FRL_PPO2_REFRESH
#define FRL_PPO2_REFRESH 1
#endif

namespace frl {

namespace {

struct Ppo2Lds {
    lds_f w1, w2, w3, b1, b2, b3, ls, ea, eb, red;
};

template <int K0B>
__device__ __forceinline__ Ppo2Lds ppo2_carve(float* smem) {
    Ppo2Lds S;
    lds_f p = (lds_f)smem;
    S.w1 = p; p += kHT * K0B * 256;
    S.w2 = p; p += kHT * kHT * 256;
    S.w3 = p; p += kHT * 256;
    S.ea = p; p += kHT * 4 * 256;
    S.eb = p; p += kHT * 4 * 256;
    S.b1 = p; p += kHid;
    S.b2 = p; p += kHid;
    S.b3 = p; p += 16;
    S.ls = p; p += 16;
    S.red = p; p += 96;
    return S;
}

}  // namespace

template <int K0B, int HACT>
__device__ __forceinline__ void ppo_update_v2_body(const EngineDesc& D, const PpoArgs& a, float* smem) {
    const int p = blockIdx.x, T = a.horizon, mb = a.minibatch;
    const bool critic = (blockIdx.y == 1);
    const bool discrete = D.n_discrete > 0;
    const NetDesc& N = D.net[critic ? 1 : 0];
    const RecordDesc& R = D.rec;
    const Ppo2Lds S = ppo2_carve<K0B>(smem);
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // i16 / q are re-declared "changed" at every phase boundary below (an empty asm the compiler cannot see through): every LDS
    // address built on them then lives exactly as long as its phase.  Left as loop invariants, hipcc hoists ~30 address registers
    // to the top of the kernel and spills them around the MFMA chains (round 6: device/chain_net.hpp, ChainNetT::lanes).
    int i16 = l & 15, q = l >> 4;
#if FRL_PPO2_REFRESH
#define PPO2_REFRESH() asm volatile("" : "+v"(i16), "+v"(q))
#else
#define PPO2_REFRESH() do {} while (0)
#endif
    const size_t off = (size_t)p * D.learner_stride + D.net_off[critic ? 1 : 0];
    g_f th_g = as_global(D.theta + off);
    g_f m_g = as_global(D.m + off);
    g_f v_g = as_global(D.v + off);
    g_cf ring = as_global(D.replay + (size_t)p * D.capacity * R.stride);
    g_cf adv = as_global(a.adv + (size_t)p * T);
    g_cf vt = as_global(a.vtarget + (size_t)p * T);
    g_cf bn = D.obs_norm_on ? as_global(D.obsnorm + (size_t)p * (1 + 3 * R.obs_dim[0])) : nullptr;
    const int O = R.obs_dim[0], A = critic ? 1 : (discrete ? D.n_discrete : R.act_dim[0]), logp_col = R.extra_off;
    const LayerDesc &L1 = N.L[0], &L2 = N.L[1], &L3 = N.L[2];
    const int np3 = L3.n_pad;                                   // 16
    int* steps = D.steps + (size_t)p * (kMaxNets + 1);
    int t_step = steps[critic ? 1 : 0];
    const int n_mb = (T + mb - 1) / mb;
    float* trace = a.trace + (size_t)p * a.k_epochs * n_mb * 2;
    constexpr float kHalfLog2PiPlusHalf = 1.41893853320467274178f, kLogSqrt2Pi_ = 0.91893853320467274178f;
    const float lr = critic ? a.critic_lr : a.actor_lr;

    // ---- parameters -> LDS (fragment order); engine layout: Wk[k][n] (n contiguous) then b[n_pad]
    for (int e = tid; e < L1.k_pad * kHid; e += kWG) {
        const int k = e / kHid, n = e - k * kHid;
        S.w1[frag_dw((n >> 4) * K0B + (k >> 4), n & 15, k & 15)] = th_g[L1.w_off + e];
    }
    for (int e = tid; e < kHid * kHid; e += kWG) {
        const int k = e / kHid, n = e - k * kHid;
        S.w2[frag_dw((n >> 4) * kHT + (k >> 4), n & 15, k & 15)] = th_g[L2.w_off + e];
    }
    for (int e = tid; e < kHid * np3; e += kWG) {
        const int k = e / np3, n = e - k * np3;
        S.w3[frag_dw(k >> 4, n, k & 15)] = th_g[L3.w_off + e];
    }
    if (tid < kHid) { S.b1[tid] = th_g[L1.b_off + tid]; S.b2[tid] = th_g[L2.b_off + tid]; }
    if (tid < 16) {
        S.b3[tid] = th_g[L3.b_off + tid];
        S.ls[tid] = (N.extra_n > 0 && tid < N.extra_n) ? th_g[N.extra_off + tid] : 0.f;
    }

    // ---- the elements this lane owns (MFMA D layout of the weight-gradient tiles: out = 16*ot + 4q + r, in = 16*kt + i16):
    // layer 2: ot in {2w, 2w+1} x kt 0..7; layer 1: ot in {2w, 2w+1} x kt < K0B; head: kt in {2w, 2w+1}; biases b1/b2 of the
    // wave's 32 outputs on lanes q == 0 (out = 16*ot + i16); b3 and log_std on wave 0
    float m2[2][kHT][4], v2[2][kHT][4], m1[2][K0B][4], v1[2][K0B][4], m3[2][4], v3[2][4];
    float mb1[2], vb1[2], mb2[2], vb2[2], mb3 = 0.f, vb3 = 0.f, mls = 0.f, vls = 0.f;
    auto gofs = [&](const LayerDesc& L, int out, int in) { return L.w_off + in * L.n_pad + out; };
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int ot = 2 * w + x;
#pragma unroll
        for (int kt = 0; kt < kHT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = gofs(L2, ot * 16 + 4 * q + r, kt * 16 + i16);
                m2[x][kt][r] = m_g[o]; v2[x][kt][r] = v_g[o];
            }
#pragma unroll
        for (int kt = 0; kt < K0B; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int in = kt * 16 + i16;
                const bool ok = in < L1.k_pad;
                const int o = gofs(L1, ot * 16 + 4 * q + r, ok ? in : 0);
                m1[x][kt][r] = ok ? m_g[o] : 0.f; v1[x][kt][r] = ok ? v_g[o] : 0.f;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = gofs(L3, 4 * q + r, (2 * w + x) * 16 + i16);
            m3[x][r] = m_g[o]; v3[x][r] = v_g[o];
        }
        mb1[x] = m_g[L1.b_off + ot * 16 + i16]; vb1[x] = v_g[L1.b_off + ot * 16 + i16];
        mb2[x] = m_g[L2.b_off + ot * 16 + i16]; vb2[x] = v_g[L2.b_off + ot * 16 + i16];
    }
    if (w == 0) {
        mb3 = m_g[L3.b_off + i16]; vb3 = v_g[L3.b_off + i16];
        if (N.extra_n > 0 && i16 < N.extra_n) { mls = m_g[N.extra_off + i16]; vls = v_g[N.extra_off + i16]; }
    }
    __syncthreads();

    // the rows of one 64-row chunk as this lane needs them: its column's observation (B operand of layer 1) and, for the
    // head's four outputs it will hold, the stored action / old log-probs; loaded one step AHEAD (before the reductions and
    // Adam of the current step) so that the dependent perm -> record round trips are off the critical path
    struct Rows { f32x4 xb[K0B]; f32x4 act, lpo; float tgt; int valid; };
    auto load_rows = [&](int k, int s, int r0, int m) {
        Rows X;
        const int row = r0 + 16 * w + i16;
        X.valid = row < m ? 1 : 0;
        const int ridx = X.valid ? as_global_i(a.perm + ((size_t)p * a.k_epochs + k) * T)[s + row] : 0;
        g_cf rec = ring + (size_t)ridx * R.stride;
#pragma unroll
        for (int kb = 0; kb < K0B; ++kb) {
            const int c0 = kb * 16 + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = 0.f;
                if (X.valid && c0 + e < O) {
                    x = rec[R.obs_off[0] + c0 + e];
                    if (bn) x = (x - bn[1 + c0 + e]) / (bn[1 + 2 * O + c0 + e] + 1e-8f);     // Batch_ObsNorm (normalization.py:78-84)
                }
                X.xb[kb][e] = x;
            }
        }
        X.act = f32x4{0.f, 0.f, 0.f, 0.f}; X.lpo = f32x4{0.f, 0.f, 0.f, 0.f}; X.tgt = 0.f;
        if (X.valid) {
            if (critic) X.tgt = vt[ridx];
            else {
                X.tgt = adv[ridx];
                if (discrete) { X.act[0] = rec[R.act_off[0]]; X.lpo[0] = rec[logp_col]; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * q + r < A) { X.act[r] = rec[R.act_off[0] + 4 * q + r]; X.lpo[r] = rec[logp_col + 4 * q + r]; }
                }
            }
        }
        return X;
    };
    double pw1 = powi_d((double)a.beta1, t_step), pw2 = powi_d((double)a.beta2, t_step);      // beta^t, advanced per step
    float last_loss = 0.f;
    Rows nxt = load_rows(0, 0, 0, min(mb, T));
    PPO_T0();
    for (int k = 0; k < a.k_epochs; ++k) {
        for (int s = 0; s < T; s += mb) {
            const int m = min(mb, T - s);
            const float invm = 1.f / (float)m;
            f32x4 g2[2][kHT], g1[2][K0B], g3[2];
            float gb1[2] = {0.f, 0.f}, gb2[2] = {0.f, 0.f}, gb3 = 0.f, lossp = 0.f;
            f32x4 gls4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int x = 0; x < 2; ++x) {
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt) g2[x][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < K0B; ++kt) g1[x][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                g3[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int r0 = 0; r0 < m; r0 += 64) {
                // ---------------------------------------------------------------- this wave's 16 rows, in registers
                PPO2_REFRESH();
                const Rows cur = (r0 == 0) ? nxt : load_rows(k, s, r0, m);
                const bool valid = cur.valid != 0;
                f32x4 xb[K0B];                                     // B operand of layer 1: X[in = 16 kb + 4q + e][row]
#pragma unroll
                for (int kb = 0; kb < K0B; ++kb) xb[kb] = cur.xb[kb];
                // layer 1
                f32x4 h1[kHT], h2[kHT];
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot) {
                    f32x4 acc = ld4((lds_cf)(S.b1 + ot * 16 + 4 * q));
#pragma unroll
                    for (int kb = 0; kb < K0B; ++kb)
                        acc = mfma4(acc, ld4((lds_cf)(S.w1 + (ot * K0B + kb) * 256 + ((q * 16 + (i16 ^ q)) << 2))), xb[kb]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[ot][r] = hact_fwd<HACT>(acc[r]);
                }
                // layer 2
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot) {
                    f32x4 acc = ld4((lds_cf)(S.b2 + ot * 16 + 4 * q));
#pragma unroll
                    for (int kb = 0; kb < kHT; ++kb)
                        acc = mfma4(acc, ld4((lds_cf)(S.w2 + (ot * kHT + kb) * 256 + ((q * 16 + (i16 ^ q)) << 2))), h1[kb]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[ot][r] = hact_fwd<HACT>(acc[r]);
                }
                // head: z[out = 4q + r][row]
                f32x4 z = ld4((lds_cf)(S.b3 + 4 * q));
#pragma unroll
                for (int kb = 0; kb < kHT; ++kb)
                    z = mfma4(z, ld4((lds_cf)(S.w3 + kb * 256 + ((q * 16 + (i16 ^ q)) << 2))), h2[kb]);

                PPO_T(0);
                PPO2_REFRESH();
                // ---------------------------------------------------------------- per-row loss and head delta dz[out][row]
                f32x4 dz = {0.f, 0.f, 0.f, 0.f};
                if (critic) {                                      // mse(v_target[idx], V(obs[idx])) (:349-351)
                    if (valid && q == 0) {
                        const float diff = z[0] - cur.tgt;
                        dz[0] = 2.f * diff * invm;
                        lossp += diff * diff;
                    }
                } else if (!discrete) {
                    // Gaussian: mean = tanh(z), per-dimension log-probs summed in the reference's order (:331-342): lane group q
                    // holds dimensions 4q..4q+3, the running sums pass from group to group
                    float lp_now = 0.f, lp_old = 0.f;
                    f32x4 mean, dm, var;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * q + r;
                        const float lsr = fminf(fmaxf(S.ls[c & 15], -20.f), 2.f);
                        mean[r] = tanhf(z[r]);
                        var[r] = expf(2.f * lsr);
                        dm[r] = (valid && c < A) ? cur.act[r] - mean[r] : 0.f;
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {                   // sequential over lane groups: dims 0..3, 4..7, ...
                        const float in_now = __shfl(lp_now, i16 + 16 * ((g + 3) & 3), 64), in_old = __shfl(lp_old, i16 + 16 * ((g + 3) & 3), 64);
                        if (q == g) {
                            if (g > 0) { lp_now = in_now; lp_old = in_old; }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 4 * g + r;
                                if (c < A) {
                                    const float lsr = fminf(fmaxf(S.ls[c], -20.f), 2.f), sd = expf(lsr);
                                    lp_now += -(dm[r] * dm[r]) / (2.f * sd * sd) - lsr - kLogSqrt2Pi_;
                                    lp_old += cur.lpo[r];
                                }
                            }
                        }
                    }
                    lp_now = __shfl(lp_now, i16 + 48, 64); lp_old = __shfl(lp_old, i16 + 48, 64);
                    float coef = 0.f;
                    if (valid) {
                        const float ratio = expf(lp_now - lp_old), Ar = cur.tgt;
                        const float s1 = ratio * Ar, s2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * Ar;
                        if (q == 0) lossp += -fminf(s1, s2);
                        coef = (s1 <= s2 ? Ar : 0.f) * (-invm) * ratio;      // d loss / d sum_c logp_now
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (4 * q + r < A) {
                            dz[r] = coef * dm[r] / var[r] * (1.f - mean[r] * mean[r]);      // through mean = tanh(z)
                            gls4[r] += coef * (dm[r] * dm[r] / var[r] - 1.f);
                        }
                    }
                } else {
                    // Categorical over the A logits of the row, spread 4 per lane group (:333-336; PPO.py:257 for logits=)
                    float mx = -3.0e38f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * q + r < A) mx = fmaxf(mx, z[r]);
                    mx = fmaxf(mx, lane_xor<16>(mx)); mx = fmaxf(mx, lane_xor<32>(mx));
                    f32x4 ex = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * q + r < A) ex[r] = expf(z[r] - mx);
                    // sums in class order (groups pass their running sum on), like a sequential loop over the classes
                    auto seq_sum = [&](const f32x4& t) {
                        float run = 0.f;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float in = __shfl(run, i16 + 16 * ((g + 3) & 3), 64);
                            if (q == g) {
                                if (g > 0) run = in;
#pragma unroll
                                for (int r = 0; r < 4; ++r) if (4 * g + r < A) run += t[r];
                            }
                        }
                        return __shfl(run, i16 + 48, 64);
                    };
                    const float sum = seq_sum(ex);
                    f32x4 pc, lg;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pc[r] = ex[r] / sum;
                    const float psum = seq_sum(pc);
                    const float lse = logf(sum);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        lg[r] = D.cat_logits ? (z[r] - mx) - lse : logf(fminf(fmaxf(pc[r] / psum, 1.1920929e-07f), 1.f - 1.1920929e-07f));
                    f32x4 et;
#pragma unroll
                    for (int r = 0; r < 4; ++r) et[r] = (4 * q + r < A) ? lg[r] * pc[r] : 0.f;
                    const float entr = -seq_sum(et);
                    const int ar = (int)cur.act[0];                 // every lane group loaded the row's action index
                    float lp_now = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * q + r == ar) lp_now = lg[r];
                    lp_now += lane_xor<16>(lp_now); lp_now += lane_xor<32>(lp_now);      // one group holds it, the others 0
                    if (valid) {
                        const float ratio = expf(lp_now - cur.lpo[0]), Ar = cur.tgt;
                        const float s1 = ratio * Ar, s2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * Ar;
                        if (q == 0) lossp += -fminf(s1, s2) - a.ent_coef * entr;
                        const float coef = (s1 <= s2 ? Ar : 0.f) * (-invm) * ratio;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (4 * q + r < A)
                                dz[r] = coef * ((4 * q + r == ar ? 1.f : 0.f) - pc[r]) + (a.ent_coef * invm) * pc[r] * (lg[r] + entr);
                    }
                }

                PPO_T(1);
                PPO2_REFRESH();
                // ---------------------------------------------------------------- exchange 1: H2 and dz -> head gradient
                lds_barrier();                                     // the previous chunk's / step's readers of ea / eb are done
                const int wslot = (((i16 >> 2) * 16) << 2) + (i16 & 3);          // + ((f16 ^ (i16 >> 2)) << 2)
                auto put_tile = [&](lds_f E, int ft, const f32x4& t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) E[(ft * 4 + w) * 256 + wslot + (((4 * q + r) ^ (i16 >> 2)) << 2)] = t[r];
                };
                auto get_frag = [&](lds_cf E, int ft, int bb) { return ld4(E + (ft * 4 + bb) * 256 + ((q * 16 + (i16 ^ q)) << 2)); };
#pragma unroll
                for (int ft = 0; ft < kHT; ++ft) put_tile(S.ea, ft, h2[ft]);
                put_tile(S.eb, 0, dz);
                lds_barrier();
                // dW3[out][in] += sum_rows dz[out][row] * H2[in][row]: this wave's in-tiles 2w, 2w+1; A = dz, B = H2^T
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const f32x4 af = get_frag(S.eb, 0, bb);
                    if (w == 0) gb3 += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
                    for (int x = 0; x < 2; ++x) g3[x] = mfma4(g3[x], af, get_frag(S.ea, 2 * w + x, bb));
                }
                // dH2 = W3^T dz, through the activation: dz2[in][row]   (A = W3^T: transposed fragment reads)
                PPO2_REFRESH();
                f32x4 d2[kHT];
#pragma unroll
                for (int it = 0; it < kHT; ++it) {
                    f32x4 wa;
#pragma unroll
                    for (int e = 0; e < 4; ++e) wa[e] = S.w3[it * 256 + ((((i16 >> 2) * 16 + ((4 * q + e) ^ (i16 >> 2)))) << 2) + (i16 & 3)];
                    f32x4 acc = mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, wa, dz);
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2[it][r] = acc[r] * hact_grad<HACT>(h2[it][r]);
                }
                PPO_T(2);
                lds_barrier();
                // ---------------------------------------------------------------- exchange 2: H1 and dz2 -> layer-2 gradient
#pragma unroll
                for (int ft = 0; ft < kHT; ++ft) { put_tile(S.ea, ft, h1[ft]); put_tile(S.eb, ft, d2[ft]); }
                lds_barrier();
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    f32x4 af[2], bf[kHT];
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        af[x] = get_frag(S.eb, 2 * w + x, bb);
                        gb2[x] += (af[x][0] + af[x][1]) + (af[x][2] + af[x][3]);
                    }
#pragma unroll
                    for (int kt = 0; kt < kHT; ++kt) bf[kt] = get_frag(S.ea, kt, bb);
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int kt = 0; kt < kHT; ++kt) g2[x][kt] = mfma4(g2[x][kt], af[x], bf[kt]);
                }
                // dH1 = W2^T dz2 -> dz1[in][row]
                PPO2_REFRESH();
                f32x4 d1[kHT];
#pragma unroll
                for (int it = 0; it < kHT; ++it) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ob = 0; ob < kHT; ++ob) {
                        f32x4 wa;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            wa[e] = S.w2[(ob * kHT + it) * 256 + ((((i16 >> 2) * 16 + ((4 * q + e) ^ (i16 >> 2)))) << 2) + (i16 & 3)];
                        acc = mfma4(acc, wa, d2[ob]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) d1[it][r] = acc[r] * hact_grad<HACT>(h1[it][r]);
                }
                PPO_T(3);
                lds_barrier();
                // ---------------------------------------------------------------- exchange 3: X and dz1 -> layer-1 gradient
#pragma unroll
                for (int ft = 0; ft < K0B; ++ft) put_tile(S.ea, ft, xb[ft]);
#pragma unroll
                for (int ft = 0; ft < kHT; ++ft) put_tile(S.eb, ft, d1[ft]);
                lds_barrier();
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    f32x4 af[2];
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        af[x] = get_frag(S.eb, 2 * w + x, bb);
                        gb1[x] += (af[x][0] + af[x][1]) + (af[x][2] + af[x][3]);
                    }
#pragma unroll
                    for (int kt = 0; kt < K0B; ++kt) {
                        const f32x4 bf = get_frag(S.ea, kt, bb);
#pragma unroll
                        for (int x = 0; x < 2; ++x) g1[x][kt] = mfma4(g1[x][kt], af[x], bf);
                    }
                }
            }

            PPO_T(4);
            PPO2_REFRESH();
            {       // the next step's rows: in flight during the reductions and Adam
                int k2 = k, s2 = s + mb;
                if (s2 >= T) { s2 = 0; ++k2; }
                if (k2 < a.k_epochs) nxt = load_rows(k2, s2, 0, min(mb, T - s2));
            }
            // -------------------------------------------------------------------- bias / log_std gradients, norm, loss
            // bias partials: lane (i16, q) summed rows 4q..4q+3 of every 16-row block: add the four lane groups
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                gb1[x] += lane_xor<16>(gb1[x]); gb1[x] += lane_xor<32>(gb1[x]);
                gb2[x] += lane_xor<16>(gb2[x]); gb2[x] += lane_xor<32>(gb2[x]);
            }
            gb3 += lane_xor<16>(gb3); gb3 += lane_xor<32>(gb3);
            // log_std: sum over the rows = over the 16 columns of every wave, then over the waves
            const bool gauss = !critic && !discrete;
            if (gauss) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) gls4[r] += lane_xor(gls4[r], o);
                }
                if (i16 == 0) st4(S.red + 32 + w * 16 + 4 * q, gls4);
            }
            float ss = 0.f;
#pragma unroll
            for (int x = 0; x < 2; ++x) {
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt) ss += (g2[x][kt][0] * g2[x][kt][0] + g2[x][kt][1] * g2[x][kt][1]) + (g2[x][kt][2] * g2[x][kt][2] + g2[x][kt][3] * g2[x][kt][3]);
#pragma unroll
                for (int kt = 0; kt < K0B; ++kt) ss += (g1[x][kt][0] * g1[x][kt][0] + g1[x][kt][1] * g1[x][kt][1]) + (g1[x][kt][2] * g1[x][kt][2] + g1[x][kt][3] * g1[x][kt][3]);
                ss += (g3[x][0] * g3[x][0] + g3[x][1] * g3[x][1]) + (g3[x][2] * g3[x][2] + g3[x][3] * g3[x][3]);
                if (q == 0) ss += gb1[x] * gb1[x] + gb2[x] * gb2[x];
            }
            if (w == 0 && q == 0) ss += gb3 * gb3;
            ss = wave_sum(ss);
            const float lsum = wave_sum(lossp);
            if (l == 0) { S.red[w] = ss; S.red[8 + w] = lsum; }
            lds_barrier();
            float gls_tot = 0.f;                                   // d loss / d log_std[i16] on wave 0's lanes
            if (gauss && w == 0 && i16 < A) {
                gls_tot = ((S.red[32 + i16] + S.red[48 + i16]) + S.red[64 + i16]) + S.red[80 + i16];
                const float raw = S.ls[i16];
                gls_tot = (raw >= -20.f && raw <= 2.f) ? (gls_tot - a.ent_coef) : 0.f;
            }
            float ss_ls = (gauss && w == 0 && q == 0 && i16 < A) ? gls_tot * gls_tot : 0.f;
            ss_ls = wave_sum(ss_ls);                               // wave 0 only holds non-zero values
            if (w == 0 && l == 0) S.red[16] = ss_ls;
            float ent_sum = 0.f;
            if (gauss) {
                for (int c = 0; c < A; ++c) ent_sum += kHalfLog2PiPlusHalf + fminf(fmaxf(S.ls[c], -20.f), 2.f);
            }
            lds_barrier();
            const float total = sqrtf((((S.red[0] + S.red[1]) + S.red[2]) + S.red[3]) + S.red[16]);
            const float loss = (((S.red[8] + S.red[9]) + S.red[10]) + S.red[11]) * invm - (gauss ? a.ent_coef * ent_sum : 0.f);
            float coef = 1.f;
            if (a.clip_norm > 0.f) coef = fminf(a.clip_norm / (total + 1e-6f), 1.f);
            ++t_step;
            pw1 *= (double)a.beta1; pw2 *= (double)a.beta2;
            const double bc1 = 1.0 - pw1, bc2 = 1.0 - pw2;
            const float step = (float)((double)lr / bc1), inv_bc2s = 1.f / (float)sqrt(bc2);
            const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2;

            PPO_T(5);
            PPO2_REFRESH();
            // -------------------------------------------------------------------- clip + Adam from the accumulators
            // (every wave finished its backward chain before the barriers above: the weights may change now)
            // 13.5 k of a step's 73 k cycles (tools/ppo_timing.py), and that is its VALU cost: 88 elements per lane x (~12 fp32 ops + the
            // quarter-rate v_sqrt / v_rcp + four accvgpr moves, m and v live in the AGPR half).  Round 5 tried three ways around the
            // ds_read-behind-ds_write pattern of `W[dw] = adam_elem(W[dw], ...)` and measured them (profiles/README.md): a tile's theta
            // read a tile ahead — 13.58 k, the same; a whole layer's slots read first, stepped, written — 21.3 ms per learn() at 256
            // learners against 19.9 (more spills); theta += step through ds_add_f32, no read at all — ~780 cycles per LDS float atomic,
            // 69 k cycles for the phase, 34.5 ms.
            auto upd = [&](lds_f W, int dw, float g, float& mm, float& vv) {
                W[dw] = adam_elem(W[dw], g * coef, mm, vv, w1, w2, a.beta2, inv_bc2s, a.adam_eps, step);
            };
            const int oslot = (((i16 >> 2) * 16) << 2) + (i16 & 3);              // transposed-owner address, as in the dX reads
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int ot = 2 * w + x;
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        upd(S.w2, (ot * kHT + kt) * 256 + oslot + (((4 * q + r) ^ (i16 >> 2)) << 2), g2[x][kt][r], m2[x][kt][r], v2[x][kt][r]);
#pragma unroll
                for (int kt = 0; kt < K0B; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        upd(S.w1, (ot * K0B + kt) * 256 + oslot + (((4 * q + r) ^ (i16 >> 2)) << 2), g1[x][kt][r], m1[x][kt][r], v1[x][kt][r]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    upd(S.w3, (2 * w + x) * 256 + oslot + (((4 * q + r) ^ (i16 >> 2)) << 2), g3[x][r], m3[x][r], v3[x][r]);
                if (q == 0) {
                    upd(S.b1, ot * 16 + i16, gb1[x], mb1[x], vb1[x]);
                    upd(S.b2, ot * 16 + i16, gb2[x], mb2[x], vb2[x]);
                }
            }
            if (w == 0 && q == 0) {
                upd(S.b3, i16, gb3, mb3, vb3);
                if (gauss && i16 < A) upd(S.ls, i16, gls_tot, mls, vls);
            }
            last_loss = loss;
            if (tid == 0) trace[2 * (k * n_mb + s / mb) + (critic ? 1 : 0)] = loss;
            lds_barrier();                                         // the next step's forward reads the new weights
            PPO_T(6);
        }
    }
    PPO_TDUMP();

    // ---- parameters and Adam state back to global memory
    for (int e = tid; e < L1.k_pad * kHid; e += kWG) {
        const int k = e / kHid, n = e - k * kHid;
        th_g[L1.w_off + e] = S.w1[frag_dw((n >> 4) * K0B + (k >> 4), n & 15, k & 15)];
    }
    for (int e = tid; e < kHid * kHid; e += kWG) {
        const int k = e / kHid, n = e - k * kHid;
        th_g[L2.w_off + e] = S.w2[frag_dw((n >> 4) * kHT + (k >> 4), n & 15, k & 15)];
    }
    for (int e = tid; e < kHid * np3; e += kWG) {
        const int k = e / np3, n = e - k * np3;
        th_g[L3.w_off + e] = S.w3[frag_dw(k >> 4, n, k & 15)];
    }
    if (tid < kHid) { th_g[L1.b_off + tid] = S.b1[tid]; th_g[L2.b_off + tid] = S.b2[tid]; }
    if (tid < 16) {
        th_g[L3.b_off + tid] = S.b3[tid];
        if (N.extra_n > 0 && tid < N.extra_n) th_g[N.extra_off + tid] = S.ls[tid];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int ot = 2 * w + x;
#pragma unroll
        for (int kt = 0; kt < kHT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = gofs(L2, ot * 16 + 4 * q + r, kt * 16 + i16);
                m_g[o] = m2[x][kt][r]; v_g[o] = v2[x][kt][r];
            }
#pragma unroll
        for (int kt = 0; kt < K0B; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int in = kt * 16 + i16;
                if (in < L1.k_pad) {
                    const int o = gofs(L1, ot * 16 + 4 * q + r, in);
                    m_g[o] = m1[x][kt][r]; v_g[o] = v1[x][kt][r];
                }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = gofs(L3, 4 * q + r, (2 * w + x) * 16 + i16);
            m_g[o] = m3[x][r]; v_g[o] = v3[x][r];
        }
        if (q == 0) {
            m_g[L1.b_off + ot * 16 + i16] = mb1[x]; v_g[L1.b_off + ot * 16 + i16] = vb1[x];
            m_g[L2.b_off + ot * 16 + i16] = mb2[x]; v_g[L2.b_off + ot * 16 + i16] = vb2[x];
        }
    }
    if (w == 0 && q == 0) {
        m_g[L3.b_off + i16] = mb3; v_g[L3.b_off + i16] = vb3;
        if (N.extra_n > 0 && i16 < N.extra_n) { m_g[N.extra_off + i16] = mls; v_g[N.extra_off + i16] = vls; }
    }
    if (tid == 0) {
        float* st = D.stats + (size_t)p * D.n_agents * ST_COUNT;
        steps[critic ? 1 : 0] = t_step;
        st[critic ? ST_CRITIC_LOSS : ST_ACTOR_LOSS] = last_loss;
    }
}

#define FRL_PPO2_KERNEL(name, K0B, HACT)                                                                            \
    __global__ __launch_bounds__(256) void name(const EngineDesc* __restrict__ Dp, PpoArgs a) {                      \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                                \
        ppo_update_v2_body<K0B, HACT>(*Dp, a, smem);                                                                \
    }
FRL_PPO2_KERNEL(ppo_update_v2_k1_relu, 1, ACT_RELU)
FRL_PPO2_KERNEL(ppo_update_v2_k2_relu, 2, ACT_RELU)
FRL_PPO2_KERNEL(ppo_update_v2_k1_tanh, 1, ACT_TANH)
FRL_PPO2_KERNEL(ppo_update_v2_k2_tanh, 2, ACT_TANH)


}  // namespace frl
