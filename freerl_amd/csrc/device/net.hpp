// Layer / MLP / optimiser building blocks of the fused update kernels.
//
// Activations of one row-chunk (rc rows) are LDS resident:
//     xin  [rc][xp]   first-layer input (zero padded to the layer's k_pad)
//     h1,h2[rc][hp]   hidden activations, overwritten IN PLACE by their deltas on the way back
//     outb [rc][op]   head output / head delta
// Pitches are (width + 4) floats: 16-byte aligned rows and conflict-light MFMA operand reads.
// Weight gradients are accumulated in the learner's global `grad` block (one owner per
// element, plain read-modify-write), parameters/Adam state/targets are streamed once per step.
#pragma once
#include "../frl_desc.h"
#include "rng.hpp"
#include "tile.hpp"

namespace frl {

// Developer instrument (tools/phase_timing.py; built only with -DFRL_PHASE_TIMING): thread 0 of a few
// sampled workgroups stamps the shader clock at every barrier-delimited phase of the gradient kernels.
#ifdef FRL_PHASE_TIMING
// Stamps go to LDS (256 extra words after S.red, see lds_bytes_for) and are dumped once at kernel end: a global store per stamp
// would put a write acknowledgement in front of every barrier's vmcnt(0) and distort what is measured.
constexpr int kPhaseMax = 44, kPhaseBlocks = 8, kPhaseWords = 5 * kPhaseMax;   // 4 waves' arrivals + wave 0's releases
__device__ int g_phase_clock[kPhaseBlocks][kPhaseWords];
__device__ int g_phase_stride = 509;
#define FRL_STAMP_(S, slot, ctr)                                                                   \
    do {                                                                                           \
        const int k_ = (int)(S).red[ctr];                                                          \
        if (k_ < kPhaseMax) ((FRL_LDS int*)((S).red + 8))[(slot) * kPhaseMax + k_] = (int)clock64(); \
        (S).red[ctr] = (float)(k_ + 1);                                                            \
    } while (0)
#define FRL_PHASE(S)                                                                               \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0) FRL_STAMP_(S, threadIdx.x >> 6, 4 + (threadIdx.x >> 6));      \
        lds_barrier();                                                                             \
        if (threadIdx.x == 0) FRL_STAMP_(S, 4, 3);                                                 \
    } while (0)
#define FRL_PHASE_INIT(S)                                                                          \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0) (S).red[4 + (threadIdx.x >> 6)] = 0.f;                        \
        if (threadIdx.x == 0) { (S).red[3] = 0.f; FRL_STAMP_(S, 4, 3); }                           \
    } while (0)
#define FRL_PHASE_DUMP(S)                                                                          \
    do {                                                                                           \
        __syncthreads();                                                                           \
        if (blockIdx.x % g_phase_stride == 0 && blockIdx.x / g_phase_stride < kPhaseBlocks)        \
            for (int i_ = threadIdx.x; i_ < kPhaseWords; i_ += kWG)                                \
                g_phase_clock[blockIdx.x / g_phase_stride][i_] = ((FRL_LDS int*)((S).red + 8))[i_]; \
    } while (0)
#else
#define FRL_PHASE(S) lds_barrier()
#define FRL_PHASE_INIT(S) do {} while (0)
#define FRL_PHASE_DUMP(S) do {} while (0)
#endif

struct Lds {
    lds_f xin; int xp;
    lds_f h1; lds_f h2; int hp;
    lds_f outb; int op;
    lds_f y;           // [batch_pad] TD targets / v_targets
    lds_f abuf;        // [rc][ap] actions produced by the actor(s)
    lds_f dabuf;       // [rc][ap] d loss / d action (or eps staging)
    int ap;
    lds_f red;         // [8] reduction scratch
    int rc;
};

// Carve the dynamic LDS region; must mirror lds_bytes() in the host code.
__device__ __forceinline__ Lds carve_lds(float* smem, int rc, int hidden, int kin_pad_max, int out_pad_max,
                                         int batch_pad, int act_pad) {
    Lds S;
    S.rc = rc;
    S.xp = kin_pad_max + 4;
    S.hp = hidden + 4;
    S.op = out_pad_max + 4;
    S.ap = act_pad;
    lds_f p = (lds_f)smem;
    S.xin = p; p += rc * S.xp;
    S.h1 = p; p += rc * S.hp;
    S.h2 = p; p += rc * S.hp;
    S.outb = p; p += rc * S.op;
    S.abuf = p; p += rc * S.ap;
    S.dabuf = p; p += rc * S.ap;
    S.y = p; p += batch_pad;
    S.red = p; p += 8;
    return S;
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_TANH) return tanhf(v);
    return v;
}
// derivative from the activation OUTPUT
__device__ __forceinline__ float act_grad(float h, int act) {
    if (act == ACT_RELU) return h > 0.f ? 1.f : 0.f;
    if (act == ACT_TANH) return 1.f - h * h;
    return 1.f;
}

// Y[rc][n_pad] = act(X[rc][k_pad] * W^T + b);  theta holds Wk[k_pad][n_pad] (contraction-major) then b[n_pad]
__device__ __forceinline__ void linear_fwd(const LayerDesc& L, g_cf theta, lds_cf X, int ldx, lds_f Y, int ldy,
                                           int act, int rc) {
    g_cf W = theta + L.w_off;
    g_cf b = theta + L.b_off;
    const int kpad = L.k_pad, npad = L.n_pad, q4 = (lane_id() >> 4) * 4;
    auto finish = [&](int r, int c4, f32x4 v, f32x4 bias) {
        v += bias;
        v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
        st4(Y + r * ldy + c4, v);
    };
#ifndef FRL_FWD_IL
#define FRL_FWD_IL 1
#endif
    if (FRL_FWD_IL && npad % 64 == 0) {
        for_il_blocks(rc / 16, npad / 64, [&](auto bm, int mt0, int g) {
            constexpr int BM = decltype(bm)::value;
            f32x4 acc[BM][4], bias[4];
            acc_zero(acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = ld4(b + g * 64 + 4 * q4 + 4 * r);      // in flight with the weights
            mma_w_any<BM, 4, W_IL>(acc, X, ldx, mt0 * 16, W, npad, g * 64, kpad);
            tile_epilogue<BM, 4, W_IL>(acc, mt0 * 16, g * 64, [&](int r, int c4, f32x4 v, int slot) { finish(r, c4, v, bias[slot]); });
        });
    } else {
        for_tile_blocks(rc / 16, npad / 16, [&](auto bm, auto bn, int mt0, int nt0) {
            constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value;
            f32x4 acc[BM][BN], bias[BN];
            acc_zero(acc);
#pragma unroll
            for (int y = 0; y < BN; ++y) bias[y] = ld4(b + (nt0 + y) * 16 + q4);
            mma_w_any<BM, BN, W_ROWS>(acc, X, ldx, mt0 * 16, W, npad, nt0 * 16, kpad);
            tile_epilogue<BM, BN, W_ROWS>(acc, mt0 * 16, nt0 * 16, [&](int r, int c4, f32x4 v, int slot) { finish(r, c4, v, bias[slot]); });
        });
    }
}

// dX[rc][k tiles ct0..ct1) = (dY[rc][n_pad] * W) (.) act'(H)   written in place over H (pitch ldh); the contraction
// runs along the rows of Wk[k_pad][n_pad].  act_prev == ACT_NONE: plain store (used for d/d(first-layer input)).
__device__ __forceinline__ void linear_bwd_dx(const LayerDesc& L, g_cf theta, lds_cf dY, int ldy, lds_f H, int ldh,
                                              int act_prev, int rc, int ct0, int ct1) {
    g_cf W = theta + L.w_off;
    const int npad = L.n_pad;
    for_tile_blocks(rc / 16, ct1 - ct0, [&](auto bm, auto bn, int mt0, int nt0) {
        constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value;
        f32x4 acc[BM][BN];
        acc_zero(acc);
        mma_w_any<BM, BN, W_COLS>(acc, dY, ldy, mt0 * 16, W, npad, (ct0 + nt0) * 16, npad);
        tile_epilogue<BM, BN, W_COLS>(acc, mt0 * 16, (ct0 + nt0) * 16, [&](int r, int c4, f32x4 v, int) {
            lds_f h = H + r * ldh + c4;
            if (act_prev != ACT_NONE) {
                const f32x4 hv = ld4((lds_cf)h);
                v.x *= act_grad(hv.x, act_prev); v.y *= act_grad(hv.y, act_prev);
                v.z *= act_grad(hv.z, act_prev); v.w *= act_grad(hv.w, act_prev);
            }
            st4(h, v);
        });
    });
}

// G.Wk[k_pad][n_pad] (=|+=) X^T * dY ;  G.b (=|+=) column sums of dY
__device__ __forceinline__ void linear_bwd_dw(const LayerDesc& L, g_f G, lds_cf dY, int ldy, lds_cf X, int ldx, int rc,
                                              bool first) {
    g_f GW = G + L.w_off;
    const int kpad = L.k_pad, npad = L.n_pad;
    auto finish = [&](int k, int n4, f32x4 v) {
        g_f g = GW + (size_t)k * npad + n4;
        if (first) { st4_stream(g, v); return; }           // a slab is written once and read once by reduce_kernel
        st4(g, v + ld4((g_cf)g));
    };
    if (npad % 64 == 0) {
        const int w = wave_id(), groups = npad / 64, tk = kpad / 16;
        if (tk % 2 == 0 && groups * (tk / 2) >= kWaves) {
            for (int blk = w; blk < groups * (tk / 2); blk += kWaves) {
                const int g = blk % groups, kt0 = (blk / groups) * 2;
                f32x4 acc[4][2];
                acc_zero(acc);
                mma_dw_il<2>(acc, dY, ldy, g * 64, X, ldx, kt0 * 16, rc);
                dw_il_epilogue<2>(acc, g * 64, kt0 * 16, finish);
            }
        } else {
            for (int blk = w; blk < groups * tk; blk += kWaves) {
                const int g = blk % groups, kt0 = blk / groups;
                f32x4 acc[4][1];
                acc_zero(acc);
                mma_dw_il<1>(acc, dY, ldy, g * 64, X, ldx, kt0 * 16, rc);
                dw_il_epilogue<1>(acc, g * 64, kt0 * 16, finish);
            }
        }
    } else {
        for_tile_blocks(kpad / 16, npad / 16, [&](auto bm, auto bn, int kt0, int nt0) {
            constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value;
            f32x4 acc[BM][BN];
            acc_zero(acc);
            mma_tn<BM, BN>(acc, X, ldx, kt0 * 16, dY, ldy, nt0 * 16, rc);
            tile_epilogue<BM, BN, W_ROWS>(acc, kt0 * 16, nt0 * 16, [&](int k, int n4, f32x4 v, int) { finish(k, n4, v); });
        });
    }
    g_f Gb = G + L.b_off;
    for (int n = threadIdx.x; n < L.n_pad; n += kWG) {
        float s = 0.f;
        for (int r = 0; r < rc; ++r) s += dY[r * ldy + n];
        Gb[n] = first ? s : (Gb[n] + s);
    }
}

// Forward of layers [l0, l0+nl) of net N: xin -> h1 [-> h2] -> outb.  Ends with a barrier.
__device__ __forceinline__ void mlp_fwd(const NetDesc& N, int l0, int nl, g_cf theta, const Lds& S, int out_act) {
    lds_cf in = S.xin;
    int ldin = S.xp;
    for (int i = 0; i < nl; ++i) {
        const bool last = (i == nl - 1);
        lds_f out = last ? S.outb : (i == 0 ? S.h1 : S.h2);
        const int ldo = last ? S.op : S.hp;
        linear_fwd(N.L[l0 + i], theta, in, ldin, out, ldo, last ? out_act : N.hidden_act, S.rc);
        FRL_PHASE(S);
        in = out;
        ldin = ldo;
    }
}

// Backward of layers [l0, l0+nl): head delta in outb (zero in padded columns and invalid rows).
// G != nullptr accumulates weight/bias gradients (first: overwrite).  want_dx0 leaves
// d loss / d xin in xin for column tiles [ct0, ct1).  Ends with a barrier.
__device__ __forceinline__ void mlp_bwd(const NetDesc& N, int l0, int nl, g_cf theta, g_f G, const Lds& S, bool first,
                                        bool want_dx0, int ct0, int ct1) {
    for (int i = nl - 1; i >= 0; --i) {
        const bool last = (i == nl - 1);
        lds_cf D = last ? S.outb : (i == 0 ? S.h1 : S.h2);
        const int ldd = last ? S.op : S.hp;
        lds_f X = (i == 0) ? S.xin : (i == 1 ? S.h1 : S.h2);
        const int ldx = (i == 0) ? S.xp : S.hp;
        const LayerDesc& L = N.L[l0 + i];
        if (G) {
            linear_bwd_dw(L, G, D, ldd, X, ldx, S.rc, first);
            FRL_PHASE(S);
        }
        if (i > 0) {
            linear_bwd_dx(L, theta, D, ldd, X, ldx, N.hidden_act, S.rc, 0, L.k_pad / 16);
            FRL_PHASE(S);
        } else if (want_dx0) {
            linear_bwd_dx(L, theta, D, ldd, X, ldx, ACT_NONE, S.rc, ct0, ct1);
            FRL_PHASE(S);
        }
    }
}

// X[r][dst0 + c] = ring[idx[r0 + r]][src0 + c] for r < nvalid, c < ncols; 0 for r >= nvalid
__device__ __forceinline__ void gather_cols(lds_f X, int ldx, int rc, int nvalid, g_ci idx, g_cf ring, int stride,
                                            int src0, int ncols, int dst0) {
    const int total = rc * ncols;
    for (int e = threadIdx.x; e < total; e += kWG) {
        const int r = e / ncols, c = e - r * ncols;
        float v = 0.f;
        if (r < nvalid) v = ring[(size_t)idx[r] * stride + src0 + c];
        X[r * ldx + dst0 + c] = v;
    }
}
// Batch_ObsNorm: X[r][c0 + c] = (X[r][c0 + c] - mean[c]) / (std[c] + 1e-8) for r < nvalid
// (Normalization_batch_size.__call__, PPO_file/normalization.py:78-84); stats = {n, mean[O], S[O], std[O]}
__device__ __forceinline__ void normalize_cols(lds_f X, int ldx, int nvalid, int c0, int ncols, g_cf stats, int O) {
    g_cf mean = stats + 1, sd = stats + 1 + 2 * O;
    for (int e = threadIdx.x; e < nvalid * ncols; e += kWG) {
        const int r = e / ncols, c = e - r * ncols;
        lds_f x = X + r * ldx + c0 + c;
        *x = (*x - mean[c]) / (sd[c] + 1e-8f);
    }
}

__device__ __forceinline__ void zero_cols(lds_f X, int ldx, int rc, int c0, int c1) {
    const int w = c1 - c0;
    if (w <= 0) return;
    for (int e = threadIdx.x; e < rc * w; e += kWG) {
        const int r = e / w, c = e - r * w;
        X[r * ldx + c0 + c] = 0.f;
    }
}

// beta^t in double by repeated squaring (torch computes `beta ** step` in Python floats)
__device__ __forceinline__ double powi_d(double b, int t) {
    double r = 1.0;
    while (t > 0) {
        if (t & 1) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

// Global-norm clip + Adam (+ optional soft target update) over one net's parameter block.
// torch semantics: clip_grad_norm_(params, clip) then optim.Adam.step() (single-tensor order),
// then theta_t <- theta_t*(1-tau) + theta*tau.  Returns the pre-clip gradient norm.
__device__ __forceinline__ float adam_net(int size, g_f theta, g_f m, g_f v, g_cf g, g_f target, float lr, float eps,
                                          float b1, float b2, float wd, float clip_norm, int t_new, float tau,
                                          lds_f red) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < size; i += kWG) {
        const float x = g[i];
        ss += x * x;
    }
    const float total = sqrtf(block_sum(ss, red));
    float coef = 1.f;
    if (clip_norm > 0.f) coef = fminf(clip_norm / (total + 1e-6f), 1.f);
    const double bc1 = 1.0 - powi_d((double)b1, t_new);
    const double bc2 = 1.0 - powi_d((double)b2, t_new);
    const float step = (float)((double)lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - b1, w2 = 1.f - b2, tk = 1.f - tau;
    for (int i = threadIdx.x; i < size; i += kWG) {
        float gi = g[i] * coef;
        float th = theta[i];
        if (wd != 0.f) gi += wd * th;
        float mi = m[i];
        mi = mi + (gi - mi) * w1;
        const float vi = v[i] * b2 + (w2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2s + eps;
        th = th - step * (mi / denom);
        m[i] = mi;
        v[i] = vi;
        theta[i] = th;
        if (target) target[i] = target[i] * tk + th * tau;
    }
    return total;
}

__device__ __forceinline__ void soft_update_net(int size, g_f target, g_cf theta, float tau) {
    const float tk = 1.f - tau;
    for (int i = threadIdx.x; i < size; i += kWG) target[i] = target[i] * tk + theta[i] * tau;
}

// Draw `batch` distinct row indices in [0,size) into idx (global, this learner's slice) using
// `lidx` (LDS int[batch]) for the duplicate check: rejection keeps the draw uniform over
// subsets, like np.random.choice(size, batch, replace=False) (DQN.py:97).
__device__ __forceinline__ void draw_indices(g_i idx, FRL_LDS int* lidx, int batch, int size, unsigned long long counter,
                                             unsigned stream, unsigned long long key) {
    for (int i = threadIdx.x; i < batch; i += kWG)
        lidx[i] = (int)uniform_index(philox4x32_10(counter, stream, (unsigned)i, key), (unsigned)size);
    __syncthreads();
    for (unsigned round = 1; round < 64; ++round) {
        int dup = 0;
        for (int i = threadIdx.x; i < batch; i += kWG) {
            const int mine = lidx[i];
            bool d = false;
            for (int j = 0; j < i; ++j) d |= (lidx[j] == mine);
            if (d) dup = 1;
        }
        // redraw AFTER everyone has finished comparing against the old values
        const int any = __syncthreads_or(dup);
        if (!any) break;
        for (int i = threadIdx.x; i < batch; i += kWG) {
            const int mine = lidx[i];
            bool d = false;
            for (int j = 0; j < i; ++j) d |= (lidx[j] == mine);
            if (d) lidx[i] = -1 - i;          // mark; distinct negative so marks never collide
        }
        __syncthreads();
        for (int i = threadIdx.x; i < batch; i += kWG)
            if (lidx[i] < 0)
                lidx[i] = (int)uniform_index(philox4x32_10(counter, stream + round * 0x10000u, (unsigned)i, key),
                                             (unsigned)size);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < batch; i += kWG) idx[i] = lidx[i];
    __syncthreads();
}

}  // namespace frl
