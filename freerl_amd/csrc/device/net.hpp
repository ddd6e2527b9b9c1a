// Layer / MLP / optimiser building blocks of the fused update kernels.
//
// Activations of one row-chunk (rc rows) are LDS resident:
//     xin  [rc][xp]   first-layer input (zero padded to the layer's k_pad)
//     h1,h2[rc][hp]   hidden activations, overwritten IN PLACE by their deltas on the way back
//     outb [rc][op]   head output / head delta
// Pitches are (width + 4) floats: 16-byte aligned rows and conflict-light MFMA operand reads.
// Weight gradients go to the block G the caller passes: a workgroup's own partial slab, written once (off-policy
// kernels), or the learner's `grad` block accumulated over a minibatch's row chunks (PPO: one owner per element,
// plain read-modify-write).  Parameters / Adam state / targets are streamed once per step by the Adam kernels.
#pragma once
#include "../frl_desc.h"
#include "rng.hpp"
#include "tile.hpp"

namespace frl {

// Developer instrument (tools/phase_timing.py; built only with -DFRL_PHASE_TIMING): every wave of a few sampled
// workgroups stamps the shader clock when it arrives at a barrier, wave 0 also when it is released.
#ifdef FRL_PHASE_TIMING
// Stamps go to (static) LDS and are dumped once at kernel end: a global store per stamp would put a write
// acknowledgement in front of the next vmcnt wait and distort what is measured.
constexpr int kPhaseMax = 64, kPhaseBlocks = 8, kPhaseWords = 5 * kPhaseMax;   // 4 waves' arrivals + wave 0's releases
__device__ int g_phase_clock[kPhaseBlocks][kPhaseWords];
__device__ int g_phase_stride = 509;
__device__ int g_phase_kernel = 0;         // which kernel dumps its stamps: 0 ac_critic_kernel, 1 ac_actor_kernel
__device__ __forceinline__ FRL_LDS int* phase_buf() {
    __shared__ int buf[kPhaseWords + 8];
    return (FRL_LDS int*)buf;
}
#define FRL_STAMP_(slot, ctr)                                                                      \
    do {                                                                                           \
        FRL_LDS int* b_ = phase_buf();                                                             \
        const int k_ = b_[kPhaseWords + (ctr)];                                                    \
        if (k_ < kPhaseMax) b_[(slot) * kPhaseMax + k_] = (int)clock64();                          \
        b_[kPhaseWords + (ctr)] = k_ + 1;                                                          \
    } while (0)
// extra stamp inside a phase (wave 0's "release" row): FRL_MARK()
#define FRL_MARK() do { if (threadIdx.x == 0) FRL_STAMP_(4, 4); } while (0)
#define FRL_PHASE(S)                                                                               \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0) FRL_STAMP_(threadIdx.x >> 6, threadIdx.x >> 6);               \
        lds_barrier();                                                                             \
        if (threadIdx.x == 0) FRL_STAMP_(4, 4);                                                    \
    } while (0)
#define FRL_PHASE_INIT(S)                                                                          \
    do {                                                                                           \
        if (threadIdx.x < 8) phase_buf()[kPhaseWords + threadIdx.x] = 0;                           \
        __syncthreads();                                                                           \
        if (threadIdx.x == 0) FRL_STAMP_(4, 4);                                                    \
    } while (0)
#define FRL_PHASE_DUMP(S, kernel_id)                                                               \
    do {                                                                                           \
        __syncthreads();                                                                           \
        if (g_phase_kernel == (kernel_id) && blockIdx.x % g_phase_stride == 0 && blockIdx.x / g_phase_stride < kPhaseBlocks) \
            for (int i_ = threadIdx.x; i_ < kPhaseWords; i_ += kWG)                                \
                g_phase_clock[blockIdx.x / g_phase_stride][i_] = phase_buf()[i_];                  \
    } while (0)
#else
#define FRL_PHASE(S) lds_barrier()
#define FRL_MARK() do {} while (0)
#define FRL_PHASE_INIT(S) do {} while (0)
#define FRL_PHASE_DUMP(S, kernel_id) do {} while (0)
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
                                         int batch_pad, int act_pad, int hbufs) {
    Lds S;
    S.rc = rc;
    S.xp = kin_pad_max + 4;
    S.hp = hidden + 4;
    S.op = out_pad_max + 4;
    S.ap = act_pad;
    lds_f p = (lds_f)smem;
    S.xin = p; p += rc * S.xp;
    S.h1 = p; p += rc * S.hp;
    S.h2 = p;                      // (hbufs == 1: an alias nobody touches — a net of two layers goes xin -> h1 -> outb)
    if (hbufs > 1) p += rc * S.hp;
    S.outb = p; p += rc * S.op;
    S.abuf = p; p += rc * S.ap;
    S.dabuf = p; p += rc * S.ap;
    S.y = p; p += batch_pad;
    S.red = p; p += 8;
    return S;
}

// compile-time activation for the hot epilogues (a runtime `act` per element costs a branch per element and drags
// the tanhf expansion into every epilogue); dispatch_act() turns the runtime value into the template argument once.
template <int ACT>
__device__ __forceinline__ f32x4 act_apply4(f32x4 v) {
    if constexpr (ACT == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if constexpr (ACT == ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
    return v;
}
template <class F>
__device__ __forceinline__ void dispatch_act(int act, F f) {
    if (act == ACT_RELU) f(std::integral_constant<int, ACT_RELU>{});
    else if (act == ACT_TANH) f(std::integral_constant<int, ACT_TANH>{});
    else f(std::integral_constant<int, ACT_NONE>{});
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
    dispatch_act(act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    auto finish = [&](int r, int c4, f32x4 v, f32x4 bias) { st4(Y + r * ldy + c4, act_apply4<ACT>(v + bias)); };
    if (npad % 64 == 0) {
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
    });
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
// How a gradient tile reaches its slab: GS_STREAM the slab is written once and read once by reduce_kernel (non-temporal
// store), GS_STORE first of several row chunks (plain store: the next chunk reads it back), GS_ADD a later chunk.
enum GradStore : int { GS_STREAM = 0, GS_STORE = 1, GS_ADD = 2 };

__device__ __forceinline__ void linear_bwd_dw(const LayerDesc& L, g_f G, lds_cf dY, int ldy, lds_cf X, int ldx, int rc,
                                              int gs) {
    g_f GW = G + L.w_off;
    const int kpad = L.k_pad, npad = L.n_pad;
    const bool first = gs != GS_ADD;
    auto finish = [&](int k, int n4, f32x4 v) {
        g_f g = GW + (size_t)k * npad + n4;
        if (gs == GS_STREAM) { st4_stream(g, v); return; }
        if (gs == GS_STORE) { st4(g, v); return; }
        st4(g, v + ld4((g_cf)g));
    };
    // the interleaved blocks fetch the running sums BEFORE their MFMA loop (GS_ADD), so the loads' latency hides
    // behind it; as part of the epilogue it was +13 % on the critic kernel
    auto run_il = [&](auto bn_c, int g, int kt0) {
        constexpr int BN = decltype(bn_c)::value;
        const int l = lane_id(), i = l & 15, q = l >> 4;
        f32x4 acc[4][BN], old[BN][4];
        acc_zero(acc);
        if (gs == GS_ADD) {
#pragma unroll
            for (int y = 0; y < BN; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[y][r] = ld4((g_cf)(GW + (size_t)(kt0 * 16 + 16 * y + 4 * q + r) * npad + g * 64 + 4 * i));
        }
        mma_dw_il<BN>(acc, dY, ldy, g * 64, X, ldx, kt0 * 16, rc);
#pragma unroll
        for (int y = 0; y < BN; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v = {acc[0][y][r], acc[1][y][r], acc[2][y][r], acc[3][y][r]};
                g_f gp = GW + (size_t)(kt0 * 16 + 16 * y + 4 * q + r) * npad + g * 64 + 4 * i;
                if (gs == GS_ADD) v += old[y][r];
                if (gs == GS_STREAM) st4_stream(gp, v); else st4(gp, v);
            }
    };
    if (npad % 64 == 0) {
        const int w = wave_id(), groups = npad / 64, tk = kpad / 16;
        if (tk % 2 == 0 && groups * (tk / 2) >= kWaves) {
            for (int blk = w; blk < groups * (tk / 2); blk += kWaves)
                run_il(std::integral_constant<int, 2>{}, blk % groups, (blk / groups) * 2);
        } else {
            for (int blk = w; blk < groups * tk; blk += kWaves)
                run_il(std::integral_constant<int, 1>{}, blk % groups, blk / groups);
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
    // bias gradient = column sums of dY.  The last wave(s) take it (the first ones own the odd tile block when the
    // tile count does not divide by 4); 8 independent partial sums keep 8 LDS reads in flight — as one dependent
    // chain of rc reads this loop was the longest thing in every dW phase.
    g_f Gb = G + L.b_off;
    for (int n = kWG - 1 - threadIdx.x; n < L.n_pad; n += kWG) {
        const float oldb = first ? 0.f : Gb[n];             // in flight during the column sum
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rc; r += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] += dY[(r + j) * ldy + n];
        }
        const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        Gb[n] = oldb + s;
    }
}

// ---- narrow heads (n <= 4 outputs: Q values, 1-4 action dimensions) ---------------------------------------------
// A 16-wide padded MFMA tile per head costs a whole barrier-delimited phase for 1/16 useful work, three times per
// head and update (forward, dW, dX) — after the 128x128 layers were pipelined these phases were a third of the
// kernel.  Instead the head rides on its neighbours: forward = dot products in the epilogue of the hidden layer
// that feeds it (each lane owns 16 of a row's columns), backward = one VALU pass over the LDS-resident activations.
__device__ __forceinline__ bool head_fusable(const NetDesc& N, int l0, int nl) {
    if (nl < 2) return false;
    const LayerDesc& LH = N.L[l0 + nl - 1];
    const LayerDesc& LP = N.L[l0 + nl - 2];
    return LH.n <= 4 && LH.n_pad == 16 && LP.n_pad % 64 == 0 && LP.n_pad <= 128 && LH.k_pad == LP.n_pad;
}

// Hidden layer L (as linear_fwd) + partial head outputs: outb[r][4*g + o] = sum over column group g of h[r][c] * WH[c][o]
// ONE: single-output head (every critic): the 16 head weights a lane needs are scalars fetched with the layer's own
// weights; otherwise they are float4s (4 outputs) fetched in the epilogue.
// STORE = false: the hidden activations are not kept (target nets: nothing is differentiated through them).
template <bool ONE, bool STORE = true>
__device__ __forceinline__ void linear_fwd_head_t(const LayerDesc& L, const LayerDesc& LH, g_cf theta, lds_cf X, int ldx, lds_f Y,
                                                  int ldy, int act, int rc, lds_f outb, int op) {
    g_cf W = theta + L.w_off;
    g_cf b = theta + L.b_off;
    g_cf WH = theta + LH.w_off;                       // Wk[L.n_pad][16]
    const int kpad = L.k_pad, npad = L.n_pad, l = lane_id(), q = l >> 4, row = l & 15;
    dispatch_act(act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    for_il_blocks(rc / 16, npad / 64, [&](auto bm, int mt0, int g) {
        constexpr int BM = decltype(bm)::value;
        f32x4 acc[BM][4], bias[4], part[BM];
        float w1[ONE ? 16 : 1];
        acc_zero(acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = ld4(b + g * 64 + 16 * q + 4 * r);
        if constexpr (ONE) {
#pragma unroll
            for (int j = 0; j < 16; ++j) w1[j] = WH[(g * 64 + 16 * q + j) * 16];
        }
        mma_w_any<BM, 4, W_IL>(acc, X, ldx, mt0 * 16, W, npad, g * 64, kpad);
#pragma unroll
        for (int x = 0; x < BM; ++x) part[x] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c4 = g * 64 + 16 * q + 4 * r;
            f32x4 w0, w1v, w2, w3;
            if constexpr (!ONE) {
                w0 = ld4(WH + (c4 + 0) * 16); w1v = ld4(WH + (c4 + 1) * 16); w2 = ld4(WH + (c4 + 2) * 16); w3 = ld4(WH + (c4 + 3) * 16);
            }
#pragma unroll
            for (int x = 0; x < BM; ++x) {
                const f32x4 v = act_apply4<ACT>(f32x4{acc[x][0][r], acc[x][1][r], acc[x][2][r], acc[x][3][r]} + bias[r]);
                if constexpr (STORE) st4(Y + (mt0 * 16 + x * 16 + row) * ldy + c4, v);
                if constexpr (ONE) part[x].x += (v.x * w1[4 * r] + v.y * w1[4 * r + 1]) + (v.z * w1[4 * r + 2] + v.w * w1[4 * r + 3]);
                else part[x] += v.x * w0 + v.y * w1v + v.z * w2 + v.w * w3;
            }
        }
#pragma unroll
        for (int x = 0; x < BM; ++x) {
#pragma unroll
            for (int c = 0; c < (ONE ? 1 : 4); ++c) {
                part[x][c] += lane_xor<16>(part[x][c]);
                part[x][c] += lane_xor<32>(part[x][c]);
            }
            if (q == 0) st4(outb + (mt0 * 16 + x * 16 + row) * op + 4 * g, part[x]);
        }
    });
    });
}
__device__ __forceinline__ void linear_fwd_head(const LayerDesc& L, const LayerDesc& LH, g_cf theta, lds_cf X, int ldx, lds_f Y,
                                                int ldy, int act, int rc, lds_f outb, int op) {
    if (LH.n == 1) linear_fwd_head_t<true>(L, LH, theta, X, ldx, Y, ldy, act, rc, outb, op);
    else linear_fwd_head_t<false>(L, LH, theta, X, ldx, Y, ldy, act, rc, outb, op);
}

// outb[r][0..16) = out_act(b + sum of the `groups` partials), zero beyond the head's n outputs; one thread per row
__device__ __forceinline__ void head_finalize(const LayerDesc& LH, g_cf theta, lds_f outb, int op, int rc, int groups, int out_act) {
    const int r = threadIdx.x;
    if (r < rc) {
        lds_f o = outb + r * op;
        f32x4 s = ld4(theta + LH.b_off);
        for (int g = 0; g < groups; ++g) s += ld4((lds_cf)(o + 4 * g));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < LH.n) v[c] = act_apply(s[c], out_act);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        st4(o, v); st4(o + 4, z); st4(o + 8, z); st4(o + 12, z);
    }
}

// Head backward in one pass: delta d[r][0..4) in outb.  Hidden unit k is owned by TWO lanes of one wave (lane
// halves 0-31 / 32-63), each taking half of the rows: dWk[k][:] = sum_r h[r][k] d[r] (halves combined by a shuffle),
// and h[r][k] <- (d[r] . WH[k]) * act'(h[r][k]) in place (the delta of the hidden layer).  The last 16 threads also
// sum the bias gradient.  G == nullptr: input gradient only.  k_pad <= 128 (32 units per wave).
__device__ __forceinline__ void head_bwd(const LayerDesc& LH, g_cf theta, g_f G, lds_cf outb, int op, lds_f H, int ldh,
                                         int act_prev, int rc, int gs) {
    const bool first = gs != GS_ADD;
    const int l = lane_id(), half = l >> 5, k = (threadIdx.x >> 6) * 32 + (l & 31);
    const int hr = rc / 2, r_lo = half * hr;
    if (k < LH.k_pad) {
        const f32x4 w = ld4(theta + LH.w_off + k * 16);
        f32x4 gw = {0.f, 0.f, 0.f, 0.f}, gold = {0.f, 0.f, 0.f, 0.f};
        if (G && !first && half == 0) gold = ld4((g_cf)(G + LH.w_off + k * 16));     // running sum, in flight during the row loop
        for (int r = r_lo; r < r_lo + hr; r += 4) {
            float h[4];
            f32x4 d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[j] = H[(r + j) * ldh + k]; d[j] = ld4(outb + (r + j) * op); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gw += h[j] * d[j];
                const float dx = d[j].x * w.x + d[j].y * w.y + d[j].z * w.z + d[j].w * w.w;
                H[(r + j) * ldh + k] = dx * act_grad(h[j], act_prev);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) gw[c] += lane_xor<32>(gw[c]);
        if (G && half == 0) {
            g_f g = G + LH.w_off + k * 16;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (gs == GS_STREAM) { st4_stream(g, gw); st4_stream(g + 4, z); st4_stream(g + 8, z); st4_stream(g + 12, z); }
            else if (first) { st4(g, gw); st4(g + 4, z); st4(g + 8, z); st4(g + 12, z); }
            else st4(g, gw + gold);
        }
    }
    const int t = kWG - 1 - threadIdx.x;
    if (G && t < 16) {
        const float oldb = first ? 0.f : G[LH.b_off + t];
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        if (t < 4)
            for (int r = 0; r < rc; r += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] += outb[(r + j) * op + t];
            }
        const float s = (p[0] + p[1]) + (p[2] + p[3]);
        g_f gb = G + LH.b_off + t;
        *gb = oldb + s;
    }
}

// Forward of layers [l0, l0+nl) of net N: xin -> h1 [-> h2] -> outb.  Ends with a barrier.
// rowf(r) runs on thread r (< rc) right after row r's head outputs are final in outb: the caller's per-row consumer of
// the output (target action, TD delta, log-prob ...) rides on the last phase instead of costing a barrier phase of its
// own (fused narrow heads; other heads get one extra phase for it).  rowf may read / rewrite outb row r and write other
// LDS buffers.  mlp_fwd = no consumer.
struct NoRowConsumer {
    __device__ __forceinline__ void operator()(int) const {}
    __device__ __forceinline__ void operator()() const {}
};
// allf() runs on every thread in that same last phase (the hidden activations in h1 / h2 are final by then).
template <class RowF, class AllF = NoRowConsumer>
__device__ __forceinline__ void mlp_fwd_rows(const NetDesc& N, int l0, int nl, g_cf theta, const Lds& S, int out_act, RowF rowf,
                                             AllF allf = AllF{}) {
    lds_cf in = S.xin;
    int ldin = S.xp;
    const bool fuse = head_fusable(N, l0, nl);
    for (int i = 0; i < nl; ++i) {
        const bool last = (i == nl - 1);
        lds_f out = last ? S.outb : (i == 0 ? S.h1 : S.h2);
        const int ldo = last ? S.op : S.hp;
        if (fuse && i == nl - 2) {
            const LayerDesc& LH = N.L[l0 + nl - 1];
            linear_fwd_head(N.L[l0 + i], LH, theta, in, ldin, out, ldo, N.hidden_act, S.rc, S.outb, S.op);
            FRL_PHASE(S);
            head_finalize(LH, theta, S.outb, S.op, S.rc, N.L[l0 + i].n_pad / 64, out_act);
            if (threadIdx.x < S.rc) rowf((int)threadIdx.x);
            if constexpr (!std::is_same<AllF, NoRowConsumer>::value) allf();
            FRL_PHASE(S);
            return;
        }
        linear_fwd(N.L[l0 + i], theta, in, ldin, out, ldo, last ? out_act : N.hidden_act, S.rc);
        FRL_PHASE(S);
        in = out;
        ldin = ldo;
    }
    if constexpr (!std::is_same<RowF, NoRowConsumer>::value) {
        if (threadIdx.x < S.rc) rowf((int)threadIdx.x);
        if constexpr (!std::is_same<AllF, NoRowConsumer>::value) allf();
        FRL_PHASE(S);
    }
}
__device__ __forceinline__ void mlp_fwd(const NetDesc& N, int l0, int nl, g_cf theta, const Lds& S, int out_act) {
    mlp_fwd_rows(N, l0, nl, theta, S, out_act, NoRowConsumer{});
}

// Twin single-output critics evaluated WITHOUT a backward pass (the target critics of TD3 / SAC / MATD3): both first
// layers in one barrier phase (xin -> h1 / h2), both second layers with their head dot products in the next (their
// activations never stored); the caller reads q_h(row) = twin_target_q(...).  Three phases instead of eight.
__device__ __forceinline__ bool twin_target_fusable(const NetDesc& N) {
    return N.heads == 2 && N.n_layers == 6 && head_fusable(N, 0, 3) && head_fusable(N, 3, 3) && N.L[2].n == 1 && N.L[5].n == 1 &&
           N.L[0].n_pad == N.L[1].k_pad && N.L[3].n_pad == N.L[4].k_pad && N.L[1].n_pad == 128 && N.L[4].n_pad == 128;
}
__device__ __forceinline__ void twin_target_fwd(const NetDesc& N, g_cf theta, const Lds& S) {
    linear_fwd(N.L[0], theta, S.xin, S.xp, S.h1, S.hp, N.hidden_act, S.rc);
    linear_fwd(N.L[3], theta, S.xin, S.xp, S.h2, S.hp, N.hidden_act, S.rc);
    FRL_PHASE(S);
    linear_fwd_head_t<true, false>(N.L[1], N.L[2], theta, S.h1, S.hp, nullptr, 0, N.hidden_act, S.rc, S.outb, S.op);
    linear_fwd_head_t<true, false>(N.L[4], N.L[5], theta, S.h2, S.hp, nullptr, 0, N.hidden_act, S.rc, S.outb + 8, S.op);
    FRL_PHASE(S);
}
// head h's value of row r after twin_target_fwd: bias + the two 64-column groups' partial dot products (head_finalize's order)
__device__ __forceinline__ float twin_target_q(const NetDesc& N, g_cf theta, const Lds& S, int r, int h) {
    lds_cf o = S.outb + r * S.op + 8 * h;
    return (theta[N.L[3 * h + 2].b_off] + o[0]) + o[4];
}

// Backward of layers [l0, l0+nl): head delta in outb (zero in padded columns and invalid rows).
// G != nullptr: weight/bias gradients go to G the way `gs` (GradStore) says.  want_dx0 leaves
// d loss / d xin in xin for column tiles [ct0, ct1).  Ends with a barrier.
__device__ __forceinline__ void mlp_bwd(const NetDesc& N, int l0, int nl, g_cf theta, g_f G, const Lds& S, int gs,
                                        bool want_dx0, int ct0, int ct1) {
    const bool fuse = head_fusable(N, l0, nl);
    for (int i = nl - 1; i >= 0; --i) {
        const bool last = (i == nl - 1);
        lds_cf D = last ? S.outb : (i == 0 ? S.h1 : S.h2);
        const int ldd = last ? S.op : S.hp;
        lds_f X = (i == 0) ? S.xin : (i == 1 ? S.h1 : S.h2);
        const int ldx = (i == 0) ? S.xp : S.hp;
        const LayerDesc& L = N.L[l0 + i];
        if (last && fuse) {
            head_bwd(L, theta, G, D, ldd, X, ldx, N.hidden_act, S.rc, gs);
            FRL_PHASE(S);
            continue;
        }
        if (G) {
            linear_bwd_dw(L, G, D, ldd, X, ldx, S.rc, gs);
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
    // Eight elements per thread and round: their row indices, then their ring loads, then the LDS stores.  One element per
    // iteration is two DEPENDENT global round trips that the compiler does not overlap with the next iteration's: 94 k cycles
    // for the 376 observation columns of 32 Humanoid rows (47 iterations), a quarter of SAC's critic kernel at those dims.
    const int total = rc * ncols;
    constexpr int U = 8;
    for (int e0 = threadIdx.x; e0 < total; e0 += kWG * U) {
        int ri[U], rr[U], cc[U];
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * kWG;
            rr[u] = e / ncols;
            cc[u] = e - rr[u] * ncols;
            ri[u] = (e < total && rr[u] < nvalid) ? idx[rr[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ri[u] >= 0 ? ring[(size_t)ri[u] * stride + src0 + cc[u]] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + u * kWG < total) X[rr[u] * ldx + dst0 + cc[u]] = v[u];
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
    // float4 lanes (the block is 16-byte aligned: net offsets are multiples of 32 floats), scalar tail for log_std
    const int n4 = size >> 2;
    const FRL_GLB f32x4* g4 = (const FRL_GLB f32x4*)g;
    float ss = 0.f;
#pragma unroll 4
    for (int i = threadIdx.x; i < n4; i += kWG) {              // unrolled: one workgroup, latency-bound — keep loads in flight
        const f32x4 x = g4[i];
        ss += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    for (int i = 4 * n4 + threadIdx.x; i < size; i += kWG) ss += g[i] * g[i];
    const float total = sqrtf(block_sum(ss, red));
    float coef = 1.f;
    if (clip_norm > 0.f) coef = fminf(clip_norm / (total + 1e-6f), 1.f);
    const double bc1 = 1.0 - powi_d((double)b1, t_new);
    const double bc2 = 1.0 - powi_d((double)b2, t_new);
    const float step = (float)((double)lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - b1, w2 = 1.f - b2, tk = 1.f - tau;
    FRL_GLB f32x4* th4 = (FRL_GLB f32x4*)theta;
    FRL_GLB f32x4* m4 = (FRL_GLB f32x4*)m;
    FRL_GLB f32x4* v4 = (FRL_GLB f32x4*)v;
    FRL_GLB f32x4* t4 = (FRL_GLB f32x4*)target;
#pragma unroll 4
    for (int i = threadIdx.x; i < n4; i += kWG) {
        f32x4 gi = g4[i] * coef, th = th4[i], mi = m4[i], vi = v4[i];
        if (wd != 0.f) gi += wd * th;
        mi = mi + (gi - mi) * w1;
        vi = vi * b2 + (w2 * gi) * gi;
        f32x4 denom;
        denom.x = sqrtf(vi.x) / bc2s + eps; denom.y = sqrtf(vi.y) / bc2s + eps;
        denom.z = sqrtf(vi.z) / bc2s + eps; denom.w = sqrtf(vi.w) / bc2s + eps;
        th = th - step * (mi / denom);
        m4[i] = mi; v4[i] = vi; th4[i] = th;
        if (target) t4[i] = t4[i] * tk + th * tau;
    }
    for (int i = 4 * n4 + threadIdx.x; i < size; i += kWG) {
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

// PPO.py's optimiser (PPO_file/c_adamw.py:80-127, "cautious" AdamW, weight_decay 0) after clip_grad_norm_ on this
// net: m = m*b1 + (1-b1) g; v = v*b2 + (1-b2) g g; denom = sqrt(v) + eps (no bias correction in denom);
// step = lr*sqrt(1-b2^t)/(1-b1^t); mask = (m*g > 0) / max(mean(m*g > 0), 1e-3) with the mean over ONE PARAMETER TENSOR
// (a weight matrix, a bias vector, log_std); p -= step * (m*mask)/denom.
__device__ __forceinline__ float cadamw_net(const NetDesc& N, g_f theta, g_f m, g_f v, g_cf g, float lr, float eps, float b1,
                                            float b2, float clip_norm, int t_new, lds_f red) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < N.size; i += kWG) {
        const float x = g[i];
        ss += x * x;
    }
    const float total = sqrtf(block_sum(ss, red));
    float coef = 1.f;
    if (clip_norm > 0.f) coef = fminf(clip_norm / (total + 1e-6f), 1.f);
    const double bc1 = 1.0 - powi_d((double)b1, t_new), bc2 = 1.0 - powi_d((double)b2, t_new);
    const float step = (float)((double)lr * sqrt(bc2) / bc1);
    const float w1 = 1.f - b1, w2 = 1.f - b2;
    const int n_tensors = 2 * N.n_layers + (N.extra_n > 0 ? 1 : 0);
    for (int ti = 0; ti < n_tensors; ++ti) {
        int off, span, numel;
        if (ti < 2 * N.n_layers) {
            const LayerDesc& L = N.L[ti >> 1];
            if (ti & 1) { off = L.b_off; span = L.n_pad; numel = L.n; }
            else { off = L.w_off; span = L.n_pad * L.k_pad; numel = L.n * L.k; }
        } else {
            off = N.extra_off; span = N.extra_n; numel = N.extra_n;
        }
        float cnt = 0.f;
        for (int i = threadIdx.x; i < span; i += kWG) {
            const float gi = g[off + i] * coef;
            const float mi = m[off + i] * b1 + w1 * gi;
            const float vi = v[off + i] * b2 + (w2 * gi) * gi;
            m[off + i] = mi;
            v[off + i] = vi;
            cnt += (mi * gi > 0.f) ? 1.f : 0.f;          // padding: m = g = 0, never counted
        }
        const float mean = block_sum(cnt, red) / (float)numel;
        const float scale = 1.f / fmaxf(mean, 1e-3f);
        for (int i = threadIdx.x; i < span; i += kWG) {  // m, v were written by this same thread
            const float gi = g[off + i] * coef, mi = m[off + i], vi = v[off + i];
            const float mask = (mi * gi > 0.f) ? scale : 0.f;
            theta[off + i] = theta[off + i] - step * ((mi * mask) / (sqrtf(vi) + eps));
        }
    }
    return total;
}

__device__ __forceinline__ void soft_update_net(int size, g_f target, g_cf theta, float tau) {
    const float tk = 1.f - tau;
    for (int i = threadIdx.x; i < size; i += kWG) target[i] = target[i] * tk + theta[i] * tau;
}

// Draw `batch` distinct row indices in [0,size) into idx (global, this learner's slice) using
// `lidx` (LDS) for the duplicate check: rejection keeps the draw uniform over subsets, like
// np.random.choice(size, batch, replace=False) (DQN.py:97): the LATER of two equal entries is redrawn, round by round.
// lidx: 2 * round_up(batch, 4) ints of LDS.  Returns with lidx[0 .. batch) final behind a barrier.  idx == nullptr: LDS only.
//
// batch <= 256 (one entry per thread; every config but MADDPG's 1024): the duplicate check is the cost — a batch of 256 from a
// 5e4-row ring collides in two calls of three, so two rounds are the rule.  Wave w reads the entries of waves 0 .. w (broadcast
// ds_read_b128, no stores in between: they pipeline) and keeps m = min over the entries in front of its lane of (entry XOR mine),
// vector ALU only; "somebody redraws" is a flag word between two LDS barriers.  Measured inside kernels_solo.hip's first section
// (tools/solo_timing.py, round 5), the whole draw: this form 3.4 us; the scan of thread i's i predecessors behind a divergent trip count
// with __syncthreads_or (rounds 1-4, the general path below) 8.6 us — __syncthreads_or alone 5.6 us per call; the later entry of
// a pair flagged by the earlier one's thread (128 reads per thread, but conditional LDS stores serialise them) 16.8 us; equality
// masks kept as scalars (ballot, s_and / s_or: ~35 cycles per entry behind the VALU -> SGPR -> SALU hazards) 8 us.
// drain = false (batch <= 256 only): return without the closing __syncthreads — lidx is final behind the last round's LDS barrier; the
// caller does not need its earlier global stores (or the idx store) performed before it goes on (kernels_solo.hip: 1.7 us)
//
// batch > 256 with a hash table (MADDPG's 1024; `table`: 2 x kDrawTable ints of LDS behind lidx's, batch <= kDrawTable / 4): a round
// inserts every entry under its row — linear probing, the slot keeps the SMALLEST position that holds the row (ds_cmpst + ds_min) —
// and an entry whose slot names another position has an earlier equal: O(batch) per round.  The scan of every entry's predecessors
// (the path below, still there for larger batches) took 92 us for draw_kernel's 3 x 1024 rows of config 5 — more than either update
// launch behind it.  Same rule, same Philox streams: the same rows.
constexpr int kDrawTable = 8192;
__device__ __forceinline__ void draw_indices(g_i idx, FRL_LDS int* lidx, int batch, int size, unsigned long long counter,
                                             unsigned stream, unsigned long long key, bool drain = true, FRL_LDS int* table = nullptr) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    if (batch > kWG && table && 4 * batch <= kDrawTable) {
        const int tid = threadIdx.x;
        int T = 1024;
        while (T < 4 * batch) T <<= 1;
        FRL_LDS int* tkey = table;                                      // the row a slot belongs to (-1: free)
        FRL_LDS int* tpos = table + kDrawTable;                         // the smallest position holding it
        FRL_LDS int* lslot = lidx + ((batch + 3) & ~3);                 // second half of the caller's 2 x batch ints: an entry's slot; then "somebody redraws"
        for (int i = tid; i < batch; i += kWG) lidx[i] = (int)uniform_index(philox4x32_10(counter, stream, (unsigned)i, key), (unsigned)size);
        for (unsigned round = 1; round < 64; ++round) {
            for (int t = tid; t < T; t += kWG) { tkey[t] = -1; tpos[t] = 0x7fffffff; }
            __syncthreads();
            for (int i = tid; i < batch; i += kWG) {
                const int v = lidx[i];
                int h = (int)((unsigned)v * 2654435761u >> 7) & (T - 1);
                for (;;) {
                    int expect = -1;
                    const bool got = __hip_atomic_compare_exchange_strong(tkey + h, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (got || expect == v) break;
                    h = (h + 1) & (T - 1);
                }
                __hip_atomic_fetch_min(tpos + h, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lslot[i] = h;
            }
            __syncthreads();
            int dup = 0;
            for (int i = tid; i < batch; i += kWG) {
                const int d = tpos[lslot[i]] != i ? 1 : 0;              // an earlier position holds the same row
                lslot[i] = d;
                dup |= d;
            }
            // redraw AFTER everyone has finished comparing against the old values
            const int any = __syncthreads_or(dup);
            if (!any) break;
            for (int i = tid; i < batch; i += kWG)
                if (lslot[i]) lidx[i] = (int)uniform_index(philox4x32_10(counter, stream + round * 0x10000u, (unsigned)i, key), (unsigned)size);
            __syncthreads();
        }
        if (idx) for (int i = tid; i < batch; i += kWG) idx[i] = lidx[i];
        __syncthreads();
        return;
    }
    if (batch <= kWG) {
        const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nq = (batch + 3) >> 2;                                // quads of entries (the last one padded)
        FRL_LDS int* fl = lidx + 4 * nq;                                // "somebody redraws" of round r in fl[r & 1]
        int mine = tid < batch ? (int)uniform_index(philox4x32_10(counter, stream, (unsigned)tid, key), (unsigned)size) : -1 - tid;
        if (tid < 4 * nq) lidx[tid] = mine;                             // (padding entries: distinct negatives, equal to nothing)
        if (tid < 2) fl[tid] = 0;
        lds_barrier();
        const FRL_LDS i32x4* l4 = (const FRL_LDS i32x4*)lidx;
        // one round's check.  FULL: batch == 256, every trip count a constant of the wave (the runtime-nq form costs 1.7 us more
        // per draw in kernels_solo.hip: tools/solo_timing.py)
        auto is_dup = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            const int nfront = FULL ? 16 * w : min(16 * w, nq);
            unsigned m = 1u;
#pragma unroll 8
            for (int j4 = 0; j4 < nfront; ++j4) {                       // entries of the waves in front of this one: every lane is behind them
                const i32x4 v = l4[j4];
                m = min(min(m, (unsigned)(v.x ^ mine)), min((unsigned)(v.y ^ mine), min((unsigned)(v.z ^ mine), (unsigned)(v.w ^ mine))));
            }
#pragma unroll
            for (int c4 = 0; c4 < 16; ++c4) {                           // this wave's own entries: entry 64 w + c counts for the lanes > c
                if (FULL || 16 * w + c4 < nq) {
                    const i32x4 v = l4[16 * w + c4];
                    m = min(m, (unsigned)(v.x ^ mine) | (unsigned)(4 * c4 >= l));
                    m = min(m, (unsigned)(v.y ^ mine) | (unsigned)(4 * c4 + 1 >= l));
                    m = min(m, (unsigned)(v.z ^ mine) | (unsigned)(4 * c4 + 2 >= l));
                    m = min(m, (unsigned)(v.w ^ mine) | (unsigned)(4 * c4 + 3 >= l));
                }
            }
            return m == 0u && tid < batch;
        };
        for (unsigned round = 1; round < 64; ++round) {
            const bool dup = batch == kWG ? is_dup(std::true_type{}) : is_dup(std::false_type{});
            if (dup) fl[round & 1] = 1;
            lds_barrier();                                              // everybody has compared against the old values and raised the flag
            if (!fl[round & 1]) break;
            if (tid == 0) fl[(round + 1) & 1] = 0;
            if (dup) {
                mine = (int)uniform_index(philox4x32_10(counter, stream + round * 0x10000u, (unsigned)tid, key), (unsigned)size);
                lidx[tid] = mine;
            }
            lds_barrier();
        }
        if (idx && tid < batch) idx[tid] = mine;
        if (drain) __syncthreads();   // (as the general path: callers count on it to have drained the workgroup's earlier global stores too)
        return;
    }
    for (int i = threadIdx.x; i < batch; i += kWG)
        lidx[i] = (int)uniform_index(philox4x32_10(counter, stream, (unsigned)i, key), (unsigned)size);
    __syncthreads();
    // does an earlier slot hold the same row?  four slots per LDS read (as a dependent chain of single reads this
    // check was most of the 33 us the kernel took)
    auto dup_before = [&](int i) {
        const int mine = lidx[i];
        bool d = false;
        const FRL_LDS i32x4* l4 = (const FRL_LDS i32x4*)lidx;
        const int full = i >> 2;
#pragma unroll 4
        for (int j4 = 0; j4 < full; ++j4) {
            const i32x4 v = l4[j4];
            d |= (v.x == mine) | (v.y == mine) | (v.z == mine) | (v.w == mine);
        }
        for (int j = full * 4; j < i; ++j) d |= (lidx[j] == mine);
        return d;
    };
    FRL_LDS int* lmark = lidx + ((batch + 3) & ~3);         // second half of the caller's 2 x batch ints of LDS
    for (unsigned round = 1; round < 64; ++round) {
        int dup = 0;
        for (int i = threadIdx.x; i < batch; i += kWG) {
            const int d = dup_before(i) ? 1 : 0;
            lmark[i] = d;
            dup |= d;
        }
        // redraw AFTER everyone has finished comparing against the old values (one scan per round: a small buffer — the
        // first thousand steps of a run — has tens of collisions per batch and takes several rounds)
        const int any = __syncthreads_or(dup);
        if (!any) break;
        for (int i = threadIdx.x; i < batch; i += kWG)
            if (lmark[i])
                lidx[i] = (int)uniform_index(philox4x32_10(counter, stream + round * 0x10000u, (unsigned)i, key),
                                             (unsigned)size);
        __syncthreads();
    }
    if (idx) for (int i = threadIdx.x; i < batch; i += kWG) idx[i] = lidx[i];
    __syncthreads();
}

}  // namespace frl
