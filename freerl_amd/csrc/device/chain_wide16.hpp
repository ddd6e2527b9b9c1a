// The K-sliced chained design (device/chain_wide.hpp) at HIDDEN 256 — north_star's "dense 256 x 256 MLP GEMMs" — kernels_criticx.hip /
// kernels_actorx.hip.  At 256 hidden units no weight image fits LDS (W2 alone is 256 KB); what fits the chip is the ACTIVATIONS:
//
//   * x-stationary sweeps: the activations of a 256-row super-chunk — four 16-row tiles per wave x sixteen feature tiles, XR: 256
//     registers in the D layout, which is the B operand of the next layer — stay in registers from the first layer to the head
//     and back down (l1_x -> sweep_x -> deltas from the ReLU mask words -> sweep_x<TR>), while the weight image streams through the
//     64 KB union two output tiles (32 KB) at a time, double-buffered, 512 MFMAs per slice and barrier; a finished pair of
//     output tiles goes to the caller's epilogue (ReLU, mask word, tile-lane store, head partials).  A head's main pass reads
//     nothing but its weights;
//   * what the weight-gradient passes need (h1, h2, d2, d1) goes to the unit's HBM scratch in TILE-LANE order — the D layout of
//     a 16 x 16 output tile, one dwordx4 per lane: tile (row block, feature tile) at ((chunk * 4 + wave) * 16 + tile) * 256
//     floats, 1 KB contiguous per wave-instruction; the gradient passes bring the tiles in as they lie and transpose them
//     through LDS (dw2_coop: once per workgroup and row block; dw_rows: per wave);
//   * a first layer wider than two k-blocks takes chain_wide.hpp's K-outer form (sweep_f, two halves of eight output tiles).
//
// Two earlier cuts of this file (round 4, profiles/README.md): two tiles per wave chained through registers with every layer
// K-sliced (W2 re-read per 128 rows: 2.4 TB/s chip-wide, ~1000 spilled VGPRs, 50.7 TFLOP/s against the row-chunk kernels'
// 55.4), and every product a K-outer sweep with the activations through tile-lane tensors in HBM between the layers (56.5).
#pragma once
#include "chain_wide.hpp"
#include "wide_timing.hpp"

namespace frl {

constexpr int kHT2 = 16;                 // hidden tiles (256 units)

struct Wide16Scratch {
    g_f xrow, xobs, dqa, yb, q1, lpn, dzt, h1t, h2t, d2t, d1t;
    int xp, op;
    __device__ __forceinline__ void init(g_f base, int bm, int xp_, int op_, int nag) {
        xp = xp_; op = op_;
        xrow = base; base += (size_t)bm * xp + 64;
        xobs = base; base += (size_t)nag * bm * op + 64;
        dqa = base; base += (size_t)bm * kWideApitch;
        yb = base; q1 = base + bm; lpn = base + 2 * (size_t)bm; base += 4 * (size_t)bm;
        dzt = base; base += 32 * (size_t)bm;          // head deltas: <= 2 tiles per 16-row block, tile-lane order
        h1t = base; base += 256 * (size_t)bm;         // hidden activations / deltas, tile-lane order: 16 tiles per 16-row block
        h2t = base; base += 256 * (size_t)bm;
        d2t = base; base += 256 * (size_t)bm;
        d1t = base;
    }
};

struct SweepNet {
    struct Slice0 { f32x4 R[8]; };     // a sweep's first weight slice in flight (sweep_fetch0)
    WideNet W;             // lane constants (W.C), the 64 KB union (W.u): the sweeps' slice buffers
    lds_f w3, w1a, b1, b2, b3, ls, red;

    __device__ __forceinline__ void init(float* smem) {
        lds_f p = (lds_f)smem;
        W.u = p; W.C.S.ea = p; W.C.S.eb = p + 8192; p += 16384;
        w3 = p; p += 2 * kHT2 * 256;                                   // the head's image: <= 2 tiles x 16 k-blocks
        w1a = p; p += 3 * kHT2 * 256;                                  // actor stage: the action k-blocks of the critic's first layer
        b1 = p; p += 256;
        b2 = p; p += 256;
        b3 = p; p += 32;
        ls = p; p += 32;
        red = p; p += 192;
        W.C.S.w3 = w3; W.C.S.b1 = b1; W.C.S.b2 = b2; W.C.S.b3 = b3; W.C.S.ls = ls; W.C.S.red = red;
        W.C.S.w1 = W.u; W.C.S.w2 = W.u; W.C.S.ab = W.u; W.C.S.yb = W.u; W.C.S.q1 = W.u; W.C.S.lpn = W.u;
        W.C.init_lanes();
    }

    // ---- the head's image (nt3 tiles x 16 k-blocks), biases, log_std of one head -> LDS
    __device__ __forceinline__ void stage3(g_cf th, const LayerDesc* L, int nt3, int extra_off, int extra_n) const {
        const int tid = W.C.tid;
        f32x4 t3[8];
        g_cf wg = th + L[2].w_off;
#pragma unroll
        for (int j = 0; j < 8; ++j) t3[j] = j < 4 * nt3 ? ld4(wg + 4 * (tid + 256 * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float bb1 = th[L[0].b_off + tid], bb2 = th[L[1].b_off + tid];
        float bb3 = 0.f, lsv = 0.f;
        if (tid < 32) {
            if (tid < L[2].n_pad) bb3 = th[L[2].b_off + tid];
            if (tid < extra_n) lsv = th[extra_off + tid];
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) st4(w3 + 4 * (tid + 256 * j), t3[j]);
        b1[tid] = bb1; b2[tid] = bb2;
        if (tid < 32) { b3[tid] = bb3; ls[tid] = lsv; }
        lds_barrier();
    }

    // this lane's slot of tile-lane tensor `base` (NT tiles per 16-row block) for row block (chunk, this wave): + tile * 256
    __device__ __forceinline__ g_f tl(g_f base, int chunk, int NT = kHT2) const { return base + ((size_t)(chunk * 4 + W.C.w) * NT) * 256 + 4 * W.C.l; }

    // ---- forward half-sweep: acc[t][j] = [relu](sum_kb W[8 hv + j][kb] x[t][kb] + bias), T x 16 rows per wave against eight
    // output tiles.  wimg = the half's image rows (tile (j, kb) at (j * KB + kb) * 256), p[t] + kb * kstride = this lane's dwordx4
    // of k-block kb (row-major rows: p = row + 4 q, kstride 16; tile-lane activations: p = tl(...), kstride 256).
    // chain_wide.hpp: l1_sweep (same pipelining rules: loads pinned in front of the slice's MFMAs, none of them conditional)
    template <int T, bool RELU>
    __device__ __forceinline__ void sweep_f(f32x4 (&acc)[T][8], const g_cf (&p)[T], int kstride, g_cf wimg, int KB, lds_cf bias) const {
        const auto K_ = W.C.lanes();
        const int l = W.C.l, w = W.C.w, q = K_.q, fslot = K_.fslot;
        const int nfull = KB / 4, tail = KB - nfull * 4, last = KB - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 bf = ld4(bias + j * 16 + 4 * q);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t][j] = bf;
        }
        f32x4 R[8], xn[4][T], xc[4][T];
        auto fetch = [&](int s) {
            int kb = 4 * s + w;
            kb = kb < last ? kb : last;
#pragma unroll
            for (int j = 0; j < 8; ++j) R[j] = ld4(wimg + ((size_t)(j * KB + kb) * 256 + 4 * l));
        };
        auto commit = [&](int s) {
            lds_f buf = W.u + (s & 1) * 8192;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(buf + (j * 4 + w) * 256 + 4 * l, R[j]);
        };
        auto xfetch = [&](int s) {
#pragma unroll
            for (int kbl = 0; kbl < 4; ++kbl) {
                int kb = 4 * s + kbl;
                kb = kb < last ? kb : last;
#pragma unroll
                for (int t = 0; t < T; ++t) xn[kbl][t] = ld4(p[t] + (size_t)kb * kstride);
            }
        };
        auto kblock = [&](lds_cf buf, int kbl, const f32x4 (&xk)[T]) {
            f32x4 wf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[j] = ld4(buf + (j * 4 + kbl) * 256 + fslot);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xk[t][e], acc[t][j], 0, 0, 0);
        };
        fetch(0);
        xfetch(0);
        lds_barrier();
        for (int s = 0; s < nfull; ++s) {
            commit(s);
            lds_barrier();
#pragma unroll
            for (int kbl = 0; kbl < 4; ++kbl)
#pragma unroll
                for (int t = 0; t < T; ++t) xc[kbl][t] = xn[kbl][t];
            fetch(s + 1);
            xfetch(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            lds_cf buf = W.u + (s & 1) * 8192;
#pragma unroll
            for (int kbl = 0; kbl < 4; ++kbl) kblock(buf, kbl, xc[kbl]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tail > 0) {
            commit(nfull);
            lds_barrier();
            lds_cf buf = W.u + (nfull & 1) * 8192;
#pragma unroll
            for (int kbl = 0; kbl < 3; ++kbl)
                if (kbl < tail) kblock(buf, kbl, xn[kbl]);
        }
        if constexpr (RELU) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][j][r] = fmaxf(acc[t][j][r], 0.f);
        }
    }

    // =========================================================================================================================
    // X-stationary form of the 256-wide products.  The activations of the wave's four tiles, XR[half][t][j] = feature tile
    // 8 half + j of row tile t (256 registers: the D layout of one layer = the B operand of the next, as in chain.hpp), stay in
    // registers; the weight image streams through the union two output tiles (32 KB) at a time and every finished pair of
    // output tiles goes to the caller's epilogue — so a layer reads nothing but its weights, its stores spread over the sweep,
    // and a 256-row super-chunk runs L1 -> L2 -> head -> deltas -> W2^T without reading an activation back.

    // first layer into XR: images of <= 32 tiles (KB1 <= 2) are staged whole, wider ones take the two K-outer half-sweeps
    __device__ __forceinline__ void l1_x(f32x4 (&X)[2][4][8], const g_cf (&p)[4], g_cf w1, int KB1) const {
        const auto K_ = W.C.lanes();
        const int l = W.C.l, w = W.C.w, q = K_.q, fslot = K_.fslot;
#ifdef FRL_WIDE_TIMING
        long long lc_[5] = {clock64(), 0, 0, 0, 0};
#endif
        if (KB1 <= 2) {
            const int lastt = 16 * KB1 - 1;
            f32x4 R[8], xin[2][4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int n = w * 8 + j;
                n = n < lastt ? n : lastt;
                R[j] = ld4(w1 + ((size_t)n * 256 + 4 * l));
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) xin[kb][t] = ld4(p[t] + (kb < KB1 ? kb : KB1 - 1) * 16);
            lds_barrier();
#ifdef FRL_WIDE_TIMING
            lc_[1] = clock64();
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(W.u + (w * 8 + j) * 256 + 4 * l, R[j]);
            lds_barrier();
#ifdef FRL_WIDE_TIMING
            lc_[2] = clock64();
            {
                float probe = xin[0][0][0] + xin[0][3][3];          // wait for the row operands here, so that the next stamp is the MFMAs' alone
                asm volatile("" :: "v"(probe));
                lc_[3] = clock64();
            }
#endif
            // (measured and not kept: the next tile's LDS reads issued a tile ahead and the ReLU a tile behind, pinned with
            // sched_barrier — 17.0 k cycles for this block against 14.5 k as the compiler orders it by itself)
            static_for<0, 16>([&](auto oc) {
                constexpr int ot = decltype(oc)::value;
                const f32x4 bf = ld4((lds_cf)(b1 + ot * 16 + 4 * q));
                f32x4 acc[4] = {bf, bf, bf, bf};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (kb < KB1) {
                        const f32x4 wf = ld4((lds_cf)(W.u + (ot * KB1 + kb) * 256 + fslot));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], xin[kb][t][e], acc[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[ot >> 3][t][ot & 7][r] = fmaxf(acc[t][r], 0.f);
            });
#ifdef FRL_WIDE_TIMING
            asm volatile("" :: "v"(X[1][3][7][0]));
            lc_[4] = clock64();
            if (threadIdx.x == 0 && blockIdx.x == 0) {
                for (int i_ = 0; i_ < 4; ++i_) g_wide_clk[1][8 + i_] += lc_[i_ + 1] - lc_[i_];
                g_wide_clk[1][12] += 1;
            }
#endif
        } else {
            sweep_f<4, true>(X[0], p, 16, w1, KB1, (lds_cf)b1);
            sweep_f<4, true>(X[1], p, 16, w1 + (size_t)8 * KB1 * 256, KB1, (lds_cf)(b1 + 128));
        }
    }

    // ReLU masks of XR: word s = tiles 2 s, 2 s + 1, bit o * 16 + t * 4 + r
    __device__ __forceinline__ void mask_bits(const f32x4 (&X)[2][4][8], unsigned (&m)[8]) const {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            unsigned v = 0;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v |= X[(2 * s + o) >> 3][t][(2 * s + o) & 7][r] > 0.f ? 1u << (o * 16 + t * 4 + r) : 0u;
            m[s] = v;
        }
    }
    // XR -> tile-lane tensor (the super-chunk's four chunks exist: the scratch is sized in multiples of 256 rows)
    __device__ __forceinline__ void store_x(g_f tensor, int sc, const f32x4 (&X)[2][4][8]) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            g_f tp = tl(tensor, 4 * sc + t);
#pragma unroll
            for (int it = 0; it < kHT2; ++it) st4(tp + it * 256, X[it >> 3][t][it & 7]);
        }
    }

    // the sweep: for s = 0..7, acc[o][t] = sum_kb Wtile(2 s + o, kb) XR[kb][t]  (TR: sum_ob Wtile(ob, 2 s + o)^T XR[ob][t]), biases
    // from `bias` when given; epi(s, acc) consumes the pair.  One barrier per slice, the next slice's loads pinned in front of
    // the slice's 512 MFMAs, nothing conditional in the loop (chain_wide.hpp's rules).
    // (sweep_fetch0: the first slice's loads, for callers that have something to run in front of the sweep — the mask words and
    // tile stores of a first layer, the delta arithmetic — while they are in flight: at 256 workgroups a sweep otherwise opens
    // with ~11 k cycles of HBM latency.  Not in front of the first layer itself: its own loads then queue behind eight more,
    // measured +17 k cycles per call)
    template <bool TR>
    __device__ __forceinline__ Slice0 sweep_fetch0(g_cf w2) const {
        Slice0 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = W.C.w * 8 + j;
            const int tile = TR ? (n & 15) * kHT2 + (n >> 4) : n;
            f.R[j] = ld4(w2 + ((size_t)tile * 256 + 4 * W.C.l));
        }
        return f;
    }
    template <bool TR, class Epi>
    __device__ __forceinline__ void sweep_x(const f32x4 (&X)[2][4][8], g_cf w2, lds_cf bias, Epi&& epi) const {
        sweep_x<TR>(X, w2, bias, epi, sweep_fetch0<TR>(w2));
    }
    template <bool TR, class Epi>
    __device__ __forceinline__ void sweep_x(const f32x4 (&X)[2][4][8], g_cf w2, lds_cf bias, Epi&& epi, const Slice0& f0) const {
        const auto K_ = W.C.lanes();
        const int l = W.C.l, w = W.C.w, q = K_.q, i16 = K_.i16, fslot = K_.fslot, tslot = K_.tslot;
        f32x4 R[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) R[j] = f0.R[j];
        auto fetch = [&](int s_) {
            const int s = s_ < 7 ? s_ : 7;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = w * 8 + j;                               // slot n of the slice = (o = n >> 4, kb = n & 15)
                const int tile = TR ? (n & 15) * kHT2 + 2 * s + (n >> 4) : 2 * s * kHT2 + n;
                R[j] = ld4(w2 + ((size_t)tile * 256 + 4 * l));
            }
        };
        auto commit = [&](int s) {
            lds_f buf = W.u + (s & 1) * 8192;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(buf + (w * 8 + j) * 256 + 4 * l, R[j]);
        };
#ifdef FRL_WIDE_TIMING
        long long sc_[5] = {clock64(), 0, 0, 0, 0}, sa_[4] = {0, 0, 0, 0};
#endif
        lds_barrier();
#ifdef FRL_WIDE_TIMING
        sc_[1] = clock64();
        sa_[0] = sc_[1] - sc_[0];
#endif
        for (int s = 0; s < 8; ++s) {
#ifdef FRL_WIDE_TIMING
            sc_[1] = clock64();
#endif
            commit(s);
            lds_barrier();
#ifdef FRL_WIDE_TIMING
            sc_[2] = clock64();
#endif
            fetch(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            lds_cf buf = W.u + (s & 1) * 8192;
            f32x4 acc[2][4];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const f32x4 bf = TR ? f32x4{0.f, 0.f, 0.f, 0.f} : ld4(bias + (2 * s + o) * 16 + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[o][t] = bf;
            }
            if constexpr (!TR) {
                // fragments one k-block ahead of their MFMAs, pinned: with XR's 256 registers live hipcc otherwise reads each
                // pair right in front of its 32 MFMAs and waits for it there (lgkmcnt(0) sixteen times per slice: -10 %)
                f32x4 wf[2][2];
#pragma unroll
                for (int o = 0; o < 2; ++o) wf[0][o] = ld4(buf + (o * 16) * 256 + fslot);
                static_for<0, 16>([&](auto kc) {
                    constexpr int kb = decltype(kc)::value;
                    if constexpr (kb + 1 < 16) {
#pragma unroll
                        for (int o = 0; o < 2; ++o) wf[(kb + 1) & 1][o] = ld4(buf + (o * 16 + kb + 1) * 256 + fslot);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int o = 0; o < 2; ++o)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                acc[o][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb & 1][o][e], X[kb >> 3][t][kb & 7][e], acc[o][t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
                float wa[2][2];
                auto frag = [&](int k_, float (&dst)[2]) {
                    const int ob = k_ >> 2, e = k_ & 3;
#pragma unroll
                    for (int o = 0; o < 2; ++o) dst[o] = buf[(o * 16 + ob) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
                };
                frag(0, wa[0]);
                static_for<0, 64>([&](auto kc) {
                    constexpr int k_ = decltype(kc)::value, ob = k_ >> 2, e = k_ & 3;
                    if constexpr (k_ + 1 < 64) frag(k_ + 1, wa[(k_ + 1) & 1]);
#pragma unroll
                    for (int o = 0; o < 2; ++o)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[o][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k_ & 1][o], X[ob >> 3][t][ob & 7][e], acc[o][t], 0, 0, 0);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef FRL_WIDE_TIMING
            asm volatile("" :: "v"(acc[1][3][0]), "v"(acc[0][0][0]));
            sc_[3] = clock64();
#endif
            epi(s, acc);
#ifdef FRL_WIDE_TIMING
            sc_[4] = clock64();
            sa_[1] += sc_[2] - sc_[1]; sa_[2] += sc_[3] - sc_[2]; sa_[3] += sc_[4] - sc_[3];
#endif
        }
#ifdef FRL_WIDE_TIMING
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            for (int i_ = 0; i_ < 4; ++i_) g_wide_clk[1][(TR ? 4 : 0) + i_] += sa_[i_];
        }
#endif
    }

    // layer-2 deltas into XR from the head's deltas and h2's ReLU masks: one-output dot-product head (dzv[t] = the row's delta)
    __device__ __forceinline__ void delta2_x_valu(f32x4 (&X)[2][4][8], const unsigned (&m2)[8], const float (&dzv)[4]) const {
        const auto K_ = W.C.lanes();
        const int q = K_.q;
#pragma unroll
        for (int it = 0; it < kHT2; ++it) {
            const f32x4 wv = ld4((lds_cf)(w3 + it * 256 + ((q * 16 + q) << 2)));          // slot (q, f = 0): W3[0][16 it + 4 q ..]
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[it >> 3][t][it & 7][r] = ((m2[it >> 1] >> ((it & 1) * 16 + t * 4 + r)) & 1u) ? wv[r] * dzv[t] : 0.f;
        }
    }
    // ... NT3 head tiles: W3^T dz through transposed fragments of the head image
    template <int NT3>
    __device__ __forceinline__ void delta2_x_tiles(f32x4 (&X)[2][4][8], const unsigned (&m2)[8], const f32x4 (&dz)[4][NT3]) const {
        const auto K_ = W.C.lanes();
        const int q = K_.q, i16 = K_.i16, tslot = K_.tslot;
#pragma unroll
        for (int it = 0; it < kHT2; ++it) {
            f32x4 wa[NT3];
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3)
#pragma unroll
                for (int e = 0; e < 4; ++e) wa[o3][e] = w3[(o3 * kHT2 + it) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o3 = 0; o3 < NT3; ++o3) d = mfma4(d, wa[o3], dz[t][o3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) X[it >> 3][t][it & 7][r] = ((m2[it >> 1] >> ((it & 1) * 16 + t * 4 + r)) & 1u) ? d[r] : 0.f;
            }
        }
    }
    // head partials of one finished pair of h2 tiles (sweep_x epilogue): dot-product head / NT3 MFMA tiles
    __device__ __forceinline__ void head_valu_pair(const f32x4 (&h)[2][4], int s, float (&zp)[4]) const {
        const auto K_ = W.C.lanes();
        const int q = K_.q;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const f32x4 wv = ld4((lds_cf)(w3 + (2 * s + o) * 256 + ((q * 16 + q) << 2)));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) zp[t] = fmaf(wv[r], h[o][t][r], zp[t]);
        }
    }
    template <int NT3>
    __device__ __forceinline__ void head_tiles_pair(const f32x4 (&h)[2][4], int s, f32x4 (&z)[4][NT3]) const {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                const f32x4 wf = ld4((lds_cf)(w3 + (o3 * kHT2 + 2 * s + o) * 256 + W.C.fslot));
#pragma unroll
                for (int t = 0; t < 4; ++t) z[t][o3] = mfma4(z[t][o3], wf, h[o][t]);
            }
    }
    // the mask words travel through the slice loops by rotation (no dynamically indexed registers)
    static __device__ __forceinline__ unsigned mask_next(unsigned (&m)[8]) {
        const unsigned v = m[0];
#pragma unroll
        for (int i = 0; i < 7; ++i) m[i] = m[i + 1];
        m[7] = v;
        return v;
    }
    static __device__ __forceinline__ void mask_push(unsigned (&m)[8], unsigned v) {
#pragma unroll
        for (int i = 0; i < 7; ++i) m[i] = m[i + 1];
        m[7] = v;
    }
    // ReLU in place on a pair; its mask word
    __device__ __forceinline__ unsigned relu_pair(f32x4 (&h)[2][4]) const {
        unsigned v = 0;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v |= h[o][t][r] > 0.f ? 1u << (o * 16 + t * 4 + r) : 0u;
                    h[o][t][r] = fmaxf(h[o][t][r], 0.f);
                }
        return v;
    }
    __device__ __forceinline__ void mask_pair(f32x4 (&d)[2][4], unsigned m) const {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[o][t][r] = ((m >> (o * 16 + t * 4 + r)) & 1u) ? d[o][t][r] : 0.f;
    }
    __device__ __forceinline__ void store_pair(g_f tensor, int sc, int s, const f32x4 (&h)[2][4]) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            g_f tp = tl(tensor, 4 * sc + t) + (2 * s) * 256;
            st4(tp, h[0][t]);
            st4(tp + 256, h[1][t]);
        }
    }

    // a tile written at the fragment slot, read back transposed: lane (i16, q) gets element (row 4 q + e, column i16)
    __device__ __forceinline__ f32x4 tr_read(lds_cf tb) const {
        const auto K_ = W.C.lanes();
        const int q = K_.q, i16 = K_.i16, tslot = K_.tslot;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tb[tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
        return o;
    }

    // ---- dW2 (256 x 256: 16 x 16 tiles) in one cooperative pass: per 16-row block the workgroup brings the block's sixteen
    // h1 and sixteen d2 tiles in ONCE (32 KB, eight dwordx4 per wave: waves 0, 1 the h1 tiles, 2, 3 the d2 tiles), double-
    // buffered through the union at the fragment slot, and every wave reads its operands back transposed: wave w owns k-tiles
    // 8 (w >> 1) + j x out tiles 8 (w & 1) + y — 64 accumulator tiles, 256 MFMAs per block and barrier.
    __device__ __forceinline__ float dw2_coop(g_f Gw, g_f Gb, g_cf h1t, g_cf d2t, int nchunks) const {
        const auto K_ = W.C.lanes();
        const int l = W.C.l, w = W.C.w, q = K_.q, i16 = K_.i16, fslot = K_.fslot;
        const int nit = nchunks * 4, kt0 = 8 * (w >> 1), ot0 = 8 * (w & 1);
        g_cf src = (w < 2 ? h1t + w * 8 * 256 : d2t + (w - 2) * 8 * 256) + 4 * l;
        f32x4 acc[8][8];
        float bsum[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            bsum[y] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][y] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 R[8];
        auto fetch = [&](int it_) {
            const int it = it_ < nit ? it_ : nit - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) R[j] = ld4(src + (size_t)it * kHT2 * 256 + j * 256);
        };
        fetch(0);
        lds_barrier();
        for (int it = 0; it < nit; ++it) {
            lds_f buf = W.u + (it & 1) * 8192;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(buf + (w * 8 + j) * 256 + fslot, R[j]);
            lds_barrier();
            fetch(it + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 a[8], b[2];                                          // the B tile of out tile y + 1 is read behind y's 32 MFMAs (pinned)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = tr_read((lds_cf)(buf + (kt0 + j) * 256));
            b[0] = tr_read((lds_cf)(buf + (kHT2 + ot0) * 256));
            static_for<0, 8>([&](auto yc) {
                constexpr int y = decltype(yc)::value;
                if constexpr (y + 1 < 8) b[(y + 1) & 1] = tr_read((lds_cf)(buf + (kHT2 + ot0 + y + 1) * 256));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j][y] = mfma4(acc[j][y], a[j], b[y & 1]);
                bsum[y] += (b[y & 1][0] + b[y & 1][1]) + (b[y & 1][2] + b[y & 1][3]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const f32x4 v = acc[j][y];
                st4(Gw + ((size_t)((ot0 + y) * kHT2 + kt0 + j) * 256 + fslot), v);
                ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            float v = bsum[y];
            v += lane_xor<16>(v);
            v += lane_xor<32>(v);
            if ((w >> 1) == 0 && q == 0) {
                Gb[(ot0 + y) * 16 + i16] = v;
                ss += v * v;
            }
        }
        return ss;
    }

    // ---- weight-gradient pass: dW^T tiles (k-tile kt0 + kstep j, out tile ot0 + y), j < NKT, y < NY, over the whole batch:
    // acc[j][y] += sum_rows A[row][k] B[row][out].  Both operands are read transposed, four dwords per lane, 16-row block and
    // tile (lane (i16 = column of the tile, q): rows 4 q + e of the block), pinned one block ahead of the MFMAs, unconditional:
    //   A  row-major rows (AROWS: arow(row) + column, rows through the LDS index table) or a tile-lane tensor (atl, NTA tiles)
    //   B  a tile-lane tensor (btl, NTB tiles per 16-row block)
    // Tiles -> Gw (image: tile (ot, kt) at (ot * KBimg + kt) * 256), input columns >= XT zeroed; returns the lane's sum of squares.
    template <int NKT, int NY> struct DwOps { f32x4 a[NKT], b[NY]; };
    template <int NKT, int NY, bool AROWS, class RowF>
    __device__ __forceinline__ float dw_pass(g_f Gw, g_f Gb, int KBimg, int XT, int kt0, int kstep, int nkt, int ot0, RowF arow, g_cf atl, int NTA, g_cf btl,
                                             int NTB, int nchunks, int B) const {
        const auto K_ = W.C.lanes();
        const int q = K_.q, i16 = K_.i16, fslot = K_.fslot;
        const int nit = nchunks * 4;
        const int lane_t = (((i16 >> 2) * 16 + 4 * q) << 2) + (i16 & 3);      // + 4 e: element (row 4 q + e, column i16) of a tile-lane tile
        int kcl[NKT], fcol[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            const int kt = j < nkt ? kt0 + kstep * j : kt0;
            kcl[j] = kt;
            const int f = 16 * kt + i16;
            fcol[j] = f < XT ? f : XT - 1;
        }
        f32x4 acc[NKT][NY];
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int y = 0; y < NY; ++y) acc[j][y] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bsum[NY];                                                // the layer's bias gradient = the column sums of the B tiles
#pragma unroll
        for (int y = 0; y < NY; ++y) bsum[y] = 0.f;
        auto rows_of = [&](int it, g_cf (&rp)[4]) {
            if constexpr (AROWS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * (it < nit ? it : nit - 1) + 4 * q + e;
                    rp[e] = arow(row < B ? row : B - 1);
                }
            }
        };
        auto load_ops = [&](int it_, const g_cf (&rp)[4], DwOps<NKT, NY>& o) {
            const int it = it_ < nit ? it_ : nit - 1;
            g_cf bb = btl + (size_t)it * NTB * 256 + lane_t;
#pragma unroll
            for (int y = 0; y < NY; ++y)
#pragma unroll
                for (int e = 0; e < 4; ++e) o.b[y][e] = bb[(ot0 + y) * 256 + 4 * e];
            if constexpr (AROWS) {
#pragma unroll
                for (int j = 0; j < NKT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.a[j][e] = rp[e][fcol[j]];
            } else {
                g_cf ab = atl + (size_t)it * NTA * 256 + lane_t;
#pragma unroll
                for (int j = 0; j < NKT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.a[j][e] = ab[kcl[j] * 256 + 4 * e];
            }
        };
        auto mma = [&](const DwOps<NKT, NY>& o) {
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int y = 0; y < NY; ++y) acc[j][y] = mfma4(acc[j][y], o.a[j], o.b[y]);
#pragma unroll
            for (int y = 0; y < NY; ++y) bsum[y] += (o.b[y][0] + o.b[y][1]) + (o.b[y][2] + o.b[y][3]);
        };
        g_cf rp0[4], rp1[4];
        DwOps<NKT, NY> A, Bo;
        rows_of(0, rp0);
        rows_of(1, rp1);
        load_ops(0, rp0, A);
        for (int it = 0; it < nit; it += 2) {
            load_ops(it + 1, rp1, Bo);
            rows_of(it + 2, rp0);
            __builtin_amdgcn_sched_barrier(0);
            mma(A);
            __builtin_amdgcn_sched_barrier(0);
            load_ops(it + 2, rp0, A);
            rows_of(it + 3, rp1);
            __builtin_amdgcn_sched_barrier(0);
            mma(Bo);
            __builtin_amdgcn_sched_barrier(0);
        }
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
                const int kt = kt0 + kstep * j;
#pragma unroll
                for (int y = 0; y < NY; ++y) {
                    f32x4 v = acc[j][y];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (16 * kt + 4 * q + r) < XT ? v[r] : 0.f;
                    st4(Gw + ((size_t)((ot0 + y) * KBimg + kt) * 256 + fslot), v);
                    ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
        }
#pragma unroll
        for (int y = 0; y < NY; ++y) {
            float v = bsum[y];
            v += lane_xor<16>(v);
            v += lane_xor<32>(v);
            if (Gb != nullptr && q == 0) {                             // (the caller names one owner wave per out tile)
                Gb[(ot0 + y) * 16 + i16] = v;
                ss += v * v;
            }
        }
        return ss;
    }
    // ---- the same for SMALL products (the head's dW3; a first layer of <= 2 k-tiles), where dw_pass's sixteen short trips per
    // wave are all latency: the waves split the ROW blocks (it = w, w + 4, ...), each accumulating all nkt x NY tiles (k-tiles
    // 0 .. nkt - 1, out tiles 0 .. NY - 1), and the four partial sums meet in the union (waves 2, 3 -> 0, 1; wave 1 -> 0; the
    // bias partials through the b1 / b2 slots, free between a head's main pass and the next stage3).  Wave 0 stores.
    template <int NKT, int NY, bool AROWS, class RowF>
    __device__ __forceinline__ float dw_rows(g_f Gw, g_f Gb, int KBimg, int XT, int nkt, RowF arow, g_cf atl, int NTA, g_cf btl, int NTB, int nchunks, int B) const {
        const auto K_ = W.C.lanes();
        const int l = W.C.l, w = W.C.w, q = K_.q, i16 = K_.i16, fslot = K_.fslot;
        int kcl[NKT], fcol[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            kcl[j] = j < nkt ? j : 0;
            const int f = 16 * kcl[j] + i16;
            fcol[j] = f < XT ? f : XT - 1;
        }
        f32x4 acc[NKT][NY];
        float bsum[NY];
#pragma unroll
        for (int y = 0; y < NY; ++y) {
            bsum[y] = 0.f;
#pragma unroll
            for (int j = 0; j < NKT; ++j) acc[j][y] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // tiles come in as they lie (one dwordx4 per lane), go through this wave's own kilobytes behind the union (the head image
        // and the actor stage's W1 blocks are idle during the gradient passes) at the fragment slot and come back transposed
        // (chain_net.hpp: fslot / tslot) — a transposed dword gather costs the texture path several times a dwordx4
        constexpr int NL = AROWS ? NY : NKT + NY;
        lds_f wb = w3 + w * 20 * 256;
        struct Raw { f32x4 t[NL]; f32x4 a[AROWS ? NKT : 1]; };
        auto fetch = [&](int i_, Raw& r) {
            const int it = 4 * (i_ < nchunks ? i_ : nchunks - 1) + w;
            g_cf bb = btl + (size_t)it * NTB * 256 + 4 * l;
#pragma unroll
            for (int y = 0; y < NY; ++y) r.t[y] = ld4(bb + y * 256);
            if constexpr (AROWS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * it + 4 * q + e;
                    g_cf rp = arow(row < B ? row : B - 1);
#pragma unroll
                    for (int j = 0; j < NKT; ++j) r.a[j][e] = rp[fcol[j]];
                }
            } else {
                g_cf ab = atl + (size_t)it * NTA * 256 + 4 * l;
#pragma unroll
                for (int j = 0; j < NKT; ++j) r.t[NY + j] = ld4(ab + kcl[j] * 256);
            }
        };
        auto turn = [&](const Raw& r, DwOps<NKT, NY>& o) {
#pragma unroll
            for (int n = 0; n < NL; ++n) st4(wb + n * 256 + fslot, r.t[n]);
#pragma unroll
            for (int y = 0; y < NY; ++y) o.b[y] = tr_read((lds_cf)(wb + y * 256));
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if constexpr (AROWS) o.a[j] = r.a[j];
                else o.a[j] = tr_read((lds_cf)(wb + (NY + j) * 256));
            }
        };
        auto mma = [&](const DwOps<NKT, NY>& o) {
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int y = 0; y < NY; ++y) acc[j][y] = mfma4(acc[j][y], o.a[j], o.b[y]);
#pragma unroll
            for (int y = 0; y < NY; ++y) bsum[y] += (o.b[y][0] + o.b[y][1]) + (o.b[y][2] + o.b[y][3]);
        };
        Raw r0, r1;
        DwOps<NKT, NY> ops;
        fetch(0, r0);
        for (int i = 0; i < nchunks; i += 2) {
            fetch(i + 1, r1);
            __builtin_amdgcn_sched_barrier(0);
            turn(r0, ops);
            mma(ops);
            __builtin_amdgcn_sched_barrier(0);
            fetch(i + 2, r0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nchunks) {
                turn(r1, ops);
                mma(ops);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int y = 0; y < NY; ++y) {
            bsum[y] += lane_xor<16>(bsum[y]);
            bsum[y] += lane_xor<32>(bsum[y]);
        }
        lds_f bsl = b1;                                                // two slots of 256 floats (b1, b2 are adjacent)
        auto put = [&](lds_f area, lds_f bslot) {
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int y = 0; y < NY; ++y) st4(area + (j * NY + y) * 256 + 4 * l, acc[j][y]);
            if (q == 0) {
#pragma unroll
                for (int y = 0; y < NY; ++y) bslot[y * 16 + i16] = bsum[y];
            }
        };
        auto add = [&](lds_cf area, lds_cf bslot) {
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int y = 0; y < NY; ++y) {
                    const f32x4 v = ld4(area + (j * NY + y) * 256 + 4 * l);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][y][r] += v[r];
                }
#pragma unroll
            for (int y = 0; y < NY; ++y) bsum[y] += bslot[y * 16 + i16];
        };
        lds_barrier();                                                 // every wave is through its rows (and the index table in the union)
        if (w >= 2) put(W.u + (w - 2) * 8192, bsl + (w - 2) * 256);
        lds_barrier();
        if (w < 2) add((lds_cf)(W.u + w * 8192), (lds_cf)(bsl + w * 256));
        lds_barrier();
        if (w == 1) put(W.u, bsl);
        lds_barrier();
        float ss = 0.f;
        if (w == 0) {
            add((lds_cf)W.u, (lds_cf)bsl);
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < nkt) {
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        f32x4 v = acc[j][y];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (16 * j + 4 * q + r) < XT ? v[r] : 0.f;
                        st4(Gw + ((size_t)(y * KBimg + j) * 256 + fslot), v);
                        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                }
            }
            if (q == 0) {
#pragma unroll
                for (int y = 0; y < NY; ++y) {
                    Gb[y * 16 + i16] = bsum[y];
                    ss += bsum[y] * bsum[y];
                }
            }
        }
        return ss;
    }
    // the three uses (Gb = the layer's bias gradient)
    __device__ __forceinline__ float dw2(g_f Gw, g_f Gb, g_cf h1t, g_cf d2t, int nchunks, int B) const { return dw2_coop(Gw, Gb, h1t, d2t, nchunks); }
    // dW3: the head's NT3 out tiles x sixteen k-tiles, rows split over the waves
    template <int NT3>
    __device__ __forceinline__ float dw3(g_f Gw, g_f Gb, g_cf h2t, g_cf dzt, int nchunks, int B) const {
        auto none = [](int) { return (g_cf) nullptr; };
        return dw_rows<kHT2, NT3, false>(Gw, Gb, kHT2, 256, kHT2, none, h2t, kHT2, dzt, NT3, nchunks, B);
    }
    // dW1: input rows row-major (arow through the LDS index table), deltas d1t; one or two k-tiles: rows split over the waves,
    // else k-tiles (w >> 1) + 2 j, out tiles as dW2
    template <class RowF>
    __device__ __forceinline__ float dw1(g_f Gw, g_f Gb_, int KB1, int XT, RowF arow, g_cf d1t, int nchunks, int B) const {
        if (KB1 == 1) return dw_rows<1, kHT2, true>(Gw, Gb_, KB1, XT, 1, arow, nullptr, 0, d1t, kHT2, nchunks, B);
        if (KB1 == 2) return dw_rows<2, kHT2, true>(Gw, Gb_, KB1, XT, 2, arow, nullptr, 0, d1t, kHT2, nchunks, B);
        float ss = 0.f;
        const int kt0 = W.C.w >> 1, nkt = (KB1 - kt0 + 1) >> 1;
        for (int hp = 0; hp < 2; ++hp) {
            const int ot0 = 8 * hp + 4 * (W.C.w & 1);
            g_f Gb = (W.C.w >> 1) == 0 ? Gb_ : nullptr;
            if (KB1 <= 6) ss += dw_pass<3, 4, true>(Gw, Gb, KB1, XT, kt0, 2, nkt, ot0, arow, nullptr, 0, d1t, kHT2, nchunks, B);
            else if (KB1 <= 14) ss += dw_pass<7, 4, true>(Gw, Gb, KB1, XT, kt0, 2, nkt, ot0, arow, nullptr, 0, d1t, kHT2, nchunks, B);
            else ss += dw_pass<kWideMaxKT, 4, true>(Gw, Gb, KB1, XT, kt0, 2, nkt, ot0, arow, nullptr, 0, d1t, kHT2, nchunks, B);
        }
        return ss;
    }
};

}  // namespace frl
