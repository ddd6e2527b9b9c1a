// The K-sliced chained design (device/chain_wide.hpp) at HIDDEN 256 — north_star's "dense 256 x 256 MLP GEMMs" — kernels_criticx.hip /
// kernels_actorx.hip.  At 256 hidden units no layer's image fits LDS next to the others (W2 alone is 256 KB), so EVERY weight
// matrix is streamed: a layer is a sweep over 32 KB slices of its image (two k-blocks x sixteen output tiles), double-buffered
// through the 64 KB union exactly like chain_wide.hpp's first layer, with the accumulators of one or two 16-row tiles per wave
// (64 registers each) in registers.  Three sweeps:
//   sweep_rows   first layer: row operand from global memory (one dwordx4 per lane and k-block)
//   sweep_regs   second layer: row operand = the previous layer's output tiles, still in registers (the chained formulation:
//                the D layout of one layer is the B layout of the next)
//   sweep_t      dH = W2^T dZ: the same image sliced by OUTPUT blocks (32 KB contiguous each), fragments read transposed
// The weight gradients of the 256 x 256 layer (256 tiles) are contracted in a pass of their own, like chain_wide.hpp's dW1:
// the chunk loop leaves h1 (row-major) and the layer-2 deltas (exchange images) in the unit's scratch, and every wave owns
// 8 k-tiles x 4 output tiles per half-pass (128 accumulator registers), both operands read without LDS.
#pragma once
#include "chain_wide.hpp"

namespace frl {

constexpr int kHT2 = 16;                 // hidden tiles (256 units)
constexpr int kSKB2 = 2;                 // k-blocks per slice: 2 x 16 tiles x 1 KB = 32 KB

struct Wide16Scratch {
    g_f xrow, xobs, dqa, yb, q1, lpn, h1s, d2i, dz1, ah1, ah2;
    int xp, op;
    __device__ __forceinline__ void init(g_f base, int bm, int xp_, int op_, int nag) {
        xp = xp_; op = op_;
        xrow = base; base += (size_t)bm * xp + 64;
        xobs = base; base += (size_t)nag * bm * op + 64;
        dqa = base; base += (size_t)bm * kWideApitch;
        yb = base; q1 = base + bm; lpn = base + 2 * (size_t)bm; base += 4 * (size_t)bm;
        h1s = base; base += 256 * (size_t)bm;         // h1 of the batch, row-major [row][256]
        d2i = base; base += 256 * (size_t)bm;         // layer-2 deltas: exchange images, 16384 floats per 64-row chunk
        dz1 = base; base += 256 * (size_t)bm;         // layer-1 deltas likewise
        ah1 = base; base += 256 * (size_t)bm;         // the actor's activations between its forward and backward passes
        ah2 = base;
    }
};

// head and bias gradients a lane owns next to the chunk loop: head k-tiles {2w, 2w+1, 8+2w, 8+2w+1}, bias tiles likewise
template <int NT3>
struct Wide16Grad {
    f32x4 g3[NT3][4];
    float gb1[4], gb2[4], gb3[NT3];
};

struct WideNet16 {
    WideNet W;             // lane constants (W.C), the 64 KB union (W.u): slice buffers | exchange buffers ea, eb
    lds_f w3, w1r, b1, b2, b3, ls, red;

    __device__ __forceinline__ void init(float* smem) {
        lds_f p = (lds_f)smem;
        W.u = p; W.C.S.ea = p; W.C.S.eb = p + 8192; p += 16384;
        w3 = p; p += 2 * kHT2 * 256;
        w1r = p; p += 2 * kHT2 * 256;                                  // the first layer's image when it has <= 2 k-blocks (resident)
        b1 = p; p += 256;
        b2 = p; p += 256;
        b3 = p; p += 32;
        ls = p; p += 32;
        red = p; p += 64;
        W.C.S.w3 = w3; W.C.S.b1 = b1; W.C.S.b2 = b2; W.C.S.b3 = b3; W.C.S.ls = ls; W.C.S.red = red;
        W.C.S.w1 = W.u; W.C.S.w2 = W.u; W.C.S.ab = W.u; W.C.S.yb = W.u; W.C.S.q1 = W.u; W.C.S.lpn = W.u;
        W.C.init_lanes();
    }

    // ---- the head's image (nt3 tiles x 16 k-blocks), biases, log_std of one head -> LDS
    __device__ __forceinline__ void stage3(g_cf th, const LayerDesc* L, int nt3, int extra_off, int extra_n) const {
        const int tid = W.C.tid;
        f32x4 t3[8], t1[8];
        g_cf wg = th + L[2].w_off, w1g = th + L[0].w_off;
        const int kb1 = L[0].k_pad >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) t3[j] = j < 4 * nt3 ? ld4(wg + 4 * (tid + 256 * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) t1[j] = (kb1 <= 2 && j < 4 * kb1) ? ld4(w1g + 4 * (tid + 256 * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float bb1 = th[L[0].b_off + tid], bb2 = th[L[1].b_off + tid];
        float bb3 = 0.f, lsv = 0.f;
        if (tid < 32) {
            if (tid < L[2].n_pad) bb3 = th[L[2].b_off + tid];
            if (tid < extra_n) lsv = th[extra_off + tid];
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) st4(w3 + 4 * (tid + 256 * j), t3[j]);
        if (kb1 <= 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(w1r + 4 * (tid + 256 * j), t1[j]);
        }
        b1[tid] = bb1; b2[tid] = bb2;
        if (tid < 32) { b3[tid] = bb3; ls[tid] = lsv; }
        lds_barrier();
    }

    // ---- slice traffic shared by the three sweeps: a slice is 32 image tiles; wave w moves tiles w, w + 4, ...
    struct SliceRegs { f32x4 r[8]; };
    // first / second layer: slice s = k-blocks 2 s, 2 s + 1 of every output tile; image tile (ot, kb) at (ot * KB + kb) * 256
    __device__ __forceinline__ void fetch_k(SliceRegs& R, g_cf img, int KB, int s) const {
        const int w = W.C.w, l = W.C.l, last = KB - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 4 * j + w, ot = t >> 1;
            int kb = kSKB2 * s + (t & 1);
            kb = kb < last ? kb : last;
            R.r[j] = ld4(img + ((size_t)(ot * KB + kb) * 256 + 4 * l));
        }
    }
    // transposed sweep: slice s = output blocks 2 s, 2 s + 1 with all 16 input tiles each: 32 contiguous tiles
    __device__ __forceinline__ void fetch_o(SliceRegs& R, g_cf img, int s) const {
        const int w = W.C.w, l = W.C.l;
#pragma unroll
        for (int j = 0; j < 8; ++j) R.r[j] = ld4(img + ((size_t)(32 * s + 4 * j + w) * 256 + 4 * l));
    }
    __device__ __forceinline__ void commit(const SliceRegs& R, int s) const {
        lds_f buf = W.u + (s & 1) * 8192;
#pragma unroll
        for (int j = 0; j < 8; ++j) st4(buf + (4 * j + W.C.w) * 256 + 4 * W.C.l, R.r[j]);
    }
    // one k-block of a forward sweep: out tile ot's fragment of k-block slot kbl at slice tile 2 ot + kbl
    template <int T>
    __device__ __forceinline__ void kblock(f32x4 (&acc)[T][kHT2], lds_cf buf, int kbl, const f32x4 (&xk)[T]) const {
        f32x4 wf[kHT2];
#pragma unroll
        for (int ot = 0; ot < kHT2; ++ot) wf[ot] = ld4(buf + (2 * ot + kbl) * 256 + W.C.fslot);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ot = 0; ot < kHT2; ++ot)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], xk[t][e], acc[t][ot], 0, 0, 0);
    }
    template <int T>
    __device__ __forceinline__ void bias_init(f32x4 (&acc)[T][kHT2], lds_cf b) const {
#pragma unroll
        for (int ot = 0; ot < kHT2; ++ot) {
            const f32x4 bf = ld4(b + ot * 16 + 4 * W.C.q);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t][ot] = bf;
        }
    }
    template <int T>
    __device__ __forceinline__ void relu(f32x4 (&acc)[T][kHT2]) const {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][ot][r] = fmaxf(acc[t][ot][r], 0.f);
    }

    // ---- first layer: h1 = relu(W1 x + b1), rows from global memory (chain_wide.hpp: l1_sweep; same pipelining rules)
    template <int T>
    __device__ __forceinline__ void sweep_rows(f32x4 (&h1)[T][kHT2], const g_cf (&rp)[T], g_cf w1, int KB1) const {
        const int nfull = KB1 / kSKB2, tail = KB1 - nfull * kSKB2, last = KB1 - 1;
        bias_init<T>(h1, (lds_cf)b1);
        SliceRegs R;
        f32x4 xn[kSKB2][T], xc[kSKB2][T];
        auto xfetch = [&](int s) {
#pragma unroll
            for (int kbl = 0; kbl < kSKB2; ++kbl) {
                int kb = kSKB2 * s + kbl;
                kb = kb < last ? kb : last;
#pragma unroll
                for (int t = 0; t < T; ++t) xn[kbl][t] = W.xfrag(rp[t], kb);
            }
        };
        fetch_k(R, w1, KB1, 0);
        xfetch(0);
        lds_barrier();
        for (int s = 0; s < nfull; ++s) {
            commit(R, s);
            lds_barrier();
#pragma unroll
            for (int kbl = 0; kbl < kSKB2; ++kbl)
#pragma unroll
                for (int t = 0; t < T; ++t) xc[kbl][t] = xn[kbl][t];
            fetch_k(R, w1, KB1, s + 1);
            xfetch(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            lds_cf buf = W.u + (s & 1) * 8192;
#pragma unroll
            for (int kbl = 0; kbl < kSKB2; ++kbl) kblock<T>(h1, buf, kbl, xc[kbl]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tail > 0) {
            commit(R, nfull);
            lds_barrier();
            kblock<T>(h1, W.u + (nfull & 1) * 8192, 0, xn[0]);
        }
        relu<T>(h1);
    }

    // ---- first layer on the RESIDENT image (KB1 <= 2: stage3 put it at w1r, tile (ot, kb) at (ot * KB1 + kb) * 256): no slice
    // traffic, no barrier — a streamed sweep of one or two k-blocks is three dependent global round trips for 64 MFMAs
    template <int T>
    __device__ __forceinline__ void layer1_resident(f32x4 (&h1)[T][kHT2], const g_cf (&rp)[T], int KB1) const {
        f32x4 x[2][T];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < T; ++t) x[kb][t] = W.xfrag(rp[t], kb < KB1 ? kb : KB1 - 1);
        bias_init<T>(h1, (lds_cf)b1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb < KB1) {
                f32x4 wf[kHT2];
#pragma unroll
                for (int ot = 0; ot < kHT2; ++ot) wf[ot] = ld4((lds_cf)(w1r + (ot * KB1 + kb) * 256 + W.C.fslot));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ot = 0; ot < kHT2; ++ot)
#pragma unroll
                        for (int t = 0; t < T; ++t) h1[t][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], x[kb][t][e], h1[t][ot], 0, 0, 0);
            }
        }
        relu<T>(h1);
    }
    template <int T>
    __device__ __forceinline__ void layer1(f32x4 (&h1)[T][kHT2], const g_cf (&rp)[T], g_cf w1, int KB1) const {
        if (KB1 <= 2) layer1_resident<T>(h1, rp, KB1); else sweep_rows<T>(h1, rp, w1, KB1);
    }

    // ---- second layer: hout = relu(W2 hin + b2), the row operand in registers: hin[t][kb][e] IS the B fragment of k-block kb.
    // In two halves of eight output tiles (a slice = four k-blocks of a half: 32 tiles, its image rows are contiguous), so that
    // the accumulators of a half (32 T registers) sit next to the 64 T of hin without spilling; four slices per half, fully
    // unrolled (the k-block index selects registers)
    template <int T, int TT = T, int T0 = 0>
    __device__ __forceinline__ void sweep_regs(f32x4 (&hout)[T][kHT2], const f32x4 (&hin)[TT][kHT2], g_cf w2) const {
        const int w = W.C.w, l = W.C.l, fslot = W.C.fslot, q = W.C.q;
        // slice (hv, s): out tiles 8 hv + j, k-blocks 4 s + kbl: wave w moves k-block 4 s + w of the eight tiles -> LDS tile j * 4 + w
        auto fetch = [&](SliceRegs& R, int hv, int s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) R.r[j] = ld4(w2 + ((size_t)((8 * hv + j) * kHT2 + 4 * s + w) * 256 + 4 * l));
        };
        auto put = [&](const SliceRegs& R, int n) {
            lds_f buf = W.u + (n & 1) * 8192;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(buf + (j * 4 + w) * 256 + 4 * l, R.r[j]);
        };
        SliceRegs R;
        fetch(R, 0, 0);
        lds_barrier();
        static_for<0, 8>([&](auto nc) {
            constexpr int n = decltype(nc)::value, hv = n >> 2, s = n & 3;
            if constexpr (s == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 bf = ld4((lds_cf)(b2 + (8 * hv + j) * 16 + 4 * q));
#pragma unroll
                    for (int t = 0; t < T; ++t) hout[t][8 * hv + j] = bf;
                }
            }
            put(R, n);
            lds_barrier();
            if constexpr (n + 1 < 8) fetch(R, (n + 1) >> 2, (n + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
            lds_cf buf = W.u + (n & 1) * 8192;
#pragma unroll
            for (int kbl = 0; kbl < 4; ++kbl) {
                f32x4 wf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wf[j] = ld4(buf + (j * 4 + kbl) * 256 + fslot);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            hout[t][8 * hv + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], hin[T0 + t][4 * s + kbl][e], hout[t][8 * hv + j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        relu<T>(hout);
    }

    // ---- dH = W2^T dZ (no activation mask: the caller applies the ReLU of its h), fragments read transposed (four ds_read_b32,
    // double-buffered one k-step ahead as ChainNet::delta1).  In two halves of eight INPUT tiles: slice (hv, s) = output blocks
    // 4 s .. 4 s + 3 x input tiles 8 hv .. 8 hv + 7 (each output block's eight tiles are contiguous); wave w moves output block
    // 4 s + w -> LDS tile j * 4 + w
    template <int T>
    __device__ __forceinline__ void sweep_t(f32x4 (&dout)[T][kHT2], const f32x4 (&din)[T][kHT2], g_cf w2) const {
        const int w = W.C.w, l = W.C.l, q = W.C.q, i16 = W.C.i16, tslot = W.C.tslot;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int it = 0; it < kHT2; ++it) dout[t][it] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](SliceRegs& R, int hv, int s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) R.r[j] = ld4(w2 + ((size_t)((4 * s + w) * kHT2 + 8 * hv + j) * 256 + 4 * l));
        };
        auto put = [&](const SliceRegs& R, int n) {
            lds_f buf = W.u + (n & 1) * 8192;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(buf + (j * 4 + w) * 256 + 4 * l, R.r[j]);
        };
        SliceRegs R;
        fetch(R, 0, 0);
        lds_barrier();
        static_for<0, 8>([&](auto nc) {
            constexpr int n = decltype(nc)::value, hv = n >> 2, s = n & 3;
            put(R, n);
            lds_barrier();
            if constexpr (n + 1 < 8) fetch(R, (n + 1) >> 2, (n + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
            lds_cf buf = W.u + (n & 1) * 8192;
            float wa[2][8];
            auto frag = [&](int obl, int e, float (&dst)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j] = buf[(j * 4 + obl) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
            };
            frag(0, 0, wa[0]);
            static_for<0, 16>([&](auto kc) {
                constexpr int k_ = decltype(kc)::value, obl = k_ >> 2, e = k_ & 3;
                if constexpr (k_ + 1 < 16) frag((k_ + 1) >> 2, (k_ + 1) & 3, wa[(k_ + 1) & 1]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        dout[t][8 * hv + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k_ & 1][j], din[t][4 * s + obl][e], dout[t][8 * hv + j], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- heads on 256 features (image tile (o3, kb) at (o3 * 16 + kb) * 256)
    template <int T>
    __device__ __forceinline__ void head_valu(const f32x4 (&h2)[T][kHT2], f32x4 (&z)[T], int hn) const {
        const int q = W.C.q;
#pragma unroll
        for (int t = 0; t < T; ++t) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < hn) {
                float acc[T];
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = 0.f;
#pragma unroll
                for (int kb = 0; kb < kHT2; ++kb) {
                    const f32x4 wv = ld4((lds_cf)(w3 + kb * 256 + ((q * 16 + (o ^ q)) << 2)));
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[t] = fmaf(wv[r], h2[t][kb][r], acc[t]);
                }
                const float bo = b3[o];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    float a = acc[t];
                    a += __shfl_xor(a, 16, 64);
                    a += __shfl_xor(a, 32, 64);
                    z[t][o] = a + bo;
                }
            }
        }
    }
    __device__ __forceinline__ void delta2_valu(const f32x4& dz, const f32x4 (&h2)[kHT2], f32x4 (&d2)[kHT2], int hn) const {
        const int q = W.C.q;
        f32x4 dzb;
#pragma unroll
        for (int o = 0; o < 4; ++o) dzb[o] = __shfl(dz[o], W.C.i16, 64);
#pragma unroll
        for (int it = 0; it < kHT2; ++it) d2[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < hn) {
#pragma unroll
                for (int it = 0; it < kHT2; ++it) {
                    const f32x4 wv = ld4((lds_cf)(w3 + it * 256 + ((q * 16 + (o ^ q)) << 2)));
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2[it][r] = fmaf(wv[r], dzb[o], d2[it][r]);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < kHT2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[it][r] = h2[it][r] > 0.f ? d2[it][r] : 0.f;
    }
    template <int T, int NT3>
    __device__ __forceinline__ void head_tiles(const f32x4 (&h2)[T][kHT2], f32x4 (&z)[T][NT3]) const {
#pragma unroll
        for (int o3 = 0; o3 < NT3; ++o3) {
            const f32x4 bb = ld4((lds_cf)(b3 + 16 * o3 + 4 * W.C.q));
#pragma unroll
            for (int t = 0; t < T; ++t) z[t][o3] = bb;
        }
#pragma unroll
        for (int kb = 0; kb < kHT2; ++kb)
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                const f32x4 wf = ld4((lds_cf)(w3 + (o3 * kHT2 + kb) * 256 + W.C.fslot));
#pragma unroll
                for (int t = 0; t < T; ++t) z[t][o3] = mfma4(z[t][o3], wf, h2[t][kb]);
            }
    }
    template <int NT3>
    __device__ __forceinline__ void delta2_tiles(const f32x4 (&dz)[NT3], const f32x4 (&h2)[kHT2], f32x4 (&d2)[kHT2]) const {
        const int q = W.C.q, i16 = W.C.i16, tslot = W.C.tslot;
#pragma unroll
        for (int it = 0; it < kHT2; ++it) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                f32x4 wa;
#pragma unroll
                for (int e = 0; e < 4; ++e) wa[e] = w3[(o3 * kHT2 + it) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
                acc = mfma4(acc, wa, dz[o3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[it][r] = h2[it][r] > 0.f ? acc[r] : 0.f;
        }
    }

    template <int NT3>
    __device__ __forceinline__ void grad_zero(Wide16Grad<NT3>& g) const {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) g.g3[o3][x] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.gb1[x] = 0.f; g.gb2[x] = 0.f;
        }
#pragma unroll
        for (int o3 = 0; o3 < NT3; ++o3) g.gb3[o3] = 0.f;
    }

    // a 16-tile register block -> an exchange image in scratch (16384 floats per chunk: tile (ft, bb) at (ft * 4 + bb) * 256), in two
    // halves of eight feature tiles through eb; gb[2 hf + x] += the column sums of feature tiles 8 hf + 2 w + x
    __device__ __forceinline__ void exchange_out(const f32x4 (&d)[kHT2], g_f img, float (&gb)[4]) const {
        const ChainNet& C = W.C;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            lds_barrier();
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) C.put_tile(C.S.eb, ft, d[8 * hf + ft]);
            lds_barrier();
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const f32x4 af = C.get_frag(C.S.eb, 2 * C.w + x, bb);
                    gb[2 * hf + x] += (af[0] + af[1]) + (af[2] + af[3]);
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(img + hf * 8192 + 4 * (C.tid + 256 * j), ld4((lds_cf)(C.S.eb + 4 * (C.tid + 256 * j))));
        }
    }

    // ---- backward of TWO 64-row chunks (tile t = chunk 2 pr + t) with ONE transposed sweep of W2 for both: per tile the head
    // gradient and the layer-2 deltas (image -> scratch), then dH1 of both tiles, then per tile the ReLU mask of h1 — re-read from
    // the row-major copy the caller stored (this lane's own row) — and the layer-1 deltas (image -> scratch).
    // A one-tile sweep streams 256 KB of weights for 1024 MFMAs per wave: 8 B per cycle and CU, 4.9 TB/s for the chip — the
    // first build's one-chunk backward and forward were bandwidth-bound at 35 % of the MFMA rate.
    template <int NT3, bool VH>
    __device__ __forceinline__ void backward_pair(Wide16Grad<NT3>& g, const f32x4 (&h2)[2][kHT2], const f32x4 (&dz)[2][NT3], int hn, g_cf w2,
                                                  const g_cf (&h1row)[2], g_f d2img, g_f dz1img) const {
        const ChainNet& C = W.C;
        const int w = C.w;
        f32x4 d2[2][kHT2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                lds_barrier();
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) C.put_tile(C.S.ea, ft, h2[t][8 * hf + ft]);
                if (hf == 0) {
#pragma unroll
                    for (int o3 = 0; o3 < NT3; ++o3) C.put_tile(C.S.eb, o3, dz[t][o3]);
                }
                lds_barrier();
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    f32x4 bf[2];
#pragma unroll
                    for (int x = 0; x < 2; ++x) bf[x] = C.get_frag(C.S.ea, 2 * w + x, bb);
#pragma unroll
                    for (int o3 = 0; o3 < NT3; ++o3) {
                        const f32x4 af = C.get_frag(C.S.eb, o3, bb);
                        if (hf == 0 && w == 0) g.gb3[o3] += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
                        for (int x = 0; x < 2; ++x) g.g3[o3][2 * hf + x] = mfma4(g.g3[o3][2 * hf + x], bf[x], af);
                    }
                }
            }
            if constexpr (VH) delta2_valu(dz[t][0], h2[t], d2[t], hn); else delta2_tiles<NT3>(dz[t], h2[t], d2[t]);
            exchange_out(d2[t], d2img + (size_t)t * 16384, g.gb2);
        }
        f32x4 d1[2][kHT2];
        sweep_t<2>(d1, d2, w2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int it = 0; it < kHT2; ++it) {
                const f32x4 hm = ld4(h1row[t] + 16 * it + 4 * C.q);
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[t][it][r] = hm[r] > 0.f ? d1[t][it][r] : 0.f;
            }
            exchange_out(d1[t], dz1img + (size_t)t * 16384, g.gb1);
        }
    }
    template <int NT3>
    __device__ __forceinline__ void grad_finish(Wide16Grad<NT3>& g) const {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            g.gb1[x] += __shfl_xor(g.gb1[x], 16, 64); g.gb1[x] += __shfl_xor(g.gb1[x], 32, 64);
            g.gb2[x] += __shfl_xor(g.gb2[x], 16, 64); g.gb2[x] += __shfl_xor(g.gb2[x], 32, 64);
        }
#pragma unroll
        for (int o3 = 0; o3 < NT3; ++o3) { g.gb3[o3] += __shfl_xor(g.gb3[o3], 16, 64); g.gb3[o3] += __shfl_xor(g.gb3[o3], 32, 64); }
    }
    // head tiles and biases -> grad; returns this lane's share of the squared norm
    template <int NT3>
    __device__ __forceinline__ float grad_store_3(g_f G, const LayerDesc* L, const Wide16Grad<NT3>& g) const {
        const int w = W.C.w, q = W.C.q, i16 = W.C.i16, fslot = W.C.fslot;
        float ss = 0.f;
        auto sq = [](const f32x4& v) { return (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]); };
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int kt = 8 * (x >> 1) + 2 * w + (x & 1);
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                st4(G + L[2].w_off + (o3 * kHT2 + kt) * 256 + fslot, g.g3[o3][x]);
                ss += sq(g.g3[o3][x]);
            }
            if (q == 0) {
                G[L[0].b_off + kt * 16 + i16] = g.gb1[x];
                G[L[1].b_off + kt * 16 + i16] = g.gb2[x];
                ss += g.gb1[x] * g.gb1[x] + g.gb2[x] * g.gb2[x];
            }
        }
        if (w == 0 && q == 0) {
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) { G[L[2].b_off + 16 * o3 + i16] = g.gb3[o3]; ss += g.gb3[o3] * g.gb3[o3]; }
        }
        return ss;
    }

    // ---- dW^T of a layer with 16 output tiles over the whole batch, its tiles -> grad (Gw = the layer's block); returns the lane's
    // share of the squared norm.  KBin = the layer's input k-tiles, img = the deltas' exchange images (16384 floats per chunk),
    // rowptr(row) = that batch row's input columns (XT of them real).  Wave w owns k-tiles (w >> 1) + 2 j (j < NKT) and, in half-pass
    // hp, output tiles 8 hp + 4 (w & 1) + y: NKT x 4 accumulator tiles.  Loads as chain_wide.hpp: dw1_grad (unconditional,
    // ping-pong operand sets, row pointers from the LDS index table one block further ahead).
    template <int NKT> struct DwOps { f32x4 a[NKT], b[4]; };
    template <int NKT, class RowF>
    __device__ __forceinline__ float dw_grad(g_f Gw, g_cf img, int nchunks, int B, int KBin, int XT, RowF rowptr) const {
        const int w = W.C.w, q = W.C.q, i16 = W.C.i16, fslot = W.C.fslot;
        const int kt0 = w >> 1, nkt = (KBin - kt0 + 1) >> 1, nit = nchunks * 4;
        int fcol[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            const int kt = kt0 + 2 * j < KBin ? kt0 + 2 * j : KBin - 1;
            const int f = 16 * kt + i16;
            fcol[j] = f < XT ? f : XT - 1;
        }
        float ss = 0.f;
        for (int hp = 0; hp < 2; ++hp) {
            const int ot0 = 8 * hp + 4 * (w & 1);
            f32x4 acc[NKT][4];
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[j][y] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto rows_of = [&](int it, g_cf (&rp)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * (it < nit ? it : nit - 1) + 4 * q + e;
                    rp[e] = rowptr(row < B ? row : B - 1);
                }
            };
            auto load_ops = [&](int it_, const g_cf (&rp)[4], DwOps<NKT>& o) {
                const int it = it_ < nit ? it_ : nit - 1;
                g_cf im = img + (size_t)(it >> 2) * 16384 + (it & 3) * 256 + fslot;
#pragma unroll
                for (int y = 0; y < 4; ++y) o.b[y] = ld4(im + (ot0 + y) * 4 * 256);
#pragma unroll
                for (int j = 0; j < NKT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.a[j][e] = rp[e][fcol[j]];
            };
            auto mma = [&](const DwOps<NKT>& o) {
#pragma unroll
                for (int j = 0; j < NKT; ++j)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[j][y] = mfma4(acc[j][y], o.a[j], o.b[y]);
            };
            g_cf rp0[4], rp1[4];
            DwOps<NKT> A, Bo;
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
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < nkt) {
                    const int kt = kt0 + 2 * j;
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        f32x4 v = acc[j][y];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (16 * kt + 4 * q + r) < XT ? v[r] : 0.f;
                        st4(Gw + ((size_t)((ot0 + y) * KBin + kt) * 256 + fslot), v);
                        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                }
            }
        }
        return ss;
    }
};
constexpr int wide16_lds_floats() { return 16384 + 4 * kHT2 * 256 + 256 + 256 + 32 + 32 + 64; }
constexpr int kWide16ScratchPerRow = kWideApitch + 4 + 5 * 256;

}  // namespace frl
