// MFMA tile toolkit for the fused update kernels (gfx950 / CDNA4 only).
//
// One workgroup = 256 threads = 4 wave64.  Activations of a row-chunk (rc rows of the batch)
// live in LDS; weights are read straight from global memory (L2 resident).  All matrix work is
// v_mfma_f32_16x16x4_f32 (exact fp32, bit-for-bit an fma chain):
//     A operand: lane l holds A[i = l&15][k = l>>4]       (one VGPR)
//     B operand: lane l holds B[k = l>>4][j = l&15]       (one VGPR)
//     C/D      : lane l, reg r holds D[row = (l>>4)*4 + r][col = l&15]
// The contraction index may be permuted freely as long as both operands agree: every fragment here
// uses k = k0 + 4*(l>>4) + e for MFMA step e of a 16-wide k-block.
//
// WEIGHT LAYOUT.  In both MFMA operands the 16-lane index is the NON-contracted dimension, so a
// weight fragment is only a coalesced load when that dimension is the contiguous one in memory
// (tools/loadpat_bench.hip: 16 rows x 16 B per wave-instruction costs 65 cycles of the CU's L1, a
// 4 x 256 B one 17).  Every weight matrix is therefore kept CONTRACTION-MAJOR for its reader:
//     forward  Y = X W^T   reads  Wk[k][n]   (the engine's layout for theta, target, Adam state and gradients)
// (the backward dX = dY W contracts over n and reads the same Wk along its rows, the slow pattern — see WMode)
// and four 16-wide output tiles are INTERLEAVED (tile y owns columns c0 + 4*c + y) so that a lane's
// fragments for the four tiles are one float4: lanes 0..15 read 256 contiguous bytes of a weight row.
//
// Pointers carry their address space in the type (lds_* = LDS, g_* = global): operand fetches
// compile to ds_read_b128 / global_load_dwordx4 instead of flat loads, and LDS / global waits
// use separate counters.  Operand fragments are fetched one k-block ahead of the MFMAs that
// consume them.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "lane.hpp"

namespace frl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FRL_LDS __attribute__((address_space(3)))
#define FRL_GLB __attribute__((address_space(1)))
typedef FRL_LDS float* lds_f;
typedef const FRL_LDS float* lds_cf;
typedef FRL_GLB float* g_f;
typedef const FRL_GLB float* g_cf;
typedef const FRL_GLB int* g_ci;
typedef FRL_GLB int* g_i;

template <class T>
__device__ __forceinline__ g_f as_global(T* p) { return (g_f)(p); }
template <class T>
__device__ __forceinline__ g_cf as_global(const T* p) { return (g_cf)(p); }
__device__ __forceinline__ g_ci as_global_i(const int* p) { return (g_ci)(p); }

__device__ __forceinline__ f32x4 ld4(lds_cf p) { return *reinterpret_cast<const FRL_LDS f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(g_cf p) { return *reinterpret_cast<const FRL_GLB f32x4*>(p); }
__device__ __forceinline__ void st4(lds_f p, f32x4 v) { *reinterpret_cast<FRL_LDS f32x4*>(p) = v; }
__device__ __forceinline__ void st4(g_f p, f32x4 v) { *reinterpret_cast<FRL_GLB f32x4*>(p) = v; }

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): global stores (the gradient
// slabs) and loads still in flight stay in flight.  __syncthreads() also drains vmcnt, which parked every wave of a
// dW phase until its slab stores were acknowledged (3-5k cycles per barrier in tools/phase_timing.py).  Use only
// where the threads of the workgroup exchange data through LDS.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// The barrier in front of a ONE-LANE release (flag store / ticket add that tells another workgroup, the next launch or the host
// "this workgroup's global stores are there").  On gfx90a+ outside threadgroup-split mode __syncthreads() is
// `s_waitcnt lgkmcnt(0); s_barrier`: it does NOT wait for a wave's outstanding global stores, and lane 0's own vmcnt wait behind
// its release fence covers wave 0 only — so every thread drains its own stores first (a per-wave wait, no cache operation).
__device__ __forceinline__ void sync_stores() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void st4_stream(g_f p, f32x4 v) {      // write-once data: do not keep it in L2
    __builtin_nontemporal_store(v, reinterpret_cast<FRL_GLB f32x4*>(p));
}

constexpr int kWG = 256;      // threads per workgroup
constexpr int kWaves = 4;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

template <int BM, int BN>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[BM][BN]) {
#pragma unroll
    for (int x = 0; x < BM; ++x)
#pragma unroll
        for (int y = 0; y < BN; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int BM, int BN>
__device__ __forceinline__ void mma_step4(f32x4 (&acc)[BM][BN], const f32x4 (&a)[BM], const f32x4 (&b)[BN]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int x = 0; x < BM; ++x)
#pragma unroll
            for (int y = 0; y < BN; ++y)
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[y][e], a[x][e], acc[x][y], 0, 0, 0);
}

// ---- C[m][c] += sum_k A[m][k] * W(k, c)   (A: LDS activations, k contiguous; W: global weights) ----------
// Three ways to read the weight fragments, K a multiple of 16:
//   W_IL    M[k][c] contraction-major, BN == 4 interleaved tiles (tile y owns columns c0 + 4*j + y): one float4
//           per lane and step, a wave-instruction covers 4 rows x 256 contiguous bytes            (forward, n_pad % 64 == 0)
//   W_ROWS  M[k][c] contraction-major, tile y owns columns c0 + 16*y + j: one dword per lane, tile and step (heads)
//   W_COLS  M[c][k]: the contraction runs ALONG the rows of M (dX = dY W out of the forward layout Wk[c][n]):
//           one float4 along the contraction per lane and tile; 16 rows x 64 B per wave-instruction — 4x the L1
//           cost of W_IL per byte, paid only by the backward pass (a second, transposed copy of the weights kept
//           by the Adam kernel was measured: its scattered stores cost more than this saves)
enum WMode : int { W_IL = 0, W_ROWS = 1, W_COLS = 2 };

template <int BN, int MODE>
__device__ __forceinline__ void fetch_w(f32x4 (&b)[BN], g_cf mp, int ld, int k) {
    if constexpr (MODE == W_IL) {        // mp -> row 4q, column c0 + 4i
        static_assert(BN % 4 == 0, "interleaved weight fragments come four tiles (64 columns) at a time");
#pragma unroll
        for (int g = 0; g < BN / 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = ld4(mp + (size_t)(k + e) * ld + 64 * g);
                b[4 * g][e] = t.x; b[4 * g + 1][e] = t.y; b[4 * g + 2][e] = t.z; b[4 * g + 3][e] = t.w;
            }
    } else if constexpr (MODE == W_ROWS) {   // mp -> row 4q, column c0 + i
#pragma unroll
        for (int y = 0; y < BN; ++y)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[y][e] = mp[(size_t)(k + e) * ld + y * 16];
    } else {                                 // mp -> row c0 + i, column 4q
#pragma unroll
        for (int y = 0; y < BN; ++y) b[y] = ld4(mp + (size_t)y * 16 * ld + k);
    }
}

// KB > 0: K == 16*KB at compile time -> straight-line code with a pinned schedule.  Two things the compiler does
// to a plain loop here: its waitcnt pass drops to vmcnt(0) at the loop header (it waits for the prefetch it has
// just issued), and, unrolled but unpinned, its scheduler sinks every load next to its first use.  So: unroll,
// keep kDepth k-blocks of fragments in flight (2: with the fetch sliced between the MFMA quarters a third block in
// flight bought nothing and cost the kernels their last spilled VGPRs) and fence "fetch block j+kDepth-1 | MFMAs of block j" with
// sched_barrier.  tools/layer_bench.hip: a 128x128 layer on one workgroup 9.1k -> 6.0k cycles (MFMA floor 4.1k).
// KB == 0: runtime K, two k-blocks per loop trip.
template <int BM, int BN, int MODE, int KB = 0>
__device__ __forceinline__ void mma_w(f32x4 (&acc)[BM][BN], lds_cf A, int lda, int m0, g_cf M, int ld, int c0, int K) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    lds_cf ap = A + (m0 + i) * lda + 4 * q;
    g_cf mp = (MODE == W_COLS) ? M + (size_t)(c0 + i) * ld + 4 * q : M + (size_t)(4 * q) * ld + c0 + (MODE == W_IL ? 4 * i : i);
    auto fetch = [&](f32x4 (&a)[BM], f32x4 (&b)[BN], int k) {
        fetch_w<BN, MODE>(b, mp, ld, k);
#pragma unroll
        for (int x = 0; x < BM; ++x) a[x] = ld4(ap + x * 16 * lda + k);
    };
    if constexpr (KB > 0) {
        constexpr int kDepth = 2;
        f32x4 a[kDepth][BM], b[kDepth][BN];
#pragma unroll
        for (int j = 0; j < kDepth - 1 && j < KB; ++j) fetch(a[j], b[j], 16 * j);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            constexpr int ahead = kDepth - 1;
            f32x4 (&an)[BM] = a[(j + ahead) % kDepth];
            f32x4 (&bn)[BN] = b[(j + ahead) % kDepth];
            const int kn = 16 * (j + ahead);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // one slice of the fetch of block j+ahead in front of each quarter of block j's MFMAs: a wave that
                // issues its loads in one bunch sits in the (CU-shared) address queue while its MFMA pipe drains
                if (j + ahead < KB) {
                    if constexpr (MODE == W_IL) {
#pragma unroll
                        for (int g = 0; g < BN / 4; ++g) {
                            const f32x4 t = ld4(mp + (size_t)(kn + e) * ld + 64 * g);
                            bn[4 * g][e] = t.x; bn[4 * g + 1][e] = t.y; bn[4 * g + 2][e] = t.z; bn[4 * g + 3][e] = t.w;
                        }
                    } else if constexpr (MODE == W_ROWS) {
#pragma unroll
                        for (int y = 0; y < BN; ++y) bn[y][e] = mp[(size_t)(kn + e) * ld + y * 16];
                    } else {
                        if (e < BN) bn[e] = ld4(mp + (size_t)e * 16 * ld + kn);
                    }
                    if (e < BM) an[e] = ld4(ap + e * 16 * lda + kn);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int x = 0; x < BM; ++x)
#pragma unroll
                    for (int y = 0; y < BN; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j % kDepth][y][e], a[j % kDepth][x][e], acc[x][y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        f32x4 a0[BM], b0[BN], a1[BM], b1[BN];
        fetch(a0, b0, 0);
        int k0 = 0;
        for (; k0 + 32 <= K; k0 += 32) {
            fetch(a1, b1, k0 + 16);
            mma_step4<BM, BN>(acc, a0, b0);
            fetch(a0, b0, min(k0 + 32, K - 16));          // last pair: harmless re-fetch of the final block
            mma_step4<BM, BN>(acc, a1, b1);
        }
        if (k0 < K) mma_step4<BM, BN>(acc, a0, b0);       // odd number of k-blocks
    }
}

// dispatch on the two contraction lengths the reference's 128-wide MLPs produce (hidden = 128, padded input = 16)
template <int BM, int BN, int MODE>
__device__ __forceinline__ void mma_w_any(f32x4 (&acc)[BM][BN], lds_cf A, int lda, int m0, g_cf M, int ld, int c0, int K) {
    if (K == 128) mma_w<BM, BN, MODE, 8>(acc, A, lda, m0, M, ld, c0, K);
    else if (K == 16) mma_w<BM, BN, MODE, 1>(acc, A, lda, m0, M, ld, c0, K);
    else mma_w<BM, BN, MODE, 0>(acc, A, lda, m0, M, ld, c0, K);
}

// f(row, col4, value4, slot): value4[j] belongs to (row, col4 + j), col4 a multiple of 4; `slot` (compile-time after
// unrolling) numbers the lane's distinct col4 values: col4 = c0 + 16q + 4*slot (W_IL) or c0 + 16*slot + 4q.
template <int BM, int BN, int MODE, class F>
__device__ __forceinline__ void tile_epilogue(const f32x4 (&acc)[BM][BN], int m0, int c0, F f) {
    const int l = lane_id(), row = l & 15, q = l >> 4;
#pragma unroll
    for (int x = 0; x < BM; ++x) {
        if constexpr (MODE == W_IL) {
#pragma unroll
            for (int g = 0; g < BN / 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    f(m0 + x * 16 + row, c0 + 64 * g + 16 * q + 4 * r,
                      f32x4{acc[x][4 * g][r], acc[x][4 * g + 1][r], acc[x][4 * g + 2][r], acc[x][4 * g + 3][r]}, 4 * g + r);
        } else {
#pragma unroll
            for (int y = 0; y < BN; ++y) f(m0 + x * 16 + row, c0 + y * 16 + 4 * q, acc[x][y], y);
        }
    }
}

// ---- weight gradient, contraction-major output: G[k][n] += sum_r X[r][k] * dY[r][n]  (both in LDS) ------
// Interleaved: the block owns 64 columns n0 + 4*j + x (x = 0..3 the four a-side tiles, fetched as one
// ds_read_b128 per step) and BN 16-wide k tiles; a lane ends up with float4s of 4 consecutive n, and a
// wave store covers 4 rows x 256 contiguous bytes of G.
template <int BN>
__device__ __forceinline__ void mma_dw_il(f32x4 (&acc)[4][BN], lds_cf dY, int ldy, int n0, lds_cf X, int ldx, int k0, int R) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    lds_cf ap = dY + (4 * q) * ldy + n0 + 4 * i;
    lds_cf bp = X + (4 * q) * ldx + k0 + i;
    for (int r0 = 0; r0 < R; r0 += 16) {
        f32x4 a[4], b[BN];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 t = ld4(ap + (r0 + e) * ldy);
            a[0][e] = t.x; a[1][e] = t.y; a[2][e] = t.z; a[3][e] = t.w;
        }
#pragma unroll
        for (int y = 0; y < BN; ++y)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[y][e] = bp[(r0 + e) * ldx + y * 16];
        mma_step4<4, BN>(acc, a, b);
    }
}
// Plain variant (n_pad not a multiple of 64, i.e. the heads): C[k][n] tiles, a-side = X (k), b-side = dY (n).
template <int BM, int BN>
__device__ __forceinline__ void mma_tn(f32x4 (&acc)[BM][BN], lds_cf A, int lda, int m0, lds_cf B, int ldb, int n0, int K) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    lds_cf ap = A + (4 * q) * lda + m0 + i;
    lds_cf bp = B + (4 * q) * ldb + n0 + i;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f32x4 a[BM], b[BN];
#pragma unroll
        for (int x = 0; x < BM; ++x)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[x][e] = ap[(k0 + e) * lda + x * 16];
#pragma unroll
        for (int y = 0; y < BN; ++y)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[y][e] = bp[(k0 + e) * ldb + y * 16];
        mma_step4<BM, BN>(acc, a, b);
    }
}

// Distribute a tm x tn grid of 16x16 output tiles over the 4 waves in register blocks of BMxBN tiles:
// 4x2 when that still gives every wave work (16 MFMAs per weight-fragment load), else 2x2, else single tiles.
// body(integral_constant<BM>, integral_constant<BN>, tile_row0, tile_col0)
template <class Body>
__device__ __forceinline__ void for_tile_blocks(int tm, int tn, Body body) {
    const int w = wave_id();
    const int total = tm * tn;
    const int need = total < kWaves ? total : kWaves;
    if ((tm % 4 == 0) && (tn % 2 == 0) && (tm / 4) * (tn / 2) >= need) {
        const int nbm = tm / 4, nb = nbm * (tn / 2);
        for (int blk = w; blk < nb; blk += kWaves)
            body(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, (blk % nbm) * 4, (blk / nbm) * 2);
    } else if ((tm % 2 == 0) && (tn % 2 == 0) && (tm / 2) * (tn / 2) >= need) {
        const int nbm = tm / 2, nb = nbm * (tn / 2);
        for (int blk = w; blk < nb; blk += kWaves)
            body(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, (blk % nbm) * 2, (blk / nbm) * 2);
    } else {
        for (int blk = w; blk < total; blk += kWaves)
            body(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, blk % tm, blk / tm);
    }
}

// Interleaved blocks: tm row tiles x (tn / 4) groups of four column tiles.  BM = 2 when every wave still
// gets a block, else 1.  body(integral_constant<BM>, tile_row0, group)
// (a BM = 4 variant for 128-row chunks was measured: -4 % on the forward layer, but its extra instantiation in every
// call site cost the 64-row kernels 50-100 spilled VGPRs and 8 % of their speed)
template <class Body>
__device__ __forceinline__ void for_il_blocks(int tm, int groups, Body body) {
    const int w = wave_id();
    if ((tm % 2 == 0) && (tm / 2) * groups >= kWaves) {
        const int nbm = tm / 2, nb = nbm * groups;
        for (int blk = w; blk < nb; blk += kWaves) body(std::integral_constant<int, 2>{}, (blk % nbm) * 2, blk / nbm);
    } else {
        const int nb = tm * groups;
        for (int blk = w; blk < nb; blk += kWaves) body(std::integral_constant<int, 1>{}, blk % tm, blk / tm);
    }
}

// ---- block-wide sum (all threads get the result); `red` = 8 floats of LDS scratch -----------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += lane_xor(v, off);
    return v;
}
__device__ __forceinline__ float block_sum(float v, lds_f red) {
    v = wave_sum(v);
    __syncthreads();                       // protect `red` from a previous use
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace frl
