// MFMA tile toolkit for the fused update kernels (gfx950 / CDNA4 only).
//
// One workgroup = 256 threads = 4 wave64.  Activations of a row-chunk (RC rows of the batch)
// live in LDS; weights are read straight from global memory (L2/MALL resident, engine-internal
// zero-padded layout W[n_pad][k_pad], n_pad and k_pad multiples of 16).  All matrix work is
// v_mfma_f32_16x16x4_f32 (exact fp32, bit-for-bit an fma chain):
//     A operand: lane l holds A[i = l&15][k = l>>4]       (one VGPR)
//     B operand: lane l holds B[k = l>>4][j = l&15]       (one VGPR)
//     C/D      : lane l, reg r holds D[row = (l>>4)*4 + r][col = l&15]
// The contraction index may be permuted freely as long as A and B agree; the "contiguous"
// operand modes load a float4 along the contraction (k = k0 + 4*(l>>4) + e for MFMA step e),
// the "strided" modes load one scalar per step from rows k0 + 4*(l>>4) + e.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace frl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// ---- C[m][n] += sum_k A[m][k] * B[n][k]  (forward: Y = X * W^T; A in LDS, B in global) -------
// A(i,kk) = A[(m0+i)*lda + kk], B(kk,j) = B[(n0+j)*ldb + kk]; K multiple of 16.
template <int BM, int BN>
__device__ __forceinline__ void mma_nt(f32x4 (&acc)[BM][BN], const float* A, int lda, int m0,
                                       const float* __restrict__ B, int ldb, int n0, int K) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    const float* ap = A + (m0 + i) * lda + 4 * q;
    const float* bp = B + (size_t)(n0 + i) * ldb + 4 * q;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f32x4 a[BM], b[BN];
#pragma unroll
        for (int x = 0; x < BM; ++x) a[x] = *reinterpret_cast<const f32x4*>(ap + x * 16 * lda + k0);
#pragma unroll
        for (int y = 0; y < BN; ++y) b[y] = *reinterpret_cast<const f32x4*>(bp + (size_t)y * 16 * ldb + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int x = 0; x < BM; ++x)
#pragma unroll
                for (int y = 0; y < BN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x][e], b[y][e], acc[x][y], 0, 0, 0);
    }
}

// ---- C[m][j] += sum_n A[m][n] * B[n][j]  (input grad: dX = dY * W; A in LDS, B in global) ----
// A(i,kk) = A[(m0+i)*lda + kk] (contiguous), B(kk,j) = B[kk*ldb + n0 + j] (strided); K mult of 16.
template <int BM, int BN>
__device__ __forceinline__ void mma_nn(f32x4 (&acc)[BM][BN], const float* A, int lda, int m0,
                                       const float* __restrict__ B, int ldb, int n0, int K) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    const float* ap = A + (m0 + i) * lda + 4 * q;
    const float* bp = B + (size_t)(4 * q) * ldb + n0 + i;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f32x4 a[BM];
        float b[BN][4];
#pragma unroll
        for (int x = 0; x < BM; ++x) a[x] = *reinterpret_cast<const f32x4*>(ap + x * 16 * lda + k0);
#pragma unroll
        for (int y = 0; y < BN; ++y)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[y][e] = bp[(size_t)(k0 + e) * ldb + y * 16];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int x = 0; x < BM; ++x)
#pragma unroll
                for (int y = 0; y < BN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x][e], b[y][e], acc[x][y], 0, 0, 0);
    }
}

// ---- C[n][k] += sum_r A[r][n] * B[r][k]  (weight grad: dW = dY^T * X; both in LDS) -----------
// A(i,kk) = A[kk*lda + m0 + i], B(kk,j) = B[kk*ldb + n0 + j]; K (rows) multiple of 4.
template <int BM, int BN>
__device__ __forceinline__ void mma_tn(f32x4 (&acc)[BM][BN], const float* A, int lda, int m0, const float* B,
                                       int ldb, int n0, int K) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    const float* ap = A + q * lda + m0 + i;
    const float* bp = B + q * ldb + n0 + i;
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 4) {
        float a[BM], b[BN];
#pragma unroll
        for (int x = 0; x < BM; ++x) a[x] = ap[k0 * lda + x * 16];
#pragma unroll
        for (int y = 0; y < BN; ++y) b[y] = bp[k0 * ldb + y * 16];
#pragma unroll
        for (int x = 0; x < BM; ++x)
#pragma unroll
            for (int y = 0; y < BN; ++y)
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
    }
}

// f(row, col, value) for every element of a BMxBN block of 16x16 tiles at tile origin (m0,n0)
template <int BM, int BN, class F>
__device__ __forceinline__ void tile_epilogue(const f32x4 (&acc)[BM][BN], int m0, int n0, F f) {
    const int l = lane_id(), col = l & 15, row = (l >> 4) * 4;
#pragma unroll
    for (int x = 0; x < BM; ++x)
#pragma unroll
        for (int y = 0; y < BN; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) f(m0 + x * 16 + row + r, n0 + y * 16 + col, acc[x][y][r]);
}

// Distribute a tm x tn grid of 16x16 output tiles over the 4 waves in register blocks of
// BMxBN tiles: 4x2 when that still gives every wave work, else 2x2, else single tiles.
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

// ---- block-wide sum (all threads get the result); `red` = 8 floats of LDS scratch -----------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                       // protect `red` from a previous use
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace frl
