// MFMA tile toolkit for the fused update kernels (gfx950 / CDNA4 only).
//
// One workgroup = 256 threads = 4 wave64.  Activations of a row-chunk (rc rows of the batch)
// live in LDS; weights are read straight from global memory (L2 resident, engine-internal
// zero-padded layout W[n_pad][k_pad], n_pad and k_pad multiples of 16).  All matrix work is
// v_mfma_f32_16x16x4_f32 (exact fp32, bit-for-bit an fma chain):
//     A operand: lane l holds A[i = l&15][k = l>>4]       (one VGPR)
//     B operand: lane l holds B[k = l>>4][j = l&15]       (one VGPR)
//     C/D      : lane l, reg r holds D[row = (l>>4)*4 + r][col = l&15]
// (the toolkit passes the operand that indexes the CONTIGUOUS output dimension as MFMA-A, so a
//  lane's 4 accumulator registers are 4 consecutive output elements: 16-byte epilogues)
// The contraction index may be permuted freely as long as A and B agree; the "contiguous"
// operand modes load a float4 along the contraction (k = k0 + 4*(l>>4) + e for MFMA step e),
// the "strided" modes load one scalar per step from rows k0 + 4*(l>>4) + e.
//
// Pointers carry their address space in the type (lds_* = LDS, g_* = global): operand fetches
// compile to ds_read_b128 / global_load_dwordx4 instead of flat loads, and LDS / global waits
// use separate counters.  Operand fragments are fetched one k-block ahead of the MFMAs that
// consume them (software pipelining: the L2 latency of the weight fragments hides behind the
// 4*BM*BN MFMAs of the previous block).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

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

// ---- C[m][n] += sum_k A[m][k] * B[n][k]  (forward: Y = X * W^T; A in LDS, B in global) -------
// A(i,kk) = A[(m0+i)*lda + kk], B(kk,j) = B[(n0+j)*ldb + kk]; K multiple of 16.
template <int BM, int BN>
__device__ __forceinline__ void mma_nt(f32x4 (&acc)[BM][BN], lds_cf A, int lda, int m0, g_cf B, int ldb, int n0, int K) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    lds_cf ap = A + (m0 + i) * lda + 4 * q;
    g_cf bp = B + (size_t)(n0 + i) * ldb + 4 * q;
    // ping-pong fragment buffers, unrolled by two k-blocks: the fetch of block k+1 is issued
    // before the MFMAs of block k and waited for (counted) only after them
    f32x4 a0[BM], b0[BN], a1[BM], b1[BN];
#pragma unroll
    for (int y = 0; y < BN; ++y) b0[y] = ld4(bp + (size_t)y * 16 * ldb);
#pragma unroll
    for (int x = 0; x < BM; ++x) a0[x] = ld4(ap + x * 16 * lda);
    int k0 = 0;
    for (; k0 + 32 <= K; k0 += 32) {
#pragma unroll
        for (int y = 0; y < BN; ++y) b1[y] = ld4(bp + (size_t)y * 16 * ldb + k0 + 16);
#pragma unroll
        for (int x = 0; x < BM; ++x) a1[x] = ld4(ap + x * 16 * lda + k0 + 16);
        mma_step4<BM, BN>(acc, a0, b0);
        const int kn = min(k0 + 32, K - 16);          // last pair: harmless re-fetch of the final block
#pragma unroll
        for (int y = 0; y < BN; ++y) b0[y] = ld4(bp + (size_t)y * 16 * ldb + kn);
#pragma unroll
        for (int x = 0; x < BM; ++x) a0[x] = ld4(ap + x * 16 * lda + kn);
        mma_step4<BM, BN>(acc, a1, b1);
    }
    if (k0 < K) mma_step4<BM, BN>(acc, a0, b0);       // odd number of k-blocks
}

// ---- C[m][j] += sum_n A[m][n] * B[n][j]  (input grad: dX = dY * W; A in LDS, B in global) ----
// A(i,kk) = A[(m0+i)*lda + kk] (contiguous), B(kk,j) = B[kk*ldb + n0 + j] (strided); K mult of 16.
template <int BM, int BN>
__device__ __forceinline__ void mma_nn(f32x4 (&acc)[BM][BN], lds_cf A, int lda, int m0, g_cf B, int ldb, int n0, int K) {
    const int l = lane_id(), i = l & 15, q = l >> 4;
    lds_cf ap = A + (m0 + i) * lda + 4 * q;
    g_cf bp = B + (size_t)(4 * q) * ldb + n0 + i;
    f32x4 a0[BM], b0[BN], a1[BM], b1[BN];
    auto fetch = [&](f32x4 (&a)[BM], f32x4 (&b)[BN], int k) {
#pragma unroll
        for (int y = 0; y < BN; ++y)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[y][e] = bp[(size_t)(k + e) * ldb + y * 16];
#pragma unroll
        for (int x = 0; x < BM; ++x) a[x] = ld4(ap + x * 16 * lda + k);
    };
    fetch(a0, b0, 0);
    int k0 = 0;
    for (; k0 + 32 <= K; k0 += 32) {
        fetch(a1, b1, k0 + 16);
        mma_step4<BM, BN>(acc, a0, b0);
        fetch(a0, b0, min(k0 + 32, K - 16));
        mma_step4<BM, BN>(acc, a1, b1);
    }
    if (k0 < K) mma_step4<BM, BN>(acc, a0, b0);
}

// ---- C[n][k] += sum_r A[r][n] * B[r][k]  (weight grad: dW = dY^T * X; both in LDS) -----------
// A(i,kk) = A[kk*lda + m0 + i], B(kk,j) = B[kk*ldb + n0 + j]; K (rows) multiple of 16.
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

// mma_step4 feeds the "B-side" fragment (second operand of mma_*) as the MFMA's A operand, so
// in D the lane's 4 registers are 4 CONSECUTIVE indices of the second (n / column) dimension
// and lane&15 is the first (m / row) dimension: every epilogue access is one 16-byte vector.
// f(row, col4, value4): value4[r] belongs to (row, col4 + r), col4 a multiple of 4.
template <int BM, int BN, class F>
__device__ __forceinline__ void tile_epilogue(const f32x4 (&acc)[BM][BN], int m0, int n0, F f) {
    const int l = lane_id(), row = l & 15, col4 = (l >> 4) * 4;
#pragma unroll
    for (int x = 0; x < BM; ++x)
#pragma unroll
        for (int y = 0; y < BN; ++y) f(m0 + x * 16 + row, n0 + y * 16 + col4, acc[x][y]);
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
__device__ __forceinline__ float block_sum(float v, lds_f red) {
    v = wave_sum(v);
    __syncthreads();                       // protect `red` from a previous use
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace frl
