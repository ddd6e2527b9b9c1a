// Building blocks of the register-chained kernels (kernels_ppo2.hip, kernels_critic2.hip): every wave carries 16 rows through
// a whole MLP in registers.  In the transposed formulation Z[out][row] = W[out][in] H[in][row] the 16x16 output tile of a
// layer (v_mfma_f32_16x16x4_f32: D[4q + r][lane & 15]) is the B operand of the next (B[4q + e][lane & 15]); weights come
// from LDS images in MFMA-fragment order, swizzled so that the forward's ds_read_b128 fragments, the backward's
// transposed ds_read_b32 fragments and the owners' ds_write_b32 are all bank-conflict free.
#pragma once
#include <type_traits>

#include "net.hpp"

namespace frl {

constexpr int kHid = 128, kHT = kHid / 16;

// Developer ablations (timing only — the results are WRONG): bit 0 no clip + Adam + soft update, bit 1 no exchange writes / barriers
// in the eight-wave backward, bit 2 no barriers in stage_commit, bit 3 no exchange writes (barriers kept)
#ifndef FRL_ABL
#define FRL_ABL 0
#endif

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N) — for bodies that index register arrays and template
// parameters with the loop variable
template <int I> using IC = std::integral_constant<int, I>;
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}

// v[c] for a runtime c without a dynamically indexed register array (which hipcc would put in scratch memory): a chain of
// selects over compile-time indices
template <int I, int N, class T>
__device__ __forceinline__ T pick_from(const T (&v)[N], int c, T r) {
    if constexpr (I < N) return pick_from<I + 1, N, T>(v, c, c == I ? v[I] : r);
    else return r;
}
template <int N, class T>
__device__ __forceinline__ T pick(const T (&v)[N], int c) { return pick_from<1, N, T>(v, c, v[0]); }

// dword offset of element (f16, k16) of 16x16 fragment tile `tile` in a fragment-ordered LDS image: the 16-byte slot of
// (q = k16 >> 2, f16) sits at q*16 + (f16 ^ q)
__device__ __forceinline__ int frag_dw(int tile, int f16, int k16) {
    const int q = k16 >> 2;
    return tile * 256 + ((q * 16 + (f16 ^ q)) << 2) + (k16 & 3);
}

template <int HACT>
__device__ __forceinline__ float hact_fwd(float x) { return HACT == ACT_TANH ? tanhf(x) : fmaxf(x, 0.f); }
template <int HACT>
__device__ __forceinline__ float hact_grad(float h) { return HACT == ACT_TANH ? 1.f - h * h : (h > 0.f ? 1.f : 0.f); }

__device__ __forceinline__ f32x4 mfma4(f32x4 acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    return acc;
}

// torch's single-tensor Adam on one element (clip coefficient already folded into g).  The moments are exact fp32
// fma chains like everywhere else; the step itself, step * m / (sqrt(v) / sqrt(bc2) + eps), uses the hardware's 1-ulp
// sqrt and reciprocal instead of the correctly rounded sequences (3 x ~10 VALU instructions per element, 88 elements per
// lane and step: the difference between a 28 k and an 8 k cycle Adam phase).  Its relative error (<= ~3 ulp of the UPDATE,
// which is itself ~lr times smaller than the parameter) is below the rounding of the subtraction that applies it.
__device__ __forceinline__ float adam_elem(float th, float g, float& m, float& v, float w1, float w2, float b2, float inv_bc2s,
                                           float eps, float step) {
    m = m + (g - m) * w1;
    v = v * b2 + (w2 * g) * g;
    const float denom = __builtin_amdgcn_sqrtf(v) * inv_bc2s + eps;
    return th - step * (m * __builtin_amdgcn_rcpf(denom));
}

}  // namespace frl
