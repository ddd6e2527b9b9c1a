// Categorical (C51) head arithmetic shared by the act and update kernels (DQN_with_tricks.py:112-126).
#pragma once
#include "tile.hpp"

namespace frl {

// Turn the head outputs of `rows` rows in outb into the logits [a][i] the reference forms (:113-120): plain head: as is
// (offset 0); Dueling: logit[a][i] = (V_i + A_a,i) - mean_a A_a,i written over the A block (offset `atoms`).  Returns the
// column offset of logit[0][0].  Ends with a barrier.
__device__ __forceinline__ int c51_combine(lds_f outb, int op, int rows, int nA, int atoms, bool duel) {
    if (!duel) return 0;
    for (int e = threadIdx.x; e < rows * atoms; e += kWG) {
        const int r = e / atoms, i = e - r * atoms;
        lds_f o = outb + r * op;
        float mean = 0.f;
        for (int b = 0; b < nA; ++b) mean += o[atoms + b * atoms + i];
        mean /= (float)nA;
        const float v = o[i];
        for (int b = 0; b < nA; ++b) o[atoms + b * atoms + i] = (v + o[atoms + b * atoms + i]) - mean;
    }
    lds_barrier();
    return atoms;
}

// softmax over the atoms of one action's logits lg[0..atoms): expected value q = sum_i z_i p_i; p_out (may be null) gets p
__device__ __forceinline__ float c51_q(lds_cf lg, int atoms, float vmin, float dz, lds_f p_out) {
    float mx = lg[0];
    for (int i = 1; i < atoms; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
    for (int i = 0; i < atoms; ++i) sum += expf(lg[i] - mx);
    float q = 0.f;
    for (int i = 0; i < atoms; ++i) {
        const float p = expf(lg[i] - mx) / sum;
        if (p_out) p_out[i] = p;
        q += p * (vmin + dz * (float)i);
    }
    return q;
}

// Expected value only, one exp per atom: q = (sum_i z_i e_i) / (sum_i e_i) — the same number as c51_q's sum_i z_i (e_i / sum)
// up to the last bits; used where only the argmax over actions is wanted.
__device__ __forceinline__ float c51_q_only(lds_cf lg, int atoms, float vmin, float dz) {
    float mx = lg[0];
    for (int i = 1; i < atoms; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f, zsum = 0.f;
    for (int i = 0; i < atoms; ++i) {
        const float e = expf(lg[i] - mx);
        sum += e;
        zsum += e * (vmin + dz * (float)i);
    }
    return zsum / sum;
}

// Softmax of R rows' logits by one WAVE (lane i = atom i, atoms <= 64), the R rows' shuffle reductions interleaved: a
// __shfl_xor is a ds_bpermute round trip (~130 cycles), and one row at a time the 18 of them were the whole phase.
// p[k] = this lane's probability in row k (0 beyond the support), q[k] = expected value (all lanes).  Sums run as shuffle
// trees instead of ascending chains: last-bit differences to c51_q.
template <int R>
__device__ __forceinline__ void wave_sum_n(float (&v)[R]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] += lane_xor(v[k], off);
}
template <int R>
__device__ __forceinline__ void c51_softmax_wave_n(lds_cf (&lg)[R], int atoms, float vmin, float dz, float (&p)[R], float (&q)[R]) {
    const int l = lane_id();
    float x[R], mx[R], e[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { x[k] = l < atoms ? lg[k][l] : -3.0e38f; mx[k] = x[k]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < R; ++k) mx[k] = fmaxf(mx[k], lane_xor(mx[k], off));
#pragma unroll
    for (int k = 0; k < R; ++k) { e[k] = l < atoms ? expf(x[k] - mx[k]) : 0.f; p[k] = e[k]; }
    wave_sum_n<R>(p);                                 // p = sum of e
#pragma unroll
    for (int k = 0; k < R; ++k) { p[k] = e[k] / p[k]; q[k] = p[k] * (vmin + dz * (float)l); }
    wave_sum_n<R>(q);
}

}  // namespace frl
