// Categorical (C51) head arithmetic shared by the act and update kernels (DQN_with_tricks.py:112-126).
#pragma once
#include "tile.hpp"

namespace frl {

// combined logit (row r, action a, atom i) out of the head outputs in outb
__device__ __forceinline__ float c51_logit(lds_cf o, int nA, int atoms, bool duel, int a, int i) {
    if (!duel) return o[a * atoms + i];
    float mean = 0.f;
    for (int b = 0; b < nA; ++b) mean += o[atoms + b * atoms + i];
    mean /= (float)nA;
    return (o[i] + o[atoms + a * atoms + i]) - mean;
}

// expected value q = sum_i z_i softmax(logits)_i of (row, action); also usable to fetch the probabilities
__device__ __forceinline__ float c51_q(lds_cf o, int nA, int atoms, bool duel, int a, float vmin, float dz, float* p_out /* [atoms] or null */) {
    float mx = -3.4e38f;
    for (int i = 0; i < atoms; ++i) mx = fmaxf(mx, c51_logit(o, nA, atoms, duel, a, i));
    float sum = 0.f;
    for (int i = 0; i < atoms; ++i) sum += expf(c51_logit(o, nA, atoms, duel, a, i) - mx);
    float q = 0.f;
    for (int i = 0; i < atoms; ++i) {
        const float p = expf(c51_logit(o, nA, atoms, duel, a, i) - mx) / sum;
        if (p_out) p_out[i] = p;
        q += p * (vmin + dz * (float)i);
    }
    return q;
}

}  // namespace frl
