// Special functions for the Beta policy head (PPO_with_tricks.py:120-151, torch.distributions.Beta): digamma and
// trigamma for arguments >= 1 (alpha, beta = softplus(.) + 1).  Evaluated in double: these run once per (row, action
// dimension) on a thread-per-row path, and torch's own fp32 lgamma/digamma then set the agreement floor, not ours.
#pragma once
#include <hip/hip_runtime.h>

namespace frl {

__device__ __forceinline__ double digamma_d(double x) {
    double r = 0.0;
    while (x < 8.0) { r -= 1.0 / x; x += 1.0; }          // psi(x) = psi(x + 1) - 1/x
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
}

__device__ __forceinline__ double trigamma_d(double x) {
    double r = 0.0;
    while (x < 8.0) { r += 1.0 / (x * x); x += 1.0; }    // psi'(x) = psi'(x + 1) + 1/x^2
    const double f = 1.0 / (x * x);
    return r + 1.0 / x + 0.5 * f + (1.0 / x) * f * (1.0 / 6 - f * (1.0 / 30 - f * (1.0 / 42 - f * (1.0 / 30))));
}

// log B(a, b)
__device__ __forceinline__ double lbeta_d(double a, double b) { return lgamma(a) + lgamma(b) - lgamma(a + b); }

}  // namespace frl
