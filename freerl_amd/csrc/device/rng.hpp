// Counter-based device RNG (Philox4x32-10) for the fast path: sample indices without
// replacement and Gaussian noise, reproducible from (seed, learner, call counter).
#pragma once
#include <hip/hip_runtime.h>

namespace frl {

struct Philox4 { unsigned x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long counter, unsigned stream, unsigned sub,
                                                 unsigned long long key) {
    unsigned c0 = (unsigned)counter, c1 = (unsigned)(counter >> 32), c2 = stream, c3 = sub;
    unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// uniform integer in [0, n) from 64 random bits (multiply-high; bias n/2^64)
__device__ __forceinline__ unsigned uniform_index(const Philox4& r, unsigned n) {
    const unsigned long long bits = ((unsigned long long)r.x << 32) | r.y;
    return (unsigned)__umul64hi(bits, (unsigned long long)n);
}

__device__ __forceinline__ float u01(unsigned bits) {            // (0,1]
    return ((float)(bits >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// two independent standard normals (Box-Muller)
__device__ __forceinline__ void normal2(const Philox4& r, float& n0, float& n1) {
    const float u = u01(r.z), v = u01(r.w);
    const float rad = sqrtf(-2.0f * logf(u));
    float s, c;
    sincosf(6.28318530717958647692f * v, &s, &c);
    n0 = rad * c;
    n1 = rad * s;
}

}  // namespace frl
