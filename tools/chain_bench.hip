// Developer microbenchmark of device/chain_net.hpp in isolation: one workgroup per CU stages a net once and then runs REPS x
// {forward<1>, forward<2>, forward<1> + backward} out of the LDS images; prints shader cycles per call and the MFMA floor
// (v_mfma_f32_16x16x4_f32: 32 cycles per SIMD).  Variants (-DVAR=n) switch one thing at a time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I freerl_amd/csrc -o tools/_bin/chain_bench tools/chain_bench.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>
#include "device/chain_net.hpp"
using namespace frl;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* theta, long long* cyc, float* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ChainNet C;
    C.init(smem);
    C.stage(as_global(theta) + (size_t)blockIdx.x * kHeadFloats, 0, 0);
    f32x4 x1[1] = {f32x4{0.01f * C.i16, 0.02f, -0.01f * C.q, 0.03f}};
    f32x4 x2[2] = {x1[0], f32x4{0.02f, -0.01f * C.i16, 0.01f, 0.f}};
    HeadGrad g;
    C.grad_zero(g);
    float acc = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if constexpr (MODE == 0) {
            f32x4 z[1], h1[1][kHT], h2[1][kHT];
            C.forward<1>(x1, h1, h2, z);
            acc += z[0][0]; x1[0][1] += 1e-6f * z[0][1];
        } else if constexpr (MODE == 1) {
            f32x4 z[2], h1[2][kHT], h2[2][kHT];
            C.forward<2>(x2, h1, h2, z);
            acc += z[0][0] + z[1][0]; x2[0][1] += 1e-6f * z[1][1];
        } else if constexpr (MODE == 2) {
            f32x4 z[1], h1[1][kHT], h2[1][kHT];
            C.forward<1>(x1, h1, h2, z);
            f32x4 dz = {z[0][0] * 1e-3f, 0.f, 0.f, 0.f};
            C.backward(g, x1[0], h1[0], h2[0], dz);
            x1[0][1] += 1e-6f * z[0][1];
        } else if constexpr (MODE == 4) {
            f32x4 z[1], h1[1][kHT], h2[1][kHT];
            C.forward_vh<1>(x1, h1, h2, z, 1);
            acc += z[0][0]; x1[0][1] += 1e-6f * z[0][1];
        } else if constexpr (MODE == 5) {
            f32x4 z[2], h1[2][kHT], h2[2][kHT];
            C.forward_vh<2>(x2, h1, h2, z, 1);
            acc += z[0][0] + z[1][0]; x2[0][1] += 1e-6f * z[1][1];
        } else if constexpr (MODE == 6) {
            f32x4 z[1], h1[1][kHT], h2[1][kHT];
            C.forward_vh<1>(x1, h1, h2, z, 1);
            f32x4 dz = {z[0][0] * 1e-3f, 0.f, 0.f, 0.f};
            C.backward(g, x1[0], h1[0], h2[0], dz, 1);
            x1[0][1] += 1e-6f * z[0][1];
        } else if constexpr (MODE == 3) {      // forward<2> + the dX-only chain of the actor stage's pass B
            f32x4 z[2], h1[2][kHT], h2[2][kHT];
            C.forward<2>(x2, h1, h2, z);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 dz = {1e-3f, 0.f, 0.f, 0.f}, d2[kHT], d1[kHT];
                C.delta2(dz, h2[t], d2);
                C.delta1(d2, h1[t], d1);
                const f32x4 dx = C.delta0(d1);
                acc += dx[0];
            }
            x2[0][1] += 1e-6f * z[1][1];
        }
    }
    const long long t1 = clock64();
    if (MODE == 2 || MODE == 6) acc += C.grad_sumsq(g);
    if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int G = 256, reps = 200;
    float* theta; long long* cyc; float* sink;
    hipMalloc(&theta, sizeof(float) * (size_t)G * kHeadFloats);
    hipMalloc(&cyc, sizeof(long long) * G);
    hipMalloc(&sink, sizeof(float) * G * 256);
    std::vector<float> h((size_t)G * kHeadFloats);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 201) - 1.f;
    hipMemcpy(theta, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    const size_t lds = (size_t)chain_lds_floats() * sizeof(float);
    const char* names[7] = {"forward<1>             (320 MFMA)", "forward<2>             (640 MFMA)", "forward<1> + backward  (928 MFMA)",
                            "forward<2> + 2 x dX    (1280 MFMA)", "forward<1>, 1-output head on VALU (288 MFMA)",
                            "forward<2>, 1-output head on VALU (576 MFMA)", "forward<1> + backward, head + its dH on VALU (864 MFMA)"};
    const int mfma[7] = {320, 640, 928, 1280, 288, 576, 864};
    auto run = [&](auto kern, int mode) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int it = 0; it < 2; ++it) {
            hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, 0, theta, cyc, sink, reps);
            hipDeviceSynchronize();
        }
        std::vector<long long> c(G);
        hipMemcpy(c.data(), cyc, sizeof(long long) * G, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += (double)v;
        s /= G;
        printf("%s  %8.0f cycles per call   MFMA floor %6d   -> %.1f %% of the issue rate\n", names[mode], s, mfma[mode] * 32, 100.0 * mfma[mode] * 32 / s);
    };
    run(k<0>, 0); run(k<1>, 1); run(k<2>, 2); run(k<3>, 3); run(k<4>, 4); run(k<5>, 5); run(k<6>, 6);
    return 0;
}
