// Developer microbenchmark of the layer primitives in device/net.hpp, in isolation: one workgroup runs
// REPS x {forward 128x128, dX 128x128, dW 128x128} on an LDS-resident 32-row chunk, weights L2 resident.
// Prints shader cycles per call for 1 and 4 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I freerl_amd/csrc -o tools/_bin/layer_bench tools/layer_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "device/net.hpp"
using namespace frl;

#ifndef RC_ROWS
#define RC_ROWS 64
#endif
constexpr int H = 128, RC = RC_ROWS;

template <int mode>
__global__ __launch_bounds__(256, 2) void k(const float* theta_all, float* slab_all, long long* cyc, int reps, LayerDesc L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds S = carve_lds(smem, RC, H, 16, 16, 64, 4);
    g_cf theta = as_global(theta_all) + (size_t)(blockIdx.x / 8) * 32768;
    g_f slab = as_global(slab_all) + (size_t)blockIdx.x * 32768;
    for (int e = threadIdx.x; e < RC * S.xp; e += kWG) S.xin[e] = 0.003f * (e % 31);
    for (int e = threadIdx.x; e < RC * S.hp; e += kWG) { S.h1[e] = 0.001f * (e % 97); S.h2[e] = 0.002f * (e % 89); }
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) linear_fwd(L, theta, (r & 1) ? S.h2 : S.h1, S.hp, (r & 1) ? S.h1 : S.h2, S.hp, ACT_RELU, RC);
        if (mode == 1) linear_bwd_dx(L, theta, (r & 1) ? S.h2 : S.h1, S.hp, (r & 1) ? S.h1 : S.h2, S.hp, ACT_RELU, RC, 0, 8);
        if (mode == 2) linear_bwd_dw(L, slab, S.h1, S.hp, S.h2, S.hp, RC, true);
        if (mode == 3) linear_fwd(L, theta, S.xin, S.xp, (r & 1) ? S.h1 : S.h2, S.hp, ACT_RELU, RC);      // K = 16 input layer
        if (mode == 4) linear_fwd(L, theta, (r & 1) ? S.h2 : S.h1, S.hp, S.outb, S.op, ACT_NONE, RC);     // N = 16 head
        __syncthreads();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    LayerDesc L{H, H, H, H, 0, H * H};
    float *theta, *slab;
    long long* cyc;
    const int maxwg = 1024;
    hipMalloc(&theta, (size_t)(maxwg / 8) * 32768 * 4);
    hipMalloc(&slab, (size_t)maxwg * 32768 * 4);
    hipMalloc(&cyc, maxwg * 8);
    std::vector<float> h((size_t)(maxwg / 8) * 32768);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * ((i * 7919) % 13) - 0.06f;
    hipMemcpy(theta, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int lds = (RC * (20 + 2 * 132 + 20 + 8) + 64 + 8) * 4;   // 80 KB at 64 rows: two workgroups per CU
    const int reps = 64;
    const char* names[5] = {"fwd 128x128", "dX 128x128", "dW 128x128", "fwd 16->128", "fwd 128->16"};
    LayerDesc L1{H, 10, H, 16, 0, 16 * H}, L3{1, H, 16, H, 0, 16 * H};
    for (int mode = 0; mode < 5; ++mode)
        for (int wg : {256, 512}) {
            for (int it = 0; it < 2; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L1);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L3);
                hipDeviceSynchronize();
            }
            std::vector<long long> c(wg);
            hipMemcpy(c.data(), cyc, wg * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : c) s += v;
            printf("%-12s %4d workgroups (%d per CU): %8.0f cycles per call  (MFMA floor %d)\n", names[mode], wg, wg / 256,
                   s / wg / reps, (mode < 3 ? 128 : 16) * 32 * (wg / 256) * (RC / 32));
        }
    return 0;
}
