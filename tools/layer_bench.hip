// Developer microbenchmark of the layer primitives in device/net.hpp, in isolation: one workgroup runs
// REPS x {forward 128x128, dX 128x128, dW 128x128} on an LDS-resident 32-row chunk, weights L2 resident.
// Prints shader cycles per call for 1 and 4 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I freerl_amd/csrc -o tools/_bin/layer_bench tools/layer_bench.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>
#include "device/net.hpp"
using namespace frl;

#ifndef WSHARE
#define WSHARE 8      // workgroups per weight set: 512 workgroups / 8 = 64 sets x 64 KB = the 4 MB of one XCD's L2
#endif
#ifndef RC_ROWS
#define RC_ROWS 64
#endif
constexpr int H = 128, RC = RC_ROWS;

template <int mode>
__global__ __launch_bounds__(256, RC_ROWS > 64 ? 1 : 2) void k(const float* theta_all, float* slab_all, long long* cyc, int reps, LayerDesc L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds S = carve_lds(smem, RC, H, 16, 16, 64, 4);
    g_cf theta = as_global(theta_all) + (size_t)(blockIdx.x / WSHARE) * 32768;
    g_f slab = as_global(slab_all) + (size_t)blockIdx.x * 32768;
    for (int e = threadIdx.x; e < RC * S.xp; e += kWG) S.xin[e] = 0.003f * (e % 31);
    for (int e = threadIdx.x; e < RC * S.hp; e += kWG) { S.h1[e] = 0.001f * (e % 97); S.h2[e] = 0.002f * (e % 89); }
    __syncthreads();
#ifdef STAGGER
    // head start for the workgroup whose wave 0 sits in an odd wave slot of its SIMD (HW_ID bits 3:0): do two
    // co-resident workgroups overlap each other's prologue / epilogue / barrier with MFMA work once out of lockstep?
    {
        __shared__ int slot;
        if (threadIdx.x == 0) slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1;   // HW_REG_HW_ID[3:0]
        __syncthreads();
        if (slot) { const long long s0 = clock64(); while (clock64() - s0 < STAGGER) {} }
        if (threadIdx.x == 0 && slot) atomicAdd((int*)(cyc + 8000), 1);
    }
#endif
    long long st[3] = {0, 0, 0};
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) linear_fwd(L, theta, (r & 1) ? S.h2 : S.h1, S.hp, (r & 1) ? S.h1 : S.h2, S.hp, ACT_RELU, RC);
        if (mode == 1) linear_bwd_dx(L, theta, (r & 1) ? S.h2 : S.h1, S.hp, (r & 1) ? S.h1 : S.h2, S.hp, ACT_RELU, RC, 0, 8);
        if (mode == 2) linear_bwd_dw(L, slab, S.h1, S.hp, S.h2, S.hp, RC, true);
        if (mode == 3) linear_fwd(L, theta, S.xin, S.xp, (r & 1) ? S.h1 : S.h2, S.hp, ACT_RELU, RC);      // K = 16 input layer
        if (mode == 4) linear_fwd(L, theta, (r & 1) ? S.h2 : S.h1, S.hp, S.outb, S.op, ACT_NONE, RC);     // N = 16 head
        if (mode == 9) head_bwd(L, theta, slab, S.outb, S.op, (r & 1) ? S.h2 : S.h1, S.hp, ACT_RELU, RC, GS_STREAM);   // 128 -> 1 head, one VALU pass
        if (mode == 5 || mode == 6) {       // the forward's MFMA loop alone: per call (5) / one long contraction (6), no epilogue, no barrier
            f32x4 acc[2][4];
            acc_zero(acc);
            const int w = wave_id();
            if (mode == 5) {
                for (int c = 0; c < 8; ++c) mma_w<2, 4, W_IL, 8>(acc, S.h1, S.hp, (w & 1) * 32, theta, H, (w >> 1) * 64, 128);
            } else {
                mma_w<2, 4, W_IL, 64>(acc, S.h1, S.hp, (w & 1) * 32, theta, H, (w >> 1) * 64, 1024);
            }
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int x = 0; x < 2; ++x) for (int y = 0; y < 4; ++y) t += acc[x][y];
            if (t.x + t.y + t.z + t.w == 12345.f) S.h2[threadIdx.x] = t.x;
            r += 7;
        }
        if (mode == 8) {       // wave-private rows: wave w owns rows 16w.., all 128 columns; no workgroup barrier between layers
            lds_cf X = (r & 1) ? S.h2 : S.h1;
            lds_f Y = (r & 1) ? S.h1 : S.h2;
            const int w = wave_id(), q4 = (lane_id() >> 4) * 4;
            for (int m0 = w * (RC / 4); m0 < (w + 1) * (RC / 4); m0 += 16) {
                f32x4 acc[1][8], bias[8];
                acc_zero(acc);
                for (int x = 0; x < 8; ++x) bias[x] = ld4(theta + L.b_off + (x >> 2) * 64 + 4 * q4 + 4 * (x & 3));
                mma_w<1, 8, W_IL, 8>(acc, X, S.hp, m0, theta + L.w_off, H, 0, 128);
                tile_epilogue<1, 8, W_IL>(acc, m0, 0, [&](int rr, int c4, f32x4 v, int slot) {
                    st4(Y + rr * S.hp + c4, act_apply4<ACT_RELU>(v + bias[slot])); });
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (mode == 7) {       // linear_fwd's interleaved path with stamps (wave 0's view, accumulated over the calls)
            lds_cf X = (r & 1) ? S.h2 : S.h1;
            lds_f Y = (r & 1) ? S.h1 : S.h2;
            const int w = wave_id(), mt0 = (w & 1) * 2, g = w >> 1, q4 = (lane_id() >> 4) * 4;
            const long long a0 = clock64();
            f32x4 acc[2][4], bias[4];
            acc_zero(acc);
            for (int x = 0; x < 4; ++x) bias[x] = ld4(theta + L.b_off + g * 64 + 4 * q4 + 4 * x);
            mma_w<2, 4, W_IL, 8>(acc, X, S.hp, mt0 * 16, theta + L.w_off, H, g * 64, 128);
            __builtin_amdgcn_sched_barrier(0);
            const long long a1 = clock64();
            __builtin_amdgcn_sched_barrier(0);
            tile_epilogue<2, 4, W_IL>(acc, mt0 * 16, g * 64, [&](int rr, int c4, f32x4 v, int slot) {
                st4(Y + rr * S.hp + c4, act_apply4<ACT_RELU>(v + bias[slot])); });
            __builtin_amdgcn_sched_barrier(0);
            const long long a2 = clock64();
            __syncthreads();
            const long long a3 = clock64();
            if (threadIdx.x == 0) { st[0] += a1 - a0; st[1] += a2 - a1; st[2] += a3 - a2; }
            continue;
        }
        __syncthreads();
    }
    if (mode == 7 && threadIdx.x == 0) for (int i = 0; i < 3; ++i) cyc[1024 + blockIdx.x * 3 + i] = st[i];
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
    hipMalloc(&cyc, maxwg * 8 * 8); hipMemset(cyc, 0, maxwg * 64);
    std::vector<float> h((size_t)(maxwg / 8) * 32768);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * ((i * 7919) % 13) - 0.06f;
    hipMemcpy(theta, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int lds = (RC * (20 + 2 * 132 + 20 + 8) + 64 + 8) * 4;   // 80 KB at 64 rows: two workgroups per CU; 160 KB at 128: one
    const int reps = 64;
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const char* names[10] = {"fwd 128x128", "dX 128x128", "dW 128x128", "fwd 16->128", "fwd 128->16", "mma 8x(K=128)", "mma K=1024", "fwd stamped", "fwd wave rows", "head bwd 128->1"};
    LayerDesc L1{H, 10, H, 16, 0, 16 * H}, L3{1, H, 16, H, 0, 16 * H};
    for (int mode = 0; mode < 10; ++mode)
        for (int wg : {256, 512}) {
            for (int it = 0; it < 2; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L1);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L3);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L);
                if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(wg), dim3(256), lds, 0, theta, slab, cyc, reps, L3);
                hipDeviceSynchronize();
            }
            std::vector<long long> c(wg);
            hipMemcpy(c.data(), cyc, wg * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : c) s += v;
            printf("%-12s %4d workgroups (%d per CU): %8.0f cycles per call  (MFMA floor %d)\n", names[mode], wg, wg / 256,
                   s / wg / reps, (mode < 3 || mode > 4 ? 128 : 16) * 32 * (wg / 256) * (RC / 32));
#ifdef STAGGER
            { long long ns = 0; hipMemcpy(&ns, cyc + 8000, 8, hipMemcpyDeviceToHost); printf("    staggered workgroups so far: %lld\n", ns); }
#endif
            if (mode == 7) {
                std::vector<long long> d(wg * 3);
                hipMemcpy(d.data(), cyc + 1024, wg * 24, hipMemcpyDeviceToHost);
                double p[3] = {0, 0, 0};
                for (int i = 0; i < wg * 3; ++i) p[i % 3] += d[i];
                printf("    mma %.0f   epilogue %.0f   barrier %.0f\n", p[0] / wg / reps, p[1] / wg / reps, p[2] / wg / reps);
            }
        }
    return 0;
}
