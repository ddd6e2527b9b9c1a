// lane_xor<OFF> (device/lane.hpp: DPP / v_permlane*_swap) against __shfl_xor on every lane, and the cost of a six-step wave sum each way.
//   hipcc --offload-arch=gfx950 -O3 -I freerl_amd/csrc tools/lane_xor_test.hip -o tools/_bin/lane_xor_test && tools/_bin/lane_xor_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "device/lane.hpp"
using namespace frl;

__global__ void check_kernel(const float* in, float* out_new, float* out_old) {
    const int t = threadIdx.x;
    const float v = in[t];
    out_new[0 * 256 + t] = lane_xor<1>(v);  out_old[0 * 256 + t] = __shfl_xor(v, 1, 64);
    out_new[1 * 256 + t] = lane_xor<2>(v);  out_old[1 * 256 + t] = __shfl_xor(v, 2, 64);
    out_new[2 * 256 + t] = lane_xor<4>(v);  out_old[2 * 256 + t] = __shfl_xor(v, 4, 64);
    out_new[3 * 256 + t] = lane_xor<8>(v);  out_old[3 * 256 + t] = __shfl_xor(v, 8, 64);
    out_new[4 * 256 + t] = lane_xor<16>(v); out_old[4 * 256 + t] = __shfl_xor(v, 16, 64);
    out_new[5 * 256 + t] = lane_xor<32>(v); out_old[5 * 256 + t] = __shfl_xor(v, 32, 64);
}
template <bool NEW>
__global__ void time_kernel(const float* in, float* out, long long* cycles, int reps) {
    float v = in[threadIdx.x];
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (NEW) {
            v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
        } else {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        }
        v = v * 0.015625f + 1.f;
    }
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    float *in, *a, *b; long long* cyc;
    hipMalloc(&in, 256 * 4); hipMalloc(&a, 6 * 256 * 4); hipMalloc(&b, 6 * 256 * 4); hipMalloc(&cyc, 8);
    std::vector<float> h(256), ha(6 * 256), hb(6 * 256);
    for (int i = 0; i < 256; ++i) h[i] = 1.f + 0.37f * i + 1e-3f * i * i;
    hipMemcpy(in, h.data(), 256 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(256), 0, 0, in, a, b);
    hipMemcpy(ha.data(), a, 6 * 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, 6 * 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 6; ++k) {
        int nb = 0;
        for (int t = 0; t < 256; ++t) nb += (ha[k * 256 + t] != hb[k * 256 + t]);
        printf("lane_xor<%d>: %d of 256 lanes differ from __shfl_xor\n", 1 << k, nb);
        bad += nb;
    }
    const int reps = 1000;
    for (int which = 0; which < 2; ++which) {
        long long c = 0;
        for (int it = 0; it < 2; ++it) {
            if (which) hipLaunchKernelGGL(time_kernel<true>, dim3(1), dim3(64), 0, 0, in, a, cyc, reps);
            else hipLaunchKernelGGL(time_kernel<false>, dim3(1), dim3(64), 0, 0, in, a, cyc, reps);
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        }
        float r0; hipMemcpy(&r0, a, 4, hipMemcpyDeviceToHost);
        printf("%s wave sum: %.1f shader cycles each (result %.6f)\n", which ? "DPP / permlane " : "ds_bpermute    ", (double)c / reps, r0);
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad != 0;
}
