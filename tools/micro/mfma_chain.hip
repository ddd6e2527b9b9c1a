// Micro-benchmark: at what rate does ONE wave per SIMD issue v_mfma_f32_16x16x4_f32 on gfx950, by the number of independent
// accumulator chains it interleaves?  (The sixteen-workgroup kernels — kernels_solo.hip, kernels_solow.hip — run four waves per
// CU, one per SIMD, with two chains per layer sweep: their sweeps measured ~74 cycles per MFMA against the 32 the peak implies.)
// One workgroup of WAVES x 64 threads, N iterations of CHAINS MFMAs (chain c accumulates into its own registers).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o tools/_bin/mfma_chain && tools/_bin/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(512) void chain(float* out, long long* clk, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x * 1e-3f, b = 1.f + (float)(threadIdx.x & 3);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int CHAINS>
void run(int waves, int grid) {
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * 512 * grid); hipMalloc(&clk, sizeof(long long) * grid);
    const int iters = 2000;
    chain<CHAINS><<<grid, waves * 64>>>(out, clk, iters);
    chain<CHAINS><<<grid, waves * 64>>>(out, clk, iters);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, clk, sizeof h, hipMemcpyDeviceToHost);
    printf("waves/CU %d (per SIMD %.1f)  chains %d: %6.1f shader-clock ticks per MFMA of one wave (s_memtime)\n", waves, waves / 4.0, CHAINS, (double)h / (iters * 8.0 * CHAINS));
    hipFree(out); hipFree(clk);
}
int main() {
    for (int waves : {4, 8}) { run<1>(waves, 1); run<2>(waves, 1); run<4>(waves, 1); run<8>(waves, 1); }
    run<2>(4, 256); run<2>(8, 256);
    return 0;
}
