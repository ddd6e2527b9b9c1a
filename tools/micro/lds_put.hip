// Micro-benchmark: what does one exchange "put" phase of the eight-wave chained kernels cost on gfx950?
// 512 threads (8 waves, 2 per SIMD), one workgroup per CU (160 KB of LDS claimed), N iterations of
//   barrier -> every wave writes T 16x16 tiles -> barrier [-> every wave reads R fragments]
// in four forms: transposed ds_write_b32 x 4 per tile (what put_tile does), ds_write_b64 x 2, ds_write_b128 x 1, nothing.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_put.hip -o tools/_bin/lds_put && tools/_bin/lds_put
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define LDS __attribute__((address_space(3)))
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int MODE, int T>
__global__ __launch_bounds__(512) void put(float* out, long long* clk, int iters, int active_waves) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    LDS float* E = (LDS float*)smem;
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), i16 = l & 15, q = l >> 4;
    const int tslot = (((i16 >> 2) * 16) << 2) + (i16 & 3);
    const int fslot = (q * 16 + (i16 ^ q)) << 2;
    f32x4 v[T];
    for (int t = 0; t < T; ++t) v[t] = f32x4{(float)tid, (float)t, 1.f, 2.f};
    f32x4 acc = {0, 0, 0, 0};
    bar();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        bar();
        if (w < active_waves) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                LDS float* tile = E + ((t % 8) * 8 + w) * 256;
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile[tslot + (((4 * q + r) ^ (i16 >> 2)) << 2)] = v[t][r];
                } else if (MODE == 1) {
                    *(LDS f32x2*)(tile + fslot) = f32x2{v[t][0], v[t][1]};
                    *(LDS f32x2*)(tile + fslot + 2) = f32x2{v[t][2], v[t][3]};
                } else if (MODE == 2) {
                    *(LDS f32x4*)(tile + fslot) = v[t];
                }
            }
        }
        bar();
        // a token consumer: one fragment read per tile row (keeps the writes observable)
        acc += *(LDS f32x4*)(E + (w * 8 + (it & 7)) * 256 + fslot);
#pragma unroll
        for (int t = 0; t < T; ++t) v[t][0] += acc[0] * 1e-30f;
    }
    const long long t1 = clock64();
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE, int T>
static void run(const char* name, float* out, long long* clk, int waves) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)put<MODE, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((put<MODE, T>), dim3(256), dim3(512), 160 * 1024, 0, out, clk, iters, waves);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-28s T=%2d tiles/wave, %d writing waves: %7.1f cycles per phase\n", name, T, waves, (double)h / iters);
}
int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 64);
    run<3, 9>("no writes (2 barriers + read)", out, clk, 8);
    run<0, 9>("ds_write_b32 x 4 (transposed)", out, clk, 8);
    run<1, 9>("ds_write_b64 x 2", out, clk, 8);
    run<2, 9>("ds_write_b128 x 1", out, clk, 8);
    run<0, 16>("ds_write_b32 x 4 (transposed)", out, clk, 4);
    run<2, 16>("ds_write_b128 x 1", out, clk, 4);
    run<0, 16>("ds_write_b32 x 4 (transposed)", out, clk, 8);
    run<0, 4>("ds_write_b32 x 4 (transposed)", out, clk, 8);
    return 0;
}
