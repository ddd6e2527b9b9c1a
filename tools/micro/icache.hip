// Micro-benchmark: what does code that runs ONCE per workgroup cost on gfx950?  Straight-line VALU code (4-byte v_add_f32,
// dependent chain: 4+ cycles each when hot) of N KB against the same instruction count in a rolled loop.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/icache.hip -o /tmp/icache && /tmp/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R1024(x) R4(R256(x))
#define ADD asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MF asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void straight(float* out, long long* clk, int mode) {
    float a = threadIdx.x, b = 1.0f;
    f32x4 acc = {0, 0, 0, 0};
    long long t0 = clock64();
    if (mode == 0) { R1024(R4(ADD)) R1024(R4(ADD)) }            // 8192 v_add = 32 KB, once
    else if (mode == 1) { for (int i = 0; i < 32; ++i) { R256(ADD) asm volatile("" ::: "memory"); } }     // 8192 v_add, 1 KB body
    else if (mode == 2) { R1024(MF) R1024(MF) }                  // 2048 MFMA = 16 KB, once (64 K cycles of matrix work)
    else { for (int i = 0; i < 32; ++i) { R16(R4(MF)) asm volatile("" ::: "memory"); } }                  // 2048 MFMA, 512 B body
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[mode] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + acc[0];
}
int main() {
    float* out; long long* clk;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 64);
    long long h[4];
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(straight, dim3(256), dim3(256), 0, 0, out, clk, mode);
            hipDeviceSynchronize();
        }
    hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost);
    printf("8192 v_add  straight (32 KB): %lld cycles   rolled (1 KB body): %lld cycles\n", h[0], h[1]);
    printf("2048 v_mfma straight (16 KB): %lld cycles   rolled (512 B body): %lld cycles\n", h[2], h[3]);
    return 0;
}
