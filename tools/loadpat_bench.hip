// Developer microbenchmark: cost of one wave-wide weight-fragment load under different lane->address
// maps, weights L2 resident (64 KB per workgroup group, re-read REPS times), 4 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/loadpat_bench tools/loadpat_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLB __attribute__((address_space(1)))

template <int PAT>
__global__ __launch_bounds__(256, 4) void k(const float* __restrict__ Wp, float* out, int reps, int nbuf) {
    const GLB float* W = (const GLB float*)(Wp) + (size_t)((blockIdx.x / 8) % nbuf) * 16384;   // 64 KB per 8 blocks
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, i = l & 15, q = l >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        int opaque;
        asm volatile("s_mov_b32 %0, 0" : "=s"(opaque));     // keeps the loads inside the loop
        W += opaque;
        // one "layer": each wave fetches its 32 columns x 128 k = 16 KB as 16 float4 (or 64 dword) per lane
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int n0 = w * 32 + y * 16;
                if (PAT == 0) {            // today's forward: row n0+i (512 B pitch), 16 B chunk q of k-block j
                    acc += *(const GLB f32x4*)(W + (n0 + i) * 128 + 16 * j + 4 * q);
                } else if (PAT == 1) {     // fragment-tiled: 1 KB block per (n-tile, k-block), lane l reads its 16 B
                    acc += *(const GLB f32x4*)(W + ((n0 >> 4) * 8 + j) * 256 + l * 4);
                } else if (PAT == 2) {     // today's dX: 4 scalar loads, rows 16j+4q+e, column n0+i
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += W[(16 * j + 4 * q + e) * 128 + n0 + i];
                } else if (PAT == 3) {     // dX out of the fragment-tiled layout: scalar, 16 pieces of 16 B per instruction
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += W[(j * 8 + (n0 >> 4)) * 256 + ((i >> 2) * 16 + 4 * q + e) * 4 + (i & 3)];
                } else if (PAT == 4) {     // k-major + 4-tile column interleave: 4 rows x 256 B contiguous
                    acc += *(const GLB f32x4*)(W + (16 * j + 4 * q + y * 2) * 128 + (w & 1) * 64 + 4 * i);
                }
            }
        asm volatile("" ::: "memory");
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int PAT>
void run(const float* W, float* out, int reps, int nbuf, const char* name) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<PAT><<<4096, 256>>>(W, out, reps, nbuf);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<PAT><<<4096, 256>>>(W, out, reps, nbuf);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // 4096 blocks x 4 waves x reps x 16 KB
    const double bytes = 4096.0 * 4 * reps * 16384;
    const double per_cu_cycles = ms * 1e-3 * 2.4e9;
    const double wave_instr_per_cu = 4096.0 * 4 * reps * (PAT == 2 || PAT == 3 ? 64 : 16) / 256;
    printf("%-34s %8.3f ms  %7.1f GB/s L2->CU aggregate  %6.1f cycles/wave-instr/CU (16 KB per wave = %s)\n", name, ms,
           bytes / ms * 1e-6, per_cu_cycles / wave_instr_per_cu, (PAT == 2 || PAT == 3) ? "64 dword" : "16 dwordx4");
}

int main() {
    const int nbuf = 64;
    float *W, *out;
    hipMalloc(&W, (size_t)nbuf * 65536);
    hipMalloc(&out, 4096);
    hipMemset(W, 0, (size_t)nbuf * 65536);
    const int reps = 50;
    run<0>(W, out, reps, nbuf, "P0 fwd today (16 rows x 16 B)");
    run<1>(W, out, reps, nbuf, "P1 fragment-tiled (1 KB contiguous)");
    run<2>(W, out, reps, nbuf, "P2 dX today (scalar, 4 x 64 B)");
    run<3>(W, out, reps, nbuf, "P3 dX from fragment tiles (scalar)");
    run<4>(W, out, reps, nbuf, "P4 k-major interleaved (4 x 256 B)");
    return 0;
}
