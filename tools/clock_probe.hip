// Developer probe: which clock does a short, nearly idle-GPU kernel run at?  One workgroup runs a dependent FMA chain and
// stamps s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around it; launched back to back and after idle gaps.
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o tools/_bin/clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void probe(int iters, unsigned long long* out, float* sink) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (x == 1234.5f) *sink = x;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1, 8, 256, 2048}) {
        for (int gap_us : {0, 200, 5000}) {
            double mhz = 0, us_kernel = 0, us_wall = 0; const int reps = 200;
            for (int r = 0; r < reps; ++r) {
                if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, 20000, out, sink);
                hipDeviceSynchronize();
                auto t1 = std::chrono::steady_clock::now();
                unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
                mhz += (double)h[0] / ((double)h[1] / 100.0);      // cycles per microsecond = MHz
                us_kernel += (double)h[1] / 100.0;
                us_wall += std::chrono::duration<double, std::micro>(t1 - t0).count();
            }
            printf("grid %4d  idle gap %5d us: s_memtime runs at %7.1f MHz of wall time; 20 k dependent FMAs take %6.1f us inside the kernel, launch + sync %6.1f us on the host\n",
                   grid, gap_us, mhz / reps, us_kernel / reps, us_wall / reps);
        }
    }
    return 0;
}
