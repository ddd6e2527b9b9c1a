#pragma once
#include <hip/hip_runtime.h>
namespace frl {
// Developer instrument (tools/wide_timing.py; -DFRL_WIDE_TIMING, unity build): thread 0 of workgroup 0 adds up the shader clock
// per section of kernels_criticw.hip (row 0) / kernels_actorw.hip (row 1).
#ifdef FRL_WIDE_TIMING
__device__ long long g_wide_clk[2][16];
#define WIDE_T0() long long wt_prev_ = clock64(); long long wt_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define WIDE_T(slot) do { const long long wt_now_ = clock64(); wt_acc_[slot] += wt_now_ - wt_prev_; wt_prev_ = wt_now_; } while (0)
#define WIDE_TDUMP(row) do { if (threadIdx.x == 0 && blockIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) g_wide_clk[row][i_] = wt_acc_[i_]; } while (0)
#else
#define WIDE_T0() do {} while (0)
#define WIDE_T(slot) do {} while (0)
#define WIDE_TDUMP(row) do {} while (0)
#endif
}  // namespace frl
