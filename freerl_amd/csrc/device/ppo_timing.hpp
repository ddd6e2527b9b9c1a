#pragma once
#include <hip/hip_runtime.h>
namespace frl {
// Developer instrument (tools/ppo_timing.py; -DFRL_PPO_TIMING, unity build): thread 0 of learner 0's two workgroups adds up
// the shader clock per section of a minibatch step.
#ifdef FRL_PPO_TIMING
__device__ long long g_ppo_clk[2][8];
#define PPO_T0() long long t_prev_ = clock64(); long long t_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PPO_T(slot) do { const long long t_now_ = clock64(); t_acc_[slot] += t_now_ - t_prev_; t_prev_ = t_now_; } while (0)
#define PPO_TDUMP() do { if (threadIdx.x == 0 && blockIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_ppo_clk[blockIdx.y][i_] = t_acc_[i_]; } while (0)
// a second, finer set of sections inside one of the above (kernels_critic2.hip's target passes): row 1 of the same array
#define PPO_U0() long long u_prev_ = clock64(); long long u_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PPO_UR() do { u_prev_ = clock64(); } while (0)
#define PPO_U(slot) do { const long long u_now_ = clock64(); u_acc_[slot] += u_now_ - u_prev_; u_prev_ = u_now_; } while (0)
#define PPO_UDUMP() do { if (threadIdx.x == 0 && blockIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_ppo_clk[1][i_] = u_acc_[i_]; } while (0)
// a helper function that stamps sections too takes / is handed the caller's clock state
#define PPO_TPARAMS , long long& t_prev_, long long (&t_acc_)[8]
#define PPO_TARGS , t_prev_, t_acc_
#define PPO_UPARAMS , long long& u_prev_, long long (&u_acc_)[8]
#define PPO_UARGS , u_prev_, u_acc_
// the quick form of the developer build (tools/build_unit_timing.sh <unit>): ONE kernels_*.hip compiled with -DFRL_PPO_TIMING
// -DFRL_CLK_COPY=1 carries the stamps and this kernel, which the host unit (compiled with -DFRL_PPO_TIMING_SPLIT) launches to fetch them
#if defined(FRL_CLK_COPY) && !defined(FRL_UNITY)
__global__ void frl_clk_copy_kernel(long long* out) {
    if (threadIdx.x < 16) out[threadIdx.x] = (&g_ppo_clk[0][0])[threadIdx.x];
}
#endif
#else
#define PPO_TPARAMS
#define PPO_TARGS
#define PPO_UPARAMS
#define PPO_UARGS
#define PPO_T0() do {} while (0)
#define PPO_T(slot) do {} while (0)
#define PPO_TDUMP() do {} while (0)
#define PPO_U0() do {} while (0)
#define PPO_UR() do {} while (0)
#define PPO_U(slot) do {} while (0)
#define PPO_UDUMP() do {} while (0)
#endif

}  // namespace frl
