// Cross-lane exchanges of a 64-wide wave without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip
// of ~130 cycles per step, six dependent steps per wave reduction — the reductions between the MFMA blocks of the latency-bound kernels
// (PPO's per-minibatch norms, the C51 softmaxes, the per-row argmax of DQN) were mostly that wait.  lane_xor<OFF>(v) returns lane
// (l ^ OFF)'s v through DPP (offsets 1 .. 8: quad_perm / row shifts / row_ror inside a row of 16 lanes) or gfx950's
// v_permlane16_swap / v_permlane32_swap (offsets 16 / 32): VALU instructions, a few cycles each, the SAME values exchanged — every
// reduction built on it is bit-identical to its __shfl_xor form (tools/lane_xor_test.hip checks the six offsets against __shfl_xor).
// The lane-16 / lane-32 halves are told apart by bits 4 / 5 of threadIdx.x: every kernel of this library is launched with a 1-D block
// whose size is a multiple of 64, where those ARE the lane number's bits (and threadIdx.x is a register every kernel has anyway; the
// v_mbcnt pair of __lane_id() would be one more live VGPR in kernels that sit at their register limit).  gfx950 only: the offsets 16 / 32
// use its v_permlane16_swap / v_permlane32_swap.
#pragma once
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "device/lane.hpp is written for gfx950 (v_permlane16_swap / v_permlane32_swap)"
#endif

namespace frl {

template <int OFF>
__device__ __forceinline__ int lane_xor_i(int v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "one bit of the lane number");
    if constexpr (OFF == 1) {
        return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);                       // quad_perm [1,0,3,2]
    } else if constexpr (OFF == 2) {
        return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);                       // quad_perm [2,3,0,1]
    } else if constexpr (OFF == 4) {
        // lanes with bit 2 clear (banks 0, 2 of a row) read lane + 4, the others lane - 4
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);       // row_shl:4 into banks 0, 2
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);              // row_shr:4 into banks 1, 3
    } else if constexpr (OFF == 8) {
        return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);                      // row_ror:8
    } else if constexpr (OFF == 16) {
        // v_permlane16_swap vdst, src: vdst's odd rows <-> src's even rows.  Both = v: r[0] = {row0, row0, row2, row2}, r[1] = {row1, row1, row3, row3}
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((threadIdx.x & 16) ? r[0] : r[1]);
    } else {
        // v_permlane32_swap vdst, src: vdst's upper half <-> src's lower half.  Both = v: r[0] = {lo, lo}, r[1] = {hi, hi}
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((threadIdx.x & 32) ? r[0] : r[1]);
    }
}
template <int OFF>
__device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor_i<OFF>(__float_as_int(v))); }
template <int OFF>
__device__ __forceinline__ int lane_xor(int v) { return lane_xor_i<OFF>(v); }
template <int OFF>
__device__ __forceinline__ unsigned lane_xor(unsigned v) { return (unsigned)lane_xor_i<OFF>((int)v); }

// runtime-looking offset of an unrolled loop (folds to one case)
__device__ __forceinline__ float lane_xor(float v, int off) {
    switch (off) {
        case 1: return lane_xor<1>(v);
        case 2: return lane_xor<2>(v);
        case 4: return lane_xor<4>(v);
        case 8: return lane_xor<8>(v);
        case 16: return lane_xor<16>(v);
        default: return lane_xor<32>(v);
    }
}
__device__ __forceinline__ int lane_xor(int v, int off) { return __float_as_int(lane_xor(__int_as_float(v), off)); }

}  // namespace frl
