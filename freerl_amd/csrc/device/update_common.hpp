// Helpers shared by the gradient kernels of the off-policy learners (kernels_dqn / _critic / _actor / _c51 .hip):
// LDS carve from the engine descriptor, block id -> (unit, slice) on one XCD, the row chunks a workgroup walks.
#pragma once
#include "net.hpp"

namespace frl {

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;
constexpr float kLog2 = 0.69314718055994530942f;

__device__ __forceinline__ float softplus_t(float x) {     // F.softplus (beta 1, threshold 20)
    return x > 20.f ? x : log1pf(expf(x));
}

// TD loss of one row's error e: value and d/de (before the 1/B of the mean).  MSE: e^2, 2e.  Huber (MAPPO.py:273-276).
__device__ __forceinline__ void td_loss_row(const LearnArgs& a, float e, float& loss, float& grad) {
    if (a.huber) {
        const float d = a.huber_delta, ae = fabsf(e);
        if (ae <= d) { loss = e * e * 0.5f; grad = e; }
        else { loss = d * (ae - d * 0.5f); grad = e > 0.f ? d : -d; }
    } else {
        loss = e * e; grad = 2.f * e;
    }
}

__device__ __forceinline__ Lds carve(const EngineDesc& D, float* smem) {
    return carve_lds(smem, D.rc, D.hidden, D.lds_kin_pad, D.lds_out_pad, D.lds_batch_pad, D.lds_act_pad, D.lds_hbufs);
}

// block id -> (unit, slice): the `ns` row chunks of unit u = learner*n_agents + agent sit on
// blocks {8*ns*g + x + 8s}, i.e. all on XCD x and adjacent in dispatch order.
struct UnitSlice { int unit, slice; };
__device__ __forceinline__ UnitSlice unit_slice(int ns) {
    const int group = 8 * ns, g = blockIdx.x / group, l = blockIdx.x - g * group;
    return UnitSlice{g * 8 + (l & 7), l >> 3};
}

// A workgroup owns `cps` consecutive row chunks of its unit (frl_create picks cps so that one round of workgroups fills
// the chip): chunk 0 stores its weight gradients in the workgroup's slab, the others add to them — one slab per
// workgroup instead of one per chunk for reduce_kernel to stream.
struct ChunkRange { int c0, c1; };
__device__ __forceinline__ ChunkRange chunk_range(const EngineDesc& D, int batch, int slab) {
    const int nchunks = (batch + D.rc - 1) / D.rc, c0 = slab * D.cps;
    return ChunkRange{c0, min(c0 + D.cps, nchunks)};
}

}  // namespace frl
