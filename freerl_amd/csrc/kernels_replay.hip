// Replay-ring kernels: scatter staged transitions into the GPU-resident ring(s) (Buffer.add),
// gather sampled rows into dense per-field tensors (Buffer.sample), synthetic fill.
//
// Layout: ring[p][row][stride] fp32, one transition record per 128-byte-aligned row (AoS), so a
// sampled transition is ONE contiguous line-aligned read instead of five scattered ones in
// the reference's five arrays (TD3_file/Buffer.py:17-21, :42-46).  HBM-bound byte work.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/rng.hpp"
#include "frl_desc.h"

namespace frl {

// staged[n][width] (device copy of the pinned staging area) -> ring rows given by slots[n]
// slots[i] = learner * capacity + row
__global__ void replay_scatter_kernel(float* __restrict__ ring, const float* __restrict__ staged,
                                      const long long* __restrict__ slots, int n, int width, int stride) {
    const int per_row = (width + 3) / 4;
    const long long total = (long long)n * per_row;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / per_row), c4 = (int)(e - (long long)i * per_row) * 4;
        float* dst = ring + slots[i] * stride + c4;
        const float* src = staged + (size_t)i * width + c4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c4 + k < width) dst[k] = src[k];
    }
}


// One launch produces every field of Buffer.sample(indices): a 16-lane group owns one sampled
// row and streams its record (coalesced 64-byte segments), writing each field's dense tensor.
__global__ void replay_gather_kernel(const float* __restrict__ ring, const long long* __restrict__ idx, int B,
                                     int stride, GatherFields F) {
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, lane = threadIdx.x & 15;
    if (g >= B) return;
    const float* rec = ring + (size_t)idx[g] * stride;
    for (int f = 0; f < F.n_fields; ++f) {
        const int nc = F.ncols[f];
        float* out = F.out[f] + (size_t)g * nc;
        for (int c = lane; c < nc; c += 16) out[c] = rec[F.col0[f] + c];
    }
}

// add(obs, action, reward, next_obs, done[, log_probs, adv_done]) of one vector step for every env (Buffer.add,
// TD3_file/Buffer.py:28-38; Buffer_for_PPO.add, PPO_file/Buffer.py:292-305): a 16-lane group owns env i = learner * E + j.
__global__ void replay_commit_kernel(float* __restrict__ ring, RecordDesc rec, int capacity, CommitArgs c) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, lane = threadIdx.x & 15;
    if (i >= c.n) return;
    float* r = ring + ((size_t)(i / c.E) * capacity + c.row[i]) * rec.stride;
    float* oc = c.obs_cur + (size_t)i * c.O;
    const unsigned char fl = c.flags[i];
    for (int k = lane; k < c.O; k += 16) {
        r[rec.obs_off[0] + k] = oc[k];
        r[rec.nobs_off[0] + k] = c.next_obs[(size_t)i * c.O + k];
        oc[k] = c.obs_next[(size_t)i * c.O + k];           // same lane read it above
    }
    for (int k = lane; k < c.aout; k += 16) r[rec.act_off[0] + k] = c.store_act[(size_t)i * c.aout + k];
    for (int k = lane; k < c.n_logp; k += 16) r[rec.extra_off + k] = c.logp[(size_t)i * c.n_logp + k];
    if (lane == 0) {
        r[rec.rew_off] = c.reward[i];
        r[rec.done_off] = (fl & 1) ? 1.f : 0.f;
        if (c.logp) r[rec.extra_off + c.n_logp] = (fl & 4) ? 1.f : 0.f;
    }
}

// out[row][width] = ring[slot0 + row][0..width)   (dense read-back of whole records)
__global__ void replay_read_kernel(const float* __restrict__ ring, long long row0, int n, int width, int stride,
                                   float* __restrict__ out) {
    const long long total = (long long)n * width;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / width;
        const int c = (int)(e - r * width);
        out[e] = ring[(row0 + r) * stride + c];
    }
}

// Synthetic transitions for benches: obs, next_obs ~ N(0,1); act ~ U(-1,1) (or an
// integer in [0,n_discrete)); reward ~ N(0,1); done ~ Bernoulli(0.05).
__global__ void replay_fill_kernel(float* __restrict__ ring, long long rows, RecordDesc rec, int n_discrete,
                                   unsigned long long seed) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < rows;
         r += (long long)gridDim.x * blockDim.x) {
        float* p = ring + r * rec.stride;
        for (int c = 0; c < rec.width; c += 2) {
            const Philox4 x = philox4x32_10((unsigned long long)r, (unsigned)c, 0x5eedu, seed);
            float n0, n1;
            normal2(x, n0, n1);
            const float u0 = u01(x.x) * 2.f - 1.f, u1 = u01(x.y) * 2.f - 1.f;
            for (int k = 0; k < 2 && c + k < rec.width; ++k) {
                const int col = c + k;
                const float nn = k ? n1 : n0, uu = k ? u1 : u0;
                float v = nn;
                if (col >= rec.act_off[0] && col < rec.act_off[0] + rec.act_total)
                    v = n_discrete > 0 ? floorf((uu * 0.5f + 0.5f) * n_discrete * 0.999999f) : uu;
                else if (col >= rec.done_off && col < rec.done_off + rec.n_agents)
                    v = (uu * 0.5f + 0.5f) < 0.05f ? 1.f : 0.f;
                p[col] = v;
            }
        }
    }
}

}  // namespace frl
