// Prioritised replay (DQN_file/Buffer.py:66-194): the reference keeps a float64 array-heap SumTree on the host, adds
// priorities one leaf at a time with an incremental walk to the root (:155-164), finds np.max over all leaves on EVERY add
// (:193-194, O(capacity)) and samples with a Python loop of tree descents (:107-114).  Here a sum-tree and a max-tree of
// the same shape live in HBM per learner; a batch of leaf writes is one launch (leaves, then their ancestors recomputed
// level by level from their children — exact child sums, no accumulated increments), a batch of stratified descents is
// one launch.  HBM-bound pointer chasing: nothing here is matrix work.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "device/net.hpp"

namespace frl {


__device__ __forceinline__ int node_depth(int i) { return 31 - __clz(i + 1); }

// Recompute the ancestors of `n` written leaves of one learner, deepest level first; leaf_of(i) = buffer index or -1.
// Every thread repairs the ancestor of its leaf at the current depth (duplicates write the same value).  (Measured and not
// taken: the top 11 levels repaired in LDS — the kernel's time is in the bottom levels' random 8-byte accesses into 1.6 MB of
// tree per learner, not in the number of barriers.)
template <class LeafOf>
__device__ __forceinline__ void per_repair(double* sum, double* mx, int cap, int n, LeafOf leaf_of) {
    const int nn = 2 * cap - 1, dmax = node_depth(nn - 1);
    for (int d = dmax - 1; d >= 0; --d) {
        for (int i = threadIdx.x; i < n; i += kWG) {
            const int li = leaf_of(i);
            if (li < 0) continue;
            const int node0 = li + cap - 1, d0 = node_depth(node0);
            if (d0 > d) {
                const int node = ((node0 + 1) >> (d0 - d)) - 1, l = 2 * node + 1, r = l + 1;
                sum[node] = sum[l] + (r < nn ? sum[r] : 0.0);
                mx[node] = fmax(mx[l], r < nn ? mx[r] : 0.0);
            }
        }
        __syncthreads();
    }
}

// PER_Buffer.add (:92-98) for the rows of one flush: every new row gets the max priority (1.0 on an empty buffer).
// The host buckets the flush's rows by learner (they are staged in order, a counting sort): bucket = [off[P + 1] |
// size_before[P] | leaf[n]], workgroup p repairs the ancestors of ITS leaf[off[p] .. off[p + 1]) only.  (Round 2 handed
// every workgroup the whole flush as learner*capacity + row words: P x n slot tests with a 64-bit division each, per tree
// level — 199 us for a 4096-row add at 512 learners.)
__global__ __launch_bounds__(256) void per_add_kernel(PerArgs a, const int* __restrict__ bucket, int P) {
    const int p = blockIdx.x, cap = a.cap, nn = 2 * cap - 1;
    const int o0 = bucket[p], n = bucket[p + 1] - o0;
    if (n <= 0) return;
    double* sum = a.sum_tree + (size_t)p * nn;
    double* mx = a.max_tree + (size_t)p * nn;
    const int* leaf = bucket + 2 * P + 1 + o0;
    const double fill = (bucket[P + 1 + p] == 0) ? 1.0 : mx[0];
    __syncthreads();                                   // everyone has read mx[0]
    for (int i = threadIdx.x; i < n; i += kWG) {
        const int li = leaf[i];
        sum[li + cap - 1] = fill; mx[li + cap - 1] = fill;
    }
    __syncthreads();
    per_repair(sum, mx, cap, n, [&](int i) { return leaf[i]; });
}

// Write n leaves of one learner and repair both trees.  One workgroup per learner; n <= kPerSetMax (frl_per_update checks).
__global__ __launch_bounds__(256) void per_set_kernel(const EngineDesc* __restrict__ Dp, PerArgs a) {
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x, cap = a.cap, nn = 2 * cap - 1;
    double* sum = a.sum_tree + (size_t)p * nn;
    double* mx = a.max_tree + (size_t)p * nn;
    const int* leaf = a.leaf + (size_t)p * a.n_pitch;
    const double fill = a.fill;
    // leaves: a later entry for the same leaf wins, as in the reference's sequential loop (:126-129).  The batch's leaf
    // indices go through LDS for that test (n <= 4096): as a loop over global memory the later-entry scan of thread 0 alone
    // was 255 dependent-latency loads, most of the kernel.
    __shared__ __attribute__((aligned(16))) int lleaf[kPerSetMax];
    for (int i = threadIdx.x; i < a.n; i += kWG) lleaf[i] = leaf[i];
    for (int i = a.n + threadIdx.x; i < ((a.n + 3) & ~3); i += kWG) lleaf[i] = -1;
    __syncthreads();
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int i = threadIdx.x; i < a.n; i += kWG) {
        const int li = lleaf[i];
        bool last = true;
        int j = i + 1;
        for (; j < a.n && (j & 3); ++j) last &= (lleaf[j] != li);
        for (; j < a.n; j += 4) {
            const i32x4 v = *(const i32x4*)(lleaf + j);
            last &= (v.x != li) & (v.y != li) & (v.z != li) & (v.w != li);
        }
        if (!last) continue;
        double v = fill;
        if (a.td) v = (double)powf(fabsf(a.td[(size_t)p * D.batch_max + i]) + a.eps, a.alpha);
        else if (a.prio) v = (double)a.prio[(size_t)p * a.n_pitch + i];
        sum[li + cap - 1] = v;
        mx[li + cap - 1] = v;
    }
    __syncthreads();
    per_repair(sum, mx, cap, a.n, [&](int i) { return leaf[i]; });
}

// PER_Buffer.sample (:99-124): stratified segments of the total priority mass, one descent per segment, float32
// priorities, P(i) = p / sum clipped at 1e-7, w = (N * P(i))^-beta / max w.  One workgroup per learner, batch <= 1024.
__global__ __launch_bounds__(256) void per_sample_kernel(const EngineDesc* __restrict__ Dp, PerArgs a) {
    __shared__ float red_s[8];
    lds_f red = (lds_f)red_s;
    const EngineDesc& D = *Dp;
    const int p = blockIdx.x, cap = a.cap, nn = 2 * cap - 1, B = a.n;
    const double* sum = a.sum_tree + (size_t)p * nn;
    int* idx = D.idx + (size_t)p * D.n_agents * D.batch_max;
    const double total = sum[0], segment = total / (double)B;
    const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
    float wmax = 0.f;
    for (int i = threadIdx.x; i < B; i += kWG) {
        double u;
        if (a.uniforms) u = a.uniforms[(size_t)p * B + i];
        else {
            const Philox4 r = philox4x32_10(a.rng_counter, 0x7000u, (unsigned)i, key);
            u = ((double)(((unsigned long long)(r.x >> 5) << 26) | (r.y >> 6))) * (1.0 / 9007199254740992.0);   // 53-bit, like NumPy
        }
        const double lo = segment * (double)i, hi = segment * (double)(i + 1);
        double s = lo + (hi - lo) * u;                        // np.random.uniform(a, b)
        int node = 0;
        while (true) {                                        // SumTree.get (:166-185)
            const int l = 2 * node + 1;
            if (l >= nn) break;
            const double sl = sum[l];
            if (s <= sl) node = l;
            else { s -= sl; node = l + 1; }
        }
        const float pr = (float)sum[node];
        idx[i] = node - cap + 1;
        a.prio_out[(size_t)p * D.batch_max + i] = pr;
        const double prob = fmax((double)pr / total, 1e-7);
        const float w = (float)pow((double)a.size[p] * prob, -a.beta);
        a.isw[(size_t)p * D.batch_max + i] = w;
        wmax = fmaxf(wmax, w);
    }
    // block max through LDS
    for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, lane_xor(wmax, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int i = threadIdx.x; i < B; i += kWG) a.isw[(size_t)p * D.batch_max + i] /= m;
}

}  // namespace frl
