// One 3-layer, hidden-128, narrow-input (<= 16 columns), narrow-head (<= 16 outputs) ReLU MLP on chip, for the kernels that
// give a whole learner to one workgroup (kernels_critic2.hip, kernels_actor2.hip): LDS images of the net in MFMA-fragment order
// (device/chain.hpp), the chained forward of 16 (or 2 x 16) rows per wave, the backward with its three activation / delta
// exchanges and the owners' weight-gradient accumulators, and clip + Adam (+ soft update) straight from those accumulators.
//
// Parameter layout in HBM (NetDesc::frag, round 3): the engines these kernels serve keep theta / target / m / v of their nets in
// the SAME fragment order as the LDS images — weight block of a layer = its 16 x 16 tiles, tile (ot, kb) at (ot * KB + kb) * 256
// floats, element (out & 15 = f, in & 15 = 4q + e) at ((q * 16 + (f ^ q)) << 2) + e (frag_dw) — instead of Wk[in][out].  Two
// things follow.  Staging a net is a linear 16-byte copy (round 2 gathered 4-byte words from four Wk rows per LDS slot: 80
// loads per thread and net, 15 k cycles x 5 nets of a critic update).  And the weight gradients are accumulated TRANSPOSED
// (dW^T[in][out] = H dZ^T: the A and B operands of round 2's dW MFMAs swapped, same fragments), so the four registers of a
// lane's accumulator tile ARE one 16-byte slot of the image: clip + Adam + soft update run register -> global with every
// wave-instruction covering one whole contiguous 1 KB tile — no transposes through LDS, no barriers, nothing that ties the
// update to the workgroup's other waves (round 2: 17 % of the critic stage, every CU in it at the same time).
#pragma once
#include "chain.hpp"
#include "ppo_timing.hpp"

// Developer instrument (-DFRL_PPO_TIMING -DFRL_BWD_TIMING, one unit: tools/build_unit_timing.sh): thread 0 of workgroup 0 adds the
// shader clock of every section of the eight-wave backward to row 1 of g_ppo_clk (summed over chunks, heads AND launches)
#if defined(FRL_PPO_TIMING) && defined(FRL_BWD_TIMING)
#define BWD_T0() long long b_prev_ = clock64()
#define BWD_T(i) do { const long long n_ = clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd((unsigned long long*)&g_ppo_clk[1][i], (unsigned long long)(n_ - b_prev_)); b_prev_ = n_; } while (0)
#else
#define BWD_T0() do {} while (0)
#define BWD_T(i) do {} while (0)
#endif
#ifndef FRL_LANES_OPAQUE4
#define FRL_LANES_OPAQUE4 1     // the four-wave kernels' lanes() opaque as well (see ChainNetT::lanes)
#endif
#ifndef FRL_FW_HALF
#define FRL_FW_HALF 1      // the 128 x 128 layer's fragments four output tiles at a time in the eight-wave kernels: 16 registers in flight, not 32 (critic 0.4692 -> 0.4659 ms, actor 0.3120 -> 0.3109, same box)
#endif

namespace frl {

constexpr int kChainBatch = 256;          // rows of per-row staging (actions, targets, ...) the carve provides

// Parameter block of one head of the shape these kernels serve (in <= 16 -> 128 -> 128 -> out <= 16; build_net's packing:
// W1[128 x 16] b1[128] W2[128 x 128] b2[128] W3[16 x 128] b3[16]) as compile-time offsets — the host checks a net's
// descriptor against them before it picks this family (chained_shape), and the kernels keep ~40 scalar registers free of
// descriptor words.  A twin critic is two such blocks; a Gaussian actor's log_std follows its single block.
// (the constants themselves: frl_desc.h, kL1w .. kHeadFloats)

struct ChainLds {
    lds_f w1, w2, w3, b1, b2, b3, ls, ea, eb, ab, yb, q1, lpn, red, ex;
};
constexpr int chain_lds_floats() { return 8 * 256 + 64 * 256 + 8 * 256 + 2 * 8192 + 128 + 128 + 16 + 16 + kChainBatch * 4 + 3 * kChainBatch + 64; }
// the eight-wave workgroups (ChainNetT<8>) carry one more kilofloat: `ex`, the narrow tile (dz / x) of each of the eight waves in
// the exchanges that put all 128 rows at once, is `ab` (dead during a backward) plus these 4 KB behind it
constexpr int chain8_lds_floats() { return chain_lds_floats() + 1024; }

// the weight-gradient accumulators one lane owns for one 3-layer head, TRANSPOSED (MFMA D layout of dW^T: in = 16*kt + 4q + r,
// out = 16*ot + i16 — the lane's four registers are the 16-byte slot (q, f = i16) of image tile (ot, kt)):
// layer 2: ot in {2w, 2w+1} x kt 0..7; layer 1 (one 16-wide input block): ot in {2w, 2w+1}; head: kt in {2w, 2w+1}
// NW = waves of the workgroup (4 or 8): a wave owns OT = 8 / NW tile rows of layers 1 / 2 and as many k-tiles of the head
template <int NW>
struct HeadGradT {
    static constexpr int OT = kHT / NW;
    f32x4 g2[OT][kHT], g1[OT], g3[OT];
    float gb1[OT], gb2[OT], gb3;
};
using HeadGrad = HeadGradT<4>;

// (no flags in here: the update runs inside MFMA chains, where a branch would cut the scheduling region — the soft target update is
// a template argument of the functions below, weight decay is applied unconditionally: g + 0 * theta = g)
struct AdamCoef { float coef, step, inv_bc2s, w1, w2, beta2, eps, wd, tk, tau; };

// The four parameter arrays of one net of one learner as raw buffer resources: the update addresses them as
// buffer_load/store_dwordx4 v, <lane offset VGPR>, s[rsrc], <tile offset: SGPR / literal> — one VGPR of address state for the
// whole update instead of a 64-bit pointer per array and tile (which the register allocator spilled inside the MFMA chains).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
struct AdamBuf { __amdgpu_buffer_rsrc_t th, mm, vv, tg; };
__device__ __forceinline__ AdamBuf adam_buf(g_f th, g_f mA, g_f vA, g_f tg) {
    AdamBuf B;
    B.th = __builtin_amdgcn_make_buffer_rsrc((float*)th, 0, 0x7fffffff, 0x00020000);
    B.mm = __builtin_amdgcn_make_buffer_rsrc((float*)mA, 0, 0x7fffffff, 0x00020000);
    B.vv = __builtin_amdgcn_make_buffer_rsrc((float*)vA, 0, 0x7fffffff, 0x00020000);
    B.tg = __builtin_amdgcn_make_buffer_rsrc((float*)tg, 0, 0x7fffffff, 0x00020000);
    return B;
}
// cache-policy bits of the update's loads / stores and of the staging loads (developer knobs: 2 = slc / non-temporal)
#ifndef FRL_ADAM_AUX_LD
#define FRL_ADAM_AUX_LD 2
#endif
#ifndef FRL_ADAM_AUX_ST
#define FRL_ADAM_AUX_ST 2
#endif
#ifndef FRL_STAGE_NT
#define FRL_STAGE_NT 0      // measured: no gain on the staging loads
#endif
__device__ __forceinline__ f32x4 ld4_stage(g_cf p) {
#if FRL_STAGE_NT
    return __builtin_nontemporal_load(reinterpret_cast<const FRL_GLB f32x4*>(p));
#else
    return ld4(p);
#endif
}
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, FRL_ADAM_AUX_LD));
}
// Stores take their whole offset in the VGPR / immediate field, never in an SGPR soffset: a 128-bit VMEM store reads its data
// registers over several cycles and a VALU write to them in the next issue slot corrupts a quarter wave of it (measured, round 3:
// the target copy's store followed by the next tile's first multiply — 16 lanes of the tile held coef * gradient afterwards).
// hipcc pads that hazard with wait states only when soffset is NOT a register (the GCN3 rule); on gfx950 it bites either way.
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, int voff, int const_off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), r, voff + const_off, 0, FRL_ADAM_AUX_ST);
}



// NW = waves per workgroup.  4 (256 threads): a wave carries its 16 (or T x 16) rows through the whole MLP at ONE wave per SIMD
// and owns two tile rows of every weight-gradient.  8 (512 threads, round 6): the same chains on twice as many waves, each
// owning ONE tile row — half the accumulators, every wave inside 256 registers, TWO waves per SIMD: one wave's exchange writes,
// row fetches, epilogues and its Adam stream overlap the MFMAs of the wave it shares the SIMD with.  The images are one learner's
// either way; the exchange buffers hold 64 rows (NW = 4: a chunk; NW = 8: half a chunk — the layer-2 exchange runs in two
// halves, the two narrow ones put all 128 rows at once: 8 x 8 wide tiles fill ea + eb, the ninth tile of a wave goes to `ex`).
template <int NW>
struct ChainNetT {
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static constexpr int OT = kHT / NW, kThreads = 64 * NW, kRows = 16 * NW;
    using Grad = HeadGradT<NW>;
    ChainLds S;
    int tid, l, w, i16, q, fslot, tslot;
    int lane2, lane1;      // byte offsets of this lane's slot inside its first owned tile of the 128 x 128 layer / of the two narrow layers

    __device__ __forceinline__ void init(float* smem) {
        lds_f p = (lds_f)smem;
        S.w1 = p; p += kHT * 256;
        S.w2 = p; p += kHT * kHT * 256;
        S.w3 = p; p += kHT * 256;
        S.ea = p; p += kHT * 4 * 256;
        S.eb = p; p += kHT * 4 * 256;
        S.b1 = p; p += kHid;
        S.b2 = p; p += kHid;
        S.b3 = p; p += 16;
        S.ls = p; p += 16;
        if constexpr (NW == 4) {
            S.ab = p; p += kChainBatch * 4;
            S.yb = p; p += kChainBatch;
            S.q1 = p; p += kChainBatch;
            S.lpn = p; p += kChainBatch;
            S.red = p; p += 64;
            S.ex = S.ab;                                               // (unused)
        } else {
            S.yb = p; p += kChainBatch;
            S.q1 = p; p += kChainBatch;
            S.lpn = p; p += kChainBatch;
            S.red = p; p += 64;
            S.ab = p; p += kChainBatch * 4;                            // ex = ab + the kilofloat behind it (chain8_lds_floats)
            S.ex = S.ab;
        }
        init_lanes();
    }
    // the lane constants alone (device/chain_wide.hpp carves its own LDS and borrows the tile helpers below)
    __device__ __forceinline__ void init_lanes() {
        tid = threadIdx.x; l = tid & 63; w = __builtin_amdgcn_readfirstlane(tid >> 6); i16 = l & 15; q = l >> 4;
        fslot = (q * 16 + (i16 ^ q)) << 2;                             // forward / exchange fragment read (16 B)
        tslot = (((i16 >> 2) * 16) << 2) + (i16 & 3);                  // transposed read / owner write: + ((f ^ (i16 >> 2)) << 2)
        lane2 = 4 * (w * OT * kHT * 256 + fslot);
        lane1 = 4 * (w * OT * 256 + fslot);
    }

    // The lane constants as short-lived LOCALS.  The members above are loop invariants of the whole kernel: hipcc hoists every
    // LDS address built on them (a VGPR per 64 KB window, per transposed-access register r, per wave-dependent tile offset: ~30) to
    // the kernel's top and, at 256 registers, spills them — and a spill RELOAD is a scratch load behind `s_waitcnt vmcnt(0)`, i.e. a
    // wait for every row / image prefetch in flight.  The eight-wave kernels re-derive them from the lane number behind an opaque
    // asm at the top of each phase (five VALU instructions) so that an address lives exactly as long as the phase that uses it.
    struct LaneK { int i16, q, fslot, tslot; };
    __device__ __forceinline__ LaneK lanes() const {
        int L = l;
        if constexpr (NW == 8 || FRL_LANES_OPAQUE4) asm volatile("" : "+v"(L));
        LaneK K;
        K.i16 = L & 15; K.q = L >> 4;
        K.fslot = (K.q * 16 + (K.i16 ^ K.q)) << 2;
        K.tslot = (((K.i16 >> 2) * 16) << 2) + (K.i16 & 3);
        return K;
    }

    // ---- one net's three layers -> LDS images: the HBM block is already in image order (NetDesc::frag), a linear copy.
    // stage_fetch issues every load of the net (84 registers per lane at four waves, 44 at eight) and can sit IN FRONT of the
    // previous net's last pass: all 256 workgroups stage at the same moment, 21 MB in one burst that HBM serves in ~9-12 k cycles —
    // under a pass's MFMAs that costs nothing.  stage_commit waits for the other waves to be done with the old images and stores.
    // th = the net's block, head = which head of it; extra_n > 0: a single-head net's log_std entries behind its block
    struct StageRegs { f32x4 t2[64 / NW], t1[8 / NW], t3[8 / NW]; float bb1, bb2, bb3, lsv; };
    __device__ __forceinline__ StageRegs stage_fetch(g_cf th, int head, int extra_n = 0) const {
        th += head * kHeadFloats;
        StageRegs R;
#pragma unroll
        for (int j = 0; j < 64 / NW; ++j) R.t2[j] = ld4_stage(th + kL2w + 4 * (tid + kThreads * j));
#pragma unroll
        for (int j = 0; j < 8 / NW; ++j) { R.t1[j] = ld4_stage(th + kL1w + 4 * (tid + kThreads * j)); R.t3[j] = ld4_stage(th + kL3w + 4 * (tid + kThreads * j)); }
        R.bb1 = R.bb2 = R.bb3 = R.lsv = 0.f;
        if (tid < kHid) { R.bb1 = th[kL1b + tid]; R.bb2 = th[kL2b + tid]; }
        if (tid < 16) { R.bb3 = th[kL3b + tid]; R.lsv = tid < extra_n ? th[kHeadFloats + tid] : 0.f; }
        __builtin_amdgcn_sched_barrier(0);                             // the loads stay here: hipcc would sink them to the stores
        return R;
    }
    __device__ __forceinline__ void stage_commit(const StageRegs& R) const {
        if constexpr (!(FRL_ABL & 4)) lds_barrier();                   // every wave is done with the previous images
#pragma unroll
        for (int j = 0; j < 64 / NW; ++j) st4(S.w2 + 4 * (tid + kThreads * j), R.t2[j]);
#pragma unroll
        for (int j = 0; j < 8 / NW; ++j) { st4(S.w1 + 4 * (tid + kThreads * j), R.t1[j]); st4(S.w3 + 4 * (tid + kThreads * j), R.t3[j]); }
        if (tid < kHid) { S.b1[tid] = R.bb1; S.b2[tid] = R.bb2; }
        if (tid < 16) { S.b3[tid] = R.bb3; S.ls[tid] = R.lsv; }
        if constexpr (!(FRL_ABL & 4)) lds_barrier();
    }
    __device__ __forceinline__ void stage(g_cf th, int head, int extra_n = 0) const { stage_commit(stage_fetch(th, head, extra_n)); }

    // ---- the chained forward of T x 16 rows per wave: x (B operand of layer 1) -> h1, h2 -> head tile z.  A tile's 32 MFMAs
    // per layer are ONE dependent accumulator chain: the 8 T chains of a layer run side by side (k-block outer, k-step
    // middle, tile inner), so that consecutive MFMAs never wait for each other's result.
    template <int T>
    __device__ __forceinline__ void forward(const f32x4 (&xb)[T], f32x4 (&h1)[T][kHT], f32x4 (&h2)[T][kHT], f32x4 (&z)[T]) const {
        forward<T, false>(xb, h1, h2, z);
    }
    // ... with the head of hn <= 4 outputs as dot products (head_valu)
    template <int T>
    __device__ __forceinline__ void forward_vh(const f32x4 (&xb)[T], f32x4 (&h1)[T][kHT], f32x4 (&h2)[T][kHT], f32x4 (&z)[T], int hn) const {
        forward<T, true>(xb, h1, h2, z, hn);
    }
    template <int T, bool VH>
    __device__ __forceinline__ void forward(const f32x4 (&xb)[T], f32x4 (&h1)[T][kHT], f32x4 (&h2)[T][kHT], f32x4 (&z)[T], int hn = 0) const {
        const LaneK K = lanes();
        // first layer: all eight fragments and biases in flight before the first MFMA.  (All four k-steps are needed whatever the
        // input width: k-step e of the fragment layout holds columns e, 4 + e, 8 + e, 12 + e, so the zero padding is spread
        // over every step — a build that skipped "unused" steps dropped real columns and failed parity by 1 %.)
        {
            f32x4 w1f[kHT], b1f[kHT];
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot) { w1f[ot] = ld4((lds_cf)(S.w1 + ot * 256 + K.fslot)); b1f[ot] = ld4((lds_cf)(S.b1 + ot * 16 + 4 * K.q)); }
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const f32x4 acc = mfma4(b1f[ot], w1f[ot], xb[t]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[t][ot][r] = fmaxf(acc[r], 0.f);
                }
        }
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            const f32x4 bb = ld4((lds_cf)(S.b2 + ot * 16 + 4 * K.q));
#pragma unroll
            for (int t = 0; t < T; ++t) h2[t][ot] = bb;
        }
#if FRL_FW_HALF
        if constexpr (NW == 8) {                                       // fragments of four output tiles at a time: 16 registers in flight, not 32
            static_for<0, 2 * kHT>([&](auto kc) {
                constexpr int kb = decltype(kc)::value >> 1, o0 = (decltype(kc)::value & 1) * 4;
                f32x4 wf[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) wf[o] = ld4((lds_cf)(S.w2 + ((o0 + o) * kHT + kb) * 256 + K.fslot));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int o = 0; o < 4; ++o)
#pragma unroll
                        for (int t = 0; t < T; ++t) h2[t][o0 + o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[o][e], h1[t][kb][e], h2[t][o0 + o], 0, 0, 0);
            });
        } else
#endif
        static_for<0, kHT>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            f32x4 wf[kHT];
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot) wf[ot] = ld4((lds_cf)(S.w2 + (ot * kHT + kb) * 256 + K.fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                    for (int t = 0; t < T; ++t) h2[t][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], h1[t][kb][e], h2[t][ot], 0, 0, 0);
        });
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[t][ot][r] = fmaxf(h2[t][ot][r], 0.f);
        if constexpr (VH) head_valu<T>(h2, z, hn); else head_mfma<T>(h2, z);
    }
    // head layer as MFMA tiles: z[t][r] = output 4q + r of this lane's row (16 outputs)
    template <int T>
    __device__ __forceinline__ void head_mfma(const f32x4 (&h2)[T][kHT], f32x4 (&z)[T]) const {
        const LaneK K = lanes();
        const f32x4 b3 = ld4((lds_cf)(S.b3 + 4 * K.q));
#pragma unroll
        for (int t = 0; t < T; ++t) z[t] = b3;
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {
            const f32x4 wf = ld4((lds_cf)(S.w3 + kb * 256 + K.fslot));
#pragma unroll
            for (int t = 0; t < T; ++t) z[t] = mfma4(z[t], wf, h2[t][kb]);
        }
    }
    // A head of hn <= 4 outputs (every critic: 1; the actors: act_dim) as dot products instead of 32 T MFMAs with 1/16 .. 4/16
    // useful columns: a lane holds 32 of its row's 128 hidden features (16 kb + 4q + r), the slot (q, f = o) of image tile kb holds
    // exactly W3[o][16 kb + 4q .. + 3], the four lane groups' partial sums meet by two xor-shuffles.  z[t][o] lands on every lane
    // group (the callers read it on group 0); entries o >= hn are zero.
    template <int T>
    __device__ __forceinline__ void head_valu(const f32x4 (&h2)[T][kHT], f32x4 (&z)[T], int hn) const {
        const LaneK K = lanes();
#pragma unroll
        for (int t = 0; t < T; ++t) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < hn) {
                float acc[T];
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = 0.f;
#pragma unroll
                for (int kb = 0; kb < kHT; ++kb) {
                    const f32x4 wv = ld4((lds_cf)(S.w3 + kb * 256 + ((K.q * 16 + (o ^ K.q)) << 2)));
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[t] = fmaf(wv[r], h2[t][kb][r], acc[t]);
                }
                const float bo = S.b3[o];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    float a = acc[t];
                    a += lane_xor<16>(a);
                    a += lane_xor<32>(a);
                    z[t][o] = a + bo;
                }
            }
        }
    }
    // dH2 = W3^T dz through the ReLU of h2 for such a head: dz[o] of the row sits on lane group 0 and is broadcast to the row's
    // other groups; the fragments are the forward's
    __device__ __forceinline__ void delta2_valu(const f32x4& dz, const f32x4 (&h2)[kHT], f32x4 (&d2)[kHT], int hn) const {
        const LaneK K = lanes();
        f32x4 dzb;
#pragma unroll
        for (int o = 0; o < 4; ++o) dzb[o] = __shfl(dz[o], K.i16, 64);
#pragma unroll
        for (int it = 0; it < kHT; ++it) d2[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < hn) {
#pragma unroll
                for (int it = 0; it < kHT; ++it) {
                    const f32x4 wv = ld4((lds_cf)(S.w3 + it * 256 + ((K.q * 16 + (o ^ K.q)) << 2)));
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2[it][r] = fmaf(wv[r], dzb[o], d2[it][r]);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < kHT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[it][r] = h2[it][r] > 0.f ? d2[it][r] : 0.f;
    }

    // dH2 = W3^T dz through the ReLU of h2 (A = W3^T: transposed fragment reads)
    __device__ __forceinline__ void delta2(const f32x4& dz, const f32x4 (&h2)[kHT], f32x4 (&d2)[kHT]) const {
        const LaneK K = lanes();
#pragma unroll
        for (int it = 0; it < kHT; ++it) {
            f32x4 wa;
#pragma unroll
            for (int e = 0; e < 4; ++e) wa[e] = S.w3[it * 256 + K.tslot + (((4 * K.q + e) ^ (K.i16 >> 2)) << 2)];
            const f32x4 acc = mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, wa, dz);
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[it][r] = h2[it][r] > 0.f ? acc[r] : 0.f;
        }
    }
    // dH1 = W2^T dz2 through the ReLU of h1; eight accumulator chains side by side.  The transposed fragments of W2 are 4-byte LDS
    // reads, eight per k-step: the reads of step s + 1 are issued in front of the MFMAs of step s (double-buffered in the source —
    // hipcc waited for every step's reads right in front of its MFMAs: ~130 exposed cycles x 32 steps per call)
    __device__ __forceinline__ void delta1(const f32x4 (&d2)[kHT], const f32x4 (&h1)[kHT], f32x4 (&d1)[kHT]) const {
        const LaneK K = lanes();
#pragma unroll
        for (int it = 0; it < kHT; ++it) d1[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        float wa[2][kHT];
        auto fetch = [&](int ob, int e, float (&dst)[kHT]) {
#pragma unroll
            for (int it = 0; it < kHT; ++it) dst[it] = S.w2[(ob * kHT + it) * 256 + K.tslot + (((4 * K.q + e) ^ (K.i16 >> 2)) << 2)];
        };
        fetch(0, 0, wa[0]);
        static_for<0, 4 * kHT>([&](auto sc) {
            constexpr int s_ = decltype(sc)::value, ob = s_ >> 2, e = s_ & 3;
            if constexpr (s_ + 1 < 4 * kHT) fetch((s_ + 1) >> 2, (s_ + 1) & 3, wa[(s_ + 1) & 1]);
#pragma unroll
            for (int it = 0; it < kHT; ++it) d1[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_ & 1][it], d2[ob][e], d1[it], 0, 0, 0);
        });
#pragma unroll
        for (int it = 0; it < kHT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[it][r] = h1[it][r] > 0.f ? d1[it][r] : 0.f;
    }
    // dX = W1^T dz1: d loss / d input column 4q + r of this lane's row (no activation in front of the input)
    __device__ __forceinline__ f32x4 delta0(const f32x4 (&d1)[kHT]) const {
        const LaneK K = lanes();
        f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ob = 0; ob < kHT; ++ob) {
            f32x4 wa;
#pragma unroll
            for (int e = 0; e < 4; ++e) wa[e] = S.w1[ob * 256 + K.tslot + (((4 * K.q + e) ^ (K.i16 >> 2)) << 2)];
            dx = mfma4(dx, wa, d1[ob]);
        }
        return dx;
    }

    // exchange tiles: E[(feature tile ft) * 4 + (wave slot)][256] in image order, 64 rows = four wave slots per buffer
    __device__ __forceinline__ void put_tile(lds_f E, int ft, const f32x4& t) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) E[(ft * 4 + (w & 3)) * 256 + tslot + (((4 * q + r) ^ (i16 >> 2)) << 2)] = t[r];
    }
    __device__ __forceinline__ f32x4 get_frag(lds_cf E, int ft, int bb) const { return ld4(E + (ft * 4 + bb) * 256 + fslot); }
    __device__ __forceinline__ void put_tile(const LaneK& K, lds_f E, int ft, const f32x4& t) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) E[(ft * 4 + (w & 3)) * 256 + K.tslot + (((4 * K.q + r) ^ (K.i16 >> 2)) << 2)] = t[r];
    }
    __device__ __forceinline__ f32x4 get_frag(const LaneK& K, lds_cf E, int ft, int bb) const { return ld4(E + (ft * 4 + bb) * 256 + K.fslot); }
    // ... and the all-rows form of the eight-wave workgroups: ea + eb as ONE buffer of [ft][8 wave slots]; a wave's narrow tile in ex
    __device__ __forceinline__ void put_tile8(const LaneK& K, lds_f E, int ft, const f32x4& t) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) E[(ft * 8 + w) * 256 + K.tslot + (((4 * K.q + r) ^ (K.i16 >> 2)) << 2)] = t[r];
    }
    __device__ __forceinline__ f32x4 get_frag8(const LaneK& K, lds_cf E, int ft, int bb) const { return ld4(E + (ft * 8 + bb) * 256 + K.fslot); }

    __device__ __forceinline__ void grad_zero(Grad& g) const {
#pragma unroll
        for (int x = 0; x < OT; ++x) {
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.g1[x] = f32x4{0.f, 0.f, 0.f, 0.f}; g.g3[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.gb1[x] = 0.f; g.gb2[x] = 0.f;
        }
        g.gb3 = 0.f;
    }

    // ---- backward of one chunk (16 rows per wave: 64 rows at four waves, 128 at eight) into the owners' accumulators: three
    // exchanges through ea / eb (H2 + dz -> head gradient; H1 + dz2 -> layer 2; X + dz1 -> layer 1), the dH chains in between
    // hn > 0: the head has hn <= 4 outputs and its dH runs as dot products (delta2_valu)
    struct NoHook { __device__ __forceinline__ void operator()() const {} };
    // late(): called once where the fewest registers are live (behind the dH1 chain: only d1 and x are left) — the callers issue
    // the NEXT image's fetch there: its 44 registers riding through the whole forward + backward were spilled by hipcc right
    // behind the loads (i.e. with a wait for HBM in the open), and so they were when issued behind the head exchange
    template <class Late = NoHook>
    __device__ __forceinline__ void backward(Grad& g, const f32x4& xb, const f32x4 (&h1)[kHT], const f32x4 (&h2)[kHT], const f32x4& dz, int hn = 0,
                                             Late&& late = NoHook{}) const {
        if constexpr (NW == 8) { backward8(g, xb, h1, h2, dz, hn, late); return; }
        late();
        lds_barrier();                                                 // the previous chunk's readers of ea / eb are done
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) put_tile(S.ea, ft, h2[ft]);
        put_tile(S.eb, 0, dz);
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 af = get_frag(S.eb, 0, bb);
            if (w == 0) g.gb3 += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
            for (int x = 0; x < OT; ++x) g.g3[x] = mfma4(g.g3[x], get_frag(S.ea, OT * w + x, bb), af);
        }
        f32x4 d2[kHT];
        if (hn > 0) delta2_valu(dz, h2, d2, hn); else delta2(dz, h2, d2);
        lds_barrier();
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) { put_tile(S.ea, ft, h1[ft]); put_tile(S.eb, ft, d2[ft]); }
        lds_barrier();
        layer2_consume(g);
        f32x4 d1[kHT];
        delta1(d2, h1, d1);
        lds_barrier();
        put_tile(S.ea, 0, xb);
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) put_tile(S.eb, ft, d1[ft]);
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 bf = get_frag(S.ea, 0, bb);
#pragma unroll
            for (int x = 0; x < OT; ++x) {
                const f32x4 af = get_frag(S.eb, OT * w + x, bb);
                g.gb1[x] += (af[0] + af[1]) + (af[2] + af[3]);
                g.g1[x] = mfma4(g.g1[x], bf, af);
            }
        }
    }
    // the layer-2 contraction over the 64 rows that lie in ea (h1) / eb (d2): this wave's OT tile rows x 8 k-tiles
    __device__ __forceinline__ void layer2_consume(Grad& g) const {
        const LaneK K = lanes();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            f32x4 af[OT], bf[kHT];
#pragma unroll
            for (int x = 0; x < OT; ++x) {
                af[x] = get_frag(K, S.eb, OT * w + x, bb);
                g.gb2[x] += (af[x][0] + af[x][1]) + (af[x][2] + af[x][3]);
            }
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) bf[kt] = get_frag(K, S.ea, kt, bb);
#pragma unroll
            for (int x = 0; x < OT; ++x)
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = mfma4(g.g2[x][kt], bf[kt], af[x]);
        }
    }
    // Eight waves, 128 rows.  The two narrow exchanges (head: H2 + dz; first layer: dz1 + X) put ALL rows at once — the wide
    // operand's 8 x 8 tiles are exactly ea + eb, the narrow one's eight tiles lie in ex — two barriers each.  The layer-2 exchange
    // needs H1 and dz2 of every row, 128 KB: it runs in two 64-row halves (the waves of the other half wait out the ~500 cycles of
    // a half's writes: 2 % of a chunk; running their dH1 chain there instead was built — it holds d1 through the contraction, 32
    // registers the kernel does not have).
    __device__ __forceinline__ void xbar() const { if constexpr (!(FRL_ABL & 2)) lds_barrier(); }
    __device__ __forceinline__ void xput8(const LaneK& K, lds_f E, int ft, const f32x4& t) const { if constexpr (!(FRL_ABL & 10)) put_tile8(K, E, ft, t); else asm volatile("" :: "v"(t)); }
    __device__ __forceinline__ void xput(const LaneK& K, lds_f E, int ft, const f32x4& t) const { if constexpr (!(FRL_ABL & 10)) put_tile(K, E, ft, t); else asm volatile("" :: "v"(t)); }
    template <class Late>
    __device__ __forceinline__ void backward8(Grad& g, const f32x4& xb, const f32x4 (&h1)[kHT], const f32x4 (&h2)[kHT], const f32x4& dz, int hn, Late&& late) const {
        const lds_f E = S.ea;
        const int hv = w >> 2;                                         // which 64-row half this wave's rows belong to
        BWD_T0();
        xbar();
        BWD_T(0);                                                        // the previous chunk's readers of ea / eb / ex are done
        {
            const LaneK K = lanes();
#pragma unroll
            for (int ft = 0; ft < kHT; ++ft) xput8(K, E, ft, h2[ft]);
            xput8(K, S.ex, 0, dz);
        }
        BWD_T(1);
        f32x4 d2[kHT];
        if (hn > 0) delta2_valu(dz, h2, d2, hn); else delta2(dz, h2, d2);
        BWD_T(2);
        xbar();
        BWD_T(0);
        {
            const LaneK K = lanes();
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                const f32x4 af = get_frag8(K, S.ex, 0, bb);
                if (w == 0) g.gb3 += (af[0] + af[1]) + (af[2] + af[3]);
                g.g3[0] = mfma4(g.g3[0], get_frag8(K, E, w, bb), af);
            }
        }
        BWD_T(3);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            xbar();
            BWD_T(0);
            if (hv == half) {
                const LaneK K = lanes();
#pragma unroll
                for (int ft = 0; ft < kHT; ++ft) { xput(K, S.ea, ft, h1[ft]); xput(K, S.eb, ft, d2[ft]); }
            }
            BWD_T(1);
            xbar();
            BWD_T(0);
            layer2_consume(g);
            BWD_T(4);
        }
        f32x4 d1[kHT];
        delta1(d2, h1, d1);
        BWD_T(5);
        late();
        xbar();
        BWD_T(0);
        {
            const LaneK K = lanes();
            xput8(K, S.ex, 0, xb);
#pragma unroll
            for (int ft = 0; ft < kHT; ++ft) xput8(K, E, ft, d1[ft]);
        }
        BWD_T(1);
        xbar();
        BWD_T(0);
        {
            const LaneK K = lanes();
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                const f32x4 bf = get_frag8(K, S.ex, 0, bb);
                const f32x4 af = get_frag8(K, E, w, bb);
                g.gb1[0] += (af[0] + af[1]) + (af[2] + af[3]);
                g.g1[0] = mfma4(g.g1[0], bf, af);
            }
        }
        BWD_T(6);
    }

    // after the last chunk: the bias partials of the four lane groups (rows 4q..4q+3 of every 16-row block) added up
    __device__ __forceinline__ void grad_finish(Grad& g) const {
#pragma unroll
        for (int x = 0; x < OT; ++x) {
            g.gb1[x] += lane_xor<16>(g.gb1[x]); g.gb1[x] += lane_xor<32>(g.gb1[x]);
            g.gb2[x] += lane_xor<16>(g.gb2[x]); g.gb2[x] += lane_xor<32>(g.gb2[x]);
        }
        g.gb3 += lane_xor<16>(g.gb3); g.gb3 += lane_xor<32>(g.gb3);
    }
    // this lane's share of the squared gradient norm (bias entries counted once: lanes q == 0 / wave 0)
    __device__ __forceinline__ float grad_sumsq(const Grad& g) const {
        float ss = 0.f;
#pragma unroll
        for (int x = 0; x < OT; ++x) {
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt)
                ss += (g.g2[x][kt][0] * g.g2[x][kt][0] + g.g2[x][kt][1] * g.g2[x][kt][1]) + (g.g2[x][kt][2] * g.g2[x][kt][2] + g.g2[x][kt][3] * g.g2[x][kt][3]);
            ss += (g.g1[x][0] * g.g1[x][0] + g.g1[x][1] * g.g1[x][1]) + (g.g1[x][2] * g.g1[x][2] + g.g1[x][3] * g.g1[x][3]);
            ss += (g.g3[x][0] * g.g3[x][0] + g.g3[x][1] * g.g3[x][1]) + (g.g3[x][2] * g.g3[x][2] + g.g3[x][3] * g.g3[x][3]);
            if (q == 0) ss += g.gb1[x] * g.gb1[x] + g.gb2[x] * g.gb2[x];
        }
        if (w == 0 && q == 0) ss += g.gb3 * g.gb3;
        return ss;
    }

    // ---- clip + Adam + soft update of one head's three layers from the owners' registers.  A lane's accumulator tile is the
    // 16-byte slot `fslot` of its image tile, and the HBM block is in image order: every load / store below is one dwordx4 per
    // lane and one whole contiguous 1 KB tile per wave-instruction.  Loads of a batch of tiles before its stores (the compiler
    // cannot prove the four arrays distinct and would wait for every store before the next load).  No LDS, no barriers.
    struct AdamIn { f32x4 th, mm, vv, tg; };
    // accumulator unit J of a head block: J < 8 OT the 128 x 128 layer's tile (ot = OT w + J / 8, kt = J % 8), then the first layer's
    // OT tiles ot = OT w + x, then the head layer's OT tiles kb = OT w + x.  Its 16-byte slot sits at lane offset
    // unit_voff<J>() + the compile-time unit_soff<J>() bytes inside the head's block.
    static constexpr int kUnits = 10 * OT;
    template <int J> __device__ __forceinline__ int unit_voff() const {
        if constexpr (NW == 8) { const int fs = lanes().fslot; return 4 * (w * OT * (J < 8 * OT ? kHT : 1) * 256 + fs); }
        else return J < 8 * OT ? lane2 : lane1;
    }
    template <int J> static constexpr int unit_soff() { return 4 * (J < 8 * OT ? kL2w + J * 256 : (J < 9 * OT ? kL1w + (J - 8 * OT) * 256 : kL3w + (J - 9 * OT) * 256)); }
    template <int J>
    __device__ __forceinline__ static const f32x4& unit_grad(const Grad& g) {
        if constexpr (J < 8 * OT) return g.g2[J / 8][J % 8];
        else if constexpr (J < 9 * OT) return g.g1[J - 8 * OT];
        else return g.g3[J - 9 * OT];
    }
    // HB = byte offset of the head's block inside the net (head * kHeadFloats * 4).  THL: theta from the head's image in LDS — the
    // last head staged is still there, in the same slot order — instead of a second trip to HBM (the update phase is the one
    // that streams HBM with every CU at once)
    template <bool SOFT, int J, int HB, bool THL>
    __device__ __forceinline__ AdamIn adam_load(const AdamBuf& B) const {
        AdamIn X;
        constexpr int so = HB + unit_soff<J>();
        const int vo = unit_voff<J>();
        if constexpr (THL) X.th = ld4((lds_cf)((J < 8 * OT ? S.w2 + (w * OT * kHT + J) * 256 : (J < 9 * OT ? S.w1 + (w * OT + J - 8 * OT) * 256 : S.w3 + (w * OT + J - 9 * OT) * 256)) + (NW == 8 ? lanes().fslot : fslot)));
        else X.th = buf_ld4(B.th, vo, so);
        X.mm = buf_ld4(B.mm, vo, so); X.vv = buf_ld4(B.vv, vo, so);
        if constexpr (SOFT) X.tg = buf_ld4(B.tg, vo, so); else X.tg = f32x4{0.f, 0.f, 0.f, 0.f};
        return X;
    }
    template <bool SOFT>
    __device__ __forceinline__ AdamIn adam_compute(const AdamCoef& c, const f32x4& gr, const AdamIn& in) const {
        AdamIn R = in;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gi = gr[r] * c.coef;
            gi += c.wd * R.th[r];
            float m1 = R.mm[r], v1 = R.vv[r];
            R.th[r] = adam_elem(R.th[r], gi, m1, v1, c.w1, c.w2, c.beta2, c.inv_bc2s, c.eps, c.step);
            R.mm[r] = m1; R.vv[r] = v1;
            if constexpr (SOFT) R.tg[r] = R.tg[r] * c.tk + R.th[r] * c.tau;
        }
        return R;
    }
    template <bool SOFT, int J, int HB>
    __device__ __forceinline__ void adam_store(const AdamBuf& B, const AdamIn& R) const {
        constexpr int so = HB + unit_soff<J>();
        const int vo = unit_voff<J>();
        buf_st4(B.th, vo, so, R.th); buf_st4(B.mm, vo, so, R.mm); buf_st4(B.vv, vo, so, R.vv);
        if constexpr (SOFT) buf_st4(B.tg, vo, so, R.tg);
    }
    template <bool SOFT>
    __device__ __forceinline__ float adam_scalar(g_f th, g_f mA, g_f vA, g_f tg, const AdamCoef& c, int o, float gr) const {
        float t1 = th[o], m1 = mA[o], v1 = vA[o];
        float gi = gr * c.coef;
        gi += c.wd * t1;
        t1 = adam_elem(t1, gi, m1, v1, c.w1, c.w2, c.beta2, c.inv_bc2s, c.eps, c.step);
        th[o] = t1; mA[o] = m1; vA[o] = v1;
        if constexpr (SOFT) tg[o] = tg[o] * c.tk + t1 * c.tau;
        return t1;
    }
    // biases (every lane group holds the full sums after grad_finish: group q == 0 writes) [, log_std behind a single head]
    template <bool SOFT>
    __device__ __forceinline__ void adam_biases(const Grad& g, g_f th, g_f mA, g_f vA, g_f tg, const AdamCoef& c, float g_extra, int extra_n) const {
        if (q == 0) {
#pragma unroll
            for (int x = 0; x < OT; ++x) {
                adam_scalar<SOFT>(th, mA, vA, tg, c, kL1b + (OT * w + x) * 16 + i16, g.gb1[x]);
                adam_scalar<SOFT>(th, mA, vA, tg, c, kL2b + (OT * w + x) * 16 + i16, g.gb2[x]);
            }
            if (w == 0) adam_scalar<SOFT>(th, mA, vA, tg, c, kL3b + i16, g.gb3);
            if (w == 1 && i16 < extra_n) adam_scalar<SOFT>(th, mA, vA, tg, c, kHeadFloats + i16, g_extra);
        }
    }
    // the whole update of head HD of a net, in the open: the units in batches of up to 8 (loads of a batch before its stores),
    // the 128 x 128 layer's tiles first, then first layer + head layer, then the biases.  th / mA / vA / tg = the NET's blocks.
    template <bool SOFT, int HD, bool THL = false>
    __device__ __forceinline__ void adam_head(const Grad& g, g_f th, g_f mA, g_f vA, g_f tg, const AdamCoef& c, float g_extra = 0.f,
                                              int extra_n = 0) const {
        const AdamBuf B = adam_buf(th, mA, vA, tg);
        constexpr int HB = HD * kHeadFloats * 4;
        constexpr int kBatch = NW == 4 ? 8 : 5;                        // (NW = 8: ten units in two batches of five — 80 registers of state)
        static_for<0, (kUnits + kBatch - 1) / kBatch>([&](auto bc) {
            constexpr int b0 = decltype(bc)::value * kBatch, nb = b0 + kBatch <= kUnits ? kBatch : kUnits - b0;
            AdamIn in[nb];
            static_for<0, nb>([&](auto j) { in[decltype(j)::value] = adam_load<SOFT, b0 + decltype(j)::value, HB, THL>(B); });
            // Every load of the batch has landed before its first store is issued: gfx9 counts loads and stores in one counter,
            // and with both kinds in flight hipcc's "all but the N youngest" waits rest on their retiring strictly in issue
            // order.  Nothing measured says they do not (the corruption first blamed on it was the store-data hazard, see
            // buf_st4) — the drain costs nothing here and takes the question off the table.
            __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0)
            static_for<0, nb>([&](auto j) {
                constexpr int J = b0 + decltype(j)::value;
                adam_store<SOFT, J, HB>(B, adam_compute<SOFT>(c, unit_grad<J>(g), in[decltype(j)::value]));
            });
        });
        adam_biases<SOFT>(g, th + HD * kHeadFloats, mA + HD * kHeadFloats, vA + HD * kHeadFloats, tg + HD * kHeadFloats, c, g_extra, extra_n);
    }
};
using ChainNet = ChainNetT<4>;

}  // namespace frl
