// The register-chained design (device/chain_net.hpp) for the shapes one LDS image per layer cannot hold: a FIRST LAYER of up to
// 416 input columns (SAC at Humanoid-v4's dims: 376 + 17; MADDPG's centralised critics: 69) and heads of up to 32 outputs, hidden
// 128 — kernels_criticw.hip / kernels_actorw.hip.  One workgroup owns one (learner, agent); parameters in fragment-image order
// in HBM (NetDesc::frag).
//
//   * W1 (up to 26 k-blocks x 8 output tiles = 208 KB) never sits in LDS as a whole: the first layer of a pass runs as a SWEEP
//     over K-slices of four k-blocks (32 KB), double-buffered — slice s + 1 travels global -> registers -> LDS under the MFMAs of
//     slice s — with the pre-activation accumulators of up to FOUR 16-row tiles per wave (the 256 rows of a super-chunk) in
//     registers, so a slice is fetched once per 256 rows and every fragment read from LDS feeds 4 x 4 MFMAs.  The row operand
//     needs no LDS at all: lane (row, q) of a tile reads columns 16 kb + 4 q .. + 3 of its row straight from the replay record
//     (or the per-row scratch that holds a policy's actions).
//   * layers 2 and 3 run on the images of W2 (64 KB) and the head (8 - 16 KB) exactly as in chain_net.hpp; the union that held
//     the W1 slices then serves the backward's activation / delta exchanges.
//   * dW1 (up to 200 16 x 16 tiles) cannot live in registers next to the backward of a chunk.  The chunk loop leaves the
//     first-layer deltas in a per-learner HBM scratch (in exchange-image order, 32 KB per 64 rows, L2-resident), and a separate
//     pass contracts them with the rows: wave w owns output half (w & 1) and the k-tiles of parity (w >> 1) — up to 13 x 4
//     accumulator tiles = 208 registers — and reads BOTH operands without LDS: the transposed row fragment is four dword loads
//     per (k-tile, 16-row block), the delta fragment one dwordx4 from the scratch image.
//   * every head's gradients go to the engine's `grad` array in image order; clip + Adam + soft update then stream the net once
//     (the accumulators of two wide heads do not fit next to each other, and the clip coefficient needs all of them).
#pragma once
#include "chain_net.hpp"
#include "update_common.hpp"

namespace frl {

// (the family's constants — kWideSliceKB, kWideSlice, kWideMaxKB1, kWideMaxKT, kWideApitch, kWideScratchPerRow, wide_lds_floats() —
// are in frl_desc.h: the host sizes LDS and scratch from them)

// per-(learner, agent) scratch in HBM (L2-resident); `bm` = batch_max rounded up to 64 rows, xp = the critic's padded input width,
// op = the widest actor's.  xrow[row] = one critic input row, built in place: [next_obs_all | a'_all] in the critic stage,
// [obs_all | act_all with a_i = actor_i(s_i)] in the actor stage — 16-byte aligned whatever the record's field offsets are, so
// that the sweeps read their row operand with ONE dwordx4 per lane and k-block; xobs = an agent's observation columns, copied
// only when they do not start on a 16-byte boundary in the record / in xrow.  Offsets in floats.
struct WideScratch {
    g_f xrow, xobs, dqa, yb, q1, lpn, dz1, ah1, ah2;
    int xp, op;
    __device__ __forceinline__ void init(g_f base, int bm, int xp_, int op_, int nag) {
        xp = xp_; op = op_;
        xrow = base; base += (size_t)bm * xp + 64;
        xobs = base; base += (size_t)nag * bm * op + 64;              // (critic stage: one copy per agent that needs it)
        dqa = base; base += (size_t)bm * kWideApitch;
        yb = base; q1 = base + bm; lpn = base + 2 * (size_t)bm; base += 4 * (size_t)bm;
        dz1 = base; base += 128 * (size_t)bm;
        ah1 = base; base += 128 * (size_t)bm;
        ah2 = base;
    }
};

// weight-gradient accumulators of one head a lane owns next to the chunk loop (layer 2 and the head; transposed, see chain_net.hpp)
template <int NT3>
struct WideGrad {
    f32x4 g2[2][kHT], g3[NT3][2];
    float gb1[2], gb2[2], gb3[NT3];
};

struct WideNet {
    ChainNet C;            // lane constants and the tile helpers; C.S points into the carve below (w1 / ab / yb / q1 / lpn unused)
    lds_f u;               // the union: W1 slice buffers u, u + kWideSlice  |  exchange buffers C.S.ea = u, C.S.eb = u + 8192

    __device__ __forceinline__ void init(float* smem) {
        lds_f p = (lds_f)smem;
        C.S.w2 = p; p += kHT * kHT * 256;
        C.S.w3 = p; p += 2 * kHT * 256;
        u = p; C.S.ea = p; C.S.eb = p + kHT * 4 * 256; p += 2 * kWideSlice;
        C.S.b1 = p; p += 128;
        C.S.b2 = p; p += 128;
        C.S.b3 = p; p += 32;
        C.S.ls = p; p += 32;
        C.S.red = p; p += 64;
        C.S.w1 = u; C.S.ab = u; C.S.yb = u; C.S.q1 = u; C.S.lpn = u;
        C.init_lanes();
    }

    // ---- the batch's ring rows -> an LDS table at the start of the union (the dW1 pass has no other use for it)
    __device__ __forceinline__ FRL_LDS int* stage_idx(g_ci idx, int B) const {
        FRL_LDS int* tab = (FRL_LDS int*)u;
        lds_barrier();
        for (int i = C.tid; i < B; i += kWG) tab[i] = idx[i];
        lds_barrier();
        return tab;
    }
    // ---- columns [src_off, src_off + n) of the batch's records -> dst[row * pitch + c], any alignment (dword loads).  tab = the
    // batch's ring rows in LDS (stage_idx).  A wave-instruction covers 64 / W rows x W columns (W = 16 / 32 / 64, the smallest
    // that holds n; wider rows take ceil(n / 64) chunks) and eight such row groups are in flight per wave: the first build walked
    // one row at a time behind its own index load — 780 k cycles for MADDPG's 1024 x 54 floats.
    __device__ __forceinline__ void copy_cols(g_f dst, int pitch, g_cf ring, int stride, const FRL_LDS int* tab, int B, int src_off, int n) const {
        const int lw = n > 32 ? 6 : (n > 16 ? 5 : 4), W = 1 << lw, rpi = 64 >> lw;      // rows per wave-instruction
        const int sub = C.l >> lw, col = C.l & (W - 1), nc = (n + 63) >> 6;
        const int ngroups = (B + rpi - 1) / rpi;
        if (nc == 1) {
            for (int g0 = C.w; g0 < ngroups; g0 += 4 * 8) {
                const int ccl = col < n ? col : n - 1;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = (g0 + 4 * k) * rpi + sub, rc = row < B ? row : B - 1;
                    v[k] = ring[(size_t)tab[rc] * stride + src_off + ccl];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = (g0 + 4 * k) * rpi + sub;
                    if (row < B && col < n) dst[(size_t)row * pitch + col] = v[k];
                }
            }
        } else {                                                       // wide rows (<= 448 columns): four rows x all chunks in flight
            for (int r0 = C.w; r0 < B; r0 += 4 * 4) {
                float v[4][7];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = r0 + 4 * k, rc = row < B ? row : B - 1;
                    g_cf src = ring + (size_t)tab[rc] * stride + src_off;
#pragma unroll
                    for (int c = 0; c < 7; ++c) {
                        const int cc = C.l + 64 * c;
                        v[k][c] = src[cc < n ? cc : n - 1];
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = r0 + 4 * k;
#pragma unroll
                    for (int c = 0; c < 7; ++c)
                        if (row < B && C.l + 64 * c < n) dst[(size_t)row * pitch + C.l + 64 * c] = v[k][c];
                }
            }
        }
    }

    // ---- layers 2 and 3 of one head -> LDS images (linear copies, in the open), biases, log_std.  th = the net's block,
    // L = the head's three LayerDescs, nt3 = head tiles
    __device__ __forceinline__ void stage23(g_cf th, const LayerDesc* L, int nt3, int extra_off, int extra_n) const {
        const int tid = C.tid;
        f32x4 t2[16], t3[4];
        g_cf w2 = th + L[1].w_off, w3 = th + L[2].w_off;
#pragma unroll
        for (int j = 0; j < 16; ++j) t2[j] = ld4(w2 + 4 * (tid + 256 * j));
#pragma unroll
        for (int j = 0; j < 4; ++j) t3[j] = j < 2 * nt3 ? ld4(w3 + 4 * (tid + 256 * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
        float bb1 = 0.f, bb2 = 0.f, bb3 = 0.f, lsv = 0.f;
        if (tid < 128) { bb1 = th[L[0].b_off + tid]; bb2 = th[L[1].b_off + tid]; }
        if (tid < 32) {
            if (tid < L[2].n_pad) bb3 = th[L[2].b_off + tid];
            if (tid < extra_n) lsv = th[extra_off + tid];
        }
        lds_barrier();                                                 // every wave is done with the previous images
#pragma unroll
        for (int j = 0; j < 16; ++j) st4(C.S.w2 + 4 * (tid + 256 * j), t2[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(C.S.w3 + 4 * (tid + 256 * j), t3[j]);
        if (tid < 128) { C.S.b1[tid] = bb1; C.S.b2[tid] = bb2; }
        if (tid < 32) { C.S.b3[tid] = bb3; C.S.ls[tid] = lsv; }
        lds_barrier();
    }

    // ---- the row operand of k-block kb: columns 16 kb + 4 q .. + 3 of this lane's row, one dwordx4.  A row source is 16-byte
    // aligned and readable for 16 KB1 columns (a replay record from its start, a scratch row): columns past the layer's real
    // inputs hold some finite value and meet zero weights (the padding invariant of W1's image).  The first build read four
    // dwords per lane and k-block, each wave-instruction touching 16 rows x 4 x 4 B: ~65 cycles of the CU's address unit apiece
    // (profiles/r01/loadpat_bench.txt), 4 waves x 72 of them per slice = more than the slice's MFMA time, and the waves stall
    // at the issue of the burst.
    __device__ __forceinline__ f32x4 xfrag(g_cf row, int kb) const { return ld4(row + 16 * kb + 4 * C.q); }

    // ---- first layer of T x 16 rows per wave as a sweep over the K-slices of W1 (w1 = the layer's weight block in image order,
    // tile (ot, kb) at (ot * KB1 + kb) * 256 floats) -> h1 = relu(W1 x + b1).  Both operand streams run one SLICE ahead of the
    // MFMAs: while slice s is multiplied, the weight tiles of slice s + 1 travel global -> registers (committed to the other LDS
    // buffer at the top of the next trip) and the row fragments of its four k-blocks global -> registers.
    // Two rules keep that overlap alive in the ISA (both learnt from the first builds' s_waitcnt placement): (1) the loads are
    // pinned in front of the slice's MFMAs (hipcc sinks each next to its first use); (2) NO load of the steady-state loop sits
    // inside a conditional — gfx9 has one in-order counter for all vector memory loads, a wait for an older load is expressed
    // as "all but the N youngest", and the compiler can only count loads that are issued unconditionally: behind a guarded
    // prefetch every wait degenerates to vmcnt(0) and the slice waits for the prefetch it has just issued.  So the full slices
    // run unguarded, the prefetch indices are clamped instead of guarded (the last trip re-reads a slice it does not need), and
    // only the MFMAs of the partial tail slice are guarded.
    template <int T>
    __device__ __forceinline__ void l1_sweep(f32x4 (&h1)[T][kHT], const g_cf (&rp)[T], g_cf w1, int KB1) const {
        const auto K_ = C.lanes();
        const int l = C.l, w = C.w, q = K_.q, fslot = K_.fslot;
        const int nfull = KB1 / kWideSliceKB, tail = KB1 - nfull * kWideSliceKB, last = KB1 - 1;
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            const f32x4 bf = ld4((lds_cf)(C.S.b1 + ot * 16 + 4 * q));
#pragma unroll
            for (int t = 0; t < T; ++t) h1[t][ot] = bf;
        }
        // slice s: wave w moves k-block 4 s + w — its eight output tiles, 1 KB contiguous each
        f32x4 R[kHT], xn[kWideSliceKB][T], xc[kWideSliceKB][T];
        auto fetch = [&](int s) {
            int kb = kWideSliceKB * s + w;
            kb = kb < last ? kb : last;
#pragma unroll
            for (int j = 0; j < kHT; ++j) R[j] = ld4(w1 + ((size_t)(j * KB1 + kb) * 256 + 4 * l));
        };
        auto commit = [&](int s) {
            lds_f buf = u + (s & 1) * kWideSlice;
#pragma unroll
            for (int j = 0; j < kHT; ++j) st4(buf + (j * kWideSliceKB + w) * 256 + 4 * l, R[j]);
        };
        auto xfetch = [&](int s) {
#pragma unroll
            for (int kbl = 0; kbl < kWideSliceKB; ++kbl) {
                int kb = kWideSliceKB * s + kbl;
                kb = kb < last ? kb : last;
#pragma unroll
                for (int t = 0; t < T; ++t) xn[kbl][t] = xfrag(rp[t], kb);
            }
        };
        auto kblock = [&](lds_cf buf, int kbl_rt, const f32x4 (&xk)[T]) {
            f32x4 wf[kHT];
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot) wf[ot] = ld4(buf + (ot * kWideSliceKB + kbl_rt) * 256 + fslot);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                    for (int t = 0; t < T; ++t) h1[t][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], xk[t][e], h1[t][ot], 0, 0, 0);
        };
        fetch(0);
        xfetch(0);
        lds_barrier();                                                 // the union's previous readers (last sweep's slice, exchanges) are done
        for (int s = 0; s < nfull; ++s) {
            commit(s);
            lds_barrier();                                             // slice s visible; every wave is past slice s - 1
#pragma unroll
            for (int kbl = 0; kbl < kWideSliceKB; ++kbl)
#pragma unroll
                for (int t = 0; t < T; ++t) xc[kbl][t] = xn[kbl][t];
            fetch(s + 1);
            xfetch(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            lds_cf buf = u + (s & 1) * kWideSlice;
#pragma unroll
            for (int kbl = 0; kbl < kWideSliceKB; ++kbl) kblock(buf, kbl, xc[kbl]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tail > 0) {
            commit(nfull);                                             // (k-blocks past the last: duplicates nobody multiplies)
            lds_barrier();
            lds_cf buf = u + (nfull & 1) * kWideSlice;
#pragma unroll
            for (int kbl = 0; kbl < kWideSliceKB - 1; ++kbl)
                if (kbl < tail) kblock(buf, kbl, xn[kbl]);
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[t][ot][r] = fmaxf(h1[t][ot][r], 0.f);
    }

    // ---- layers 2 and 3 of T tiles: h1 -> h2 = relu(W2 h1 + b2), z = W3 h2 + b3.  VH: the head has hn <= 4 outputs and runs as
    // dot products (z[t][0][o] on every lane group, chain_net.hpp: head_valu); else NT3 MFMA tiles, z[t][o3][r] = output
    // 16 o3 + 4 q + r of this lane's row
    // (h1 may be a window [T0, T0 + T) of a sweep's TT tiles)
    template <int T, int NT3, bool VH, int TT = T, int T0 = 0>
    __device__ __forceinline__ void l23(const f32x4 (&h1)[TT][kHT], f32x4 (&h2)[T][kHT], f32x4 (&z)[T][NT3], int hn) const {
        const auto K_ = C.lanes();
        const int q = K_.q, fslot = K_.fslot;
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            const f32x4 bb = ld4((lds_cf)(C.S.b2 + ot * 16 + 4 * q));
#pragma unroll
            for (int t = 0; t < T; ++t) h2[t][ot] = bb;
        }
        static_for<0, kHT>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            f32x4 wf[kHT];
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot) wf[ot] = ld4((lds_cf)(C.S.w2 + (ot * kHT + kb) * 256 + fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                    for (int t = 0; t < T; ++t) h2[t][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], h1[T0 + t][kb][e], h2[t][ot], 0, 0, 0);
        });
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[t][ot][r] = fmaxf(h2[t][ot][r], 0.f);
        if constexpr (VH) {
            static_assert(NT3 == 1, "a dot-product head is one tile");
            f32x4 z1[T];
            C.head_valu<T>(h2, z1, hn);
#pragma unroll
            for (int t = 0; t < T; ++t) z[t][0] = z1[t];
        } else {
            head_tiles<T, NT3>(h2, z);
        }
    }
    template <int T, int NT3>
    __device__ __forceinline__ void head_tiles(const f32x4 (&h2)[T][kHT], f32x4 (&z)[T][NT3]) const {
#pragma unroll
        for (int o3 = 0; o3 < NT3; ++o3) {
            const f32x4 b3 = ld4((lds_cf)(C.S.b3 + 16 * o3 + 4 * C.q));
#pragma unroll
            for (int t = 0; t < T; ++t) z[t][o3] = b3;
        }
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb)
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                const f32x4 wf = ld4((lds_cf)(C.S.w3 + (o3 * kHT + kb) * 256 + C.fslot));
#pragma unroll
                for (int t = 0; t < T; ++t) z[t][o3] = mfma4(z[t][o3], wf, h2[t][kb]);
            }
    }
    // dH2 = W3^T dz through the ReLU of h2 for a head of NT3 tiles (transposed fragment reads, as ChainNet::delta2)
    template <int NT3>
    __device__ __forceinline__ void delta2_tiles(const f32x4 (&dz)[NT3], const f32x4 (&h2)[kHT], f32x4 (&d2)[kHT]) const {
        const auto K_ = C.lanes();
        const int q = K_.q, i16 = K_.i16, tslot = K_.tslot;
#pragma unroll
        for (int it = 0; it < kHT; ++it) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                f32x4 wa;
#pragma unroll
                for (int e = 0; e < 4; ++e) wa[e] = C.S.w3[(o3 * kHT + it) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
                acc = mfma4(acc, wa, dz[o3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[it][r] = h2[it][r] > 0.f ? acc[r] : 0.f;
        }
    }

    // dH1 = W2^T dz2 through the ReLU of h1 for T tiles at once (ChainNet::delta1 with every transposed fragment feeding T MFMAs;
    // h1 may be a window [T0, T0 + T) of TT tiles)
    template <int T, int TT, int T0>
    __device__ __forceinline__ void delta1_t(const f32x4 (&d2)[T][kHT], const f32x4 (&h1)[TT][kHT], f32x4 (&d1)[T][kHT]) const {
        const auto K_ = C.lanes();
        const int q = K_.q, i16 = K_.i16, tslot = K_.tslot;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int it = 0; it < kHT; ++it) d1[t][it] = f32x4{0.f, 0.f, 0.f, 0.f};
        float wa[2][kHT];
        auto fetch = [&](int ob, int e, float (&dst)[kHT]) {
#pragma unroll
            for (int it = 0; it < kHT; ++it) dst[it] = C.S.w2[(ob * kHT + it) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
        };
        fetch(0, 0, wa[0]);
        static_for<0, 4 * kHT>([&](auto sc) {
            constexpr int s_ = decltype(sc)::value, ob = s_ >> 2, e = s_ & 3;
            if constexpr (s_ + 1 < 4 * kHT) fetch((s_ + 1) >> 2, (s_ + 1) & 3, wa[(s_ + 1) & 1]);
#pragma unroll
            for (int it = 0; it < kHT; ++it)
#pragma unroll
                for (int t = 0; t < T; ++t) d1[t][it] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_ & 1][it], d2[t][ob][e], d1[t][it], 0, 0, 0);
        });
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int it = 0; it < kHT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[t][it][r] = h1[T0 + t][it][r] > 0.f ? d1[t][it][r] : 0.f;
    }

    template <int NT3>
    __device__ __forceinline__ void grad_zero(WideGrad<NT3>& g) const {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) g.g3[o3][x] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.gb1[x] = 0.f; g.gb2[x] = 0.f;
        }
#pragma unroll
        for (int o3 = 0; o3 < NT3; ++o3) g.gb3[o3] = 0.f;
    }

    // ---- backward of one 64-row chunk (16 rows per wave): head and layer-2 gradients into the owners' accumulators (exchanges
    // through ea / eb as ChainNet::backward), the first-layer deltas d1 -> `dz1_dst` (8192 floats, exchange-image order:
    // tile (out tile, 16-row block) at (ot * 4 + bb) * 256) for the dW1 pass; their bias sums into gb1.
    // VH: dot-product head of hn outputs (dz[0] on lane group 0)
    template <int NT3, bool VH>
    __device__ __forceinline__ void backward(WideGrad<NT3>& g, const f32x4 (&h1)[kHT], const f32x4 (&h2)[kHT], const f32x4 (&dz)[NT3], int hn,
                                             g_f dz1_dst) const {
        const int w = C.w, tid = C.tid;
        lds_barrier();                                                 // the union's previous readers are done
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) C.put_tile(C.S.ea, ft, h2[ft]);
#pragma unroll
        for (int o3 = 0; o3 < NT3; ++o3) C.put_tile(C.S.eb, o3, dz[o3]);
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            f32x4 bf[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) bf[x] = C.get_frag(C.S.ea, 2 * w + x, bb);
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                const f32x4 af = C.get_frag(C.S.eb, o3, bb);
                if (w == 0) g.gb3[o3] += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
                for (int x = 0; x < 2; ++x) g.g3[o3][x] = mfma4(g.g3[o3][x], bf[x], af);
            }
        }
        f32x4 d2[kHT];
        if constexpr (VH) C.delta2_valu(dz[0], h2, d2, hn); else delta2_tiles<NT3>(dz, h2, d2);
        lds_barrier();
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) { C.put_tile(C.S.ea, ft, h1[ft]); C.put_tile(C.S.eb, ft, d2[ft]); }
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            f32x4 af[2], bf[kHT];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                af[x] = C.get_frag(C.S.eb, 2 * w + x, bb);
                g.gb2[x] += (af[x][0] + af[x][1]) + (af[x][2] + af[x][3]);
            }
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) bf[kt] = C.get_frag(C.S.ea, kt, bb);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = mfma4(g.g2[x][kt], bf[kt], af[x]);
        }
        f32x4 d1[kHT];
        C.delta1(d2, h1, d1);
        lds_barrier();
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) C.put_tile(C.S.eb, ft, d1[ft]);
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const f32x4 af = C.get_frag(C.S.eb, 2 * w + x, bb);
                g.gb1[x] += (af[0] + af[1]) + (af[2] + af[3]);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) st4(dz1_dst + 4 * (tid + 256 * j), ld4((lds_cf)(C.S.eb + 4 * (tid + 256 * j))));
    }
    template <int NT3>
    __device__ __forceinline__ void grad_finish(WideGrad<NT3>& g) const {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            g.gb1[x] += lane_xor<16>(g.gb1[x]); g.gb1[x] += lane_xor<32>(g.gb1[x]);
            g.gb2[x] += lane_xor<16>(g.gb2[x]); g.gb2[x] += lane_xor<32>(g.gb2[x]);
        }
#pragma unroll
        for (int o3 = 0; o3 < NT3; ++o3) { g.gb3[o3] += lane_xor<16>(g.gb3[o3]); g.gb3[o3] += lane_xor<32>(g.gb3[o3]); }
    }

    // ---- dW1^T of one head over the whole batch, then its tiles -> grad; returns this lane's share of the squared norm.
    // Wave w owns out tiles 4 (w & 1) + y and the k-tiles (w >> 1) + 2 j: acc[j][y], NKT >= ceil(KB1 / 2) of them per y.
    // dz1 = the scratch images the chunk loop left (one per 64-row chunk), rowptr(row) = that batch row's XT contiguous columns,
    // G + L[0].w_off = the layer's block of the grad array.
    // One trip = one 16-row block: NKT x 4 dword loads (the transposed row fragments of this wave's k-tiles) + 4 dwordx4 (the
    // delta fragments of its out tiles) for NKT x 16 MFMAs.  The loads of block it + 1 are issued in front of the MFMAs of block
    // it (two operand sets, alternating by name), the row pointers one block further ahead, and — as in l1_sweep — every load is
    // unconditional: indices are clamped (a k-tile past the last re-reads the last one into accumulators nobody stores; the trip
    // behind the last block re-reads it) instead of guarded.  The first build left the placement to the compiler: every k-tile's
    // loads sat in front of its MFMAs, 13 exposed HBM round trips per block, 7x the MFMA time of the pass.
    template <int NKT> struct Dw1Ops { f32x4 a[NKT], b[4]; };
    template <int NKT, class RowF>
    __device__ __forceinline__ float dw1_grad(g_f G, const LayerDesc* L, g_cf dz1, int nchunks, int B, int KB1, int XT, RowF rowptr) const {
        const auto K_ = C.lanes();
        const int w = C.w, q = K_.q, i16 = K_.i16, fslot = K_.fslot;
        const int oh = w & 1, kt0 = w >> 1, nkt = (KB1 - kt0 + 1) >> 1;
        f32x4 acc[NKT][4];
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[j][y] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nit = nchunks * 4;                                   // it = 4 * chunk + 16-row block
        int fcol[NKT];                                                 // this lane's column of k-tile j (clamped into the row)
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            const int kt = kt0 + 2 * j < KB1 ? kt0 + 2 * j : KB1 - 1;
            const int f = 16 * kt + i16;
            fcol[j] = f < XT ? f : XT - 1;
        }
        // (rowptr reads the batch's row index from an LDS table: a GLOBAL index load here — one vmcnt counter for everything —
        // made every trip wait for the operand loads it had just issued)
        auto rows_of = [&](int it, g_cf (&rp)[4]) {                    // rows 16 it + 4 q + e of the batch (clamped: their deltas are zero)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * (it < nit ? it : nit - 1) + 4 * q + e;
                rp[e] = rowptr(row < B ? row : B - 1);
            }
        };
        auto load_ops = [&](int it_, const g_cf (&rp)[4], Dw1Ops<NKT>& o) {
            const int it = it_ < nit ? it_ : nit - 1;
            g_cf img = dz1 + (size_t)(it >> 2) * 8192 + (it & 3) * 256 + fslot;
#pragma unroll
            for (int y = 0; y < 4; ++y) o.b[y] = ld4(img + (4 * oh + y) * 4 * 256);
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) o.a[j][e] = rp[e][fcol[j]];
        };
        auto mma = [&](const Dw1Ops<NKT>& o) {
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[j][y] = mfma4(acc[j][y], o.a[j], o.b[y]);
        };
        g_cf rp0[4], rp1[4];
        Dw1Ops<NKT> A, Bo;
        rows_of(0, rp0);
        rows_of(1, rp1);
        load_ops(0, rp0, A);
        for (int it = 0; it < nit; it += 2) {                          // (nit is a multiple of 4)
            load_ops(it + 1, rp1, Bo);
            rows_of(it + 2, rp0);
            __builtin_amdgcn_sched_barrier(0);
            mma(A);
            __builtin_amdgcn_sched_barrier(0);
            load_ops(it + 2, rp0, A);
            rows_of(it + 3, rp1);
            __builtin_amdgcn_sched_barrier(0);
            mma(Bo);
            __builtin_amdgcn_sched_barrier(0);
        }
        // a lane's registers of tile (., kt) are input columns 16 kt + 4 q + r: the padding columns (>= XT) keep a zero gradient
        // (their row fragments were clamped, not zeroed, at the loads)
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            if (j < nkt) {
                const int kt = kt0 + 2 * j;
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    f32x4 v = acc[j][y];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (16 * kt + 4 * q + r) < XT ? v[r] : 0.f;
                    st4(G + L[0].w_off + ((size_t)((4 * oh + y) * KB1 + kt) * 256 + fslot), v);
                    ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
        }
        return ss;
    }
    // (one instantiation per size class of the first layer: a 13-tile build would make MADDPG's 5-block critics load and
    // multiply 13 k-tiles per wave where they own 3)
    template <class RowF>
    __device__ __forceinline__ float dw1_grad_any(g_f G, const LayerDesc* L, g_cf dz1, int nchunks, int B, int KB1, int XT, RowF rowptr) const {
        if (KB1 <= 2) return dw1_grad<1>(G, L, dz1, nchunks, B, KB1, XT, rowptr);
        if (KB1 <= 6) return dw1_grad<3>(G, L, dz1, nchunks, B, KB1, XT, rowptr);
        if (KB1 <= 14) return dw1_grad<7>(G, L, dz1, nchunks, B, KB1, XT, rowptr);
        return dw1_grad<kWideMaxKT>(G, L, dz1, nchunks, B, KB1, XT, rowptr);
    }

    // ---- one head's layer-2 / head gradients and biases -> the engine's grad array (image order, G = the net's block; L = the
    // head's LayerDescs), BEFORE the dW1 pass (its accumulators want the registers); returns this lane's share of the squared norm
    template <int NT3>
    __device__ __forceinline__ float grad_store_23(g_f G, const LayerDesc* L, const WideGrad<NT3>& g) const {
        const auto K_ = C.lanes();
        const int w = C.w, q = K_.q, i16 = K_.i16, fslot = K_.fslot;
        float ss = 0.f;
        auto sq = [](const f32x4& v) { return (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]); };
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) {
                st4(G + L[1].w_off + ((2 * w + x) * kHT + kt) * 256 + fslot, g.g2[x][kt]);
                ss += sq(g.g2[x][kt]);
            }
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) {
                st4(G + L[2].w_off + (o3 * kHT + 2 * w + x) * 256 + fslot, g.g3[o3][x]);
                ss += sq(g.g3[o3][x]);
            }
            if (q == 0) {
                G[L[0].b_off + (2 * w + x) * 16 + i16] = g.gb1[x];
                G[L[1].b_off + (2 * w + x) * 16 + i16] = g.gb2[x];
                ss += g.gb1[x] * g.gb1[x] + g.gb2[x] * g.gb2[x];
            }
        }
        if (w == 0 && q == 0) {
#pragma unroll
            for (int o3 = 0; o3 < NT3; ++o3) { G[L[2].b_off + 16 * o3 + i16] = g.gb3[o3]; ss += g.gb3[o3] * g.gb3[o3]; }
        }
        return ss;
    }

    // ---- clip + Adam (+ soft target update) of a whole net, streamed: n4 = NetDesc::size / 4.  The gradient was written by this
    // workgroup's own waves (grad_store); the caller has synchronised (vmcnt + barrier) in between.
    template <bool SOFT>
    __device__ __forceinline__ void adam_stream(g_f th, g_f mA, g_f vA, g_f tg, g_cf gr, int n4, const AdamCoef& c) const {
        for (int i0 = C.tid; i0 < n4; i0 += 4 * 256) {
            f32x4 G4[4], T4[4], M4[4], V4[4], X4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 256 * k;
                if (i < n4) {
                    G4[k] = ld4(gr + 4 * (size_t)i); T4[k] = ld4((g_cf)(th + 4 * (size_t)i)); M4[k] = ld4((g_cf)(mA + 4 * (size_t)i));
                    V4[k] = ld4((g_cf)(vA + 4 * (size_t)i));
                    if constexpr (SOFT) X4[k] = ld4((g_cf)(tg + 4 * (size_t)i));
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 256 * k;
                if (i < n4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gi = G4[k][r] * c.coef;
                        gi += c.wd * T4[k][r];
                        float m1 = M4[k][r], v1 = V4[k][r];
                        T4[k][r] = adam_elem(T4[k][r], gi, m1, v1, c.w1, c.w2, c.beta2, c.inv_bc2s, c.eps, c.step);
                        M4[k][r] = m1; V4[k][r] = v1;
                        if constexpr (SOFT) X4[k][r] = X4[k][r] * c.tk + T4[k][r] * c.tau;
                    }
                    st4(th + 4 * (size_t)i, T4[k]); st4(mA + 4 * (size_t)i, M4[k]); st4(vA + 4 * (size_t)i, V4[k]);
                    if constexpr (SOFT) st4(tg + 4 * (size_t)i, X4[k]);
                }
            }
        }
    }
};

}  // namespace frl
