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

namespace frl {

constexpr int kChainBatch = 256;          // rows of per-row staging (actions, targets, ...) the carve provides

struct ChainLds {
    lds_f w1, w2, w3, b1, b2, b3, ls, ea, eb, ab, yb, q1, lpn, red;
};
constexpr int chain_lds_floats() { return 8 * 256 + 64 * 256 + 8 * 256 + 2 * 8192 + 128 + 128 + 16 + 16 + kChainBatch * 4 + 3 * kChainBatch + 64; }

// the weight-gradient accumulators one lane owns for one 3-layer head, TRANSPOSED (MFMA D layout of dW^T: in = 16*kt + 4q + r,
// out = 16*ot + i16 — the lane's four registers are the 16-byte slot (q, f = i16) of image tile (ot, kt)):
// layer 2: ot in {2w, 2w+1} x kt 0..7; layer 1 (one 16-wide input block): ot in {2w, 2w+1}; head: kt in {2w, 2w+1}
struct HeadGrad {
    f32x4 g2[2][kHT], g1[2], g3[2];
    float gb1[2], gb2[2], gb3;
};

struct AdamCoef { float coef, step, inv_bc2s, w1, w2, beta2, eps, wd, tk, tau; bool soft; };

struct ChainNet {
    ChainLds S;
    int tid, l, w, i16, q, fslot, tslot;

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
        S.ab = p; p += kChainBatch * 4;
        S.yb = p; p += kChainBatch;
        S.q1 = p; p += kChainBatch;
        S.lpn = p; p += kChainBatch;
        S.red = p; p += 64;
        tid = threadIdx.x; l = tid & 63; w = __builtin_amdgcn_readfirstlane(tid >> 6); i16 = l & 15; q = l >> 4;
        fslot = (q * 16 + (i16 ^ q)) << 2;                             // forward / exchange fragment read (16 B)
        tslot = (((i16 >> 2) * 16) << 2) + (i16 & 3);                  // transposed read / owner write: + ((f ^ (i16 >> 2)) << 2)
    }

    // ---- one net's three layers -> LDS images: the HBM block is already in image order (NetDesc::frag), a linear copy
    __device__ __forceinline__ void stage(g_cf th, const NetDesc& N, int l0) const {
        const LayerDesc &L1 = N.L[l0], &L2 = N.L[l0 + 1], &L3 = N.L[l0 + 2];
        lds_barrier();                                                 // every wave is done with the previous images
        f32x4 t2[16], t1[2], t3[2];                                    // all loads of the net in flight before the first store
#pragma unroll
        for (int j = 0; j < 16; ++j) t2[j] = ld4(th + L2.w_off + 4 * (tid + 256 * j));
#pragma unroll
        for (int j = 0; j < 2; ++j) { t1[j] = ld4(th + L1.w_off + 4 * (tid + 256 * j)); t3[j] = ld4(th + L3.w_off + 4 * (tid + 256 * j)); }
        float bb1 = 0.f, bb2 = 0.f, bb3 = 0.f, lsv = 0.f;
        if (tid < kHid) { bb1 = th[L1.b_off + tid]; bb2 = th[L2.b_off + tid]; }
        if (tid < 16) { bb3 = th[L3.b_off + tid]; lsv = (N.extra_n > 0 && tid < N.extra_n) ? th[N.extra_off + tid] : 0.f; }
#pragma unroll
        for (int j = 0; j < 16; ++j) st4(S.w2 + 4 * (tid + 256 * j), t2[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) { st4(S.w1 + 4 * (tid + 256 * j), t1[j]); st4(S.w3 + 4 * (tid + 256 * j), t3[j]); }
        if (tid < kHid) { S.b1[tid] = bb1; S.b2[tid] = bb2; }
        if (tid < 16) { S.b3[tid] = bb3; S.ls[tid] = lsv; }
        lds_barrier();
    }

    // ---- the chained forward of T x 16 rows per wave: x (B operand of layer 1) -> h1, h2 -> head tile z.  A tile's 32 MFMAs
    // per layer are ONE dependent accumulator chain: the 8 T chains of a layer run side by side (k-block outer, k-step
    // middle, tile inner), so that consecutive MFMAs never wait for each other's result.
    template <int T>
    __device__ __forceinline__ void forward(const f32x4 (&xb)[T], f32x4 (&h1)[T][kHT], f32x4 (&h2)[T][kHT], f32x4 (&z)[T]) const {
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            const f32x4 wf = ld4((lds_cf)(S.w1 + ot * 256 + fslot)), bb = ld4((lds_cf)(S.b1 + ot * 16 + 4 * q));
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 acc = mfma4(bb, wf, xb[t]);
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[t][ot][r] = fmaxf(acc[r], 0.f);
            }
        }
#pragma unroll
        for (int ot = 0; ot < kHT; ++ot) {
            const f32x4 bb = ld4((lds_cf)(S.b2 + ot * 16 + 4 * q));
#pragma unroll
            for (int t = 0; t < T; ++t) h2[t][ot] = bb;
        }
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {
            f32x4 wf[kHT];
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot) wf[ot] = ld4((lds_cf)(S.w2 + (ot * kHT + kb) * 256 + fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                    for (int t = 0; t < T; ++t) h2[t][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][e], h1[t][kb][e], h2[t][ot], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int ot = 0; ot < kHT; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[t][ot][r] = fmaxf(h2[t][ot][r], 0.f);
        const f32x4 b3 = ld4((lds_cf)(S.b3 + 4 * q));
#pragma unroll
        for (int t = 0; t < T; ++t) z[t] = b3;
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {
            const f32x4 wf = ld4((lds_cf)(S.w3 + kb * 256 + fslot));
#pragma unroll
            for (int t = 0; t < T; ++t) z[t] = mfma4(z[t], wf, h2[t][kb]);
        }
    }

    // dH2 = W3^T dz through the ReLU of h2 (A = W3^T: transposed fragment reads)
    __device__ __forceinline__ void delta2(const f32x4& dz, const f32x4 (&h2)[kHT], f32x4 (&d2)[kHT]) const {
#pragma unroll
        for (int it = 0; it < kHT; ++it) {
            f32x4 wa;
#pragma unroll
            for (int e = 0; e < 4; ++e) wa[e] = S.w3[it * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
            const f32x4 acc = mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, wa, dz);
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[it][r] = h2[it][r] > 0.f ? acc[r] : 0.f;
        }
    }
    // dH1 = W2^T dz2 through the ReLU of h1; eight accumulator chains side by side
    __device__ __forceinline__ void delta1(const f32x4 (&d2)[kHT], const f32x4 (&h1)[kHT], f32x4 (&d1)[kHT]) const {
#pragma unroll
        for (int it = 0; it < kHT; ++it) d1[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ob = 0; ob < kHT; ++ob) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float wa[kHT];
#pragma unroll
                for (int it = 0; it < kHT; ++it) wa[it] = S.w2[(ob * kHT + it) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
#pragma unroll
                for (int it = 0; it < kHT; ++it) d1[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[it], d2[ob][e], d1[it], 0, 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < kHT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[it][r] = h1[it][r] > 0.f ? d1[it][r] : 0.f;
    }
    // dX = W1^T dz1: d loss / d input column 4q + r of this lane's row (no activation in front of the input)
    __device__ __forceinline__ f32x4 delta0(const f32x4 (&d1)[kHT]) const {
        f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ob = 0; ob < kHT; ++ob) {
            f32x4 wa;
#pragma unroll
            for (int e = 0; e < 4; ++e) wa[e] = S.w1[ob * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
            dx = mfma4(dx, wa, d1[ob]);
        }
        return dx;
    }

    __device__ __forceinline__ void put_tile(lds_f E, int ft, const f32x4& t) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) E[(ft * 4 + w) * 256 + tslot + (((4 * q + r) ^ (i16 >> 2)) << 2)] = t[r];
    }
    __device__ __forceinline__ f32x4 get_frag(lds_cf E, int ft, int bb) const { return ld4(E + (ft * 4 + bb) * 256 + fslot); }

    __device__ __forceinline__ void grad_zero(HeadGrad& g) const {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.g1[x] = f32x4{0.f, 0.f, 0.f, 0.f}; g.g3[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            g.gb1[x] = 0.f; g.gb2[x] = 0.f;
        }
        g.gb3 = 0.f;
    }

    // ---- backward of one 64-row chunk (16 rows per wave) into the owners' accumulators: three exchanges through ea / eb
    // (H2 + dz -> head gradient; H1 + dz2 -> layer 2; X + dz1 -> layer 1), the dH chains in between
    __device__ __forceinline__ void backward(HeadGrad& g, const f32x4& xb, const f32x4 (&h1)[kHT], const f32x4 (&h2)[kHT], const f32x4& dz) const {
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
            for (int x = 0; x < 2; ++x) g.g3[x] = mfma4(g.g3[x], get_frag(S.ea, 2 * w + x, bb), af);
        }
        f32x4 d2[kHT];
        delta2(dz, h2, d2);
        lds_barrier();
#pragma unroll
        for (int ft = 0; ft < kHT; ++ft) { put_tile(S.ea, ft, h1[ft]); put_tile(S.eb, ft, d2[ft]); }
        lds_barrier();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            f32x4 af[2], bf[kHT];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                af[x] = get_frag(S.eb, 2 * w + x, bb);
                g.gb2[x] += (af[x][0] + af[x][1]) + (af[x][2] + af[x][3]);
            }
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) bf[kt] = get_frag(S.ea, kt, bb);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt) g.g2[x][kt] = mfma4(g.g2[x][kt], bf[kt], af[x]);
        }
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
            for (int x = 0; x < 2; ++x) {
                const f32x4 af = get_frag(S.eb, 2 * w + x, bb);
                g.gb1[x] += (af[0] + af[1]) + (af[2] + af[3]);
                g.g1[x] = mfma4(g.g1[x], bf, af);
            }
        }
    }

    // after the last chunk: the bias partials of the four lane groups (rows 4q..4q+3 of every 16-row block) added up
    __device__ __forceinline__ void grad_finish(HeadGrad& g) const {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            g.gb1[x] += __shfl_xor(g.gb1[x], 16, 64); g.gb1[x] += __shfl_xor(g.gb1[x], 32, 64);
            g.gb2[x] += __shfl_xor(g.gb2[x], 16, 64); g.gb2[x] += __shfl_xor(g.gb2[x], 32, 64);
        }
        g.gb3 += __shfl_xor(g.gb3, 16, 64); g.gb3 += __shfl_xor(g.gb3, 32, 64);
    }
    // this lane's share of the squared gradient norm (bias entries counted once: lanes q == 0 / wave 0)
    __device__ __forceinline__ float grad_sumsq(const HeadGrad& g) const {
        float ss = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
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
    __device__ __forceinline__ AdamIn adam_load(g_f th, g_f mA, g_f vA, g_f tg, const AdamCoef& c, int o) const {
        AdamIn X;
        X.th = ld4((g_cf)(th + o)); X.mm = ld4((g_cf)(mA + o)); X.vv = ld4((g_cf)(vA + o));
        X.tg = c.soft ? ld4((g_cf)(tg + o)) : f32x4{0.f, 0.f, 0.f, 0.f};
        return X;
    }
    __device__ __forceinline__ void adam_apply(g_f th, g_f mA, g_f vA, g_f tg, const AdamCoef& c, int o, const f32x4& gr, const AdamIn& in) const {
        f32x4 t4 = in.th, mm = in.mm, vv = in.vv, tt = in.tg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gi = gr[r] * c.coef;
            if (c.wd != 0.f) gi += c.wd * t4[r];
            float m1 = mm[r], v1 = vv[r];
            t4[r] = adam_elem(t4[r], gi, m1, v1, c.w1, c.w2, c.beta2, c.inv_bc2s, c.eps, c.step);
            mm[r] = m1; vv[r] = v1;
            tt[r] = tt[r] * c.tk + t4[r] * c.tau;
        }
        st4(th + o, t4); st4(mA + o, mm); st4(vA + o, vv);
        if (c.soft) st4(tg + o, tt);
    }
    __device__ __forceinline__ float adam_scalar(g_f th, g_f mA, g_f vA, g_f tg, const AdamCoef& c, int o, float gr) const {
        float t1 = th[o], m1 = mA[o], v1 = vA[o];
        float gi = gr * c.coef;
        if (c.wd != 0.f) gi += c.wd * t1;
        t1 = adam_elem(t1, gi, m1, v1, c.w1, c.w2, c.beta2, c.inv_bc2s, c.eps, c.step);
        th[o] = t1; mA[o] = m1; vA[o] = v1;
        if (c.soft) tg[o] = tg[o] * c.tk + t1 * c.tau;
        return t1;
    }
    __device__ __forceinline__ void adam_head(const HeadGrad& g, const LayerDesc& L1, const LayerDesc& L2, const LayerDesc& L3, g_f th,
                                              g_f mA, g_f vA, g_f tg, const AdamCoef& c, float g_extra, int extra_off, int extra_n) const {
        // ---- the 128 x 128 layer: 16 tiles per lane, in two batches of 8
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            AdamIn in[kHT];
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) in[kt] = adam_load(th, mA, vA, tg, c, L2.w_off + ((2 * w + x) * kHT + kt) * 256 + fslot);
#pragma unroll
            for (int kt = 0; kt < kHT; ++kt) adam_apply(th, mA, vA, tg, c, L2.w_off + ((2 * w + x) * kHT + kt) * 256 + fslot, g.g2[x][kt], in[kt]);
        }
        // ---- first layer (tiles ot = 2w + x), head (tiles kb = 2w + x)
        {
            AdamIn in[4];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                in[x] = adam_load(th, mA, vA, tg, c, L1.w_off + (2 * w + x) * 256 + fslot);
                in[2 + x] = adam_load(th, mA, vA, tg, c, L3.w_off + (2 * w + x) * 256 + fslot);
            }
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                adam_apply(th, mA, vA, tg, c, L1.w_off + (2 * w + x) * 256 + fslot, g.g1[x], in[x]);
                adam_apply(th, mA, vA, tg, c, L3.w_off + (2 * w + x) * 256 + fslot, g.g3[x], in[2 + x]);
            }
        }
        // ---- biases (every lane group holds the full sums after grad_finish: group q == 0 writes) [, log_std]
        if (q == 0) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                adam_scalar(th, mA, vA, tg, c, L1.b_off + (2 * w + x) * 16 + i16, g.gb1[x]);
                adam_scalar(th, mA, vA, tg, c, L2.b_off + (2 * w + x) * 16 + i16, g.gb2[x]);
            }
            if (w == 0) adam_scalar(th, mA, vA, tg, c, L3.b_off + i16, g.gb3);
            if (w == 1 && i16 < extra_n) adam_scalar(th, mA, vA, tg, c, extra_off + i16, g_extra);
        }
    }
};

}  // namespace frl
