// ONE learner's update on SIXTEEN workgroups (kernels_solo.hip): the latency design for the reference's own use case — a single
// `TD3 / DDPG / SAC.learn()` per env step (TD3_file/TD3.py:403-450, SAC_file/SAC.py:519-576) — where the population kernels have
// nothing to batch.  The register-chained kernels (chain_net.hpp) give a whole learner to one workgroup: 82 MFLOP on one CU is
// 133 us at the MFMA peak.  The row-chunk kernels (net.hpp) split the batch over eight 32-row workgroups, but walk 23
// barrier-delimited phases per chunk, each behind a dependent L2 round trip for its weight block (55 us), and hand their slabs to a
// one-workgroup reduce + Adam launch (25 us).
//
// Here a 256-row batch is sixteen 16-row tiles, one workgroup each, and inside a workgroup the four waves split every 128-wide
// layer's OUTPUT features (wave w owns output tiles 2w, 2w + 1):
//   * weights: the net's fragment-order LDS images of chain_net.hpp (NetDesc::frag = 1 in HBM: staging is a linear copy),
//     fetched global -> registers one pass AHEAD (stage_fetch) and committed when the pass in front is done;
//   * forward of a layer: 2 x 8 k-blocks of MFMAs per wave on the row tile's activations, which the waves exchange through LDS
//     in the MFMA D layout (one ds_write_b128 of each own tile, one barrier, eight ds_read_b128: the D tile of layer L IS the B
//     operand of layer L + 1, chain.hpp) — two barriers per three-layer pass instead of the row-chunk kernels' five phases;
//   * backward: weight gradients of the wave's own output tiles by MFMAs that contract over the tile's 16 rows (operands
//     transposed through LDS, wave-local for the deltas), dH through the transposed reads of the W2 image;
//   * the sixteen partial gradients go to per-workgroup slabs in image order; behind a GRID barrier (one atomic counter per
//     learner, monotonic across launches) every workgroup sums ONE SIXTEENTH of the net over the slabs in workgroup order
//     (bitwise deterministic), the partial squared norms meet through sixteen mailboxes, and clip + Adam + soft update run on
//     the same sixteenth — torch's single-tensor Adam with exact sqrt / division, as the row-chunk family's adam_kernel.
// Shape: chained_shape() (single agent, hidden 128 ReLU, obs + act <= 16 columns, act <= 4, batch <= 256), up to kSoloMaxP learners
// (all their workgroups must be resident at once: the grid barrier spins).
#pragma once
#include "chain_net.hpp"
#include "update_common.hpp"

namespace frl {

// (kSoloWG, kSoloMaxP, solo_lds_floats(): frl_desc.h — the host sizes the launch from them)

// (SoloArgs: kernels.h)

// Developer instrument (tools/solo_timing.py; kernels_solo.hip compiled with -DFRL_SOLO_TIMING): thread 0 of every workgroup leaves
// wall-clock stamps (100 MHz ticks since its start) in the tail of its `part` row
constexpr int kSoloPart = 32;            // floats per workgroup in SoloArgs::part: 0 loss / Q sum, 1 log-pi sum, 2-3 the norm mailbox {partial, epoch}, 8.. stamps
#ifdef FRL_SOLO_TIMING
#define SOLO_T0() const unsigned long long solo_t0_ = wall_clock64()
#define SOLO_T(slot) do { if (threadIdx.x == 0) part[b * kSoloPart + 8 + (slot)] = (float)(wall_clock64() - solo_t0_); } while (0)
#define SOLO_TARG , solo_t0_
#define SOLO_TARGP , part, solo_t0_
#else
#define SOLO_T0() do {} while (0)
#define SOLO_T(slot) do {} while (0)
#define SOLO_TARG
#define SOLO_TARGP
#endif

// (slab stores are plain stores.  Measured: write-through ones — global_store_dwordx4 ... sc0 sc1, so that the barrier's release fence
// finds nothing dirty in this XCD's L2 — take 1.4 us off grid barrier 1 and put 1.1 us on the backward that issues them; `sc1` alone,
// with the one-lane fences: nothing for one learner, -3 % / -6 % at 8 / 16 — and wrong numbers at sixteen learners: inline-asm stores
// are outside hipcc's wait-count bookkeeping, the fragile kind of fast.)
__device__ __forceinline__ void st4_slab(g_f p, const f32x4& v) { st4(p, v); }

struct SoloNet {
    ChainNet C;             // images (S.w1 / w2 / w3 / b1 / b2 / b3 / ls), lane constants, stage_fetch / stage_commit, delta0
    lds_f ea, eb;           // activation / delta exchange, MFMA D layout: tile ft at ft * 256 + 4 * lane
    lds_f th1;              // h1 of the row tile, TRANSPOSED fragment image (all 8 feature tiles): operand of dW2
    lds_f td;               // a delta's own tiles, transposed (wave-local): operand of dW2 / dW1
    lds_f tx;               // the input rows, transposed (one tile)
    lds_f red;

    __device__ __forceinline__ void init(float* smem) {
        lds_f p = (lds_f)smem;
        C.S.w1 = p; p += kHT * 256;
        C.S.w2 = p; p += kHT * kHT * 256;
        C.S.w3 = p; p += kHT * 256;
        C.S.b1 = p; p += kHid;
        C.S.b2 = p; p += kHid;
        C.S.b3 = p; p += 16;
        C.S.ls = p; p += 16;
        ea = p; p += kHT * 256;
        eb = p; p += kHT * 256;
        th1 = p; p += kHT * 256;
        td = p; p += kHT * 256;
        tx = p; p += 256;
        red = p; p += 128;
        C.S.ea = ea; C.S.eb = eb; C.S.ab = ea; C.S.yb = ea; C.S.q1 = ea; C.S.lpn = ea; C.S.red = red;
        C.init_lanes();
    }

    // element (feature 4q + r of tile ft, row i16) of a D-layout tile -> its place in a transposed fragment image (chain_net.hpp: put_tile
    // with one row block): get_t() then returns lane (i16', q') the four values (feature i16', rows 4q' .. 4q' + 3)
    __device__ __forceinline__ void put_t(lds_f E, int ft, const f32x4& t) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) E[ft * 256 + C.tslot + (((4 * C.q + r) ^ (C.i16 >> 2)) << 2)] = t[r];
    }
    __device__ __forceinline__ f32x4 get_t(lds_cf E, int ft) const { return ld4(E + ft * 256 + C.fslot); }
    __device__ __forceinline__ void put_d(lds_f E, int ft, const f32x4& t) const { st4(E + ft * 256 + 4 * C.l, t); }
    __device__ __forceinline__ f32x4 get_d(lds_cf E, int ft) const { return ld4(E + ft * 256 + 4 * C.l); }

    // ---- forward of the row tile through the staged head: xb = the input columns 4q .. 4q + 3 of this lane's row (zero past the
    // net's inputs).  Out: the wave's own tiles of h1 / h2 (for the ReLU masks of a backward), h2 of all tiles (D layout) and the
    // head's outputs z[o], o < hn <= 4, on EVERY lane of the row.  KEEP: h1 is also left transposed in th1 and xb in tx.
    template <bool KEEP>
    __device__ __forceinline__ void forward(const f32x4& xb, f32x4 (&h1o)[2], f32x4 (&h2o)[2], f32x4 (&h2f)[kHT], f32x4& z, int hn) const {
        const ChainLds& S = C.S;
        const int w = C.w, q = C.q, fslot = C.fslot;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ot = 2 * w + x;
            const f32x4 acc = mfma4(ld4((lds_cf)(S.b1 + ot * 16 + 4 * q)), ld4((lds_cf)(S.w1 + ot * 256 + fslot)), xb);
#pragma unroll
            for (int r = 0; r < 4; ++r) h1o[x][r] = fmaxf(acc[r], 0.f);
            put_d(ea, ot, h1o[x]);
            if constexpr (KEEP) put_t(th1, ot, h1o[x]);
        }
        if constexpr (KEEP) { if (w == 0) put_t(tx, 0, xb); }
        lds_barrier();
        f32x4 h1f[kHT], acc[2];
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) h1f[kb] = get_d(ea, kb);
#pragma unroll
        for (int x = 0; x < 2; ++x) acc[x] = ld4((lds_cf)(S.b2 + (2 * w + x) * 16 + 4 * q));
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {
            f32x4 wf[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) wf[x] = ld4((lds_cf)(S.w2 + ((2 * w + x) * kHT + kb) * 256 + fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int x = 0; x < 2; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[x][e], h1f[kb][e], acc[x], 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h2o[x][r] = fmaxf(acc[x][r], 0.f);
            put_d(eb, 2 * w + x, h2o[x]);
        }
        lds_barrier();
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) h2f[kb] = get_d(eb, kb);
        // head of hn <= 4 outputs as dot products (chain_net.hpp: head_valu): a lane holds 32 of its row's 128 hidden features
        z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < hn) {
                float acc1 = 0.f;
#pragma unroll
                for (int kb = 0; kb < kHT; ++kb) {
                    const f32x4 wv = ld4((lds_cf)(S.w3 + kb * 256 + ((q * 16 + (o ^ q)) << 2)));
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc1 = fmaf(wv[r], h2f[kb][r], acc1);
                }
                acc1 += lane_xor<16>(acc1);
                acc1 += lane_xor<32>(acc1);
                z[o] = acc1 + S.b3[o];
            }
        }
    }

    // ---- forward-ONLY passes (the target nets): the wave's A fragments straight from the net's block in HBM / L2 (fragment-image
    // order: a lane's 16-byte slot of tile (ot, kb) is one global_load_dwordx4) into registers — no LDS image, no staging barriers —
    // and NHD heads (the twin target critics) in ONE pass sharing its two layer barriers.  Same MFMA / fma order per accumulator as
    // forward(): bit-identical outputs.  HN = head outputs the fragments provide for (1: a critic head; 4: an actor, o < hn loaded).
    template <int HN>
    struct Frag { f32x4 w1[2], b1[2], b2[2], w2[2][kHT], w3[HN][kHT]; float b3[HN], ls[HN]; };
    template <int HN>
    __device__ __forceinline__ Frag<HN> frag_fetch(g_cf th, int hn, int extra_n) const {
        Frag<HN> F;
        const int w = C.w, q = C.q, fslot = C.fslot;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ot = 2 * w + x;
            F.w1[x] = ld4(th + kL1w + ot * 256 + fslot);
            F.b1[x] = ld4(th + kL1b + ot * 16 + 4 * q);
            F.b2[x] = ld4(th + kL2b + ot * 16 + 4 * q);
#pragma unroll
            for (int kb = 0; kb < kHT; ++kb) F.w2[x][kb] = ld4(th + kL2w + (ot * kHT + kb) * 256 + fslot);
        }
#pragma unroll
        for (int o = 0; o < HN; ++o) {
            F.b3[o] = 0.f; F.ls[o] = 0.f;
#pragma unroll
            for (int kb = 0; kb < kHT; ++kb) F.w3[o][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (o < hn) {
#pragma unroll
                for (int kb = 0; kb < kHT; ++kb) F.w3[o][kb] = ld4(th + kL3w + kb * 256 + ((q * 16 + (o ^ q)) << 2));
                F.b3[o] = th[kL3b + o];
                if (o < extra_n) F.ls[o] = th[kHeadFloats + o];
            }
        }
        __builtin_amdgcn_sched_barrier(0);                             // the loads stay here: a pass ahead of their use
        return F;
    }
    // xb[hd]: the heads' input columns; z[hd][o]: their outputs on every lane of the row.  Exchange buffers: head 0 ea / eb, head 1
    // th1 / td (free outside a training pass).
    template <int NHD, int HN>
    __device__ __forceinline__ void forward_g(const Frag<HN> (&F)[NHD], const f32x4 (&xb)[NHD], f32x4 (&z)[NHD], int hn) const {
        const int w = C.w, q = C.q;
        lds_f EA[2] = {ea, th1}, EB[2] = {eb, td};
#pragma unroll
        for (int hd = 0; hd < NHD; ++hd)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const f32x4 acc = mfma4(F[hd].b1[x], F[hd].w1[x], xb[hd]);
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = fmaxf(acc[r], 0.f);
                put_d(EA[hd], 2 * w + x, h);
            }
        lds_barrier();
        f32x4 h1f[NHD][kHT], acc[NHD][2];
#pragma unroll
        for (int hd = 0; hd < NHD; ++hd) {
#pragma unroll
            for (int kb = 0; kb < kHT; ++kb) h1f[hd][kb] = get_d(EA[hd], kb);
#pragma unroll
            for (int x = 0; x < 2; ++x) acc[hd][x] = F[hd].b2[x];
        }
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int hd = 0; hd < NHD; ++hd)
#pragma unroll
                    for (int x = 0; x < 2; ++x)
                        acc[hd][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(F[hd].w2[x][kb][e], h1f[hd][kb][e], acc[hd][x], 0, 0, 0);
#pragma unroll
        for (int hd = 0; hd < NHD; ++hd)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = fmaxf(acc[hd][x][r], 0.f);
                put_d(EB[hd], 2 * w + x, h);
            }
        lds_barrier();
#pragma unroll
        for (int hd = 0; hd < NHD; ++hd) {
            f32x4 h2f[kHT];
#pragma unroll
            for (int kb = 0; kb < kHT; ++kb) h2f[kb] = get_d(EB[hd], kb);
            z[hd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < HN; ++o) {
                if (o < hn) {
                    float acc1 = 0.f;
#pragma unroll
                    for (int kb = 0; kb < kHT; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc1 = fmaf(F[hd].w3[o][kb][r], h2f[kb][r], acc1);
                    acc1 += lane_xor<16>(acc1);
                    acc1 += lane_xor<32>(acc1);
                    z[hd][o] = acc1 + F[hd].b3[o];
                }
            }
        }
    }

    // sum over the 16 rows of the tile (lanes of one q group): every lane of the group gets it
    __device__ __forceinline__ static float rows_sum(float v) {
        v += lane_xor<1>(v); v += lane_xor<2>(v); v += lane_xor<4>(v); v += lane_xor<8>(v);
        return v;
    }

    // ---- the head layer's share of a backward: dz[o] = d loss / d output o of this lane's row (o < hn, on every lane of the row)
    // -> the own tiles' deltas d2o through the ReLU of h2; WG (weight gradients wanted): the head's gradient tiles / bias -> slab
    template <bool WG>
    __device__ __forceinline__ void head_bwd(g_f hs, const f32x4& dz, const f32x4 (&h2o)[2], f32x4 (&d2o)[2], int hn) const {
        const ChainLds& S = C.S;
        const int w = C.w, q = C.q, i16 = C.i16;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ot = 2 * w + x;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, g3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < hn) {
                    const f32x4 wv = ld4((lds_cf)(S.w3 + ot * 256 + ((q * 16 + (o ^ q)) << 2)));      // W3[o][16 ot + 4q .. + 3]
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = fmaf(wv[r], dz[o], acc[r]);
                    if constexpr (WG) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float s = rows_sum(h2o[x][r] * dz[o]);           // dW3[o][16 ot + 4q + r] of this row tile
                            if (i16 == o) g3[r] = s;                              // ... lives in the slot of lane (i16 = o, q)
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) d2o[x][r] = h2o[x][r] > 0.f ? acc[r] : 0.f;
            if constexpr (WG) st4_slab(hs + kL3w + ot * 256 + C.fslot, g3);            // the whole tile: zeros in the slots of outputs >= hn
        }
        if constexpr (WG) {
            float gb = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < hn) { const float s = rows_sum(dz[o]); if (i16 == o) gb = s; }
            }
            if (w == 0 && q == 0) hs[kL3b + i16] = gb;
        }
    }

    // ---- layers 2 and 1 of a backward from the own tiles' d2o.  WG: dW2 / db2 / dW1 / db1 of the wave's output tiles -> slab
    // (needs forward<true>: th1, tx).  Returns d1o (own tiles, through the ReLU of h1).
    template <bool WG>
    __device__ __forceinline__ void hidden_bwd(g_f hs, const f32x4 (&d2o)[2], const f32x4 (&h1o)[2], f32x4 (&d1o)[2]) const {
        const ChainLds& S = C.S;
        const int w = C.w, q = C.q, i16 = C.i16, fslot = C.fslot, tslot = C.tslot;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            put_d(ea, 2 * w + x, d2o[x]);
            if constexpr (WG) put_t(td, 2 * w + x, d2o[x]);
        }
        lds_barrier();
        if constexpr (WG) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int ot = 2 * w + x;
                const f32x4 af = get_t(td, ot);
                float gb = (af[0] + af[1]) + (af[2] + af[3]);
                gb += lane_xor<16>(gb); gb += lane_xor<32>(gb);
                if (q == 0) hs[kL2b + ot * 16 + i16] = gb;
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt) {
                    const f32x4 g = mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, get_t(th1, kt), af);      // dW2^T tile (ot, kt): chain_net.hpp's accumulator layout
                    st4_slab(hs + kL2w + (ot * kHT + kt) * 256 + fslot, g);
                }
            }
        }
        f32x4 d2f[kHT], acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ob = 0; ob < kHT; ++ob) d2f[ob] = get_d(ea, ob);
#pragma unroll
        for (int ob = 0; ob < kHT; ++ob)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const float wa = S.w2[(ob * kHT + 2 * w + x) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];      // W2[16 ob + 4q + e][16 (2w + x) + i16]
                    acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, d2f[ob][e], acc[x], 0, 0, 0);
                }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) d1o[x][r] = h1o[x][r] > 0.f ? acc[x][r] : 0.f;
        if constexpr (WG) {
            const f32x4 xt = get_t(tx, 0);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int ot = 2 * w + x;
                put_t(td, ot, d1o[x]);                                            // (this wave's own tiles: its reads of d2 above are done, in order)
                const f32x4 af = get_t(td, ot);
                float gb = (af[0] + af[1]) + (af[2] + af[3]);
                gb += lane_xor<16>(gb); gb += lane_xor<32>(gb);
                if (q == 0) hs[kL1b + ot * 16 + i16] = gb;
                st4_slab(hs + kL1w + ot * 256 + fslot, mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, xt, af));
            }
        }
    }

    // dX = W1^T d1 of the row tile: the waves' d1 tiles meet through eb, then chain_net.hpp's delta0 (every wave computes the
    // same tile: 32 MFMAs).  Returns d loss / d input column 4q + r of this lane's row.
    __device__ __forceinline__ f32x4 input_bwd(const f32x4 (&d1o)[2]) const {
#pragma unroll
        for (int x = 0; x < 2; ++x) put_d(eb, 2 * C.w + x, d1o[x]);
        lds_barrier();
        f32x4 d1f[kHT];
#pragma unroll
        for (int ob = 0; ob < kHT; ++ob) d1f[ob] = get_d(eb, ob);
        return C.delta0(d1f);
    }
};

// ---- "every workgroup of the learner has written its slab": sixteen FLAG words, not a counter.  Workgroup b, behind a workgroup
// barrier (sync_stores: every thread's stores acknowledged), publishes flag[b] = epoch with ONE agent-scope release store (L2 write-back:
// the readers sit on other XCDs); sixteen threads of every workgroup each poll one flag until it holds this launch's epoch, then the
// workgroup takes an agent-scope acquire fence (invalidate) and reads the slabs.  Against the counter form (atomic add, then spin on
// the count): one memory round trip less on the critical path — the add had to return before the spin could start.
// epoch: unique per launch (SoloArgs::bar_base + kSoloWG, the host advances bar_base by kSoloWG per launch).  A thread that waits
// 2 s gives up and raises *err: the launch then finishes with wrong numbers instead of hanging the queue.
__device__ __forceinline__ void solo_grid_sync(unsigned* flags, int b, unsigned epoch, int* err, int nw = kSoloWG) {
    sync_stores();
    if (threadIdx.x == 0) {
        // (every wave's stores are acknowledged: each drained its own vmcnt in front of the barrier)  release fence, an explicit wait —
        // hipcc may drop the fence's own when it believes the wave has nothing outstanding, and the flag would overtake the write-back — then the flag
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flags + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((int)threadIdx.x < nw) {
        const unsigned long long t0 = wall_clock64();                      // 100 MHz
        while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) { *err = 1; break; }
        }
    }
    __syncthreads();
    // ONE lane's acquire serves the CU (buffer_inv sc1 drops the CU's L1; MI355X_MICROARCH.md, inter-workgroup visibility): 256 threads
    // running __threadfence() here each wrote the L2 back again and invalidated again
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
}

// What the update of one net needs besides the slabs
struct SoloUpdate {
    g_f th, mm, vv, tg;     // the net's blocks
    int size;               // floats (multiple of 32)
    float lr, wd;
    int soft;               // also theta_target <- (1 - tau) theta_target + tau theta
    int t_new;              // Adam step count of this update
};

// ---- behind grid barrier 1: this workgroup's sixteenth of the net — slab sum in workgroup order, partial squared norm ->
// grid barrier 2 -> clip coefficient, Adam, soft update.  Returns the gradient norm.
// W = workgroups of the learner (16: one row tile each; 8 / 4: two / four tiles each, every tile with a slab of its own): a workgroup's
// share of the net is 1 / W of it, 48 / W float4 per thread, summed over the nb tiles' slabs in tile order, W slabs (48 loads per thread) in
// flight at a time — the same sum in the same order whatever W is
template <int W = kSoloWG>
__device__ __forceinline__ float solo_update(const SoloArgs& s, const LearnArgs& a, const SoloUpdate& u, int p, int b, int nb, lds_f red,
                                             unsigned bar2_target
#ifdef FRL_SOLO_TIMING
                                             , unsigned long long solo_t0_
#endif
                                             ) {
    const int tid = threadIdx.x;
    const int n4 = u.size >> 2, per = (n4 + W - 1) / W, i0 = b * per, i1 = min(n4, i0 + per);
    constexpr int KM = 3 * kSoloWG / W;                                    // float4 per thread: nets of up to 16 x 3 x 256 x 4 = 49 k floats
    g_cf slab = as_global(s.slab + (size_t)p * kSoloWG * s.slab_stride);
    f32x4 g[KM], th[KM], mi[KM], vi[KM], tg[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int i = i0 + tid + kWG * k, ic = i < i1 ? i : (i1 > i0 ? i1 - 1 : 0);
        th[k] = ld4((g_cf)(u.th + 4 * ic)); mi[k] = ld4((g_cf)(u.mm + 4 * ic)); vi[k] = ld4((g_cf)(u.vv + 4 * ic)); tg[k] = ld4((g_cf)(u.tg + 4 * ic));
    }
    // every slab's share in flight at once — 16 x KM independent 16-byte loads per thread, ONE round trip to the memory side (the slabs
    // were written by other XCDs: nothing of them is in this L2) — then summed in workgroup order.  A runtime loop over the slabs with
    // the sum inside was four dependent round trips: 14 us of the 48 us launch (tools/solo_timing.py).
#pragma unroll
    for (int s0 = 0; s0 < kSoloWG; s0 += W) {
        f32x4 sl[W][KM];
#pragma unroll
        for (int sb = 0; sb < W; ++sb) {
            const int sc = s0 + sb < nb ? s0 + sb : nb - 1;                // (slabs past the batch's tiles: a harmless re-read, dropped below)
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int i = i0 + tid + kWG * k, ic = i < i1 ? i : (i1 > i0 ? i1 - 1 : 0);
                sl[sb][k] = ld4(slab + (size_t)sc * s.slab_stride + 4 * ic);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sb = 0; sb < W; ++sb) {
            if (s0 + sb < nb) {
#pragma unroll
                for (int k = 0; k < KM; ++k) g[k] += sl[sb][k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KM; ++k)
        if (i0 + tid + kWG * k >= i1) g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) ss += (g[k][0] * g[k][0] + g[k][1] * g[k][1]) + (g[k][2] * g[k][2] + g[k][3] * g[k][3]);
    ss = wave_sum(ss);
    __syncthreads();
    if ((tid & 63) == 0) red[64 + (tid >> 6)] = ss;
    __syncthreads();
    float* part = s.part + ((size_t)p * kSoloWG) * kSoloPart;
    // The sixteen partial norms meet through MAILBOXES, not a second grid barrier: workgroup b publishes {its partial, this launch's
    // epoch} as ONE 64-bit agent-scope atomic store, sixteen threads of every workgroup each poll one mailbox until its epoch is
    // this launch's.  Nothing but these words is exchanged here, so no release / acquire fence is needed (L2 write-back and
    // invalidate: most of a grid barrier's 3.3 us — tools/solo_timing.py).  epoch = the barrier target this launch would have used.
    typedef unsigned long long u64;
    if (tid == 0) {
        const float mine = ((red[64] + red[65]) + red[66]) + red[67];
        __hip_atomic_store((u64*)(part + b * kSoloPart + 2), ((u64)bar2_target << 32) | (u64)__float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SOLO_T(5);
    if (tid < W) {
        const u64* box = (const u64*)(part + tid * kSoloPart + 2);
        const unsigned long long t0 = wall_clock64();
        u64 v = __hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((unsigned)(v >> 32) != bar2_target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) { *s.err = 1; break; }
            v = __hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        red[80 + tid] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
    SOLO_T(6);
    float tot = 0.f;
#pragma unroll
    for (int sb = 0; sb < W; ++sb) tot += red[80 + sb];
    const float total = sqrtf(tot);
    const float coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    const double bc1 = 1.0 - powi_d((double)a.beta1, u.t_new), bc2 = 1.0 - powi_d((double)a.beta2, u.t_new);
    const float step = (float)((double)u.lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        const int i = i0 + tid + kWG * k;
        if (i < i1) {
            f32x4 gi = g[k] * coef, t4 = th[k], m4 = mi[k], v4 = vi[k];
            if (u.wd != 0.f) gi += u.wd * t4;
            m4 = m4 + (gi - m4) * w1;
            v4 = v4 * a.beta2 + (w2 * gi) * gi;
            f32x4 den;
#pragma unroll
            for (int r = 0; r < 4; ++r) den[r] = sqrtf(v4[r]) / bc2s + a.adam_eps;
            t4 = t4 - step * (m4 / den);
            st4(u.th + 4 * i, t4); st4(u.mm + 4 * i, m4); st4(u.vv + 4 * i, v4);
            if (u.soft) st4(u.tg + 4 * i, tg[k] * tk + t4 * a.tau);
        }
    }
    return total;
}

}  // namespace frl
