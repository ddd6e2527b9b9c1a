// ONE learner's update on sixteen workgroups (device/solo.hpp) for the shapes whose FIRST LAYER does not fit an LDS image: up to
// 416 input columns (SAC at Humanoid-v4's dims, config 4 of BASELINE.json: 376 + 17) and actor heads of up to 32 outputs, hidden
// 128 — kernels_solow.hip.  What a single `SAC.learn()` per vector step is on one GPU (SAC_file/SAC.py:519-576); the row-chunk
// chain took 330-390 us for it (eight 32-row workgroups, each phase behind an L2 round trip for its weight block, two more
// launches for the reduce and Adam), the K-sliced population kernels (chain_wide.hpp) give the whole learner to ONE workgroup.
//
// Same decomposition as solo.hpp — a 256-row batch is sixteen 16-row tiles, one workgroup each; wave w owns output tiles 2w, 2w + 1
// of every 128-wide layer; partial gradients to per-workgroup slabs in image order, summed in workgroup order behind a flag
// hand-over — with these differences:
//   * W1 (up to 26 k-tiles x 8 output tiles = 208 KB) never sits in LDS: a wave's A fragments of the first layer (its two output
//     tiles x every k-tile) go from the net's block (fragment-image order: one global_load_dwordx4 per lane and tile) straight
//     into registers, four k-tiles at a time, fetched two batches ahead of their MFMAs (the first two a whole pass ahead).
//   * the tile's input rows sit in LDS ONCE, as fragment images (26 KB; wave w fetches the k-tiles kb = w mod 4 from the replay
//     records — a field that starts off a 16-byte boundary as the two aligned dwordx4 around a fragment): read as stored they are
//     the B operand of a first layer, read transposed (as a weight image is in a backward) the A operand of dW1.  The k-tiles
//     that hold a policy's action columns are composed next to them ([s | a]: three tiles) for the critic passes on a = pi(s).
//   * dW1 of the wave's two output tiles = (x^T of the row tile) x (delta tiles): 2 x KB1 tiles of four MFMAs each, stored
//     straight to the slab.
//   * actor heads of up to 32 outputs run as MFMA tiles (every wave computes the head of the tile's 16 rows: 64 MFMAs), their
//     backward (dW3 = h2^T dz, d2 = W3^T dz) on the wave's own tiles with wave-local transposes.
//   * dQ/da for the policy step: only the k-tiles of W1 that hold action columns (<= 3) are walked backward, one per wave,
//     transposed fragments as dword loads from the block.
//   * the update: nets of up to 2 x 75 k floats do not fit a thread's registers — the slab sum goes through EngineDesc::grad
//     (phase 1: sum + squared norm; mailboxes; phase 2: clip, Adam, soft update).
#pragma once
#include "solo.hpp"

#ifndef FRL_SOLOW_G
#define FRL_SOLOW_G 4       // k-tiles of W1 per fetched batch of a first-layer pass (three batches live: 24 G registers)
#endif

namespace frl {

// (kSoloWMaxKB, solow_lds_floats(): frl_desc.h)

// where the input columns of a row tile come from: columns [0, ng) = the row's record from float `off` on, zero behind them
struct SoloWX {
    g_cf rec;               // this lane's row: its record in the ring
    int off, ng;
    int stride;             // floats per record (a multiple of 4)
};

struct SoloWNet {
    ChainNet C;             // lane constants; S.w2 / w3 / b1 / b2 / b3 / ls point into the carve below
    lds_f ea, eb;           // activation / delta exchange, MFMA D layout: tile ft at ft * 256 + 4 * lane
    lds_f th1;              // h1 of the row tile, transposed fragment image (all 8 feature tiles): operand of dW2
    lds_f td;               // a delta's (or h2's) own tiles, transposed (wave-local)
    lds_f xs;               // the tile's input rows as fragment images (k-tile kb at kb * 256: lane (row, q)'s columns 16 kb + 4 q .. + 3 at fslot):
                            // B operand of a first layer (one ds_read_b128) AND, read transposed like a weight image, the A operand of dW1
    lds_f xa;               // ... the k-tiles that hold a policy's action columns, [s | a] composed (<= 3 tiles), for the critic passes on them
    lds_f zt;               // a tile of zeros: the rows of the k-tiles past a first layer's last one, in the sweeps' padded last batch
    lds_f tz;               // per wave: the head's dz tiles, transposed (operand of dW3)
    lds_f ar;               // [16 rows][32] the tile's policy / target-policy actions
    lds_f dxa;              // [16 rows][48] dQ/d(input columns of the action k-tiles)
    lds_f red;

    __device__ __forceinline__ void init(float* smem) {
        lds_f p = (lds_f)smem;
        C.S.w2 = p; p += kHT * kHT * 256;
        C.S.w3 = p; p += 2 * kHT * 256;
        C.S.b1 = p; p += kHid;
        C.S.b2 = p; p += kHid;
        C.S.b3 = p; p += 32;
        C.S.ls = p; p += 32;
        ea = p; p += kHT * 256;
        eb = p; p += kHT * 256;
        th1 = p; p += kHT * 256;
        td = p; p += kHT * 256;
        xs = p; p += kSoloWMaxKB * 256;
        xa = p; p += 3 * 256;
        zt = p; p += 256;
        tz = p; p += 4 * 2 * 256;
        ar = p; p += 16 * 32;
        dxa = p; p += 16 * 48;
        red = p; p += 128;
        zt[threadIdx.x] = 0.f;                                         // (visible behind the first stage_commit's barriers)
        C.S.w1 = ea; C.S.ea = ea; C.S.eb = eb; C.S.ab = ea; C.S.yb = ea; C.S.q1 = ea; C.S.lpn = ea; C.S.red = red;
        C.init_lanes();
    }

    __device__ __forceinline__ void put_t(lds_f E, int ft, const f32x4& t) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) E[ft * 256 + C.tslot + (((4 * C.q + r) ^ (C.i16 >> 2)) << 2)] = t[r];
    }
    __device__ __forceinline__ f32x4 get_t(lds_cf E, int ft) const { return ld4(E + ft * 256 + C.fslot); }
    __device__ __forceinline__ void put_d(lds_f E, int ft, const f32x4& t) const { st4(E + ft * 256 + 4 * C.l, t); }
    __device__ __forceinline__ f32x4 get_d(lds_cf E, int ft) const { return ld4(E + ft * 256 + 4 * C.l); }

    // ---- columns 16 kb + 4 q .. + 3 of this lane's row: every load unconditional (clamped inside the record, selected afterwards —
    // a conditional load is a branch with its own wait: the first build's per-element conditions put sixteen L2 round trips in a
    // row per batch of four k-tiles, 12 us per pass).  A field that starts off a 16-byte boundary (next_obs behind [obs | act |
    // rew | done]) is read as the two aligned dwordx4 around the fragment
    __device__ __forceinline__ f32x4 xload(const SoloWX& X, int kb) const {
        const int c0 = 16 * kb + 4 * C.q, mis = X.off & 3;
        f32x4 v;
        if (mis == 0) {
            v = ld4(X.rec + min(X.off + c0, X.stride - 4));
        } else {
            const int fo = X.off - mis + c0;
            const f32x4 lo = ld4(X.rec + min(fo, X.stride - 4)), hi = ld4(X.rec + min(fo + 4, X.stride - 4));
            if (mis == 1) v = f32x4{lo[1], lo[2], lo[3], hi[0]};
            else if (mis == 2) v = f32x4{lo[2], lo[3], hi[0], hi[1]};
            else v = f32x4{lo[3], hi[0], hi[1], hi[2]};
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c0 + e < X.ng ? v[e] : 0.f;
        return v;
    }
    // the tile's rows -> xs: wave w holds k-tiles kb = w (mod 4) (fetch: up to seven loads in flight; commit: behind a barrier that
    // says every wave is done with the old rows, and in front of one — the caller's next stage_commit — before anybody reads)
    struct XRegs { f32x4 v[(kSoloWMaxKB + 3) / 4]; };
    __device__ __forceinline__ XRegs x_fetch(const SoloWX& X, int KB1) const {
        XRegs R;
#pragma unroll
        for (int j = 0; j < (kSoloWMaxKB + 3) / 4; ++j) { const int kb = C.w + 4 * j; R.v[j] = xload(X, kb < KB1 ? kb : KB1 - 1); }
        return R;
    }
    // (xb: first tile of xs the rows go to — MADDPG's actor stage keeps agent i's observation rows, tiles kSoloWActorBase .., next to
    // the joint [s | a] rows of its critic passes, tiles 0 ..)
    __device__ __forceinline__ void x_commit(const XRegs& R, int KB1, int xb = 0) const {
#pragma unroll
        for (int j = 0; j < (kSoloWMaxKB + 3) / 4; ++j) { const int kb = C.w + 4 * j; if (kb < KB1) st4(xs + (xb + kb) * 256 + C.fslot, R.v[j]); }
    }
    // xa <- the k-tiles ka0 .. ka0 + nka - 1 of xs with columns [O, O + A) replaced by ar[0 .. A) of this row (wave j < nka: tile
    // ka0 + j; the other columns as they are in xs: zero behind a single agent's observation, the batch's joint actions around
    // the updating agent's in MADDPG's actor stage).  xs and ar must be visible (a barrier in front), xa is visible behind the
    // caller's next barrier.
    __device__ __forceinline__ void xa_compose(int O, int A, int ka0, int nka) const {
        if (C.w < nka) {
            const int kb = ka0 + C.w, c0 = 16 * kb + 4 * C.q;
            f32x4 v = ld4((lds_cf)(xs + kb * 256 + C.fslot));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = c0 + e - O;
                const float av = ar[C.i16 * 32 + (j < 0 ? 0 : (j > 31 ? 31 : j))];
                v[e] = (j >= 0 && j < A) ? av : v[e];
            }
            st4(xa + C.w * 256 + C.fslot, v);
        }
    }

    // ---- layers 2 and 3 of one head -> LDS images (linear copies of the block's image order), the three biases, log_std.
    // fetch issues the loads (84 registers per lane) and can sit a pass ahead; commit waits for the other waves to be done
    // with the old images.  L = the head's three LayerDescs, nt3 = head tiles
    struct Stage { f32x4 t2[16], t3[4]; float bb, b3v, lsv; };
    __device__ __forceinline__ Stage stage_fetch(g_cf th, const LayerDesc* L, int nt3, int extra_off, int extra_n) const {
        Stage R;
        const int tid = C.tid;
#pragma unroll
        for (int j = 0; j < 16; ++j) R.t2[j] = ld4(th + L[1].w_off + 4 * (tid + kWG * j));
#pragma unroll
        for (int j = 0; j < 4; ++j) R.t3[j] = j < 2 * nt3 ? ld4(th + L[2].w_off + 4 * (tid + kWG * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
        R.bb = tid < kHid ? th[L[0].b_off + tid] : th[L[1].b_off + tid - kHid];
        R.b3v = tid < 16 * nt3 ? th[L[2].b_off + tid] : 0.f;
        R.lsv = tid < extra_n ? th[extra_off + tid] : 0.f;
        __builtin_amdgcn_sched_barrier(0);                             // the loads stay here: hipcc would sink them to the stores
        return R;
    }
    __device__ __forceinline__ void stage_commit(const Stage& R) const {
        const int tid = C.tid;
        lds_barrier();                                                 // every wave is done with the previous images
#pragma unroll
        for (int j = 0; j < 16; ++j) st4(C.S.w2 + 4 * (tid + kWG * j), R.t2[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(C.S.w3 + 4 * (tid + kWG * j), R.t3[j]);
        if (tid < kHid) C.S.b1[tid] = R.bb; else C.S.b2[tid - kHid] = R.bb;
        if (tid < 32) { C.S.b3[tid] = R.b3v; C.S.ls[tid] = R.lsv; }
        lds_barrier();
    }

    // ---- first layer of the row tile: acc[x] (the bias on entry) += W1[tiles 2w + x] x, K = 16 KB1 columns: W1's fragments from
    // the block, G k-tiles at a time; the rows from xs (k-tiles < ka), xa (the rest) and the zero tile (past the last one).
    // The first two batches of a sweep are fetched a pass AHEAD (l1_fetch next to the stage_fetch of the same net).
    // (k-tiles past the last one are NOT clamped: such a fragment is the head of the next tile row, or of the b1 / W2 that follow every
    // W1 in its block — at most 3 G KB past the layer, finite numbers that meet the zero tile — and one base address per tile row
    // with the k-tile as an immediate offset is all the address arithmetic a sweep has left)
    template <int G>
    struct L1Pre { f32x4 b0[2][G], b1[2][G]; };
    template <int G>
    __device__ __forceinline__ L1Pre<G> l1_fetch(g_cf w1, int KB1) const {
        L1Pre<G> P;
        g_cf p0 = w1 + ((size_t)(2 * C.w) * KB1 * 256 + C.fslot), p1 = p0 + (size_t)KB1 * 256;
#pragma unroll
        for (int g = 0; g < G; ++g) { P.b0[0][g] = ld4(p0 + g * 256); P.b0[1][g] = ld4(p1 + g * 256); }
        __builtin_amdgcn_sched_barrier(0);                             // (in this order, pinned: vmcnt counts in issue order)
#pragma unroll
        for (int g = 0; g < G; ++g) { P.b1[0][g] = ld4(p0 + (G + g) * 256); P.b1[1][g] = ld4(p1 + (G + g) * 256); }
        __builtin_amdgcn_sched_barrier(0);
        return P;
    }
    template <int G>
    __device__ __forceinline__ void l1(g_cf w1, int KB1, int ka, int xb, f32x4 (&acc)[2], L1Pre<G>& P) const {
        const int w = C.w, fslot = C.fslot;
        f32x4 (&b0)[2][G] = P.b0, (&b1)[2][G] = P.b1;
        f32x4 b2[2][G];
        g_cf nx0 = w1 + ((size_t)(2 * w) * KB1 * 256 + fslot) + G * 256, nx1 = nx0 + (size_t)KB1 * 256;      // (the batch in front of the next one fetched)
        // One sweep = G k-tiles on the batch W, the batch two sweeps ahead fetched into Wn on the way: two fragment loads in front of
        // each k-tile's eight MFMAs, so that their issue sits in the MFMAs' shadow (one wave per SIMD: nothing else would fill it).
        // No branch and no select in here.  Measured on the way (tools/solow_timing.py, a 24-tile sweep, us): fragments fetched right
        // in front of their MFMAs behind per-tile branches 5.7; one batch ahead 5.9 (the loop header drained everything: batches
        // issued in the wrong order); two ahead, loads in front of the batch's 32 MFMAs and the padded tiles' rows zeroed by
        // v_cndmask into ONE register 4.5 — of which the MFMAs alone 3.7 (46 cycles each: the select's write waits for the MFMA in
        // front to have read the register) and the load phase alone 2.1; this form 3.8 (3.6 without the loads: 44 cycles per
        // MFMA, the LDS round trip of a sweep's first rows; reading them a sweep ahead: 3.6 with loads, at 40 more registers: not kept).
        auto sweep = [&](int kb0, const f32x4 (&W)[2][G], f32x4 (&Wn)[2][G]) {
            f32x4 xf[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int kb = kb0 + g;
                xf[g] = ld4((lds_cf)((kb < ka ? xs + (xb + kb) * 256 : (kb < KB1 ? xa + (kb - ka) * 256 : zt)) + fslot));
            }
            nx0 += G * 256; nx1 += G * 256;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#if defined(FRL_SOLOW_ABL) && (FRL_SOLOW_ABL & 1)
                Wn[0][g] = W[0][g]; Wn[1][g] = W[1][g];                // (timing only: no fragment loads behind the first two batches)
#else
                Wn[0][g] = ld4(nx0 + g * 256); Wn[1][g] = ld4(nx1 + g * 256);
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int x = 0; x < 2; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[x][g][e], xf[g][e], acc[x], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                     // (a k-tile's loads stay in front of its MFMAs: hipcc sinks them to their uses)
            }
        };
        // three buffers in fixed roles, the loop unrolled by three: a batch is fetched TWO sweeps ahead of its use (a rotation by
        // register moves made every sweep wait for the batch fetched at its own top: one ahead again)
#ifdef FRL_SOLO_TIMING
        if (threadIdx.x == 0) red[121] = (float)(unsigned)(wall_clock64() & 0xFFFFFFull);
#endif
        for (int kb0 = 0; kb0 < KB1; kb0 += 3 * G) {
            sweep(kb0, b0, b2);
#ifdef FRL_SOLO_TIMING
            if (threadIdx.x == 0 && kb0 == 0) red[122] = (float)(unsigned)(wall_clock64() & 0xFFFFFFull);
#endif
            if (kb0 + G >= KB1) break;
            sweep(kb0 + G, b1, b0);
            if (kb0 + 2 * G >= KB1) break;
            sweep(kb0 + 2 * G, b2, b1);
        }
    }

    // ---- forward of the row tile through the staged head (W2 / W3 images in LDS, W1 from the block w1, rows in xs / xa).  Out: the
    // wave's own tiles of h1 / h2 (ReLU masks of a backward) and h2 of all tiles (D layout).  KEEP: h1 is also left transposed in th1.
    typedef L1Pre<FRL_SOLOW_G> Pre;
    __device__ __forceinline__ Pre pre_fetch(g_cf w1, int KB1) const { return l1_fetch<FRL_SOLOW_G>(w1, KB1); }
    template <bool KEEP>
    __device__ __forceinline__ void forward(g_cf w1, int KB1, int ka, Pre& P, f32x4 (&h1o)[2], f32x4 (&h2o)[2], f32x4 (&h2f)[kHT], int xb = 0) const {
        const ChainLds& S = C.S;
        const int w = C.w, q = C.q, fslot = C.fslot;
        f32x4 acc[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) acc[x] = ld4((lds_cf)(S.b1 + (2 * w + x) * 16 + 4 * q));
        l1<FRL_SOLOW_G>(w1, KB1, ka, xb, acc, P);
#ifdef FRL_SOLO_TIMING
        if (threadIdx.x == 0) red[120] = (float)(unsigned)(wall_clock64() & 0xFFFFFFull);      // (tools/solow_timing.py: end of the last first-layer sweep)
#endif
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h1o[x][r] = fmaxf(acc[x][r], 0.f);
            put_d(ea, 2 * w + x, h1o[x]);
            if constexpr (KEEP) put_t(th1, 2 * w + x, h1o[x]);
        }
        lds_barrier();
        f32x4 h1f[kHT];
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
    }

    // a critic's single output as a dot product (solo.hpp: forward): on every lane of the row
    __device__ __forceinline__ float head_q(const f32x4 (&h2f)[kHT]) const {
        const ChainLds& S = C.S;
        float acc1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < kHT; ++kb) {
            const f32x4 wv = ld4((lds_cf)(S.w3 + kb * 256 + ((C.q * 16 + C.q) << 2)));
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1 = fmaf(wv[r], h2f[kb][r], acc1);
        }
        acc1 += lane_xor<16>(acc1);
        acc1 += lane_xor<32>(acc1);
        return acc1 + S.b3[0];
    }
    // an actor's NT3 head tiles: z[t][r] = output 16 t + 4 q + r of this lane's row (every wave computes all of them)
    template <int NT3>
    __device__ __forceinline__ void head_tiles(const f32x4 (&h2f)[kHT], f32x4 (&z)[NT3]) const {
        const ChainLds& S = C.S;
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            f32x4 acc = ld4((lds_cf)(S.b3 + 16 * t + 4 * C.q));
#pragma unroll
            for (int kb = 0; kb < kHT; ++kb) acc = mfma4(acc, ld4((lds_cf)(S.w3 + (t * kHT + kb) * 256 + C.fslot)), h2f[kb]);
            z[t] = acc;
        }
    }

    // ---- a critic head's share of a backward: dz = d loss / d Q of this lane's row (on every lane of the row) -> the own tiles'
    // deltas d2o through the ReLU of h2; WG: the head's gradient tiles / bias -> slab (hs = the net's slab, L = the head's layers)
    template <bool WG>
    __device__ __forceinline__ void head_bwd_q(g_f hs, const LayerDesc* L, float dz, const f32x4 (&h2o)[2], f32x4 (&d2o)[2]) const {
        const ChainLds& S = C.S;
        const int w = C.w, q = C.q, i16 = C.i16;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ot = 2 * w + x;
            const f32x4 wv = ld4((lds_cf)(S.w3 + ot * 256 + ((q * 16 + q) << 2)));                     // W3[0][16 ot + 4q .. + 3]
            f32x4 g3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d2o[x][r] = h2o[x][r] > 0.f ? wv[r] * dz : 0.f;
                if constexpr (WG) {
                    const float s = SoloNet::rows_sum(h2o[x][r] * dz);
                    if (i16 == 0) g3[r] = s;
                }
            }
            if constexpr (WG) st4_slab(hs + L[2].w_off + ot * 256 + C.fslot, g3);
        }
        if constexpr (WG) {
            const float s = SoloNet::rows_sum(dz);
            if (w == 0 && q == 0) hs[L[2].b_off + i16] = i16 == 0 ? s : 0.f;
        }
    }

    // ---- an actor head's share: dz[t][r] = d loss / d output 16 t + 4 q + r of this lane's row (the same on every wave)
    template <int NT3>
    __device__ __forceinline__ void head_bwd_a(g_f hs, const LayerDesc* L, const f32x4 (&dz)[NT3], const f32x4 (&h2o)[2], f32x4 (&d2o)[2]) const {
        const ChainLds& S = C.S;
        const int w = C.w, q = C.q, i16 = C.i16, tslot = C.tslot;
        lds_f tzw = tz + w * 2 * 256;
#pragma unroll
        for (int t = 0; t < NT3; ++t) put_t(tzw, t, dz[t]);                                            // (wave-local: read back below, in order)
#pragma unroll
        for (int x = 0; x < 2; ++x) put_t(td, 2 * w + x, h2o[x]);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            const f32x4 zt = get_t(tzw, t);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int ot = 2 * w + x;
                // dW3 tile (t, ot) = dz^T h2: accumulator layout of chain_net.hpp (in = 16 ot + 4q + r, out = 16 t + i16)
                st4_slab(hs + L[2].w_off + (t * kHT + ot) * 256 + C.fslot, mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, get_t(td, ot), zt));
                f32x4 wa;
#pragma unroll
                for (int e = 0; e < 4; ++e) wa[e] = S.w3[(t * kHT + ot) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];      // W3[16 t + 4q + e][16 ot + i16]
                acc[x] = mfma4(acc[x], wa, dz[t]);
            }
            float gb = (zt[0] + zt[1]) + (zt[2] + zt[3]);                                              // db3[16 t + i16]: the tile's 16 rows
            gb += lane_xor<16>(gb); gb += lane_xor<32>(gb);
            if (w == 0 && q == 0) hs[L[2].b_off + 16 * t + i16] = gb;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) d2o[x][r] = h2o[x][r] > 0.f ? acc[x][r] : 0.f;
    }

    // ---- layers 2 and 1 of a backward from the own tiles' d2o.  WG: dW2 / db2 / dW1 / db1 of the wave's output tiles -> slab
    // (needs forward<true>: th1; the pass's rows in xs).  Returns d1o (own tiles, through the ReLU of h1).
    template <bool WG>
    __device__ __forceinline__ void hidden_bwd(g_f hs, const LayerDesc* L, int KB1, const f32x4 (&d2o)[2], const f32x4 (&h1o)[2], f32x4 (&d1o)[2], int xb = 0) const {
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
                if (q == 0) hs[L[1].b_off + ot * 16 + i16] = gb;
#pragma unroll
                for (int kt = 0; kt < kHT; ++kt)
                    st4_slab(hs + L[1].w_off + (ot * kHT + kt) * 256 + fslot, mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, get_t(th1, kt), af));
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
            f32x4 af[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int ot = 2 * w + x;
                put_t(td, ot, d1o[x]);                                            // (this wave's own tiles: its reads of d2 above are done, in order)
                af[x] = get_t(td, ot);
                float gb = (af[x][0] + af[x][1]) + (af[x][2] + af[x][3]);
                gb += lane_xor<16>(gb); gb += lane_xor<32>(gb);
                if (q == 0) hs[L[0].b_off + ot * 16 + i16] = gb;
            }
            // four k-tiles at a time: their sixteen transposed reads in flight before the first MFMA (one tile at a time, every
            // tile's MFMAs waited out an LDS round trip: 25 of them per head)
            for (int kt0 = 0; kt0 < KB1; kt0 += 4) {
                f32x4 xt[4];                                                      // x^T of k-tile kt: rows 4q .. 4q + 3 of column 16 kt + i16
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kt = kt0 + j < KB1 ? kt0 + j : KB1 - 1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xt[j][e] = xs[(xb + kt) * 256 + tslot + (((4 * q + e) ^ (i16 >> 2)) << 2)];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kt0 + j < KB1) {
#pragma unroll
                        for (int x = 0; x < 2; ++x)
                            st4_slab(hs + L[0].w_off + ((size_t)((2 * w + x) * KB1 + kt0 + j) * 256 + fslot), mfma4(f32x4{0.f, 0.f, 0.f, 0.f}, xt[j], af[x]));
                    }
                }
            }
        }
    }

    // ---- dX of the k-tiles ka0 .. ka0 + nka - 1 (nka <= 3: the ones that hold action columns) = W1^T d1 of the row tile: the
    // waves' d1 tiles meet through eb; wave j < nka walks k-tile ka0 + j (transposed fragments as dword loads from the block, issued
    // by input_bwd_fetch a pass ahead) and leaves dxa[row][16 j + 4 q + r].  A workgroup barrier behind it makes dxa readable.
    struct DxRegs { f32x4 wa[kHT]; };
    __device__ __forceinline__ DxRegs input_bwd_fetch(g_cf w1, int KB1, int ka0, int nka) const {
        DxRegs R;
        const int kt = ka0 + (C.w < nka ? C.w : 0);
#pragma unroll
        for (int ob = 0; ob < kHT; ++ob)
#pragma unroll
            for (int e = 0; e < 4; ++e) R.wa[ob][e] = w1[(size_t)(ob * KB1 + kt) * 256 + C.tslot + (((4 * C.q + e) ^ (C.i16 >> 2)) << 2)];
        return R;
    }
    __device__ __forceinline__ void input_bwd(const DxRegs& R, int nka, const f32x4 (&d1o)[2]) const {
#pragma unroll
        for (int x = 0; x < 2; ++x) put_d(eb, 2 * C.w + x, d1o[x]);
        lds_barrier();
        if (C.w < nka) {
            f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ob = 0; ob < kHT; ++ob) dx = mfma4(dx, R.wa[ob], get_d(eb, ob));
            st4(dxa + C.i16 * 48 + 16 * C.w + 4 * C.q, dx);
        }
        lds_barrier();
    }
};

// ---- "the unit's slabs are written" (solo.hpp: solo_grid_sync) by its NT workgroups with row tiles (16 for batches of up to 256 rows
// — 8, with two tiles each, for populations of 17 .. 32 units —, 64 for MADDPG's 1024) with HELPER workgroups: b >= NT has no row
// tile and publishes nothing — it waits for the NT flags like the others and takes its share of the update
__device__ __forceinline__ void solow_grid_sync(unsigned* flags, int b, int NT, unsigned epoch, int* err) {
    sync_stores();
    if (threadIdx.x == 0 && b < NT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flags + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((int)threadIdx.x < NT) {
        const unsigned long long t0 = wall_clock64();                      // 100 MHz
        // ("has reached", not "equals": in the fused step a workgroup without rows flags its empty slab, walks through the actor half and
        //  flags the NEXT epoch before a slow helper has looked — epochs only grow)
        while ((int)(__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) { *err = 1; break; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
}

// ---- the fused policy step: the workgroups with row tiles wait here — behind the policy's forward, which needs nothing of it — until
// the unit's H helpers have stepped the critic (flags2[h] = epoch, published like the slab flags)
__device__ __forceinline__ void solow_wait_flags(const unsigned* flags2, int H, unsigned epoch, int* err) {
    if ((int)threadIdx.x < H) {
        const unsigned long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(flags2 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) { *err = 1; break; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
}
__device__ __forceinline__ void solow_publish(unsigned* flag, unsigned epoch) {
    sync_stores();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- behind the slab hand-over: this workgroup's share (one Wt-th: the learner's sixteen workgroups and its Wt - 16 helpers — a
// CU pulls ~65 GB/s of slabs through the fabric, three dependent round trips per workgroup for the twin critic of config 4 on
// sixteen: 8.4 us, and the other 240 CUs idle) of a net of any size — phase 1: slab sum in workgroup order -> gsum
// (EngineDesc::grad), partial squared norm; the sixteen partial norms meet through the mailboxes of solo_update; phase 2: clip
// coefficient, Adam, soft update.  Returns the gradient norm.
// (b of Wt: this workgroup's place among the ones that share the update — all of the unit's, or, in the fused policy step, its helpers
// alone; part: the unit's rows of SoloArgs::part, whose mailboxes are read by place)
__device__ __forceinline__ float solow_update(const SoloArgs& s, const LearnArgs& a, const SoloUpdate& u, g_f gsum, int unit, float* part, int b, int nb, int NT, int Wt, lds_f red,
                                              lds_f box, unsigned bar2_target
#ifdef FRL_SOLO_TIMING
                                              , unsigned long long solo_t0_
#endif
                                              ) {
    constexpr int W = kSoloWG, KM = 3;
    const int tid = threadIdx.x;
    const int n4 = u.size >> 2, per = (n4 + Wt - 1) / Wt, i0 = b * per, i1 = min(n4, i0 + per);
    g_cf slab = as_global(s.slab + (size_t)unit * NT * s.slab_stride);
    float ss = 0.f;
    const bool many_slabs = NT > W && per <= kWG / 2;
    if (many_slabs) {
        // MADDPG's 64 slabs of a 7-29 k-float net: a workgroup's share is ~100 float4 — one per thread of a HALF of the workgroup,
        // the halves take slabs [0, 32) and [32, 64), all 32 loads of a thread in flight at once, the two partial sums meet in LDS
        // (four dependent rounds of sixteen slabs with 113 of 256 threads busy were 15 us of a 53 us launch)
        const int half = tid >> 7, t = tid & 127, i = i0 + t, ic = i < i1 ? i : (i1 > i0 ? i1 - 1 : 0);
        f32x4 sl[32], acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sb = 0; sb < 32; ++sb) {
            const int sx = 32 * half + sb, sc = sx < nb ? sx : nb - 1;
            sl[sb] = ld4(slab + (size_t)sc * s.slab_stride + 4 * (size_t)ic);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sb = 0; sb < 32; ++sb) if (32 * half + sb < nb) acc += sl[sb];
        st4(box + 256 + 4 * tid, acc);
        __syncthreads();
        if (tid < kWG / 2 && i < i1) {
            const f32x4 g = ld4((lds_cf)(box + 256 + 4 * tid)) + ld4((lds_cf)(box + 256 + 4 * (tid + 128)));
            st4(gsum + 4 * (size_t)i, g);
            ss += (g[0] * g[0] + g[1] * g[1]) + (g[2] * g[2] + g[3] * g[3]);
        }
    }
    for (int c0 = i0; c0 < i1 && !many_slabs; c0 += kWG * KM) {
        f32x4 g[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < nb; s0 += W) {                               // sixteen slabs in flight at a time, summed in tile order
            f32x4 sl[W][KM];
#pragma unroll
            for (int sb = 0; sb < W; ++sb) {
                const int sc = s0 + sb < nb ? s0 + sb : nb - 1;            // (slabs past the batch's tiles: a harmless re-read, dropped below)
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    const int i = c0 + tid + kWG * k, ic = i < i1 ? i : i1 - 1;
                    sl[sb][k] = ld4(slab + (size_t)sc * s.slab_stride + 4 * (size_t)ic);
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
        for (int k = 0; k < KM; ++k) {
            const int i = c0 + tid + kWG * k;
            if (i < i1) {
                st4(gsum + 4 * (size_t)i, g[k]);
                ss += (g[k][0] * g[k][0] + g[k][1] * g[k][1]) + (g[k][2] * g[k][2] + g[k][3] * g[k][3]);
            }
        }
    }
    ss = wave_sum(ss);
    sync_stores();                                                         // (phase 2 reads gsum back: the stores are acknowledged)
    if ((tid & 63) == 0) red[64 + (tid >> 6)] = ss;
    __syncthreads();
    SOLO_T(5);
    typedef unsigned long long u64;
    if (tid == 0) {
        const float mine = ((red[64] + red[65]) + red[66]) + red[67];
        __hip_atomic_store((u64*)(part + b * kSoloPart + 2), ((u64)bar2_target << 32) | (u64)__float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < Wt) {
        const u64* mbox = (const u64*)(part + tid * kSoloPart + 2);
        const unsigned long long t0 = wall_clock64();
        u64 v = __hip_atomic_load(mbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((unsigned)(v >> 32) != bar2_target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) { *s.err = 1; break; }
            v = __hip_atomic_load(mbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        box[tid] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
    SOLO_T(6);
    float tot = 0.f;
#pragma unroll
    for (int sb = 0; sb < Wt; ++sb) tot += box[sb];                       // (workgroup order: the same sum in every workgroup)
    const float total = sqrtf(tot);
    const float coef = a.clip_norm > 0.f ? fminf(a.clip_norm / (total + 1e-6f), 1.f) : 1.f;
    const double bc1 = 1.0 - powi_d((double)a.beta1, u.t_new), bc2 = 1.0 - powi_d((double)a.beta2, u.t_new);
    const float step = (float)((double)u.lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2, tk = 1.f - a.tau;
    for (int c0 = i0; c0 < i1; c0 += kWG * KM) {
        f32x4 g[KM], th[KM], mi[KM], vi[KM], tg[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int i = c0 + tid + kWG * k, ic = i < i1 ? i : i1 - 1;
            g[k] = ld4((g_cf)(gsum + 4 * (size_t)ic));                  // (this thread's own stores of phase 1)
            th[k] = ld4((g_cf)(u.th + 4 * (size_t)ic)); mi[k] = ld4((g_cf)(u.mm + 4 * (size_t)ic)); vi[k] = ld4((g_cf)(u.vv + 4 * (size_t)ic));
            tg[k] = ld4((g_cf)(u.tg + 4 * (size_t)ic));
        }
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int i = c0 + tid + kWG * k;
            if (i < i1) {
                f32x4 gi = g[k] * coef, t4 = th[k], m4 = mi[k], v4 = vi[k];
                if (u.wd != 0.f) gi += u.wd * t4;
                m4 = m4 + (gi - m4) * w1;
                v4 = v4 * a.beta2 + (w2 * gi) * gi;
                f32x4 den;
#pragma unroll
                for (int r = 0; r < 4; ++r) den[r] = sqrtf(v4[r]) / bc2s + a.adam_eps;
                t4 = t4 - step * (m4 / den);
                st4(u.th + 4 * (size_t)i, t4); st4(u.mm + 4 * (size_t)i, m4); st4(u.vv + 4 * (size_t)i, v4);
                if (u.soft) st4(u.tg + 4 * (size_t)i, tg[k] * tk + t4 * a.tau);
            }
        }
    }
    return total;
}

}  // namespace frl
