// select_action's shared device code: the action / exploration rules behind a policy's head outputs (act_epilogue) and the
// register-chained forward of the fragment-image engines in front of it (act_frag_body) — kernels_act.hip's two kernels, and the
// tail of kernels_solo.hip's launches in the rollout loop (the next step's select_action folded into the update's launch).
#pragma once
#include "c51.hpp"
#include "net.hpp"
#include "chain_net.hpp"

namespace frl {

// Everything after the head's output rows are in LDS (outb[row * op + column], rows r0 .. r0 + nv of learner p): the mode's
// action rule and, for the collectors, the exploration rule.  Shared by act_kernel (row-chunk forward, Wk parameters) and
// act_frag_kernel (register-chained forward, fragment-image parameters).
__device__ __forceinline__ void act_epilogue(const EngineDesc& D, const ActArgs& a, const NetDesc& N, g_cf theta, lds_f outb, int op,
                                             int nv, int r0, int p, int nout) {
    // device-side draws of this launch: Philox keyed like draw_kernel's (seed, learner), one counter value per launch
    const unsigned long long key = D.seed + 0x9E3779B97F4A7C15ull * (p + 1);
    auto normal_at = [&](unsigned stream, unsigned e) {
        float n0, n1;
        normal2(philox4x32_10(a.rng_counter, stream, e, key), n0, n1);
        return n0;
    };
    // epsilon-greedy on the greedy index of row r (DQN.py:307-310: np.random.rand() < epsilon -> np.random.randint(action_dim))
    auto eps_greedy = [&](int row, int best, int n_act) {
        if (a.env_out && a.explore == EXPL_EPS_GREEDY) {
            const Philox4 u = philox4x32_10(a.rng_counter, 0x9000u, (unsigned)row, key);
            if (u01(u.z) <= a.epsilon) best = (int)uniform_index(u, (unsigned)n_act);      // u01 is (0,1]
        }
        return best;
    };
    if (a.mode == ACTM_ARGMAX && D.c51_atoms && D.algo == ALGO_DQN) {      // argmax_a sum_i z_i p_i(s, a) (DQN_with_tricks.py:122-130)
        const float dz = (D.c51_vmax - D.c51_vmin) / (float)(D.c51_atoms - 1);
        const int lb = c51_combine(outb, op, nv, D.n_discrete, D.c51_atoms, D.dueling != 0);
        for (int r = threadIdx.x; r < nv; r += kWG) {
            int best = 0;
            float mx = 0.f;
            for (int j = 0; j < D.n_discrete; ++j) {
                const float q = c51_q(outb + r * op + lb + j * D.c51_atoms, D.c51_atoms, D.c51_vmin, dz, nullptr);
                if (j == 0 || q > mx) { mx = q; best = j; }
            }
            best = eps_greedy(r0 + r, best, D.n_discrete);
            a.out[(size_t)p * a.n_rows + r0 + r] = (float)best;
            if (a.env_out) a.env_out[(size_t)p * a.n_rows + r0 + r] = (float)best;
        }
        return;
    }
    if (a.mode == ACTM_ARGMAX) {
        const bool duel = D.dueling && D.algo == ALGO_DQN;       // Q = V + A - mean(A) (DQN_with_tricks.py:79): head = [V ; A]
        for (int r = threadIdx.x; r < nv; r += kWG) {
            lds_cf o = outb + r * op;
            const int nq = duel ? nout - 1 : nout;
            float mean = 0.f;
            if (duel) { for (int j = 0; j < nq; ++j) mean += o[1 + j]; mean /= (float)nq; }
            int best = 0;
            float mx = duel ? (o[0] + o[1]) - mean : o[0];
            for (int j = 1; j < nq; ++j) {         // first maximum wins, like torch.argmax
                const float v = duel ? (o[0] + o[1 + j]) - mean : o[j];
                if (v > mx) { mx = v; best = j; }
            }
            best = eps_greedy(r0 + r, best, nq);
            a.out[(size_t)p * a.n_rows + r0 + r] = (float)best;
            if (a.env_out) a.env_out[(size_t)p * a.n_rows + r0 + r] = (float)best;
        }
        return;
    }
    if (a.mode == ACTM_CAT_SAMPLE) {        // PPO_with_tricks.py:249-251 (torch single-draw multinomial)
        for (int r = threadIdx.x; r < nv; r += kWG) {
            float mx = outb[r * op];
            for (int j = 1; j < nout; ++j) mx = fmaxf(mx, outb[r * op + j]);
            float sum = 0.f;
            for (int j = 0; j < nout; ++j) sum += expf(outb[r * op + j] - mx);
            const size_t row = (size_t)p * a.n_rows + r0 + r;
            int best = 0;
            float bestv = -1.f, pbest = 0.f, psum = 0.f;
            for (int j = 0; j < nout; ++j) {
                const float pj = expf(outb[r * op + j] - mx) / sum;
                psum += pj;
                // q ~ Exp(1): injected, or -log(u) from the launch's Philox stream
                const float qj = a.device_eps ? -logf(u01(philox4x32_10(a.rng_counter, 0x9100u, (unsigned)((r0 + r) * nout + j), key).x))
                                              : a.eps[row * nout + j];
                const float v = pj / qj;
                if (v > bestv) { bestv = v; best = j; pbest = pj; }
            }
            a.out[row] = (float)best;
            if (a.env_out) a.env_out[row] = (float)best;
            if (a.out_logp) a.out_logp[row] = D.cat_logits ? (outb[r * op + best] - mx) - logf(sum)          // Categorical(logits=)
                                                           : logf(fminf(fmaxf(pbest / psum, 1.1920929e-07f), 1.f - 1.1920929e-07f));
        }
        return;
    }
    for (int e = threadIdx.x; e < nv * nout; e += kWG) {
        const int r = e / nout, c = e - r * nout;
        const size_t o = ((size_t)p * a.n_rows + r0 + r) * nout + c;
        float v = outb[r * op + c];
        if (a.mode == ACTM_SAC_SAMPLE || a.mode == ACTM_PPO_SAMPLE) {
            const float ls = fminf(fmaxf(theta[N.extra_off + c], -20.f), 2.f);
            const float sd = expf(ls);
            const float eps = a.eps ? a.eps[o] : (a.device_eps ? normal_at(0x9200u, (unsigned)((r0 + r) * nout + c)) : 0.f);
            const float u = v + sd * eps;
            if (a.mode == ACTM_SAC_SAMPLE) {
                v = tanhf(u);
            } else {
                const float du = u - v;
                if (a.out_logp) a.out_logp[o] = -(du * du) / (2.f * sd * sd) - ls - 0.91893853320467274178f;
                v = u;
            }
        }
        a.out[o] = v;
        if (a.env_out) {
            // action_ = clip(action * max_action [+ exploration noise], -max_action, max_action): TD3.py:412 (Gaussian),
            // SAC.py:528-533 (OU / Gaussian / none), PPO_with_tricks.py:529-530 (none)
            const float ma = a.max_action, sc = a.scale ? a.scale[p] : a.scale0;
            float x = v * ma;
            if (a.explore == EXPL_GAUSS) {
                x += sc * (normal_at(0x9300u, (unsigned)((r0 + r) * nout + c)) * (a.sigma * ma));
            } else if (a.explore == EXPL_OU) {
                float st = a.ou_state[o];
                if (a.flags && (a.flags[(size_t)p * a.n_rows + r0 + r] & 2)) st = 0.f;             // OUNoise.reset() at the episode's end
                st = st + (a.ou_theta * (0.f - st) + sqrtf(a.ou_dt) * a.ou_sigma * normal_at(0x9300u, (unsigned)((r0 + r) * nout + c)));
                a.ou_state[o] = st;
                x += (st * sc) * ma;
            }
            a.env_out[o] = fminf(fmaxf(x, -ma), ma);
        }
    }
}

// The same for the engines of the register-chained kernels (NetDesc::frag: parameters in fragment-image order): the net's
// images staged linearly into LDS, 64 rows per workgroup carried through the MLP in registers (device/chain_net.hpp), the
// head tile written to LDS for the shared epilogue.  Shape: 3 layers, hidden 128, <= 16 inputs and outputs, ReLU (what
// chained_shape() admits); grid = (ceil(n_rows / 64), learners).
// (the body as a device function: kernels_solo.hip runs it at the tail of its launches for the rollout loop's next select_action)
__device__ __forceinline__ void act_frag_body(const EngineDesc& D, const ActArgs& a, float* smem, int p, int r0) {
    const NetDesc& N = D.net[a.net];
    ChainNet C;
    C.init(smem);
    const int nv = min(64, a.n_rows - r0);
    const size_t off = (size_t)p * D.learner_stride + D.net_off[a.net];
    g_cf theta = as_global((a.use_target ? D.target : D.theta) + off);
    const int l0 = a.head * 3, K = a.in_dim;
    const int row = 16 * C.w + C.i16;
    f32x4 xb[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if (row < nv) {
        g_cf in = as_global(a.in + ((size_t)p * a.n_rows + r0 + row) * K);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * C.q + e < K) xb[0][e] = in[4 * C.q + e];
    }
    C.stage(theta, a.head, N.heads == 1 ? N.extra_n : 0);
    f32x4 z[1], h1[1][kHT], h2[1][kHT];
    C.forward<1>(xb, h1, h2, z);
    const bool th = (a.mode == ACTM_TANH || a.mode == ACTM_PPO_SAMPLE);
    lds_f outb = C.S.ea;                                               // [64][20]
    constexpr int op = 20;
#pragma unroll
    for (int r = 0; r < 4; ++r) outb[row * op + 4 * C.q + r] = th ? tanhf(z[0][r]) : z[0][r];
    __syncthreads();
    act_epilogue(D, a, N, theta, outb, op, nv, r0, p, N.L[l0 + 2].n);
}


}  // namespace frl
