// Descriptors shared by host code and kernels (plain structs; no device code here).
#pragma once
#include <stdint.h>

namespace frl {

constexpr int kMaxLayers = 6;     // twin critic = 2 x 3 layers in ONE net (one optimiser, one clip norm)
constexpr int kMaxAgents = 8;
constexpr int kMaxNets = 2 * kMaxAgents;

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };
enum Algo : int { ALGO_DQN = 0, ALGO_DDPG = 1, ALGO_TD3 = 2, ALGO_SAC = 3, ALGO_MADDPG = 4, ALGO_PPO = 5 };

// One nn.Linear in the engine-internal layout: Wk[k_pad][n_pad] (CONTRACTION-major for the forward pass: row =
// input feature, n contiguous), zero padded, then b[n_pad].  theta / target / m / v / grad / slab all use it;
// see device/tile.hpp (WMode) for why.
struct LayerDesc {
    int n, k;            // logical out / in features
    int n_pad, k_pad;    // padded to multiples of 16
    int w_off, b_off;    // float offsets inside the net's parameter block
};

// A net = the unit one optimiser (and one clip_grad_norm_) covers.
struct NetDesc {
    int n_layers;
    int size;            // floats per learner (multiple of 32)
    int n_params;        // logical parameter count (unpadded), for the algorithmic-bytes figure
    int hidden_act;      // Act after every layer but the head(s)
    int out_act;         // Act after the head
    int extra_off;       // state-independent log_std [extra_n] (Gaussian actors), else -1
    int extra_n;
    int heads;           // 1, or 2 for a twin critic (layers [0,n_layers/2) and [n_layers/2,n_layers))
    int n_shadow;        // parameter-only layers behind the forward ones: L[n_layers] = the sigma of a NoisyLinear head
    int frag;            // 1: every weight block of this net is stored in MFMA-fragment IMAGE order instead of Wk[k][n] — 16 x 16
                         // tiles, tile (n >> 4, k >> 4) at ((n >> 4) * (k_pad / 16) + (k >> 4)) * 256 floats, element (n & 15, k & 15)
                         // at weight_index() below (device/chain_net.hpp: what the register-chained kernels stage linearly and
                         // step from their accumulators).  Same offsets and sizes; biases / log_std unchanged.
    LayerDesc L[kMaxLayers];
};

// One head of the register-chained kernels' shape (device/chain_net.hpp): in <= 16 -> 128 -> 128 -> out <= 16, packed by
// build_net as W1[128 x 16] b1[128] W2[128 x 128] b2[128] W3[16 x 128] b3[16]
constexpr int kL1w = 0, kL1b = kL1w + 128 * 16, kL2w = kL1b + 128, kL2b = kL2w + 128 * 128, kL3w = kL2b + 128,
              kL3b = kL3w + 16 * 128, kHeadFloats = kL3b + 16;

constexpr int kDrawTableHost = 8192;     // ints per half of draw_indices' duplicate table (device/net.hpp: kDrawTable), batches of 257 .. 2048 rows

// One learner on sixteen workgroups (device/solo.hpp, kernels_solo.hip): the single-learner latency path
constexpr int kSoloWG = 16;              // workgroups per learner = 16-row tiles of a 256-row batch
constexpr int kSoloPartHost = 32;        // floats per workgroup of SoloArgs::part (device/solo.hpp: kSoloPart)
constexpr int kSoloMaxP = 16;            // learners per engine at sixteen workgroups each: every workgroup of a launch must be resident (flag hand-overs): 16 x 16 = 256 CUs (8 workgroups per learner, two tiles each: up to 32).
                                         // Measured (tools/small_pop_bench.py, TD3): 9 / 12 / 16 learners 82 / 90 / 101 us per learn() against 144 / 150 / 150 on the row-chunk kernels
constexpr int solo_lds_floats() { return 8 * 256 + 64 * 256 + 8 * 256 + 128 + 128 + 16 + 16 + 4 * 8 * 256 + 256 + 128; }

// ... and its form for wide first layers / heads of up to 32 outputs (device/solo_wide.hpp, kernels_solow.hip): W1 stays in the block
constexpr int kSoloWMaxKB = 26;          // first layer: <= 416 input columns
constexpr int kSoloWActorBase = 13;      // multi-agent engines: tile of the LDS row images where the updating agent's own observation rows start (the joint rows: tile 0);
                                         // both first layers then have at most 13 k-tiles
constexpr int solow_lds_floats() { return 64 * 256 + 2 * 8 * 256 + 128 + 128 + 32 + 32 + 4 * 8 * 256 + kSoloWMaxKB * 256 + 3 * 256 + 256 + 4 * 2 * 256 + 16 * 32 + 16 * 48 + 128; }

// The K-sliced chained family (device/chain_wide.hpp)
constexpr int kWideSliceKB = 4;          // k-blocks of W1 per streamed slice (4 x 8 tiles x 1 KB = 32 KB)
constexpr int kWideSlice = kWideSliceKB * 8 * 256;
constexpr int kWideMaxKB1 = 26;          // first layer: <= 416 input columns
constexpr int kWideMaxKT = 13;           // k-tiles of dW1 one wave owns
constexpr int kWideApitch = 32;          // floats per row in the action scratch arrays (all agents' actions: <= 32 columns)
constexpr int kWideScratchPerRow = kWideApitch + 4 + 3 * 128;  // floats of scratch per batch row next to the two row copies (WideScratch)
constexpr int wide_lds_floats() { return 8 * 8 * 256 + 2 * 8 * 256 + 2 * kWideSlice + 128 + 128 + 32 + 32 + 64; }
// ... at hidden 256 (device/chain_wide16.hpp: every layer streamed)
constexpr int kWide16ScratchPerRowHost = kWideApitch + 4 + 32 + 5 * 256;      // Wide16Scratch + the actor stage's third critic tensor
constexpr int wide16_lds_floats_host() { return 16384 + 2 * 16 * 256 + 3 * 16 * 256 + 256 + 256 + 32 + 32 + 192; }

// float index of W[out n][in k] inside a layer's weight block (host and device)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int weight_index(const NetDesc& N, const LayerDesc& L, int n, int k) {
    if (!N.frag) return k * L.n_pad + n;
    const int q = (k & 15) >> 2;
    return ((n >> 4) * (L.k_pad >> 4) + (k >> 4)) * 256 + ((q * 16 + ((n & 15) ^ q)) << 2) + (k & 3);
}

// Replay record (one transition of ALL agents, 128-byte aligned stride):
//   [ obs_0..obs_{n-1} | act_0..act_{n-1} | rew_0..rew_{n-1} | done_0..done_{n-1} | next_obs_0.. | extra ]
struct RecordDesc {
    int n_agents;
    int stride;          // floats, multiple of 32 (128 B)
    int width;           // floats actually used
    int obs_total, act_total, extra;
    int obs_off[kMaxAgents], obs_dim[kMaxAgents];
    int act_off[kMaxAgents], act_dim[kMaxAgents];    // act_dim = A' (1 for discrete)
    int rew_off, done_off;
    int nobs_off[kMaxAgents];
    int extra_off;
};

// Per-learner statistics slots written by the update kernels.
enum Stat : int {
    ST_CRITIC_LOSS = 0, ST_ACTOR_LOSS = 1, ST_ALPHA_LOSS = 2, ST_ALPHA = 3, ST_CRITIC_GNORM = 4,
    ST_ACTOR_GNORM = 5, ST_ENTROPY = 6, ST_COUNT = 8
};

struct EngineDesc {
    int algo, P, n_agents, hidden, rc;       // rc = batch rows per LDS chunk (multiple of 16)
    int capacity;                            // ring rows per learner
    int batch_max;
    int n_nets;
    int learner_stride;                      // floats per learner in theta/target/m/v/grad
    int net_off[kMaxNets];
    NetDesc net[kMaxNets];                   // MADDPG: net 2i = actor_i, 2i+1 = critic_i; else 0 actor/Q, 1 critic
    RecordDesc rec;
    // device pointers
    float* theta;
    float* target;
    float* m;
    float* v;
    float* grad;          // reduced gradients (sum of the partial slabs)
    float* slab;          // [P][S][learner_stride] per-row-chunk partial gradients
    float* act_spill;     // [P][n_agents][S][2][rc][hidden + 4] the actor's hidden activations of a row chunk, parked in HBM while
                          // the critic pass of ac_actor_kernel reuses h1 / h2 (instead of a second actor forward)
    float* part;          // [P][n_agents][S][4] per-row-chunk partial sums {loss, entropy, -, -}
    int S;                // gradient slabs per unit = workgroups per unit = ceil(ceil(batch_max / rc) / cps)
    int cps;              // consecutive row chunks one gradient workgroup works through (summing into its slab)
    float* gsq;           // [P][n_agents][Gmax] per-workgroup sum of squared gradients (reduce -> adam)
    int Gmax;             // workgroups per net in the reduce/adam launches
    float* replay;        // [P][capacity][rec.stride]
    int* idx;             // [P][n_agents][batch_max] sampled row indices
    float* noise;         // [P][n_agents][noise_sets][batch_max][act_max] standard-normal draws: set 0 = TD3 policy noise /
                          // SAC eps', set 1 = SAC eps; MATD3: set j = the policy noise on agent j's target action
    int noise_sets;       // max(2, n_agents)
    float* stats;         // [P][n_agents][ST_COUNT]
    int* steps;           // [P][kMaxNets + 1] Adam step counters (+1: SAC alpha)
    int* ticket;          // [P + 1] arrival counters in dqn_fused_kernel: a learner's workgroups; [P]: the learners (zero between launches)
    float* alpha;         // [P][4]: log_alpha, m, v, alpha (SAC)
    unsigned long long seed;
    int act_max;          // max act_dim over agents (row pitch of `noise`)
    // LDS carve parameters (must match between host lds_bytes() and device carve_lds())
    int lds_kin_pad, lds_out_pad, lds_batch_pad, lds_act_pad;
    int lds_hbufs;        // hidden-activation buffers in the carve: 2, or 1 when no head of any net has more than one hidden layer (DQN's Q-net)
    int n_discrete;       // DQN: number of discrete actions (0 otherwise)
    float* isw;           // [P][batch_max] PER importance weights of the current sample (DQN_with_tricks.py:276-279)
    float* td_err;        // [P][batch_max] TD errors Q(s,a) - y left by the last DQN learn (PER priorities)
    // NoisyLinear head (DQN_file/Noisy_net.py:17-76): head layer = mu, shadow layer = sigma; every forward of the reference
    // draws fresh factorised noise, so learn() materialises up to three effective parameter sets (online on s', target on
    // s', online on s) into theta_eff before the gradient kernel; noisy_eps[P][3][2*(k_pad + n_pad)] holds, per set,
    // eps_in / eps_out of sub-layer 0 then of sub-layer 1 (Dueling: V then A)
    int noisy;
    int noisy_split;      // head rows [0, noisy_split) belong to sub-layer 0 (Dueling's V), the rest to sub-layer 1
    float* theta_eff;     // [P][3][learner_stride]
    float* noisy_eps;
    int c51_atoms;        // > 0: Categorical DQN head (DQN_with_tricks.py:82-158): action_dim x atoms logits (+ atoms for Dueling's V)
    float c51_vmin, c51_vmax;
    int dueling;          // DQN: head = [V ; A] (1 + n_discrete outputs), Q = V + A - mean(A) (DQN_with_tricks.py:60-79)
    int cat_logits;       // PPO discrete: Categorical(logits=l3) (PPO_file/PPO.py:176,257: log-softmax, unclamped) instead of
                          // PPO_with_tricks.py's Categorical(probs=softmax(l3)) (probabilities clamped to [eps, 1-eps])
    int beta_actor;       // PPO: the actor is Actor_Beta (head = [alpha_layer ; beta_layer], 2*act_dim outputs)
    // Batch_ObsNorm (Normalization_batch_size, PPO_file/normalization.py:53-84): per learner
    // [1 + 3*O] floats = {n, mean[O], S[O], std[O]}; obs_norm_on switches every gather / act to
    // (x - mean) / (std + 1e-8)
    float* obsnorm;
    int obs_norm_on;
    // multi-agent (MADDPG.py:155-156,194-196): statistics per agent, and one VERSION per updating agent — inside one learn()
    // agent i's sample() updates every agent's statistics before agent i's update reads them.  Address of (learner p,
    // version i, agent j) = obsnorm + ((p*n + i)*n + j) * obsnorm_w, obsnorm_w = 1 + 3*max obs_dim; the last version is the
    // state carried to the next call (and what select_action reads).  n = 1: exactly the single-agent layout.
    int obsnorm_w;
    // The K-sliced chained family (device/chain_wide.hpp, kernels_criticw.hip / kernels_actorw.hip): per-(learner, agent) scratch —
    // target / policy actions, TD targets, dQ/da, the first-layer deltas of the batch (exchange-image order), the actor's
    // hidden activations between its forward and backward passes — [P][n_agents][wide_unit] floats, L2-resident
    int wide;             // 1: this engine's actor-critic updates run on that family (every net in fragment-image order); 2: its
                          // hidden-256 form (kernels_criticx.hip / kernels_actorx.hip)
    int wide_bm;          // batch_max rounded up to 64 rows
    int wide_xp, wide_op; // row pitches of the scratch's critic-input rows / observation copies (the padded first-layer widths)
    int wide_unit;        // floats per (learner, agent): (wide_xp + n_agents * wide_op + kWideScratchPerRow) * wide_bm + 128, rounded up to 64
    float* wide_scr;
    int solow;            // > 0 (16): DDPG / TD3 / SAC updates of this engine run on kernels_solow.hip (single agent, hidden 128, first layers of up to 416
                          // columns, heads of up to 32 outputs, batches of up to 256 rows, <= kSoloMaxP learners); every net in fragment-image order
    int solo;             // > 0 (the workgroups per learner: 16 / 8): DDPG / TD3 / SAC updates of this engine run on kernels_solo.hip (<= kSoloMaxP learners of the narrow standard
                          // shape, sixteen workgroups per learner); parameters in fragment-image order like the chained family's
};

// Hyper-parameters of one learn() call (passed by value to the kernels).
struct LearnArgs {
    int batch;
    int size;             // rows currently valid in every ring (for device-side index draws)
    int device_rng;       // 1: draw indices / noise on the device (Philox); 0: use desc.idx / desc.noise as uploaded
    int do_actor;         // TD3: total_it % policy_freq == 0; others 1
    float gamma, tau;
    float actor_lr, critic_lr, alpha_lr;
    float adam_eps, beta1, beta2;
    float critic_wd;      // DDPG.py supplement weight_decay (L2-in-grad), 0 otherwise
    float clip_norm;      // 0.5, or <= 0 for no clipping (DQN)
    float policy_noise, noise_clip, max_action, policy_noise_scale;   // TD3
    int use_policy_noise;
    float target_entropy; // SAC
    unsigned long long rng_counter;   // device_rng: Philox counter of this call (host increments)
    int p0, p_count;      // this launch covers learners [p0, p0 + p_count): frl_learn pipelines two halves of a population
    int double_dqn;       // DQN trick['Double']: a* = argmax_a Q(s',a), y uses Q_target(s', a*) (DQN_with_tricks.py:263-265)
    int use_isw;          // DQN trick['PER']: loss = mean(w * td^2) with the weights in desc.isw (:276-278)
    int dqn_split;        // dqn_fused_kernel: workgroups per learner (its 64-row chunks dealt round-robin)
    int stagger;          // kernels_critic2: s_sleep(32) units (2 k cycles) between the start phases of the first round's workgroups (0: none)
    int stagger_groups;   // ... how many start phases (a power of two), and how many workgroups make the first round (the device's CUs)
    int stagger_wgs;
    int fuse_actor;       // kernels_solow.hip: this call's critic AND actor stage run in one launch (solow_step_*); frl_learn skips the actor launch
    int huber;            // TD loss: 0 F.mse_loss (every hot-path loss of the reference), 1 Huber with `huber_delta`
    float huber_delta;    // (the reference's huber_loss, MAPPO_file/MAPPO.py:273-276: e^2/2 if |e| <= d else d(|e| - d/2), mean)
};

}  // namespace frl
