// Argument structs and prototypes of every kernel of the library.  Each kernels_*.hip is its own translation unit
// (compiled in parallel by freerl_amd/_native.py:build; no relocatable device code needed: a kernel's device code is
// complete in the unit that defines it); the host API (frl_api.hip) launches them through these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "frl_desc.h"

#ifndef FRL_GRAD_WGS
#define FRL_GRAD_WGS 2      // gradient-kernel workgroups per CU the register budget is sized for (frl_create picks rc to match)
#endif

namespace frl {

// ---- kernels_act.hip
enum ActMode : int {
    ACTM_RAW = 0,        // head output as is (Q values, V(s), Gaussian mean before squashing)
    ACTM_ARGMAX = 1,     // DQN greedy action (index as float)
    ACTM_TANH = 2,       // tanh(head): deterministic actors, SAC/PPO evaluate_action
    ACTM_SAC_SAMPLE = 3, // tanh(mean + std*eps)
    ACTM_PPO_SAMPLE = 4, // a = tanh(head) + std*eps ; logp per dimension
    ACTM_CAT_SAMPLE = 5  // Categorical(softmax(head)).sample() = argmax(p / q), q ~ Exp(1); logp of the draw
};

struct ActArgs {
    int net;             // net index
    int use_target;
    int mode;
    int n_rows;          // rows per learner
    int head;            // critic head (0/1) for ACTM_RAW on twin critics
    int in_dim;          // logical input width (obs dim, or obs+act for critics)
    int normalize;       // apply Batch_ObsNorm to the first obs_dim input columns (policy / value nets on raw obs)
    const float* in;     // [P][n_rows][in_dim] dense
    const float* eps;    // [P][n_rows][out_dim] standard normal draws, or nullptr
    float* out;          // [P][n_rows][out_dim]   (ARGMAX: out_dim = 1)
    float* out_logp;     // [P][n_rows][out_dim] (PPO sample) or nullptr
    // ---- exploration folded into the launch (frl_act_explore, the rollout collectors).  env_out == nullptr: none of it runs.
    int explore;         // ExploreKind
    int device_eps;      // 1: the sampling modes draw their own N(0,1) / Exp(1) variates (Philox) instead of reading `eps`
    float epsilon;       // EXPL_EPS_GREEDY: P(random action)                                   (DQN.py:307-310)
    float sigma;         // EXPL_GAUSS: std of the action noise in units of max_action           (TD3.py:412 gauss_sigma)
    float scale0;        // multiplier of the Gaussian noise / of the OU state when `scale` is nullptr (gauss_scale, OUNoise.scale)
    float max_action;
    float ou_theta, ou_sigma, ou_dt;   // EXPL_OU: x += theta*(0 - x) + sqrt(dt)*sigma*N(0,1)     (SAC.py:334-356)
    const float* scale;  // [P] per-learner multiplier (the reference decays it per episode, TD3.py:425-427, SAC.py:548-551) or nullptr
    float* ou_state;     // [P][n_rows][out_dim]
    const unsigned char* flags;   // [P][n_rows] bit 1: the row's episode ended on the previous step -> OU state reset first (SAC.py:546-547)
    float* env_out;      // [P][n_rows][out_dim] env-unit action = clip(a*max_action + noise, +-max_action); discrete: [P][n_rows] index
    unsigned long long rng_counter;
    const float* theta_alt;   // non-null: [P][NetDesc::size] Wk-layout copy of the net to read instead of theta / target (frag_to_wk_kernel)
};
enum ExploreKind : int { EXPL_NONE = 0, EXPL_EPS_GREEDY = 1, EXPL_GAUSS = 2, EXPL_OU = 3 };

// ---- kernels_per.hip
constexpr int kPerSetMax = 4096;      // rows of one per_set_kernel launch (its later-entry-wins scan stages them in LDS)
struct PerArgs {
    double* sum_tree;      // [P][2*cap-1]
    double* max_tree;      // [P][2*cap-1]
    int cap;               // leaves per learner
    int n;                 // entries in this launch
    const int* leaf;       // [P][n_pitch] buffer indices to write (per_set) / out: sampled indices (per_sample writes D.idx)
    int n_pitch;
    const float* prio;     // per_set: [P][n_pitch] priorities, or nullptr: use `fill` for every entry
    double fill;
    const int* size;       // [P] rows valid per learner
    // sampling
    const double* uniforms;   // [P][n] draws in [0,1) or nullptr (Philox)
    float* isw;               // [P][batch_max] importance weights out
    float* prio_out;          // [P][batch_max] float32 priorities of the sampled leaves
    double beta;
    unsigned long long rng_counter;
    const float* td;          // per_update: [P][batch_max] TD errors -> priority (|td| + eps)^alpha in float32
    float alpha, eps;
};

// ---- kernels_ppo.hip
struct PpoArgs {
    int horizon, minibatch, k_epochs, adv_norm;
    float gamma, lmbda, clip, ent_coef;
    float actor_lr, critic_lr, adam_eps, beta1, beta2, clip_norm;
    int optimizer;    // 0 torch Adam per net, 1 PPO.py's cautious AdamW (lr = actor_lr for both nets)
    // device scratch, per learner blocks of `horizon` floats
    float* td;        // [P][T] td_delta, then (after GAE) unused
    float* vs;        // [P][T] V(s)
    float* adv_raw;   // [P][T] GAE advantages before normalisation
    float* adv;       // [P][T] advantages used by the surrogate
    float* vtarget;   // [P][T]
    float* trace;     // [P][k_epochs * n_mb][2] per-minibatch (actor, critic) losses
    const int* perm;  // [P][k_epochs][T]
    const float* last_value;   // [P] (gae_mode 1)
    double gamma_d, lmbda_d;   // the float64 scan's discount and lambda (gae_mode 1)
};

// ---- kernels_replay.hip
struct GatherFields {
    int n_fields;
    int col0[8];
    int ncols[8];
    float* out[8];        // out[f][b][ncols[f]] dense
};

// Rollout commit: one launch writes the transitions of a vector step into the ring(s) from what is already on the device
// (the observations the policy saw, the actions it chose, their log-probs) plus the one block the host uploaded after
// env.step (next_obs, obs_next, reward, flags, ring rows), and advances the device copy of the current observations.
struct CommitArgs {
    float* obs_cur;               // [n][O] in: obs of the transition; out: obs_next (what the policy sees next)
    const float* store_act;       // [n][aout] action as stored by add() (policy output in (-1,1) / action index)
    const float* logp;            // [n][n_logp] or nullptr
    const int* row;               // [n] ring row of env i inside its learner's ring
    const float* next_obs;        // [n][O]
    const float* obs_next;        // [n][O]
    const float* reward;          // [n]
    const unsigned char* flags;   // [n] bit 0 terminated (the stored `done`), bit 1 episode ended, bit 2 adv_done (PPO)
    int n, E, O, aout, n_logp;
};

// ---- kernels_update.hip: reduce + clip + Adam.  which = 0: critic / Q-net, 1: actor.
struct AdamArgs {
    int which, ns, batch, soft, sac_alpha, G, p0;
    float lr, eps, beta1, beta2, wd, clip, tau, alpha_lr, target_entropy;
};
constexpr int kAdamVec = 8;                              // float4 per thread per workgroup (reduce_kernel / adam_kernel)
constexpr int kFusedThreads = 1024, kFusedVec = 12;      // adam_fused_kernel: one workgroup per net, gradient in registers
constexpr int kFusedVecWide = 18;                        // adam_fused_wide_kernel: nets up to 73 k parameters (element-major slab sums)

// kernels_act.hip
__global__ void act_kernel(const EngineDesc* __restrict__ Dp, ActArgs a);
__global__ void act_frag_kernel(const EngineDesc* __restrict__ Dp, ActArgs a);

// kernels_actor.hip
__global__ void ac_actor_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns);

// kernels_c51.hip
__global__ void c51_grad_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns);

// kernels_critic.hip
__global__ void ac_critic_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns);

// kernels_dqn.hip
__global__ void dqn_grad_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int ns);

// kernels_noisy.hip
__global__ void noisy_materialise_kernel(const EngineDesc* __restrict__ Dp, int set0, int n_sets, int target_mask);
__global__ void noisy_draw_kernel(const EngineDesc* __restrict__ Dp, int set0, int n_sets, unsigned long long counter);

__global__ void per_add_kernel(PerArgs a, const int* __restrict__ bucket, int P);
__global__ void per_set_kernel(const EngineDesc* __restrict__ Dp, PerArgs a);
__global__ void per_sample_kernel(const EngineDesc* __restrict__ Dp, PerArgs a);

// kernels_ppo.hip
__global__ void ppo_values_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a);
__global__ void ppo_gae_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a);
__global__ void ppo_gae_sb3_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a);
__global__ void ppo_perm_kernel(int* __restrict__ perm, int T, int Tpad, unsigned long long counter, unsigned long long seed);
__global__ void gae_dense_kernel(const float* __restrict__ delta, const float* __restrict__ adv_done, int T, float c, float* __restrict__ adv);
__global__ void ppo_update_kernel(const EngineDesc* __restrict__ Dp, PpoArgs a);

// kernels_replay.hip
__global__ void replay_scatter_kernel(float* __restrict__ ring, const float* __restrict__ staged, const long long* __restrict__ slots, int n, int width, int stride);
__global__ void replay_gather_kernel(const float* __restrict__ ring, const long long* __restrict__ idx, int B, int stride, GatherFields F);
__global__ void replay_read_kernel(const float* __restrict__ ring, long long row0, int n, int width, int stride, float* __restrict__ out);
__global__ void replay_commit_kernel(float* __restrict__ ring, RecordDesc rec, int capacity, CommitArgs c);
__global__ void replay_fill_kernel(float* __restrict__ ring, long long rows, RecordDesc rec, int n_discrete, unsigned long long seed);

// kernels_update.hip
__global__ void relayout_to_wk_kernel(const EngineDesc* __restrict__ Dp, float* scratch);      // scratch: [4][P][learner_stride]
__global__ void frag_to_wk_kernel(const EngineDesc* __restrict__ Dp, int net, int use_target, float* dst);   // dst: [P][NetDesc::size]
__global__ void draw_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, int want_noise);
__global__ void obsnorm_kernel(const EngineDesc* __restrict__ Dp, int batch, int all_rows, int p0);
__global__ void reduce_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a);
__global__ void adam_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a);
__global__ void adam_fused_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a);
__global__ void adam_fused_wide_kernel(const EngineDesc* __restrict__ Dp, AdamArgs a);
__global__ void soft_update_kernel(const EngineDesc* __restrict__ Dp, float tau, int p0);

// kernels_critic2.hip: the critic stage of DDPG / TD3 / SAC for one learner per workgroup (register-chained, Adam fused)
__global__ void ac_critic_v2_twin_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_v2_single_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_v2_twin_nv_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);      // (_nv: aligned next_obs / (reward, done) loads)
__global__ void ac_critic_v2_single_nv_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
// (round 2-5's four-wave form of the same stage, one wave per SIMD: FRL_CHAIN_WAVES=4, same-box A/B runs)
__global__ void ac_critic_v2w4_twin_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_v2w4_single_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
constexpr int critic2_lds_floats() { return 8 * 256 + 64 * 256 + 8 * 256 + 2 * 8192 + 128 + 128 + 16 + 16 + 256 * 4 + 3 * 256 + 64; }
constexpr int critic8_lds_floats() { return critic2_lds_floats() + 1024; }      // the eight-wave workgroups' carve (device/chain_net.hpp: chain8_lds_floats)

// kernels_criticw.hip / kernels_actorw.hip: the same two stages on the K-sliced chained design (device/chain_wide.hpp) for wide
// first layers / heads and multi-agent critics; h<critic heads>a<actor head tiles>
__global__ void ac_critic_wide_h1a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_wide_h1a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_wide_h2a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_wide_h2a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_actor_wide_a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_actor_wide_a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);

// kernels_criticx.hip / kernels_actorx.hip: ... and at hidden 256 (device/chain_wide16.hpp)
__global__ void ac_critic_x_h1a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_x_h1a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_x_h2a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_critic_x_h2a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_actor_x_a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_actor_x_a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);

// kernels_actor2.hip: the actor stage of DDPG / TD3 likewise
__global__ void ac_actor_v2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);
__global__ void ac_actor_v2w4_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a);

// ---- kernels_solo.hip: a single learner's DDPG / TD3 / SAC update on kSoloWG workgroups (device/solo.hpp)
struct SoloArgs {
    float* slab;            // [P][kSoloWG][slab_stride] partial gradients of the net being trained, fragment-image order
    float* part;            // [P][kSoloWG][8] per-workgroup partial sums: 0 loss / Q sum, 1 log-pi sum, 2 squared gradient norm
    unsigned* bar;          // [P][kSoloWG] "my slab is written" flags: the epoch of the last launch that wrote it
    int* err;               // [1] set to 1 by a barrier that timed out (a workgroup of the learner never arrived)
    unsigned bar_base;      // epoch of this launch = bar_base + kSoloWG (the host advances bar_base by kSoloWG per launch)
    int slab_stride;
    // The NEXT call's batch rows, drawn by a spare workgroup of THIS launch on a CU the learner's sixteen leave idle (round 6: the
    // draw was 6.5 us at the head of every critic launch's critical path).  Two slots per learner of kSoloPre ints, alternating by
    // launch: [0..1] the Philox counter the rows were drawn for, [2] the ring size, [3] the batch, [8 ..) the rows.  A launch uses
    // pre_read only if all three match its own arguments — same draw_indices(), same bits — and draws for itself otherwise.
    const int* pre_read;    // [P][kSoloPre] or NULL
    int* pre_write;         // [P][kSoloPre] or NULL: the grid carries p_count extra workgroups behind the learners'
    unsigned long long pre_counter;   // the counter the next frl_learn call will take if nothing else draws in between
    int tiles;              // kernels_solow.hip: row tiles = slabs = flags per unit (16: batches of up to 256 rows; 64: MADDPG's 1024); slab / bar are [units][tiles]
    unsigned* bar2;         // kernels_solow.hip, fused policy step: [units][64] "critic stepped" flags of the unit's helper workgroups (or NULL)
    int row_wgs;            // kernels_solow.hip: workgroups per unit that own row tiles (tiles / row_wgs each; flags are polled for these)
    int update_wgs;         // kernels_solow.hip: workgroups per learner in the grid (16 with row tiles + helpers that only take a share of the update); `part` is [P][update_wgs][..]
};
constexpr int kSoloPre = 8 + 256;
// The rollout loop's step folded into these launches (frl_rollout; what dqn_fused_kernel does for DQN): `head` — Buffer.add of the
// vector step's transitions at the start of the critic launch (every workgroup of the learner writes the same ring rows: it
// samples from them next) — and `tail` — at the end of the step's LAST launch: the device copy of the current observations moves
// on to obs_next, select_action + the exploration rule run on it (kernels_act.hip's own act_frag_body: the same draws, bit for
// bit, as the separate act launch), the env actions land in host memory and a host-visible word is flagged.
struct SoloStepArgs {
    int head, tail, act;          // act: the tail also selects the next actions (0 on a rollout call's last step)
    CommitArgs c;                 // c.obs_cur / store_act / row / next_obs / obs_next / reward / flags, n = P * E
    ActArgs act_args;             // as launch_act would have built them for the separate launch (in = c.obs_cur)
    int* ticket;                  // [1] learners that have finished their tail (zero between launches)
    int* done_flag;               // host-visible word (or nullptr): set to done_value once every learner's env actions are out
    int done_value;
    unsigned* bar2;               // [P][kSoloWG] "my slice of the actor is stepped" flags (the tail reads the whole actor)
    const int* go_flag;           // pre-armed launch (DqnStepArgs::go_flag; on the pool's second stream): the critic launch draws its rows, waits
    int go_value;                 // until dev_cnt has reached dev_wait (every workgroup of the step in front has left), stages its first image, then
    int* dev_cnt;                 // spins on this host-visible word until it holds go_value (-1: the host gave the step up, nothing is touched).
    int dev_wait;                 // dev_cnt: device word (or nullptr) every workgroup of a folded step's launches adds one to on its way out
};
__global__ void solo_critic_twin_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st);
__global__ void solo_critic_single_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st);
__global__ void solo_actor_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st);
// ... with eight workgroups per learner, two row tiles each (populations of 17 .. 32 learners)
__global__ void solo_critic_twin_w8_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st);
__global__ void solo_critic_single_w8_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st);
__global__ void solo_actor_w8_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s, SoloStepArgs st);
// kernels_solow.hip: the same decomposition for wide first layers / heads of up to 32 outputs (device/solo_wide.hpp), behind draw_kernel;
// h<critic heads>a<actor head tiles>
__global__ void solow_critic_h1a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_critic_h1a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_critic_h2a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_critic_h2a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_actor_a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_actor_a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
// ... MADDPG / MATD3: a unit = (learner, agent)
__global__ void solow_critic_ma_h1a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_critic_ma_h1a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_critic_ma_h2a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_critic_ma_h2a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_actor_ma_a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_actor_ma_a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
// ... a policy step (critic + actor stage) of single-agent engines with helper workgroups in one launch
__global__ void solow_step_h1a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_step_h1a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_step_h2a1_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
__global__ void solow_step_h2a2_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, SoloArgs s);
// kernels_dqn2.hip: draw + DQN / Double-DQN update + Adam + soft update of one learner in one launch
constexpr int kDqn2Batch = 256;
constexpr int dqn2_lds_floats() { return 4 * 8 * 256 + 8 * 4 * 256 + 4 * 256 + 2 * (128 + 16) + 64 + 64 * 16 + 2 * kDqn2Batch; }
// The rollout loop's neighbours of learn() folded into the same launch (frl_rollout, DQN.py:294-339): add() of the vector
// step's transitions before the sample, select_action + epsilon-greedy on the post-update net for the next step after it.
struct DqnStepArgs {
    int commit, act;              // which of the two run (0 / 0: plain learn())
    int E, O;                     // envs per learner, obs_dim
    const float* obs_cur;         // [P*E][O] obs of the transition
    const float* store_act;       // [P*E] action index as stored by add()
    const int* row;               // [P*E] ring row of env i inside its learner's ring
    const float* next_obs;        // [P*E][O] observation the transition ended in
    const float* obs_next;        // [P*E][O] observation the policy sees next (the reset observation after an episode end)
    const float* reward;          // [P*E]
    const unsigned char* flags;   // [P*E] bit 0 terminated
    float* act_out;               // [P*E] explored action index (what the next add() stores)
    float* env_out;               // [P*E] the same, for the env
    float epsilon;
    unsigned long long act_counter;   // Philox counter of the act draw (act_kernel's stream 0x9000)
    int* done_flag;               // host-visible word (or nullptr): set to done_value once env_out is written, so that the
    int done_value;               // host can pick the actions up without waiting for the launch to retire
    // PRE-ARMED launch (frl_rollout, small populations): the launch is enqueued a vector step AHEAD on the pool's second stream, so it
    // starts while the previous step's launch is still running: it draws the batch's rows (launch arguments only), waits on the DEVICE
    // word dev_done for the previous launch's done_value (that launch's parameters are final), stages both nets, then spins on the
    // host-visible word go_flag until the host — env stepped, block filled — sets it to go_value: launch latency, the gap between two
    // launches of a stream, the draw and the weight staging all run under the previous launch and the host's turn.  -1 in go_flag: the
    // host gave the step up, the launch returns without having touched anything.
    const int* go_flag;
    int go_value;
    int* dev_done;                // device word (or nullptr): set to done_value together with done_flag
    int dev_wait;                 // pre-armed: the done_value of the launch in front
    int* err;                     // pinned word (or nullptr): set to 2 when a pre-armed launch gives up after 2 s of waiting (NOT when the host cancels it):
};                                // its update and the next actions were dropped — frl_rollout / frl_sync report FRL_ERR_STATE instead of stepping the envs on stale actions
__global__ void dqn_fused_kernel(const EngineDesc* __restrict__ Dp, LearnArgs a, DqnStepArgs s);

// kernels_ppo2.hip: the on-chip variant of ppo_update_kernel, <first-layer k-blocks, hidden activation>
__global__ void ppo_update_v2_k1_relu(const EngineDesc* __restrict__ Dp, PpoArgs a);
__global__ void ppo_update_v2_k2_relu(const EngineDesc* __restrict__ Dp, PpoArgs a);
__global__ void ppo_update_v2_k1_tanh(const EngineDesc* __restrict__ Dp, PpoArgs a);
__global__ void ppo_update_v2_k2_tanh(const EngineDesc* __restrict__ Dp, PpoArgs a);
constexpr int ppo2_lds_floats(int k0b) { return 8 * k0b * 256 + 64 * 256 + 8 * 256 + 2 * 8192 + 128 + 128 + 16 + 16 + 96; }

}  // namespace frl
