/* freerl_hip.h — C ABI of the MI355X-native replay-sample + batched-update engine.
 *
 * FreeRL (the reference) is pure Python and has NO plugin / FFI interface: its
 * boundary for this path is the duck-typed class surface `Buffer.add / Buffer.sample /
 * Agent.update_* / <ALGO>.select_action / <ALGO>.learn`.  This header is what a ctypes (or
 * cffi / pybind) binding on the reference side binds in order to keep that surface and run it
 * on hand-written HIP kernels; `freerl_amd/` is exactly such a binding and INTEGRATION.md shows
 * the stub.  Each entry point names the reference interface it replaces (paths relative to
 * the FreeRL tree).
 *
 * Conventions: opaque handle, `int` status (0 = FRL_OK), no exceptions or torch types across
 * the boundary, plain pointers + sizes, caller-allocated output buffers.  One engine = one HIP
 * stream; an engine is thread-compatible (one thread at a time).  "host" pointers are ordinary
 * process memory, "device" pointers are HIP device memory on the engine's GPU.
 *
 * An engine holds P independent learners ("population": seeds / env-instance sets on this
 * GPU); P = 1 is the reference-compatible single learner.  Every call below
 * that takes [P][...] arrays processes all learners in ONE kernel launch.
 */
#ifndef FREERL_HIP_H
#define FREERL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRL_MAX_AGENTS 8
#define FRL_STAT_COUNT 8

enum frl_status { FRL_OK = 0, FRL_ERR_INVALID = 1, FRL_ERR_HIP = 2, FRL_ERR_NO_DEVICE = 3, FRL_ERR_STATE = 4 };

enum frl_algo {
    FRL_ALGO_REPLAY_ONLY = -1, /* stand-alone Buffer.py replacement, no networks */
    FRL_ALGO_DQN = 0,          /* DQN_file/DQN.py:62-138 */
    FRL_ALGO_DDPG = 1,         /* DDPG_file/DDPG_simple.py:100-179, DDPG.py */
    FRL_ALGO_TD3 = 2,          /* TD3_file/TD3.py:150-256 */
    FRL_ALGO_SAC = 3,          /* SAC_file/SAC.py:171-282 */
    FRL_ALGO_MADDPG = 4,       /* MADDPG_file/MADDPG_simple.py:107-210 */
    FRL_ALGO_PPO = 5           /* PPO_file/PPO_with_tricks.py:211-374, PPO.py */
};

enum frl_activation { FRL_ACT_NONE = 0, FRL_ACT_RELU = 1, FRL_ACT_TANH = 2 };

/* which copy of a net's parameters frl_params_get/set addresses.  FRL_PARAM_GRAD is a debugging view of the reduced
 * gradient block: PPO engines, noisy DQN engines and nets wider than the fused Adam kernel materialise it; the other
 * off-policy updates keep the reduced gradient in registers and leave this block untouched. */
enum frl_param_kind { FRL_PARAM_ONLINE = 0, FRL_PARAM_TARGET = 1, FRL_PARAM_ADAM_M = 2, FRL_PARAM_ADAM_V = 3, FRL_PARAM_GRAD = 4 };

/* frl_act modes */
enum frl_act_mode {
    FRL_ACT_RAW = 0,        /* head output: Q values (DQN.py:83), V(s) (PPO_with_tricks.py:304-305), Gaussian mean */
    FRL_ACT_ARGMAX = 1,     /* DQN.select_action, DQN.py:70-84 */
    FRL_ACT_TANHHEAD = 2,   /* TD3/DDPG/MADDPG select_action (TD3.py:163-170), SAC/PPO evaluate_action */
    FRL_ACT_SAC_SAMPLE = 3, /* SAC.select_action, SAC.py:192-198: tanh(mean + std*eps) */
    FRL_ACT_PPO_SAMPLE = 4, /* PPO.select_action, PPO_with_tricks.py:234-255: a = mean + std*eps, per-dim log-prob */
    FRL_ACT_CAT_SAMPLE = 5  /* discrete PPO (:249-251): Categorical(softmax).sample() = argmax(p/q), q ~ Exp(1) in eps
                               [P][n_rows][n_actions]; out / logp are [P][n_rows] (index as float, log-prob of the draw) */
};

/* OR into `mode`: skip Batch_ObsNorm for this call — the reference's evaluate_action does not normalise
 * although select_action does (SAC.py:200-204 vs :194-195; DDPG.py:173-181 vs :165-166) */
#define FRL_ACT_NO_OBSNORM 0x100

/* per-(learner, agent) statistics written by frl_learn; index into stats[.][FRL_STAT_COUNT] */
enum frl_stat {
    FRL_STAT_CRITIC_LOSS = 0, FRL_STAT_ACTOR_LOSS = 1, FRL_STAT_ALPHA_LOSS = 2, FRL_STAT_ALPHA = 3,
    FRL_STAT_CRITIC_GNORM = 4, FRL_STAT_ACTOR_GNORM = 5, FRL_STAT_ENTROPY = 6
};

typedef struct frl_engine frl_engine;

/* Constructor arguments; mirrors `<ALGO>(dim_info, is_continue, lrs, buffer_size, device, ...)`
 * (DQN.py:63-68, TD3.py:151-161, SAC.py:172-190, MADDPG_simple.py:108-120, PPO_with_tricks.py:212-232). */
typedef struct frl_config {
    int algo;                     /* enum frl_algo */
    int n_learners;               /* P >= 1 */
    int n_agents;                 /* 1, or the MADDPG agent count (<= FRL_MAX_AGENTS) */
    int obs_dim[FRL_MAX_AGENTS];
    int act_dim[FRL_MAX_AGENTS];  /* continuous: action width; discrete (DQN): number of actions */
    int discrete;                 /* 1: actions are stored as ONE float index (Buffer.py:3-9 act_dim vs action_dim) */
    int hidden;                   /* 128 = the reference's hard-coded width (DQN.py:38, TD3.py:53 ...) */
    int hidden_act;               /* FRL_ACT_RELU, or FRL_ACT_TANH for PPO trick['tanh'] */
    int twin_critic;              /* Critic_TD3 / SAC Critic: l1..l6 in one module */
    int capacity;                 /* replay rows per learner (PPO: horizon) */
    int batch_max;                /* largest batch / minibatch a learn call will use */
    int extra_cols;               /* extra record columns (PPO: act_dim log-probs + 1 adv_done) */
    int actor_dist;               /* PPO, discrete: 0 Categorical(probs=softmax(l3)) (PPO_with_tricks.py:107-118,333-336), 2
                                     Categorical(logits=l3) (PPO_file/PPO.py:78-90,176,257: no clamp at float eps).
                                     PPO, continuous: 0 Gaussian `Actor` (PPO_with_tricks.py:79-108), 1 `Actor_Beta` (:120-151):
                                     the actor's head is [alpha_layer ; beta_layer] = 2*act_dim outputs, no log_std */
    int dueling;                  /* DQN trick['Dueling'] (DQN_with_tricks.py:60-79): the head is [V ; A] = 1 + n_actions outputs and
                                     Q = V + A - mean(A) */
    int noisy;                    /* DQN trick['Noisy'] (Noisy_net.py:17-76): the head (l2, or Dueling's V and A) is NoisyLinear: parameters
                                     mu + sigma, fresh factorised noise at every forward */
    int c51_atoms;                /* DQN trick['Categorical'] (DQN_with_tricks.py:82-158): atoms of the value distribution (51), 0 = off */
    float c51_vmin, c51_vmax;     /* its support [v_min, v_max] (-100, 100) */
    int device_id;
    uint64_t seed;                /* device Philox key (fast path only) */
} frl_config;

/* Column layout of one replay record (all agents of one transition, fp32):
 * [obs_0..|act_0..|rew_0..|done_0..|next_obs_0..|extra] — see DESIGN.md "Data layout". */
typedef struct frl_record_layout {
    int n_agents, width, stride;
    int obs_off[FRL_MAX_AGENTS], obs_dim[FRL_MAX_AGENTS];
    int act_off[FRL_MAX_AGENTS], act_dim[FRL_MAX_AGENTS];
    int rew_off, done_off;
    int next_obs_off[FRL_MAX_AGENTS];
    int extra_off, extra;
} frl_record_layout;

/* Arguments of one learn() call: `<ALGO>.learn(batch_size, gamma, tau, ...)`
 * (DQN.py:104, DDPG_simple.py:137, TD3.py:189, SAC.py:222, MADDPG_simple.py:165). */
typedef struct frl_learn_args {
    int batch;                 /* min(len(buffer), batch_size) is the caller's job (DQN.py:95-96) */
    int do_actor;              /* TD3 / MATD3: total_it % policy_freq == 0 (TD3.py:224, MATD3_simple.py:236,245: actor step
                                  AND target update); else 1 */
    int use_policy_noise;      /* TD3 / MATD3 realize['policy_noise'] (FRL_ALGO_MADDPG + twin_critic + these two = MATD3_simple.py) */
    float gamma, tau;
    float actor_lr, critic_lr; /* DQN: critic_lr = Qnet_lr */
    float alpha_lr;            /* SAC Alpha, 1e-4 (SAC.py:155) */
    float adam_eps;            /* 1e-8; 1e-5 with PPO trick['adam_eps'] */
    float critic_weight_decay; /* DDPG.py:131-134 supplement; 0 otherwise */
    float clip_norm;           /* clip_grad_norm_ max_norm 0.5 (TD3.py:140); <= 0: none (DQN.py:56-59) */
    float policy_noise, noise_clip, max_action, policy_noise_scale;   /* TD3.py:196-198 */
    float target_entropy;      /* SAC: -act_dim (SAC.py:160) */
    int double_dqn;            /* DQN trick['Double'] (DQN_with_tricks.py:263-265) */
    int per;                   /* DQN trick['PER'] (:276-279): rows and importance weights of the last frl_per_sample; the TD errors
                                  stay on the device for frl_per_update.  1 = the reference's arithmetic: its [B] weights times
                                  [B,1] squared errors broadcast to [B,B], so loss = mean(w) * mean(td^2); 2 = mean(w_i * td_i^2) */
    const float* noisy_eps;    /* noisy engines: host [P][3][S] factorised noise of the three forwards of learn() in the reference's
                                  order (online on s' when double_dqn, target on s', online on s); S = frl_noisy_eps_size(): per
                                  NoisyLinear eps_in[hidden] then eps_out[rows] = f(randn) (Noisy_net.py:66-76), V before A.  NULL:
                                  drawn on the device */
    const int64_t* idx;        /* host [P][n_agents][batch] rows drawn by the caller (np.random.choice,
                                  DQN.py:97) for bit-identical sampling; NULL: drawn on the device */
    const float* noise;        /* host [P][n_agents][S][batch][act_max], S = max(2, n_agents): N(0,1) draws the reference
                                  takes from torch's generator (TD3.py:197 slot 0; SAC.py:227 slot 0, :244 slot 1;
                                  MATD3_simple.py:200: slot j = randn_like(action[agent j]) inside agent i's sample());
                                  NULL: drawn on the device */
    float* stats_out;          /* host [P][n_agents][FRL_STAT_COUNT] or NULL (NULL: call is asynchronous) */
    int loss_kind;             /* TD loss of the Q / critic update: FRL_LOSS_MSE = F.mse_loss, what every hot-path learn() of the
                                  reference uses (DQN.py:116, TD3.py:212, SAC.py:237); FRL_LOSS_HUBER = the reference's
                                  huber_loss(e, d).mean() (MAPPO_file/MAPPO.py:273-276): e^2/2 if |e| <= d else d(|e| - d/2) */
    float huber_delta;         /* d (MAPPO_attention.py:518 defaults to 10) */
} frl_learn_args;
enum frl_loss_kind { FRL_LOSS_MSE = 0, FRL_LOSS_HUBER = 1 };

/* ---------------------------------------------------------------- library / engine lifetime */
const char* frl_last_error(void);          /* message of the last failing call on this thread */
int frl_version(void);
int frl_device_count(int* n_out);
int frl_create(const frl_config* cfg, frl_engine** out);
int frl_destroy(frl_engine* e);
int frl_sync(frl_engine* e);               /* wait for the engine's stream */
int frl_lds_bytes(const frl_engine* e, int* bytes_out, int* row_chunk_out);
/* Which kernel family frl_learn() launches for this engine at `batch` rows: chained = 1 when the critic and actor stages run
 * as one workgroup per learner with the weights in LDS images (kernels_critic2.hip / kernels_actor2.hip), 0 for the row-chunk
 * kernels + reduce / Adam launches; the dynamic LDS bytes and the batch rows of one workgroup of that family.  16 (or 32) rows with
 * chained = 1: the sixteen-workgroups-per-learner families of small populations (kernels_solo.hip; kernels_solow.hip for wide first
 * layers and MADDPG: 160512 bytes of LDS). */
int frl_learn_path(const frl_engine* e, int batch, int* chained_out, int* lds_bytes_out, int* rows_per_workgroup_out);

/* ---------------------------------------------------------------- replay ring (Buffer.py)
 * Replaces class Buffer (TD3_file/Buffer.py:11-61 = DQN_file/Buffer.py:12-62) and
 * Buffer_for_PPO (PPO_file/Buffer.py:266-323). */
int frl_record_layout_get(const frl_engine* e, frl_record_layout* out);
/* Buffer.add (Buffer.py:28-38): one record (host, `width` floats) appended at the learner's cursor;
 * staged in pinned memory and pushed by the next flush / sample / learn. */
int frl_buffer_add(frl_engine* e, int learner, const float* record);
/* vectorised add: record i goes to learner learners[i] (NULL: all to learner 0 in order) */
int frl_buffer_add_batch(frl_engine* e, int n, const int* learners, const float* records);
int frl_buffer_flush(frl_engine* e);
/* Buffer._index / Buffer._size / __len__ / clear (Buffer.py:23-24,60-61; PPO Buffer.py:303-306) */
int frl_buffer_cursor_get(const frl_engine* e, int learner, int* index_out, int* size_out);
int frl_buffer_cursor_set(frl_engine* e, int learner, int index, int size);
/* Buffer.sample(indices) (Buffer.py:40-57): gather rows `idx` (host int64[B]) of one learner into
 * n_fields dense DEVICE tensors out[f][B][ncols[f]] = record[:, col0[f] : col0[f]+ncols[f]] — one launch. */
int frl_buffer_sample(frl_engine* e, int learner, const int64_t* idx, int batch, int n_fields, const int* col0,
                      const int* ncols, float* const* out_device);
/* read `n` whole records starting at ring row `row0` into host memory [n][width] (Buffer_for_PPO.all, views) */
int frl_buffer_read(frl_engine* e, int learner, int row0, int n, float* out_host);
/* synthetic fill of the first `rows` rows of every learner's ring (bench only) */
int frl_buffer_fill_synthetic(frl_engine* e, int rows, uint64_t seed);

/* ---------------------------------------------------------------- parameters (state_dict)
 * Nets: DQN 0 = Qnet; DDPG/TD3/SAC/PPO 0 = actor, 1 = critic; MADDPG 2i = actor_i, 2i+1 = critic_i.
 * Flat layout = the reference state_dict tensors in layer order (l1.weight[out][in], l1.bias, l2...,
 * twin critic l1..l6), then log_std for Gaussian actors (DQN.py:131-138, SAC.py:274-282). */
int frl_net_count(const frl_engine* e, int* n_out);
int frl_net_num_params(const frl_engine* e, int net, int* n_out);
int frl_params_get(frl_engine* e, int learner, int net, int kind, float* out_host);
int frl_params_set(frl_engine* e, int learner, int net, int kind, const float* in_host);
/* invariant check: max |x| over the PADDING slots of one net's block of `kind` (weight rows / columns and bias entries past a
 * layer's real dims; layers are padded to multiples of 16).  The update kernels rely on the padding staying exactly zero — the
 * K-sliced first-layer sweeps read a record for 16 * ceil(in / 16) columns and let the zero weights cancel what lies past the
 * layer's inputs (which is why every field of a stored transition must be FINITE: 0 * inf is NaN; a non-finite value in a sampled
 * row poisons the reference's update as well, through the TD target) — and Adam keeps it: zero gradient, zero moments, zero step. */
int frl_params_pad_max(frl_engine* e, int learner, int net, int kind, float* max_abs_out);
int frl_opt_step_get(frl_engine* e, int learner, int net, int* t_out);   /* Adam step count */
int frl_opt_step_set(frl_engine* e, int learner, int net, int t);
/* SAC Alpha (SAC.py:154-169): vals = {log_alpha, exp_avg, exp_avg_sq, alpha} */
int frl_alpha_get(frl_engine* e, int learner, float* vals4_out, int* step_out);
int frl_alpha_set(frl_engine* e, int learner, const float* vals4, int step);

/* Batch_ObsNorm (`Normalization_batch_size`, PPO_file/normalization.py:53-84; SAC.py:181-182,215-217;
 * DDPG.py:160-161,190-192; PPO_with_tricks.py:225-226,297-299): when enabled, every learn call first
 * updates the running statistics with the batch mean of the sampled observations, then obs and
 * next_obs are normalised wherever the path reads them; select_action normalises without updating.
 * stats = {n, mean[O], S[O], std[O]}; a MADDPG engine (MADDPG.py:155-156,194-196) keeps one such block per agent, each 1 +
 * 3*max(obs_dim) floats wide, and get/set move all n_agents blocks. */
int frl_obsnorm_enable(frl_engine* e, int on);
int frl_obsnorm_get(frl_engine* e, int learner, float* stats_out);
int frl_obsnorm_set(frl_engine* e, int learner, const float* stats);

/* ---------------------------------------------------------------- forward (select_action)
 * in_host [P][n_rows][in_dim], eps_host [P][n_rows][out_dim] or NULL, out_host [P][n_rows][out_dim]
 * (ARGMAX: [P][n_rows] indices as float), logp_host like out or NULL.  Synchronous.
 * use_target: 0 online parameters, 1 target parameters, 2 the noisy net's effective set of the last frl_noisy_resample. */
int frl_act(frl_engine* e, int net, int mode, int head, int use_target, int n_rows, int in_dim, const float* in_host,
            const float* eps_host, float* out_host, float* logp_host);
/* same with DEVICE pointers, asynchronous on the engine stream (vectorised env pool path) */
int frl_act_device(frl_engine* e, int net, int mode, int head, int use_target, int n_rows, int in_dim,
                   const float* in_dev, const float* eps_dev, float* out_dev, float* logp_dev);

/* ---------------------------------------------------------------- learn (the hot path) */
int frl_learn(frl_engine* e, const frl_learn_args* args);
int frl_stats_get(frl_engine* e, float* out_host);          /* [P][n_agents][FRL_STAT_COUNT] */
/* the ring rows the last frl_learn trained on — `indices` of `<ALGO>.sample` (DQN.py:97): uploaded, device-drawn or (per = 1)
 * the last frl_per_sample's — as host int64 [P][n_agents][batch] */
int frl_last_indices(frl_engine* e, int batch, int64_t* out_host);
/* algorithmic work of one frl_learn launch (for the roofline figure): flops and HBM bytes, by SURVEY.md 8(d)'s formula —
 * 2 B sum(in x out) x (#forward + 2 x #backward passes), i.e. a dX is counted for every layer of every backward pass */
int frl_learn_work(const frl_engine* e, int batch, int do_actor, double* flops_out, double* bytes_out);
/* the same launch's EXECUTED flops: what torch autograd (and these kernels) actually compute — a trained net's backward is
 * dW for every layer but dX only from the second layer up (nothing needs d loss / d input), and the policy loss's pass through
 * the frozen critic (Q(s, pi(s)), TD3.py:225, SAC.py:245-249) is forward + dX with the first layer's dX on the agent's action
 * columns only.  8(d)'s figure is 1.6 % above this at the narrow bench shape and 30 % above it at config 4's 393-column first
 * layers; roofline fractions quote both, named. */
int frl_learn_work_executed(const frl_engine* e, int batch, int do_actor, double* flops_out);

/* ---------------------------------------------------------------- PPO (PPO_file/PPO_with_tricks.py)
 * `PPO.learn(minibatch_size, gamma, lmbda, clip_param, K_epochs, entropy_coefficient)` (:290-354):
 * value pass over the stored horizon, GAE scan, v_target, optional advantage normalisation, then
 * K_epochs x ceil(horizon/minibatch) actor + critic steps — three launches in total. */
typedef struct frl_ppo_args {
    int horizon;               /* rows 0..horizon-1 of the ring, in time order (Buffer_for_PPO.all, Buffer.py:312-323) */
    int minibatch, k_epochs;
    int adv_norm;              /* trick['adv_norm'] (:314-315) */
    float gamma, lmbda, clip, ent_coef;
    float actor_lr, critic_lr; /* per call so that PPO.lr_decay (:357-363) is the caller's arithmetic */
    float adam_eps;            /* 1e-5 with trick['adam_eps'] (:191-196) */
    float clip_norm;           /* 0.5 */
    int optimizer;             /* 0: torch.optim.Adam per net (PPO_with_tricks.py:191-196);
                                  1: PPO.py's combined cautious AdamW over actor + critic parameters (PPO.py:121,145-152,
                                     c_adamw.py:80-127): lr = actor_lr for both nets, eps = adam_eps (1e-6 there), mask =
                                     (exp_avg*grad > 0) / max(mean, 1e-3) PER PARAMETER TENSOR, no bias correction in denom */
    const int64_t* perms;      /* host [P][k_epochs][horizon] np.random.permutation draws (:320), or NULL: drawn on the device */
    float* loss_trace_out;     /* host [P][k_epochs*n_mb][2] (actor, critic) losses or NULL */
    float* adv_out;            /* host [P][horizon] raw GAE advantages or NULL */
    float* vtarget_out;        /* host [P][horizon] or NULL */
    int gae_mode;              /* 0: the value pass + TD-delta scan described above.
                                  1: PPO_advance/PPO_2.py:213-224 — no value pass: V(s_t) was stored at rollout time in the
                                     extra column before adv_done (Buffer_for_PPO_2.add, PPO_advance/Buffer.py:462-478) and
                                     the advantages / returns come from stable-baselines3's scan (:480-507) in float64:
                                     delta = r + gamma*next_value*(1-done) - V, A = delta + gamma*lambda*(1-adv_done)*A',
                                     returns = A + V, both cast to float32 once */
    const float* last_value;   /* gae_mode 1: host [P], the critic's value of the state after the last stored step */
    double gae_gamma, gae_lmbda; /* gae_mode 1: the scan's gamma / lambda as the caller's doubles (0: use gamma / lmbda) */
} frl_ppo_args;
int frl_ppo_learn(frl_engine* e, const frl_ppo_args* args);
/* stand-alone GAE scan (K3) on device arrays [n_seq][horizon]: replaces the host loop at
 * PPO_with_tricks.py:308-311 / PPO.py:229-231 */
int frl_gae(frl_engine* e, const float* td_delta_dev, const float* adv_done_dev, int n_seq, int horizon,
            float gamma, float lmbda, float* adv_out_dev);

/* floats of factorised noise ONE forward of a noisy net consumes (0 for other engines) */
int frl_noisy_eps_size(const frl_engine* e, int* n_out);
/* one forward's worth of fresh noise for select_action (DQN_with_tricks.py:213-216 through Noisy_net.py:41-44): eps_host
 * [P][frl_noisy_eps_size] or NULL (device draw); afterwards frl_act(..., use_target = 2, ...) runs the net with it */
int frl_noisy_resample(frl_engine* e, const float* eps_host);

/* ---------------------------------------------------------------- prioritised replay
 * PER_Buffer + SumTree (DQN_file/Buffer.py:66-194) on the engine's ring: a float64 sum-tree and max-tree per learner in HBM.
 * add (frl_buffer_add*) gives new rows the current maximum priority, 1.0 on an empty buffer (:92-98). */
/* PER_Buffer.__init__ (:81-90).  beta and its increment stay doubles (the reference's Python floats); alpha and epsilon act
 * on float32 TD errors. */
int frl_per_enable(frl_engine* e, double alpha, double beta, double beta_increment, double epsilon);
/* PER_Buffer.sample (:99-124): beta += increment; `batch` stratified descents with s = a + (b-a)*u.  uniforms: host
 * [P][batch] draws of np.random.uniform's underlying random_sample(), or NULL (device Philox).  The sampled rows become the
 * engine's current sample (frl_learn with per = 1); idx_out / is_weight_out: host [P][batch] or NULL. */
int frl_per_sample(frl_engine* e, int batch, const double* uniforms, int64_t* idx_out, float* is_weight_out);
/* PER_Buffer.update_priorities (:126-129): priority = (|td| + epsilon)^alpha in float32.  idx / td_error: host [P][batch],
 * or NULL = the rows of the last frl_per_sample / the TD errors the last frl_learn(per = 1) left on the device. */
int frl_per_update(frl_engine* e, int batch, const int64_t* idx, const float* td_error);
int frl_per_state(frl_engine* e, int learner, double* sum_out, double* max_out, double* beta_out);   /* sumtree.sum(), .max(), beta */

/* ---------------------------------------------------------------- env pool + rollout
 * The step BEFORE the path: the reference steps one Python env inline per update (DQN.py:316) and
 * pays H2D + D2H per step in select_action (DQN.py:77,83).  The pool steps n env instances on
 * host worker threads into pinned staging; frl_rollout runs the whole per-step order of the
 * reference loop (select_action -> exploration -> env.step -> add -> learn; DQN.py:294-339,
 * TD3.py:403-450) for P learners x E instances per launch chain. */
enum frl_env_kind { FRL_ENV_PENDULUM = 0, FRL_ENV_CARTPOLE = 1, FRL_ENV_SYNLINEAR = 2, FRL_ENV_SYNLINEAR_DISCRETE = 3,
                    FRL_ENV_PENDULUM_SHORT = 4,
                    FRL_ENV_SYNBAND_WIDE = 5 /* obs 376 / act 17 (Humanoid-v4's dims, SAC.py:519-576 at BASELINE config 4) on banded
                                                linear dynamics: sizes the rollout path, not the physics */ };
typedef struct frl_envpool frl_envpool;
/* params: optional SynLinear matrices A[8][8] then B[8][2] (doubles), else NULL */
int frl_envpool_create(int kind, int n_envs, int n_threads, uint64_t seed, const double* params, int n_params,
                       frl_envpool** out);
/* A pool over caller-supplied environments (the gymnasium protocol the reference's loops drive, DQN.py:292,316: reset() ->
 * obs, step(a) -> obs, reward, terminated, truncated): ONE vectorised callback steps all n envs into the pool's pinned
 * staging and resets the finished ones (obs_next = the reset observation, like the built-in kinds).  Return 0 on success.
 * n_actions > 0: discrete (actions = [n] indices as float, act_dim 1). */
typedef int (*frl_env_step_fn)(void* user, const float* actions, float* next_obs, float* reward, uint8_t* terminated,
                               uint8_t* truncated, float* obs_next);
typedef int (*frl_env_reset_fn)(void* user, float* obs_out);
int frl_envpool_create_callback(int n_envs, int obs_dim, int act_dim, int n_actions, float max_action, frl_env_step_fn step,
                                frl_env_reset_fn reset, void* user, frl_envpool** out);
int frl_envpool_destroy(frl_envpool* p);
int frl_envpool_dims(const frl_envpool* p, int* n_envs, int* obs_dim, int* act_dim, int* n_actions, float* max_action,
                     int* max_steps);
int frl_envpool_reset(frl_envpool* p, float* obs_out);                       /* host [n][obs_dim] */
int frl_envpool_set_state(frl_envpool* p, int env, const double* state);     /* test hook: physical state of one env */
/* actions host [n][act_dim] in env units (discrete: [n] indices as float); outputs host arrays:
 * next_obs = observation of the transition, obs_next = what the policy sees next (reset obs after a done) */
int frl_envpool_step(frl_envpool* p, const float* actions, float* next_obs, float* reward, uint8_t* terminated,
                     uint8_t* truncated, float* obs_next);
/* The reference loops' per-step exploration rules, applied INSIDE the act launch from the engine's Philox stream (the
 * reference draws them from NumPy's global generator on the host, one env at a time):
 *   EPS_GREEDY  DQN.py:307-310    np.random.rand() < epsilon -> np.random.randint(action_dim), else the greedy action
 *   GAUSS       TD3.py:412        action_ = clip(a*max_action + scale * N(0, sigma*max_action), +-max_action)
 *   OU          SAC.py:334-356,529  x += theta*(0 - x) + sqrt(dt)*sigma*N(0,1); action_ = clip(a*max_action + x*scale*max_action)
 *   NONE        SAC.py:533, PPO_with_tricks.py:529-530  action_ = clip(a*max_action)
 * The action handed to add() is the policy's own output (continuous) or the explored index (discrete), as in the reference. */
enum frl_explore_kind { FRL_EXPLORE_NONE = 0, FRL_EXPLORE_EPS_GREEDY = 1, FRL_EXPLORE_GAUSS = 2, FRL_EXPLORE_OU = 3,
                        FRL_EXPLORE_OFF = 4 /* frl_rollout_args.explore_kind only: explicitly no exploration noise */ };
/* frl_rollout_args.explore_kind counts 0 differently from frl_explore_args.kind (since frl_version 101): there 0 is what a
 * zero-initialised struct carries and selects the algorithm's loop default, so "no noise" has its own value, FRL_EXPLORE_OFF.
 * Write FRL_EXPLORE_DEFAULT, never FRL_EXPLORE_NONE, into rollout args; frl_act_explore rejects FRL_EXPLORE_OFF. */
#define FRL_EXPLORE_DEFAULT 0
typedef struct frl_explore_args {
    int kind;
    float epsilon;
    float sigma;              /* gauss_sigma */
    float scale;              /* gauss_scale / OUNoise.scale (1 = none) */
    float max_action;
    float ou_theta, ou_sigma, ou_dt;
} frl_explore_args;
/* obs host [P][n_rows][obs_dim]; ended host [P][n_rows] or NULL (1: reset that row's OU state first, SAC.py:546-547);
 * outputs host [P][n_rows][act_dim] (FRL_ACT_ARGMAX: [P][n_rows]): the action add() stores, and the env-unit action. */
int frl_act_explore(frl_engine* e, int mode, int n_rows, const float* obs_host, const frl_explore_args* x,
                    const uint8_t* ended_host, float* store_act_out, float* env_act_out);

typedef struct frl_rollout_args {
    int n_steps;             /* vector steps (each steps every env once) */
    int envs_per_learner;    /* E; the pool must hold P*E envs, env i feeds learner i/E */
    int start_steps;         /* learn once every ring holds more rows than this (`step > start_steps`) */
    int learn_every;         /* vector steps per frl_learn call; 0: collect only */
    int policy_freq;         /* TD3 delayed actor update (TD3.py:224) */
    float epsilon;           /* DQN epsilon-greedy (DQN.py:307) */
    float explore_sigma;     /* Gaussian action-noise std as a fraction of max_action (gauss_scale*gauss_sigma, TD3.py:412) */
    frl_learn_args learn;    /* idx / noise / stats_out must be NULL; per = 1|2 (PER engines): frl_per_sample / frl_per_update around every learn */
    int host_explore;        /* 0: exploration inside the act launch — per vector step ONE D2H (env actions) and ONE H2D (the
                              * env outputs), records assembled on the device; 1: round 1's loop (host generator, obs H2D +
                              * action D2H + staged records H2D) kept for comparison */
    int explore_kind;        /* FRL_EXPLORE_DEFAULT (0, a zero-initialised struct) or -1: the algorithm's loop default (DQN
                              * epsilon-greedy, DDPG/TD3 Gaussian, SAC none); FRL_EXPLORE_EPS_GREEDY / _GAUSS / _OU: that rule;
                              * FRL_EXPLORE_OFF: none (a deterministic evaluation rollout).  NOT FRL_EXPLORE_NONE: that is 0 */
    float gauss_init_scale, gauss_final_scale;   /* with max_episodes > 0: a learner's noise multiplier decays with ITS finished episodes, */
    int max_episodes;                            /* scale = final + (init - final) * max(0, max_episodes - episodes) / max_episodes (TD3.py:425-427, SAC.py:548-556) */
    float ou_theta, ou_sigma, ou_dt;             /* OUNoise(theta 0.15, sigma, dt) (SAC.py:334-356) */
} frl_rollout_args;
typedef struct frl_rollout_stats {
    long long env_steps, updates, episodes;
    double return_sum, seconds;
} frl_rollout_stats;
int frl_rollout(frl_engine* e, frl_envpool* p, const frl_rollout_args* args, frl_rollout_stats* out);
/* On-policy counterpart for PPO engines (BASELINE config 3, PPO with vectorised envs): the reference steps one env for
 * `horizon` steps and calls learn() (PPO_with_tricks.py:524-569); here every learner steps E env instances for
 * horizon / E vector steps per cycle — select_action (:234-255) batched over all envs, action_ = clip(action *
 * max_action) (:529-530), add(obs, action, reward, next_obs, terminated, log_pi, terminated or truncated) (:541-545) —
 * then frl_ppo_learn.  Env j's steps fill ring rows [j*steps_per_env, (j+1)*steps_per_env); the last row of each
 * segment is marked adv_done so that the GAE scan restarts there as it does at the end of the reference's horizon.
 * stats: updates = minibatch steps (one actor + one critic step each). */
typedef struct frl_ppo_rollout_args {
    int n_iters;             /* collect + learn cycles */
    int envs_per_learner;    /* E; the pool must hold P*E envs */
    int steps_per_env;       /* vector steps per cycle; learn.horizon must equal E * steps_per_env */
    frl_ppo_args learn;      /* perms / outputs NULL (device-side permutations), gae_mode 0 */
} frl_ppo_rollout_args;
int frl_ppo_rollout(frl_engine* e, frl_envpool* p, const frl_ppo_rollout_args* args, frl_rollout_stats* out);

/* Algorithmic flops / bytes of one frl_ppo_learn over all learners (the figure a PPO roofline fraction is computed from). */
int frl_ppo_work(const frl_engine* e, int horizon, int k_epochs, double* flops_out, double* bytes_out);

/* ---------------------------------------------------------------- multi-GPU: the path's ONE collective
 * The reference is single-process / single-device and has no distributed code (its only trace is the dead import
 * DDPG_file/misc(lose).py:4): independent seeds / env-instance sets are sharded one process per GPU and nothing on the data
 * path is exchanged.  What IS exchanged is a short metrics vector per reporting interval — counters summed, wall-clock maxed —
 * by ncclAllReduce over RCCL (xGMI inside a node).  Bootstrap like NCCL's: ONE rank calls frl_comm_unique_id and ships the
 * 128 bytes to the others by any host channel (freerl_amd/dist.py: the launcher's store); every rank then calls
 * frl_comm_create collectively.  RCCL is bound at run time (dlopen), so the library loads on a box without it. */
#define FRL_COMM_ID_BYTES 128
#define FRL_COMM_MAX_VALUES 64
typedef struct frl_comm frl_comm;
int frl_comm_unique_id(uint8_t* id_out /* [FRL_COMM_ID_BYTES] */);
int frl_comm_create(const uint8_t* id, int rank, int world, int device_id, frl_comm** out);
int frl_comm_destroy(frl_comm* c);
int frl_comm_info(const frl_comm* c, int* rank_out, int* world_out);      /* NULL comm: rank 0 of 1 */
/* sums[n_sum] <- sum over ranks, maxes[n_max] <- max over ranks, in place (float64: counters exact to 2^53); at most
 * FRL_COMM_MAX_VALUES each.  comm == NULL: the single-process case, the vectors are already the job's totals.  Synchronous. */
int frl_metrics_allreduce(frl_comm* c, double* sums, int n_sum, double* maxes, int n_max);

/* developer read-back of the single-learner kernels' per-workgroup block (partial sums; section stamps in a timing build):
 * [16][32] floats of learner 0 (tools/solo_timing.py) */
int frl_solo_debug_read(frl_engine* e, float* out_host, int n_floats);

/* ---------------------------------------------------------------- timing on the engine stream */
int frl_timer_start(frl_engine* e);
int frl_timer_stop(frl_engine* e, float* ms_out);           /* synchronises */
/* per-kernel durations of frl_learn's launch chain (HIP events around every launch while enabled):
 * slots 0 draw, 1 grad(critic/Q), 2 adam(critic/Q), 3 grad(actor), 4 adam(actor), 5 soft update */
int frl_profile_enable(frl_engine* e, int on);
int frl_profile_read(frl_engine* e, double* ms_sum8_out, long long* count8_out);

#ifdef __cplusplus
}
#endif
#endif /* FREERL_HIP_H */
