// Host side of the C ABI (include/freerl_hip.h): engine construction, the replay ring's pinned
// staging, parameter import/export, kernel launches.  Compiled with hipcc for gfx950 only.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/freerl_hip.h"
#include "frl_desc.h"

#include "kernels.h"
#ifdef FRL_UNITY      // developer variants (tools/phase_timing.py): one translation unit, so that the phase-timing symbols exist once
#include "kernels_replay.hip"
#include "kernels_update.hip"
#include "kernels_dqn.hip"
#include "kernels_critic.hip"
#include "kernels_actor.hip"
#include "kernels_act.hip"
#include "kernels_ppo.hip"
#include "kernels_ppo2.hip"
#include "kernels_critic2.hip"
#include "kernels_actor2.hip"
#include "kernels_criticw.hip"
#include "kernels_actorw.hip"
#include "kernels_criticx.hip"
#include "kernels_actorx.hip"
#include "kernels_dqn2.hip"
#include "kernels_per.hip"
#include "kernels_noisy.hip"
#include "kernels_c51.hip"
#include "kernels_solo.hip"
#include "kernels_solow.hip"
#endif

using namespace frl;

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(FRL_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

static inline int pad16(int x) { return (x + 15) / 16 * 16; }
static inline int pad32(int x) { return (x + 31) / 32 * 32; }

struct frl_engine {
    frl_config cfg;
    EngineDesc h;                 // host mirror of the device descriptor
    EngineDesc* d = nullptr;      // device copy
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_stage = nullptr;
    int lds_bytes = 0;
    // replay cursors (host authoritative, reference semantics Buffer.py:23-24,36-38)
    std::vector<int> index, size;
    // pinned staging of not-yet-flushed adds
    float* stage_rows = nullptr;          // [stage_cap][width] pinned
    long long* stage_slots = nullptr;     // [stage_cap] pinned
    float* d_stage_rows = nullptr;
    long long* d_stage_slots = nullptr;
    int stage_cap = 0, stage_n = 0;
    bool stage_inflight = false;
    std::vector<int> staged_per_learner;  // rows staged since the last flush (a flush may not hold one ring row twice)
    // pinned scratch for idx / noise uploads
    int* h_idx = nullptr;
    long long* h_idx64 = nullptr;
    long long* d_idx64 = nullptr;
    float* h_noise = nullptr;
    size_t idx_count = 0, noise_count = 0;
    unsigned long long rng_counter = 0;
    // act scratch (device)
    // [slot][P][net size]: fragment-image nets re-laid out to Wk for act_kernel (wide chained engines), one slot per
    // (net, online | target), sized once at frl_create; a slot is re-laid out only when the engine's parameters have changed since
    // (param_version: bumped by frl_learn, frl_params_set and the obs-norm relayout)
    float* d_act_wk = nullptr;
    size_t act_wk_slot = 0;               // floats per slot = P * largest net
    std::vector<unsigned long long> act_wk_version;   // [2 * n_nets], 0 = never laid out
    unsigned long long param_version = 1;
    // kernels_solo.hip (h.solo): per-workgroup gradient slabs, partial sums, the learners' grid-barrier counters
    float* d_solo_slab = nullptr;
    float* d_solo_part = nullptr;
    unsigned* d_solo_bar = nullptr;
    int* d_solo_err = nullptr;            // device address of h_solo_err
    int* d_solo_pre = nullptr;            // [2][P][kSoloPre]: the next call's rows, drawn a launch ahead (SoloArgs::pre_read / pre_write)
    unsigned solo_pre_seq = 0;            // critic-stage launches so far: which of the two slots is read / written
    // kernels_solow.hip, multi-agent: what the rows in the slot the next launch reads were drawn for (the HOST decides whether draw_kernel runs)
    bool ma_pre_valid = false;
    unsigned long long ma_pre_counter = 0;
    int ma_pre_size = 0, ma_pre_batch = 0;
    int* h_solo_err = nullptr;            // pinned: a solo workgroup that waited 2 s for its learner's others sets it (checked after syncs: solo_err_check)
    int* d_solo_ticket = nullptr;         // the rollout tail's learner ticket
    unsigned solo_bar_base = 0;           // arrivals every counter has seen (one counting barrier per launch: kSoloWG)
    int solo_stride = 0;
    int solow_wgs = 0;                    // kernels_solow.hip: workgroups per unit in its grids (row-tile workgroups + helpers for the update)
    unsigned* d_solow_bar2 = nullptr;     // [units][64] "critic stepped" flags of the helper workgroups (the fused policy step)
    int solow_row_wgs = 0;                // ... of which own row tiles (16: one tile each; 8: two each, populations of 17 .. 32 units; 64: MADDPG's batches of 1024)
    float* d_act_in = nullptr;
    float* d_act_eps = nullptr;
    float* d_act_out = nullptr;
    float* d_act_logp = nullptr;
    size_t act_in_cap = 0, act_out_cap = 0;
    float* d_ou = nullptr;                // frl_act_explore's Ornstein-Uhlenbeck state [P][n_rows][A]
    unsigned char* d_ou_flags = nullptr;
    size_t ou_n = 0;
    // ppo scratch
    float* d_ppo = nullptr;
    size_t ppo_cap = 0;
    int* d_perm = nullptr;
    size_t perm_cap = 0;
    bool has_nets = false;
    int noisy_per_set = 0;                // floats of device noise per (learner, forward): 2 * (k_pad + n_pad) of the head
    float* h_noisy = nullptr;             // pinned staging for uploaded noise
    // prioritised replay (frl_per_*): sum-tree + max-tree per learner, float64 like the reference's SumTree
    bool per_on = false;
    double* d_per_sum = nullptr;
    double* d_per_max = nullptr;
    float per_alpha = 0.5f, per_eps = 0.01f;
    double per_beta = 0.4, per_beta_inc = 0.001;
    std::vector<int> bucket_cursor;
    std::vector<int> size_flushed;        // rows valid per learner as of the last flush (PER_Buffer.add's `len(self.buffer) == 0`)
    int* d_size = nullptr;                // [2][P]: size before the flush being applied / current size
    int n_cus = 256;                      // compute units of the device (how many one-per-CU workgroups are resident at once)
    int lds_per_cu = 160 * 1024;          // LDS bytes of one compute unit
    int chain_waves = 8;                  // waves per workgroup of the register-chained actor-critic kernels (FRL_CHAIN_WAVES=4: round 5's)
    int* stage_bucket = nullptr;          // PER, pinned: [off[P + 1] | size_before[P] | leaf[stage_cap]] of the flush being applied
    int* d_stage_bucket = nullptr;
    float* d_per_prio = nullptr;          // [P][batch_max] float32 priorities of the last sample
    double* d_uniforms = nullptr;         // [P][batch_max]
    // optional per-kernel timing (frl_profile_*): event pairs recorded around each launch
    bool profile = false;
    std::vector<hipEvent_t> prof_ev;      // pool, pairs
    std::vector<int> prof_kind;           // kernel kind per recorded pair
    size_t prof_used = 0;
    double prof_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

enum ProfKind { PK_DRAW = 0, PK_GRAD_CRITIC = 1, PK_ADAM_CRITIC = 2, PK_GRAD_ACTOR = 3, PK_ADAM_ACTOR = 4, PK_SOFT = 5, PK_PPO = 6 };

static void prof_begin(frl_engine* e, int kind) {
    if (!e->profile) return;
    if (e->prof_used + 2 > e->prof_ev.size()) {
        for (int i = 0; i < 2; ++i) { hipEvent_t ev; hipEventCreate(&ev); e->prof_ev.push_back(ev); }
    }
    hipEventRecord(e->prof_ev[e->prof_used], e->stream);
    e->prof_kind.push_back(kind);
}
static void prof_end(frl_engine* e) {
    if (!e->profile) return;
    hipEventRecord(e->prof_ev[e->prof_used + 1], e->stream);
    e->prof_used += 2;
}
static void prof_collect(frl_engine* e) {
    if (e->prof_used == 0) return;
    hipStreamSynchronize(e->stream);
    for (size_t i = 0; i < e->prof_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]) == hipSuccess) {
            const int k = e->prof_kind[i / 2];
            e->prof_ms[k] += ms;
            e->prof_n[k]++;
        }
    }
    e->prof_used = 0;
    e->prof_kind.clear();
}

// ------------------------------------------------------------------------------ descriptors
static int build_net(NetDesc& N, const std::vector<std::pair<int, int>>& layers_in /* (out,in) */, int heads,
                     int hidden_act, int out_act, int extra_n, int n_shadow = 0) {
    memset(&N, 0, sizeof N);
    std::vector<std::pair<int, int>> layers = layers_in;
    for (int i = 0; i < n_shadow; ++i) layers.push_back(layers_in.back());       // sigma of a NoisyLinear head: same shape as the head
    N.n_layers = (int)layers.size();
    N.heads = heads;
    N.hidden_act = hidden_act;
    N.out_act = out_act;
    int off = 0, np = 0;
    for (int i = 0; i < N.n_layers; ++i) {
        LayerDesc& L = N.L[i];
        L.n = layers[i].first;
        L.k = layers[i].second;
        L.n_pad = pad16(L.n);
        L.k_pad = pad16(L.k);
        L.w_off = off;
        off += L.n_pad * L.k_pad;
        L.b_off = off;
        off += L.n_pad;
        np += L.n * L.k + L.n;
    }
    N.extra_off = -1;
    N.extra_n = extra_n;
    if (extra_n > 0) {
        N.extra_off = off;
        off += pad16(extra_n);
        np += extra_n;
    }
    N.size = pad32(off);
    N.n_params = np;
    N.n_layers -= n_shadow;
    N.n_shadow = n_shadow;
    return 0;
}

static void build_record(RecordDesc& R, const frl_config& c) {
    memset(&R, 0, sizeof R);
    R.n_agents = c.n_agents;
    int off = 0;
    for (int j = 0; j < c.n_agents; ++j) { R.obs_off[j] = off; R.obs_dim[j] = c.obs_dim[j]; off += c.obs_dim[j]; }
    R.obs_total = off;
    for (int j = 0; j < c.n_agents; ++j) {
        R.act_off[j] = off;
        R.act_dim[j] = c.discrete ? 1 : c.act_dim[j];
        off += R.act_dim[j];
    }
    R.act_total = off - R.obs_total;
    R.rew_off = off; off += c.n_agents;
    R.done_off = off; off += c.n_agents;
    for (int j = 0; j < c.n_agents; ++j) { R.nobs_off[j] = off; off += c.obs_dim[j]; }
    R.extra_off = off;
    R.extra = c.extra_cols;
    off += c.extra_cols;
    R.width = off;
    R.stride = pad32(off);
}

static int lds_bytes_for(const EngineDesc& h, int rc) {
    const int xp = h.lds_kin_pad + 4, hp = h.hidden + 4, op = h.lds_out_pad + 4, ap = h.lds_act_pad;
    long long fl = (long long)rc * (xp + h.lds_hbufs * hp + op + 2 * ap) + h.lds_batch_pad + 8;
    return (int)(fl * 4);
}

// --------------------------------------------------------------------------------- lifetime
extern "C" const char* frl_last_error(void) { return g_err.c_str(); }
extern "C" int frl_version(void) { return 102; }      // 101: frl_rollout_args.explore_kind 0 = FRL_EXPLORE_DEFAULT; 102: frl_learn_work_executed

extern "C" int frl_device_count(int* n_out) {
    if (!n_out) return fail(FRL_ERR_INVALID, "n_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *n_out = 0; return fail(FRL_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n_out = n;
    return FRL_OK;
}

extern "C" int frl_destroy(frl_engine* e) {
    if (!e) return FRL_OK;
    hipSetDevice(e->cfg.device_id);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->d_per_sum) hipFree(e->d_per_sum);
    if (e->d_per_max) hipFree(e->d_per_max);
    if (e->d_size) hipFree(e->d_size);
    if (e->d_stage_bucket) hipFree(e->d_stage_bucket);
    if (e->stage_bucket) hipHostFree(e->stage_bucket);
    if (e->d_per_prio) hipFree(e->d_per_prio);
    if (e->d_uniforms) hipFree(e->d_uniforms);
    if (e->h_noisy) hipHostFree(e->h_noisy);
    if (e->d_solo_slab) hipFree(e->d_solo_slab);
    if (e->d_solo_part) hipFree(e->d_solo_part);
    if (e->d_solo_pre) hipFree(e->d_solo_pre);
    if (e->d_solo_bar) hipFree(e->d_solo_bar);
    if (e->d_solow_bar2) hipFree(e->d_solow_bar2);
    if (e->h_solo_err) hipHostFree(e->h_solo_err);
    float* dev[] = {e->h.act_spill, e->h.theta_eff, e->h.noisy_eps, e->h.isw, e->h.td_err, e->h.theta, e->h.target, e->h.m, e->h.v, e->h.grad, e->h.replay, e->h.noise, e->h.stats, e->h.alpha,
                    e->d_stage_rows, e->d_act_in, e->d_act_eps, e->d_act_out, e->d_act_logp, e->d_ppo, e->d_act_wk, e->h.wide_scr};
    for (float* p : dev) if (p) hipFree(p);
    if (e->h.idx) hipFree(e->h.idx);
    if (e->h.steps) hipFree(e->h.steps);
    if (e->h.ticket) hipFree(e->h.ticket);
    if (e->d_stage_slots) hipFree(e->d_stage_slots);
    if (e->d_ou) hipFree(e->d_ou);
    if (e->d_ou_flags) hipFree(e->d_ou_flags);
    if (e->d_idx64) hipFree(e->d_idx64);
    if (e->d_perm) hipFree(e->d_perm);
    if (e->d) hipFree(e->d);
    if (e->stage_rows) hipHostFree(e->stage_rows);
    if (e->stage_slots) hipHostFree(e->stage_slots);
    if (e->h_idx) hipHostFree(e->h_idx);
    if (e->h_idx64) hipHostFree(e->h_idx64);
    if (e->h_noise) hipHostFree(e->h_noise);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->ev_stage) hipEventDestroy(e->ev_stage);
    for (hipEvent_t ev : e->prof_ev) hipEventDestroy(ev);
    if (e->h.slab) hipFree(e->h.slab);
    if (e->h.part) hipFree(e->h.part);
    if (e->h.gsq) hipFree(e->h.gsq);
    if (e->h.obsnorm) hipFree(e->h.obsnorm);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return FRL_OK;
}

template <class T>
static hipError_t dalloc_zero(T** p, size_t count, hipStream_t s) {
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, 0, count * sizeof(T), s);
}

static bool chained_shape(const EngineDesc& h);
static bool wide_shape(const EngineDesc& h);

extern "C" int frl_create(const frl_config* cfg, frl_engine** out) {
    if (!cfg || !out) return fail(FRL_ERR_INVALID, "cfg/out is NULL");
    *out = nullptr;
    const frl_config& c = *cfg;
    if (c.n_learners < 1) return fail(FRL_ERR_INVALID, "n_learners must be >= 1");
    if (c.n_agents < 1 || c.n_agents > FRL_MAX_AGENTS) return fail(FRL_ERR_INVALID, "n_agents out of range");
    if (c.algo != FRL_ALGO_MADDPG && c.n_agents != 1) return fail(FRL_ERR_INVALID, "n_agents > 1 needs FRL_ALGO_MADDPG");
    if (c.capacity < 1) return fail(FRL_ERR_INVALID, "capacity must be >= 1");
    if (c.algo < FRL_ALGO_REPLAY_ONLY || c.algo > FRL_ALGO_PPO) return fail(FRL_ERR_INVALID, "unknown algo %d", c.algo);
    for (int j = 0; j < c.n_agents; ++j)
        if (c.obs_dim[j] < 1 || c.act_dim[j] < 1) return fail(FRL_ERR_INVALID, "obs_dim/act_dim must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FRL_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (c.device_id < 0 || c.device_id >= ndev) return fail(FRL_ERR_INVALID, "device_id %d of %d", c.device_id, ndev);
    HIP_TRY(hipSetDevice(c.device_id));

    frl_engine* e = new frl_engine();
    e->cfg = c;
    {   // what the device can hold resident at once: the solo kernels' flag hand-overs spin on every workgroup of a launch
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c.device_id) == hipSuccess) {
            e->n_cus = std::max(1, prop.multiProcessorCount);
            e->lds_per_cu = (int)std::max<size_t>(prop.maxSharedMemoryPerMultiProcessor, prop.sharedMemPerBlock);
        }
        const char* fc = getenv("FRL_ASSUME_CUS");                      // (tests: a smaller / partitioned device)
        if (fc && atoi(fc) > 0) e->n_cus = atoi(fc);
    }
    EngineDesc& h = e->h;
    memset(&h, 0, sizeof h);
    h.algo = c.algo;
    h.P = c.n_learners;
    h.n_agents = c.n_agents;
    h.hidden = c.hidden > 0 ? c.hidden : 128;
    if (h.hidden % 16) { delete e; return fail(FRL_ERR_INVALID, "hidden must be a multiple of 16"); }
    h.capacity = c.capacity;
    h.batch_max = c.batch_max > 0 ? c.batch_max : 256;
    h.seed = c.seed;
    h.n_discrete = (c.algo == FRL_ALGO_DQN || (c.algo == FRL_ALGO_PPO && c.discrete)) ? c.act_dim[0] : 0;
    build_record(h.rec, c);
    const RecordDesc& R = h.rec;
    const int H = h.hidden;
    const int hact = c.hidden_act == FRL_ACT_TANH ? ACT_TANH : ACT_RELU;
    e->has_nets = c.algo != FRL_ALGO_REPLAY_ONLY;
    if (c.algo == FRL_ALGO_DQN) {
        h.n_nets = 1;
        h.dueling = c.dueling ? 1 : 0;
        h.noisy = c.noisy ? 1 : 0;
        h.c51_atoms = c.c51_atoms > 1 ? c.c51_atoms : 0;
        if (h.c51_atoms > 64) { delete e; return fail(FRL_ERR_INVALID, "c51_atoms %d: the distribution arithmetic runs one wave lane per atom (<= 64; the reference uses 51)", h.c51_atoms); }
        h.c51_vmin = c.c51_vmin; h.c51_vmax = c.c51_vmax;
        const int per_out = h.c51_atoms ? h.c51_atoms : 1;        // head rows: [V (per_out) ;] A (act_dim x per_out)
        build_net(h.net[0], {{H, c.obs_dim[0]}, {(c.act_dim[0] + (c.dueling ? 1 : 0)) * per_out, H}}, 1, ACT_RELU, ACT_NONE, 0,
                  c.noisy ? 1 : 0);                                                                      // MLP, DQN.py:32-45
        h.noisy_split = c.dueling ? per_out : h.net[0].L[1].n_pad;
    } else if (c.algo == FRL_ALGO_PPO) {
        h.n_nets = 2;
        if (c.discrete && c.actor_dist == 2) h.cat_logits = 1;     // PPO_file/PPO.py:78-90,176,257: raw logits into Categorical(logits=)
        if (c.discrete)     // Actor_discrete (PPO_with_tricks.py:110-121): ReLU body, softmax over n_actions logits
            build_net(h.net[0], {{H, c.obs_dim[0]}, {H, H}, {c.act_dim[0], H}}, 1, ACT_RELU, ACT_NONE, 0);
        else if (c.actor_dist == 1) {   // Actor_Beta (:120-151): alpha_layer and beta_layer share the trunk = one 2A-wide head
            build_net(h.net[0], {{H, c.obs_dim[0]}, {H, H}, {2 * c.act_dim[0], H}}, 1, hact, ACT_NONE, 0);
            h.beta_actor = 1;
        } else
            build_net(h.net[0], {{H, c.obs_dim[0]}, {H, H}, {c.act_dim[0], H}}, 1, hact, ACT_TANH, c.act_dim[0]);
        build_net(h.net[1], {{H, c.obs_dim[0]}, {H, H}, {1, H}}, 1, hact, ACT_NONE, 0);
    } else if (e->has_nets) {
        h.n_nets = 2 * c.n_agents;
        const int cin = R.obs_total + R.act_total;
        const int heads = c.twin_critic ? 2 : 1;
        for (int j = 0; j < c.n_agents; ++j) {
            const bool sac = c.algo == FRL_ALGO_SAC;
            build_net(h.net[2 * j], {{H, c.obs_dim[j]}, {H, H}, {c.act_dim[j], H}}, 1, ACT_RELU,
                      sac ? ACT_NONE : ACT_TANH, sac ? c.act_dim[j] : 0);
            std::vector<std::pair<int, int>> ls;
            for (int hd = 0; hd < heads; ++hd) { ls.push_back({H, cin}); ls.push_back({H, H}); ls.push_back({1, H}); }
            build_net(h.net[2 * j + 1], ls, heads, ACT_RELU, ACT_NONE, 0);
        }
    }
    int off = 0, kin = 16, outp = 16;
    for (int i = 0; i < h.n_nets; ++i) {
        h.net_off[i] = off;
        off += h.net[i].size;
        const int nl = h.net[i].n_layers / h.net[i].heads;
        for (int hd = 0; hd < h.net[i].heads; ++hd) {
            kin = std::max(kin, h.net[i].L[hd * nl].k_pad);
            outp = std::max(outp, h.net[i].L[hd * nl + nl - 1].n_pad);
        }
    }
    h.learner_stride = pad32(off);
    if (e->has_nets && chained_shape(h) && !(getenv("FRL_SOLOW_NARROW") && atoi(getenv("FRL_SOLOW_NARROW")) != 0)) {       // kernel family (and with it the parameter layout in HBM), fixed for the engine's life
        // (FRL_SOLOW_NARROW=1, developer knob: the narrow standard shape on kernels_solow.hip — one first-layer k-tile — for A/Bs against kernels_solo.hip)
        const char* force = getenv("FRL_CRITIC_V2");
        // measured (bench workload, updates/s): 128 learners are exactly one round of the row-chunk kernels' 512 resident
        // workgroups — 484 k against 373 k for 128 one-learner workgroups on half the CUs; from 129 up the chained kernels win
        // (160: 459 k / 393 k, 256: 694 k / 566 k) or tie (320: 472 k / 480 k)
        // up to kSoloMaxP learners: one learner on sixteen workgroups (kernels_solo.hip; FRL_SOLO=0/1 overrides, FRL_CRITIC_V2 set
        // means the caller asked for one of the other two families by name)
        const char* solo = getenv("FRL_SOLO");
        // (every one of its P x 16 workgroups — 156 KB of LDS each: one per CU — has to be RESIDENT: they wait for each other's flags.
        //  On a device with fewer CUs, a CU-masked or partitioned one, the row-chunk kernels take the engine instead)
        // h.solo = workgroups per learner: 16 (one 16-row tile each) up to 16 learners; 8 (two tiles each, a slab per tile) up to 32
        // learners — populations the row-chunk kernels used to take at 150-160 us per learn() (kernels_solo.hip has the numbers;
        // FRL_SOLO_MAXP: the largest population on this family, default 32)
        const char* smp = getenv("FRL_SOLO_MAXP");
        const int solo_maxp = smp ? std::min(atoi(smp), 2 * kSoloMaxP) : 2 * kSoloMaxP;
        int wgs = 0;
        for (int cand : {16, 8})
            if (wgs == 0 && (long long)h.P * cand <= e->n_cus && h.P * cand <= kSoloMaxP * kSoloWG) wgs = cand;
        const bool solo_fits = wgs > 0 && h.P <= solo_maxp && e->lds_per_cu >= (int)(std::max(solo_lds_floats(), critic2_lds_floats()) * sizeof(float));
        h.solo = (solo ? atoi(solo) != 0 : !force) && solo_fits ? wgs : 0;
        if (h.solo || (force ? atoi(force) != 0 : h.P > 128)) h.net[0].frag = h.net[1].frag = 1;
    } else if (e->has_nets && wide_shape(h)) {   // the K-sliced chained family (kernels_criticw.hip / kernels_actorw.hip): one workgroup per (learner, agent)
        const char* force = getenv("FRL_CRITIC_V2");
        // a handful of single-agent learners at hidden 128: sixteen workgroups per learner, W1 streamed from the block (kernels_solow.hip;
        // FRL_SOLOW=0/1 overrides, FRL_CRITIC_V2 set means the caller asked for one of the other two families by name).  Every one of
        // the P x 16 workgroups has to be resident, as for kernels_solo.hip.  [s | a] must be the record's first columns, 16-byte aligned
        const char* sw = getenv("FRL_SOLOW");
        // ... MADDPG / MATD3 (config 5: three agents, batches of 1024): a unit = (learner, agent), 64 row tiles = 64 workgroups per unit; the
        // updating agent's own observation rows sit behind the joint rows in LDS, so both first layers have at most kSoloWActorBase k-tiles
        bool solow_shape = h.hidden == 128 && h.batch_max <= (h.n_agents == 1 ? 256 : 1024) && h.rec.stride % 4 == 0 && h.rec.obs_off[0] % 4 == 0 &&
                           h.rec.act_off[0] == h.rec.obs_off[0] + h.rec.obs_total;
        for (int i = 0; i < h.n_nets; ++i) solow_shape = solow_shape && h.net[i].L[0].k_pad <= 16 * (h.n_agents == 1 ? kSoloWMaxKB : kSoloWActorBase);
        const int solow_tiles = h.batch_max <= 256 ? kSoloWG : 4 * kSoloWG;
        const long long solow_units = (long long)h.P * h.n_agents;
        // (two row tiles per workgroup for 17 .. 32 units: measured level with the row-chunk chain — kernels_solow.hip — and not built)
        const int solow_rw = solow_tiles;
        const bool solow_fits = solow_units <= kSoloMaxP && solow_units * solow_rw <= e->n_cus && e->lds_per_cu >= (int)(solow_lds_floats() * sizeof(float) + 256);
        if ((sw ? atoi(sw) != 0 : !force) && solow_shape && solow_fits) {
            for (int i = 0; i < h.n_nets; ++i) h.net[i].frag = 1;
            h.solow = solow_tiles;
            e->solow_row_wgs = solow_rw;
        } else
        // from 129 (learner, agent) units up: hidden 128 (chain_wide.hpp) SAC at Humanoid dims 85.5 TFLOP/s against the row-chunk
        // kernels' 46.2, MADDPG simple_spread 77.2 / 55.5; hidden 256 (chain_wide16.hpp: x-stationary sweeps) 71.1 / 56.9
        // (profiles/r04)
        // (profiles/r04/family_crossover.txt: hidden 128 ties at ~110-129 units; hidden 256 at 128 units 38.9 against 53.1, at 192
        // 55.3 / 51.5, at 256 68.0 / 55.7 — its workgroups are twice as long, so the half-empty chip costs more: from 177 up)
        if (force ? atoi(force) != 0 : (long long)h.P * h.n_agents > (h.hidden == 256 ? 176 : 128)) {
            for (int i = 0; i < h.n_nets; ++i) h.net[i].frag = 1;
            h.wide = h.hidden == 256 ? 2 : 1;
            h.wide_bm = h.wide == 2 ? (h.batch_max + 255) / 256 * 256 : (h.batch_max + 63) / 64 * 64;      // (hidden 256 works in super-chunks of 256 rows)
            h.wide_xp = h.net[1].L[0].k_pad;
            h.wide_op = 16;
            for (int j = 0; j < h.n_agents; ++j) h.wide_op = std::max(h.wide_op, h.net[2 * j].L[0].k_pad);
            h.wide_unit = ((h.wide_xp + h.n_agents * h.wide_op + (h.wide == 2 ? kWide16ScratchPerRowHost : kWideScratchPerRow)) * h.wide_bm + 128 + 63) / 64 * 64;
        }
    }
    h.act_max = 1;
    for (int j = 0; j < c.n_agents; ++j) h.act_max = std::max(h.act_max, R.act_dim[j]);
    h.lds_kin_pad = kin;
    // one hidden-activation buffer is enough when every head is input -> hidden -> output (DQN's Q-net): 17 KB less per 32-row
    // chunk, which puts the Categorical update's 32-row workgroups at two per CU
    h.lds_hbufs = 1;
    for (int i = 0; i < h.n_nets; ++i)
        if (h.net[i].n_layers / std::max(1, h.net[i].heads) > 2) h.lds_hbufs = 2;
    h.lds_out_pad = outp;
    h.lds_batch_pad = 128;                                  // y holds one row chunk (rc <= 128) of TD targets
    h.lds_act_pad = (std::max(R.act_total, 1) + 3) / 4 * 4; // abuf / dabuf are scalar-accessed: no tile padding
    if (c.algo == FRL_ALGO_PPO && c.discrete) h.lds_act_pad = std::max(h.lds_act_pad, pad16(c.act_dim[0]));   // logits' delta staging
    if (c.algo == FRL_ALGO_PPO && c.actor_dist == 1) h.lds_act_pad = std::max(h.lds_act_pad, pad16(2 * c.act_dim[0]));
    if (h.c51_atoms) h.lds_act_pad = std::max(h.lds_act_pad, (h.c51_atoms + 3) / 4 * 4);      // projected distribution / probabilities per row
    // row chunk: the largest of {64,32,16} whose LDS footprint still lets TWO workgroups share a CU.  Measured
    // (profiles/README.md v4): 64 rows x 2 workgroups beats 32 x 3, 32 x 4 and 128 x 1 — more rows per weight fragment
    // fetched and half the gradient slabs, while two workgroups still overlap each other's barrier phases
    h.rc = 64;
    while (h.rc > 16 && lds_bytes_for(h, h.rc) > 80 * 1024) h.rc /= 2;   // two workgroups per CU (160 KB LDS)
    // wide inputs (SAC on Humanoid: 393 input columns): 16 rows re-read every weight 16x per batch; 32 rows at ONE
    // workgroup per CU measured +8 % over 16 rows at three (tools/config_bench.py, SAC C4)
    if (h.rc == 16 && lds_bytes_for(h, 32) <= 160 * 1024) h.rc = 32;
    // small populations cannot fill 256 CUs with 64-row chunks (one learner = batch/64 workgroups): 32-row chunks double
    // the workgroup count and measured +19 % (P = 1) / +13 % (P = 8) updates/s.  PPO's persistent kernel is one workgroup
    // per net whatever rc is, and prefers the whole minibatch in one chunk.
    if (c.algo != FRL_ALGO_PPO && h.rc == 64 && (long long)h.P * ((h.batch_max + 63) / 64) < 512) h.rc = 32;
    if (const char* force = getenv("FRL_RC")) {             // developer knob: rows per workgroup (16 / 32 / 64 / 128)
        const int v = atoi(force);
        if (v == 16 || v == 32 || v == 64 || v == 128) h.rc = v;
    }
    if (lds_bytes_for(h, h.rc) > 160 * 1024) { delete e; return fail(FRL_ERR_INVALID, "network too wide for LDS (%d B at 16 rows)", lds_bytes_for(h, h.rc)); }
    e->lds_bytes = lds_bytes_for(h, h.rc);
    // Row chunks per gradient workgroup.  With `units` (learner, agent) pairs and n_chunks chunks each, s slabs per unit cost
    // ceil(units * s / slots) rounds of n_chunks / s chunk-times on the chip's resident-workgroup slots, and the reduce
    // pass streams s slabs: take the fewest slabs (but two) among the cheapest schedules.  512 learners x 4 chunks of 64
    // rows on 512 slots: two workgroups per learner walk 2 chunks each (2 slabs) instead of 4 workgroups writing 4 slabs.
    {
        const int n_chunks = (h.batch_max + h.rc - 1) / h.rc;
        // resident gradient workgroups: 256 CUs x (2 by registers — the kernels are built for FRL_GRAD_WGS = 2 — or 1 by LDS)
        const long long slots = 256LL * std::max(1, std::min(2, (160 * 1024) / std::max(1, e->lds_bytes)));
        const long long units = (long long)h.P * h.n_agents;
        long long best_cost = -1;
        h.cps = 1;
        for (int cps = n_chunks; cps >= 1; --cps) {
            if (n_chunks % cps) continue;
            const int s = n_chunks / cps;
            // measured (P = 512, 4 chunks): 2 slabs 529 k updates/s, 1 slab 524 k, 4 slabs 495 k — with a single slab every
            // workgroup of the launch starts at once and they stay in step (gather bursts, barrier phases coincide)
            if (s < 2 && n_chunks >= 2) continue;
            const long long cost = ((units * s + slots - 1) / slots) * cps;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; h.cps = cps; }
        }
        if (c.algo == FRL_ALGO_PPO) h.cps = 1;
        if (const char* force = getenv("FRL_CPS")) {        // developer knob
            const int v = atoi(force);
            if (v >= 1 && n_chunks % v == 0) h.cps = v;
        }
        h.S = n_chunks / h.cps;
    }

#define CREATE_TRY(expr)                                                                           \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            int rc_ = fail(FRL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));                   \
            frl_destroy(e);                                                                        \
            return rc_;                                                                            \
        }                                                                                          \
    } while (0)

    CREATE_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&e->ev0));
    CREATE_TRY(hipEventCreate(&e->ev1));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_stage, hipEventDisableTiming));
    const size_t P = h.P;
    CREATE_TRY(dalloc_zero(&h.replay, P * (size_t)h.capacity * R.stride, e->stream));
    if (e->has_nets) {
        const size_t ls = h.learner_stride;
        CREATE_TRY(dalloc_zero(&h.theta, P * ls, e->stream));
        CREATE_TRY(dalloc_zero(&h.target, P * ls, e->stream));
        CREATE_TRY(dalloc_zero(&h.m, P * ls, e->stream));
        CREATE_TRY(dalloc_zero(&h.v, P * ls, e->stream));
        CREATE_TRY(dalloc_zero(&h.grad, P * ls, e->stream));
        CREATE_TRY(dalloc_zero(&h.slab, P * (size_t)h.S * ls, e->stream));
        CREATE_TRY(dalloc_zero(&h.part, P * (size_t)h.n_agents * h.S * 4, e->stream));
        if (c.algo != FRL_ALGO_DQN && c.algo != FRL_ALGO_PPO)
            CREATE_TRY(dalloc_zero(&h.act_spill, P * (size_t)h.n_agents * h.S * 2 * h.rc * (h.hidden + 4), e->stream));
        h.Gmax = 1;
        for (int i = 0; i < h.n_nets; ++i) h.Gmax = std::max(h.Gmax, (h.net[i].size / 4 + 256 * kAdamVec - 1) / (256 * kAdamVec));
        CREATE_TRY(dalloc_zero(&h.gsq, P * (size_t)h.n_agents * h.Gmax, e->stream));
        e->idx_count = P * h.n_agents * h.batch_max;
        if (h.noisy) {
            const LayerDesc& HL = h.net[0].L[h.net[0].n_layers - 1];
            e->noisy_per_set = 2 * (HL.k_pad + HL.n_pad);
            CREATE_TRY(dalloc_zero(&h.theta_eff, P * 3 * ls, e->stream));
            CREATE_TRY(dalloc_zero(&h.noisy_eps, P * 3 * (size_t)e->noisy_per_set, e->stream));
            CREATE_TRY(hipHostMalloc((void**)&e->h_noisy, P * 3 * (size_t)e->noisy_per_set * sizeof(float)));
        }
        CREATE_TRY(dalloc_zero(&h.isw, P * (size_t)h.batch_max, e->stream));
        CREATE_TRY(dalloc_zero(&h.td_err, P * (size_t)h.batch_max, e->stream));
        h.noise_sets = std::max(2, h.n_agents);
        e->noise_count = P * h.n_agents * h.noise_sets * (size_t)h.batch_max * h.act_max;
        CREATE_TRY(dalloc_zero(&h.idx, e->idx_count, e->stream));
        CREATE_TRY(dalloc_zero(&h.noise, e->noise_count, e->stream));
        CREATE_TRY(dalloc_zero(&h.stats, P * h.n_agents * ST_COUNT, e->stream));
        CREATE_TRY(dalloc_zero(&h.steps, P * (kMaxNets + 1), e->stream));
        CREATE_TRY(dalloc_zero(&h.ticket, P + 1, e->stream));
        CREATE_TRY(dalloc_zero(&h.alpha, P * 4, e->stream));
        if (h.solo || h.solow) {
            e->solo_stride = 0;
            for (int i = 0; i < h.n_nets; ++i) e->solo_stride = std::max(e->solo_stride, h.net[i].size);
            const size_t U = P * (size_t)h.n_agents, NT = h.solow ? (size_t)h.solow : (size_t)kSoloWG;      // (kernels_solow.hip: units, row tiles per unit)
            CREATE_TRY(dalloc_zero(&e->d_solo_slab, U * NT * e->solo_stride, e->stream));
            // (kernels_solow.hip: helper workgroups on the CUs a small population leaves idle take a share of the slab sum and of Adam —
            //  FRL_SOLOW_HELPERS=0 switches them off; every workgroup of a launch must be resident)
            e->solow_wgs = (int)NT;
            if (h.solow) {
                const char* hp = getenv("FRL_SOLOW_HELPERS");
                const int rw = e->solow_row_wgs;
                const int per = std::min(64 / rw > 0 ? 64 / rw : 1, std::max(1, e->n_cus / (rw * (int)U)));      // (at most 64 workgroups per unit: the mailboxes are polled by one wave)
                e->solow_wgs = (hp && atoi(hp) == 0) ? rw : rw * per;
            }
            CREATE_TRY(dalloc_zero(&e->d_solo_part, U * (size_t)std::max((int)NT, e->solow_wgs) * kSoloPartHost, e->stream));
            { float* z = nullptr; CREATE_TRY(dalloc_zero(&z, 2 * U * (size_t)std::max(kSoloPre, 8 + h.batch_max), e->stream)); e->d_solo_pre = (int*)z; }
            float* z = nullptr;
            CREATE_TRY(dalloc_zero(&z, 2 * U * NT + 2, e->stream));
            e->d_solo_bar = (unsigned*)z;                                  // [units][tiles] slab flags, then as many "actor slice stepped" flags (kernels_solo.hip: offset P * 16),
            e->d_solo_ticket = (int*)(z + 2 * U * NT);                     // ... and the rollout tail's learner ticket
            if (h.solow) { float* z2 = nullptr; CREATE_TRY(dalloc_zero(&z2, U * 64, e->stream)); e->d_solow_bar2 = (unsigned*)z2; }
        }
        if (h.solo || h.solow || h.algo == ALGO_DQN) {                     // the pinned give-up word of the spinning launches (solo hand-overs, pre-armed DQN steps)
            CREATE_TRY(hipHostMalloc((void**)&e->h_solo_err, 64, hipHostMallocCoherent | hipHostMallocMapped));
            *e->h_solo_err = 0;
            CREATE_TRY(hipHostGetDevicePointer((void**)&e->d_solo_err, e->h_solo_err, 0));
        }
        if (h.wide) CREATE_TRY(dalloc_zero(&h.wide_scr, P * (size_t)h.n_agents * h.wide_unit, e->stream));
        if (h.wide || h.solow) {                                           // select_action's Wk-layout copies of the nets (launch_act)
            int biggest = 0;
            for (int i = 0; i < h.n_nets; ++i) biggest = std::max(biggest, h.net[i].size);
            e->act_wk_slot = P * (size_t)biggest;
            e->act_wk_version.assign(2 * (size_t)h.n_nets, 0ULL);
            CREATE_TRY(hipMalloc((void**)&e->d_act_wk, 2 * (size_t)h.n_nets * e->act_wk_slot * sizeof(float)));
        }
        {
            int omax = 1;
            for (int j = 0; j < c.n_agents; ++j) omax = std::max(omax, c.obs_dim[j]);
            h.obsnorm_w = 1 + 3 * omax;
            CREATE_TRY(dalloc_zero(&h.obsnorm, P * (size_t)c.n_agents * c.n_agents * h.obsnorm_w, e->stream));
        }
        CREATE_TRY(hipHostMalloc((void**)&e->h_idx, e->idx_count * sizeof(int)));
        CREATE_TRY(hipHostMalloc((void**)&e->h_noise, e->noise_count * sizeof(float)));
    }
    e->stage_cap = 4096;
    CREATE_TRY(hipHostMalloc((void**)&e->stage_rows, (size_t)e->stage_cap * R.width * sizeof(float)));
    CREATE_TRY(hipHostMalloc((void**)&e->stage_slots, (size_t)e->stage_cap * sizeof(long long)));
    CREATE_TRY(hipMalloc((void**)&e->d_stage_rows, (size_t)e->stage_cap * R.width * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&e->d_stage_slots, (size_t)e->stage_cap * sizeof(long long)));
    CREATE_TRY(hipHostMalloc((void**)&e->h_idx64, (size_t)h.batch_max * 16 * sizeof(long long)));
    CREATE_TRY(hipMalloc((void**)&e->d_idx64, (size_t)h.batch_max * 16 * sizeof(long long)));
    CREATE_TRY(hipMalloc((void**)&e->d, sizeof(EngineDesc)));
    CREATE_TRY(hipMemcpyAsync(e->d, &h, sizeof h, hipMemcpyHostToDevice, e->stream));
    e->index.assign(P, 0);
    e->size.assign(P, 0);
    e->staged_per_learner.assign(P, 0);
    { const char* cw = getenv("FRL_CHAIN_WAVES"); e->chain_waves = (cw && atoi(cw) == 4) ? 4 : 8; }
    if (h.algo == ALGO_DQN)
        CREATE_TRY(hipFuncSetAttribute((const void*)dqn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dqn2_lds_floats() * (int)sizeof(float)));
    if (h.batch_max > 256 && 4 * h.batch_max <= kDrawTableHost)     // draw_kernel's duplicate table for batches of 257 .. 2048 rows (device/net.hpp)
        CREATE_TRY(hipFuncSetAttribute((const void*)draw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 2048 + 2 * kDrawTableHost) * (int)sizeof(int)));
    if (h.wide == 2) {
        const int lb = wide16_lds_floats_host() * (int)sizeof(float);
        for (auto k : {ac_critic_x_h1a1_kernel, ac_critic_x_h1a2_kernel, ac_critic_x_h2a1_kernel, ac_critic_x_h2a2_kernel, ac_actor_x_a1_kernel, ac_actor_x_a2_kernel})
            CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    } else if (h.wide) {
        const int lb = wide_lds_floats() * (int)sizeof(float);
        for (auto k : {ac_critic_wide_h1a1_kernel, ac_critic_wide_h1a2_kernel, ac_critic_wide_h2a1_kernel, ac_critic_wide_h2a2_kernel,
                       ac_actor_wide_a1_kernel, ac_actor_wide_a2_kernel})
            CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    } else if (h.solow) {
        const int lb = solow_lds_floats() * (int)sizeof(float);
        for (auto k : {solow_critic_h1a1_kernel, solow_critic_h1a2_kernel, solow_critic_h2a1_kernel, solow_critic_h2a2_kernel, solow_actor_a1_kernel, solow_actor_a2_kernel,
                       solow_critic_ma_h1a1_kernel, solow_critic_ma_h1a2_kernel, solow_critic_ma_h2a1_kernel, solow_critic_ma_h2a2_kernel, solow_actor_ma_a1_kernel, solow_actor_ma_a2_kernel,
                       solow_step_h1a1_kernel, solow_step_h1a2_kernel, solow_step_h2a1_kernel, solow_step_h2a2_kernel})
            CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    } else if (h.solo) {
        const int lb = std::max(solo_lds_floats(), critic2_lds_floats()) * (int)sizeof(float);      // (critic2: the rollout tail's act_frag_body)
        for (auto k : {solo_critic_twin_kernel, solo_critic_single_kernel, solo_actor_kernel, solo_critic_twin_w8_kernel, solo_critic_single_w8_kernel, solo_actor_w8_kernel})
            CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
        CREATE_TRY(hipFuncSetAttribute((const void*)act_frag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, critic2_lds_floats() * (int)sizeof(float)));
    } else if (h.net[0].frag) {        // the register-chained family: one workgroup per learner with the nets as LDS images (156 KB)
        const int lb = critic2_lds_floats() * (int)sizeof(float), lb8 = critic8_lds_floats() * (int)sizeof(float);
        for (auto k : {ac_critic_v2_twin_kernel, ac_critic_v2_single_kernel, ac_critic_v2_twin_nv_kernel, ac_critic_v2_single_nv_kernel, ac_actor_v2_kernel})
            CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb8));
        for (auto k : {ac_critic_v2w4_twin_kernel, ac_critic_v2w4_single_kernel, ac_actor_v2w4_kernel})
            CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
        CREATE_TRY(hipFuncSetAttribute((const void*)act_frag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    }
    if (e->lds_bytes > 64 * 1024) {
        CREATE_TRY(hipFuncSetAttribute((const void*)dqn_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
        CREATE_TRY(hipFuncSetAttribute((const void*)ac_critic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
        CREATE_TRY(hipFuncSetAttribute((const void*)ac_actor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
        CREATE_TRY(hipFuncSetAttribute((const void*)act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
        CREATE_TRY(hipFuncSetAttribute((const void*)ppo_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
        if (h.algo == ALGO_DDPG || h.algo == ALGO_TD3 || h.algo == ALGO_SAC) {
            const int lb = critic2_lds_floats() * (int)sizeof(float), lb8 = critic8_lds_floats() * (int)sizeof(float);
            for (auto k : {ac_critic_v2_twin_kernel, ac_critic_v2_single_kernel, ac_critic_v2_twin_nv_kernel, ac_critic_v2_single_nv_kernel, ac_actor_v2_kernel})
                CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb8));
            for (auto k : {ac_critic_v2w4_twin_kernel, ac_critic_v2w4_single_kernel, ac_actor_v2w4_kernel})
                CREATE_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
        }
        if (h.algo == ALGO_PPO) {
            const int lb = ppo2_lds_floats(2) * (int)sizeof(float);
            CREATE_TRY(hipFuncSetAttribute((const void*)ppo_update_v2_k1_relu, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            CREATE_TRY(hipFuncSetAttribute((const void*)ppo_update_v2_k2_relu, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            CREATE_TRY(hipFuncSetAttribute((const void*)ppo_update_v2_k1_tanh, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            CREATE_TRY(hipFuncSetAttribute((const void*)ppo_update_v2_k2_tanh, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
        }
    }
    CREATE_TRY(hipStreamSynchronize(e->stream));
    *out = e;
    return FRL_OK;
}

#define ENG(e)                                                  \
    if (!(e)) return fail(FRL_ERR_INVALID, "engine is NULL");   \
    HIP_TRY(hipSetDevice((e)->cfg.device_id))

// kernels_solo.hip's hand-overs spin on flags of the learner's other workgroups, which therefore must all be resident; a workgroup that
// has waited 2 s gives up, sets the pinned word and the launch's numbers are not valid.  That takes something else holding this GPU's CUs
// for seconds (another process' long kernel): reported at the next synchronising call instead of passing silently.
static int solo_err_check(frl_engine* e) {
    if (!e->h_solo_err || *(volatile int*)e->h_solo_err == 0) return FRL_OK;
    const int what = *(volatile int*)e->h_solo_err;
    *(volatile int*)e->h_solo_err = 0;
    if (what == 2)
        return fail(FRL_ERR_STATE, "dqn_fused_kernel: a pre-armed rollout launch waited 2 s for the previous launch or for the host's doorbell and gave up; "
                                   "its update and the actions it was to select were dropped");
    return fail(FRL_ERR_STATE, "kernels_solo.hip / kernels_solow.hip: a workgroup waited 2 s for the other workgroups of its learner (is something else holding this GPU's "
                               "CUs?); the updates since the last synchronising call are not valid");
}

extern "C" int frl_sync(frl_engine* e) {
    ENG(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    return solo_err_check(e);
}

extern "C" int frl_lds_bytes(const frl_engine* e, int* bytes_out, int* rc_out) {
    if (!e) return fail(FRL_ERR_INVALID, "engine is NULL");
    if (bytes_out) *bytes_out = e->lds_bytes;
    if (rc_out) *rc_out = e->h.rc;
    return FRL_OK;
}

static bool chained_path(const EngineDesc& h, int batch, int pc);
static bool dqn_fused_path(const EngineDesc& h, int batch, bool per_weights);
// workgroups per learner of the one-launch DQN update (kernels_dqn2.hip).  A few learners: one 64-row chunk per workgroup (the
// last to arrive reduces and steps); populations: one workgroup each (measured: P = 64 x 4 workgroups 74 us, x 1 45 us).
static int dqn_split_for(const EngineDesc& h, int batch, int pc) {
    const int nchunks = (batch + 63) / 64;
    int split = (pc <= 16) ? std::min(std::min(4, nchunks), h.S) : 1;
    if (const char* sp = getenv("FRL_DQN_SPLIT")) split = std::max(1, std::min(std::min(atoi(sp), nchunks), h.S));
    return split;
}

extern "C" int frl_learn_path(const frl_engine* e, int batch, int* chained_out, int* bytes_out, int* rows_out) {
    if (!e) return fail(FRL_ERR_INVALID, "engine is NULL");
    if (e->h.algo == ALGO_PPO) return fail(FRL_ERR_INVALID, "frl_learn_path describes frl_learn(); PPO updates go through frl_ppo_learn");
    if (batch <= 0 || batch > e->h.batch_max) return fail(FRL_ERR_INVALID, "batch out of range");
    if (dqn_fused_path(e->h, batch, e->per_on)) {           // kernels_dqn2.hip
        if (chained_out) *chained_out = 1;
        if (bytes_out) *bytes_out = dqn2_lds_floats() * (int)sizeof(float);
        if (rows_out) { const int sp = dqn_split_for(e->h, batch, e->h.P); *rows_out = ((batch + 63) / 64 + sp - 1) / sp * 64 < batch ? ((batch + 63) / 64 + sp - 1) / sp * 64 : batch; }
        return FRL_OK;
    }
    const bool v2 = chained_path(e->h, batch, e->h.P);
    if (v2 && e->h.solo) {                                  // kernels_solo.hip: 16-row tiles, 16 / h.solo of them per workgroup
        if (chained_out) *chained_out = 1;
        if (bytes_out) *bytes_out = solo_lds_floats() * (int)sizeof(float);
        if (rows_out) *rows_out = 16 * (kSoloWG / e->h.solo);
        return FRL_OK;
    }
    if (v2 && e->h.solow) {                                 // kernels_solow.hip: one 16-row tile per workgroup
        if (chained_out) *chained_out = 1;
        if (bytes_out) *bytes_out = solow_lds_floats() * (int)sizeof(float);
        if (rows_out) *rows_out = 16 * (e->h.solow / std::max(1, e->solow_row_wgs));
        return FRL_OK;
    }
    if (chained_out) *chained_out = v2 ? 1 : 0;
    if (bytes_out) *bytes_out = v2 ? (e->h.wide == 2 ? wide16_lds_floats_host() : (e->h.wide ? wide_lds_floats() : (e->chain_waves == 8 && !e->h.solo ? critic8_lds_floats() : critic2_lds_floats()))) * (int)sizeof(float) : e->lds_bytes;
    if (rows_out) *rows_out = v2 ? batch : e->h.rc;
    return FRL_OK;
}

// ----------------------------------------------------------------------------------- replay
extern "C" int frl_record_layout_get(const frl_engine* e, frl_record_layout* out) {
    if (!e || !out) return fail(FRL_ERR_INVALID, "NULL argument");
    const RecordDesc& R = e->h.rec;
    memset(out, 0, sizeof *out);
    out->n_agents = R.n_agents;
    out->width = R.width;
    out->stride = R.stride;
    for (int j = 0; j < R.n_agents; ++j) {
        out->obs_off[j] = R.obs_off[j]; out->obs_dim[j] = R.obs_dim[j];
        out->act_off[j] = R.act_off[j]; out->act_dim[j] = R.act_dim[j];
        out->next_obs_off[j] = R.nobs_off[j];
    }
    out->rew_off = R.rew_off;
    out->done_off = R.done_off;
    out->extra_off = R.extra_off;
    out->extra = R.extra;
    return FRL_OK;
}

static int flush_stage(frl_engine* e) {
    if (e->stage_n == 0) return FRL_OK;
    const RecordDesc& R = e->h.rec;
    const int n = e->stage_n;
    HIP_TRY(hipMemcpyAsync(e->d_stage_rows, e->stage_rows, (size_t)n * R.width * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_stage_slots, e->stage_slots, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, e->stream));
    const int per_row = (R.width + 3) / 4;
    const long long total = (long long)n * per_row;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(replay_scatter_kernel, dim3(blocks), dim3(256), 0, e->stream, e->h.replay, e->d_stage_rows,
                       e->d_stage_slots, n, R.width, R.stride);
    HIP_TRY(hipGetLastError());
    if (e->per_on) {                 // PER_Buffer.add (Buffer.py:92-98): the new rows enter at the current maximum priority
        // bucket the flush's rows by learner (counting sort; rows keep their staging order inside a bucket)
        const int P = e->h.P, cap = e->h.capacity;
        int* off = e->stage_bucket;
        int* before = off + P + 1;
        int* leaf = before + P;
        std::fill(off, off + P + 1, 0);
        for (int i = 0; i < n; ++i) off[e->stage_slots[i] / cap + 1]++;
        for (int p = 0; p < P; ++p) off[p + 1] += off[p];
        std::vector<int>& cur = e->bucket_cursor;
        cur.assign(off, off + P);
        for (int i = 0; i < n; ++i) { const long long s = e->stage_slots[i]; const int p = (int)(s / cap); leaf[cur[p]++] = (int)(s - (long long)p * cap); }
        memcpy(before, e->size_flushed.data(), (size_t)P * sizeof(int));
        HIP_TRY(hipMemcpyAsync(e->d_stage_bucket, e->stage_bucket, (size_t)(2 * P + 1 + n) * sizeof(int), hipMemcpyHostToDevice, e->stream));
        PerArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.sum_tree = e->d_per_sum; pa.max_tree = e->d_per_max; pa.cap = cap; pa.n = n;
        hipLaunchKernelGGL(per_add_kernel, dim3(P), dim3(256), 0, e->stream, pa, (const int*)e->d_stage_bucket, P);
        HIP_TRY(hipGetLastError());
    }
    e->size_flushed = e->size;
    HIP_TRY(hipEventRecord(e->ev_stage, e->stream));
    e->stage_inflight = true;
    e->stage_n = 0;
    std::fill(e->staged_per_learner.begin(), e->staged_per_learner.end(), 0);
    return FRL_OK;
}

static int stage_one(frl_engine* e, int learner, const float* record) {
    if (learner < 0 || learner >= e->h.P) return fail(FRL_ERR_INVALID, "learner %d out of range", learner);
    // one scatter launch writes its rows in no particular order: never stage the same ring row twice
    if (e->stage_n == e->stage_cap || e->staged_per_learner[learner] >= e->h.capacity) { int rc = flush_stage(e); if (rc) return rc; }
    if (e->stage_inflight) {     // the pinned area may still be read by the previous flush's copy
        HIP_TRY(hipEventSynchronize(e->ev_stage));
        e->stage_inflight = false;
    }
    const RecordDesc& R = e->h.rec;
    memcpy(e->stage_rows + (size_t)e->stage_n * R.width, record, (size_t)R.width * sizeof(float));
    int& ix = e->index[learner];
    e->stage_slots[e->stage_n] = (long long)learner * e->h.capacity + ix;
    e->stage_n++;
    e->staged_per_learner[learner]++;
    ix = (ix + 1) % e->h.capacity;                        // Buffer.py:36
    if (e->size[learner] < e->h.capacity) e->size[learner]++;   // Buffer.py:37-38
    return FRL_OK;
}

extern "C" int frl_buffer_add(frl_engine* e, int learner, const float* record) {
    ENG(e);
    if (!record) return fail(FRL_ERR_INVALID, "record is NULL");
    return stage_one(e, learner, record);
}

extern "C" int frl_buffer_add_batch(frl_engine* e, int n, const int* learners, const float* records) {
    ENG(e);
    if (n < 0 || (n > 0 && !records)) return fail(FRL_ERR_INVALID, "bad batch");
    const int w = e->h.rec.width;
    for (int i = 0; i < n; ++i) {
        int rc = stage_one(e, learners ? learners[i] : 0, records + (size_t)i * w);
        if (rc) return rc;
    }
    return FRL_OK;
}

extern "C" int frl_buffer_flush(frl_engine* e) {
    ENG(e);
    return flush_stage(e);
}

extern "C" int frl_buffer_cursor_get(const frl_engine* e, int learner, int* index_out, int* size_out) {
    if (!e) return fail(FRL_ERR_INVALID, "engine is NULL");
    if (learner < 0 || learner >= e->h.P) return fail(FRL_ERR_INVALID, "learner out of range");
    if (index_out) *index_out = e->index[learner];
    if (size_out) *size_out = e->size[learner];
    return FRL_OK;
}

extern "C" int frl_buffer_cursor_set(frl_engine* e, int learner, int index, int size) {
    ENG(e);
    if (learner < 0 || learner >= e->h.P) return fail(FRL_ERR_INVALID, "learner out of range");
    if (index < 0 || index >= e->h.capacity || size < 0 || size > e->h.capacity)
        return fail(FRL_ERR_INVALID, "cursor (%d,%d) outside capacity %d", index, size, e->h.capacity);
    // rows staged before the cursor moves are committed first: afterwards the same ring slot may be staged again, and one
    // scatter launch must never hold a slot twice (it writes its rows in no particular order)
    int rc = flush_stage(e);
    if (rc) return rc;
    e->index[learner] = index;
    e->size[learner] = size;
    return FRL_OK;
}

extern "C" int frl_buffer_sample(frl_engine* e, int learner, const int64_t* idx, int batch, int n_fields,
                                 const int* col0, const int* ncols, float* const* out_device) {
    ENG(e);
    if (learner < 0 || learner >= e->h.P) return fail(FRL_ERR_INVALID, "learner out of range");
    if (batch < 0 || batch > e->h.batch_max * 16) return fail(FRL_ERR_INVALID, "batch %d exceeds 16*batch_max", batch);
    if (n_fields < 1 || n_fields > 8 || !idx || !col0 || !ncols || !out_device) return fail(FRL_ERR_INVALID, "bad field list");
    if (batch == 0) return FRL_OK;
    const RecordDesc& R = e->h.rec;
    GatherFields F;
    memset(&F, 0, sizeof F);
    F.n_fields = n_fields;
    for (int f = 0; f < n_fields; ++f) {
        if (col0[f] < 0 || ncols[f] < 1 || col0[f] + ncols[f] > R.width) return fail(FRL_ERR_INVALID, "field %d outside the record", f);
        F.col0[f] = col0[f]; F.ncols[f] = ncols[f]; F.out[f] = out_device[f];
    }
    for (int i = 0; i < batch; ++i)     // NumPy fancy indexing semantics: negative wraps, out of range raises
        if (idx[i] < -(int64_t)e->h.capacity || idx[i] >= e->h.capacity) return fail(FRL_ERR_INVALID, "index %lld out of range", (long long)idx[i]);
    int rc = flush_stage(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));            // h_idx64 reuse
    for (int i = 0; i < batch; ++i) e->h_idx64[i] = idx[i] < 0 ? idx[i] + e->h.capacity : idx[i];
    HIP_TRY(hipMemcpyAsync(e->d_idx64, e->h_idx64, (size_t)batch * sizeof(long long), hipMemcpyHostToDevice, e->stream));
    const float* ring = e->h.replay + (size_t)learner * e->h.capacity * R.stride;
    const int threads = 256, groups_per_block = threads / 16;
    hipLaunchKernelGGL(replay_gather_kernel, dim3((batch + groups_per_block - 1) / groups_per_block), dim3(threads), 0,
                       e->stream, ring, e->d_idx64, batch, R.stride, F);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));            // outputs are consumed on the caller's stream
    return FRL_OK;
}

extern "C" int frl_buffer_read(frl_engine* e, int learner, int row0, int n, float* out_host) {
    ENG(e);
    if (learner < 0 || learner >= e->h.P || row0 < 0 || n < 0 || row0 + n > e->h.capacity || !out_host)
        return fail(FRL_ERR_INVALID, "bad read range");
    if (n == 0) return FRL_OK;
    int rc = flush_stage(e);
    if (rc) return rc;
    const RecordDesc& R = e->h.rec;
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)n * R.width * sizeof(float)));
    const long long total = (long long)n * R.width;
    hipLaunchKernelGGL(replay_read_kernel, dim3((int)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, e->stream,
                       e->h.replay, (long long)learner * e->h.capacity + row0, n, R.width, R.stride, tmp);
    hipError_t err = hipMemcpyAsync(out_host, tmp, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    hipFree(tmp);
    if (err != hipSuccess) return fail(FRL_ERR_HIP, "frl_buffer_read: %s", hipGetErrorString(err));
    return FRL_OK;
}

extern "C" int frl_buffer_fill_synthetic(frl_engine* e, int rows, uint64_t seed) {
    ENG(e);
    if (rows < 0 || rows > e->h.capacity) return fail(FRL_ERR_INVALID, "rows out of range");
    if (e->per_on)       // priorities are assigned by add (PER_Buffer.add, Buffer.py:92-98): a bulk fill would leave the trees empty
        return fail(FRL_ERR_STATE, "frl_buffer_fill_synthetic bypasses the priority trees: fill a PER engine through frl_buffer_add_batch");
    int rc = flush_stage(e);
    if (rc) return rc;
    const RecordDesc& R = e->h.rec;
    for (int p = 0; p < e->h.P; ++p) {
        float* ring = e->h.replay + (size_t)p * e->h.capacity * R.stride;
        hipLaunchKernelGGL(replay_fill_kernel, dim3(std::min((rows + 255) / 256, 2048)), dim3(256), 0, e->stream, ring,
                           (long long)rows, R, e->h.n_discrete, seed + 0x632BE59BD9B4E019ull * (p + 1));
        e->index[p] = rows % e->h.capacity;
        e->size[p] = rows;
    }
    HIP_TRY(hipGetLastError());
    return FRL_OK;
}

// ------------------------------------------------------------------------------- parameters
extern "C" int frl_net_count(const frl_engine* e, int* n_out) {
    if (!e || !n_out) return fail(FRL_ERR_INVALID, "NULL argument");
    *n_out = e->h.n_nets;
    return FRL_OK;
}

extern "C" int frl_net_num_params(const frl_engine* e, int net, int* n_out) {
    if (!e || !n_out) return fail(FRL_ERR_INVALID, "NULL argument");
    if (net < 0 || net >= e->h.n_nets) return fail(FRL_ERR_INVALID, "net %d out of range", net);
    *n_out = e->h.net[net].n_params;
    return FRL_OK;
}

static float* kind_ptr(frl_engine* e, int kind) {
    switch (kind) {
        case FRL_PARAM_ONLINE: return e->h.theta;
        case FRL_PARAM_TARGET: return e->h.target;
        case FRL_PARAM_ADAM_M: return e->h.m;
        case FRL_PARAM_ADAM_V: return e->h.v;
        case FRL_PARAM_GRAD: return e->h.grad;
    }
    return nullptr;
}

static int params_xfer(frl_engine* e, int learner, int net, int kind, float* host, bool to_device) {
    ENG(e);
    if (!e->has_nets) return fail(FRL_ERR_STATE, "replay-only engine has no networks");
    if (learner < 0 || learner >= e->h.P) return fail(FRL_ERR_INVALID, "learner out of range");
    if (net < 0 || net >= e->h.n_nets) return fail(FRL_ERR_INVALID, "net %d out of range", net);
    float* base = kind_ptr(e, kind);
    if (!base || !host) return fail(FRL_ERR_INVALID, "bad kind / NULL buffer");
    const NetDesc& N = e->h.net[net];
    float* dev = base + (size_t)learner * e->h.learner_stride + e->h.net_off[net];
    // host order = the reference's state_dict order: per nn.Linear weight[out][in] then bias; device block =
    // Wk[k_pad][n_pad], or the fragment-image order of the register-chained engines (frl_desc.h: weight_index)
    std::vector<float> blk(N.size, 0.f);
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!to_device) {
        const int rs = solo_err_check(e);
        if (rs) return rs;
        HIP_TRY(hipMemcpy(blk.data(), dev, (size_t)N.size * sizeof(float), hipMemcpyDeviceToHost));
    }
    size_t o = 0;
    for (int i = 0; i < N.n_layers + N.n_shadow; ++i) {       // shadow layers (a noisy head's sigma) follow the forward ones
        const LayerDesc& L = N.L[i];
        for (int r = 0; r < L.n; ++r)
            for (int c = 0; c < L.k; ++c) {
                float& d = blk[L.w_off + weight_index(N, L, r, c)];
                if (to_device) d = host[o]; else host[o] = d;
                ++o;
            }
        for (int r = 0; r < L.n; ++r) {
            float& d = blk[L.b_off + r];
            if (to_device) d = host[o]; else host[o] = d;
            ++o;
        }
    }
    for (int j = 0; j < N.extra_n; ++j) {
        float& d = blk[N.extra_off + j];
        if (to_device) d = host[o]; else host[o] = d;
        ++o;
    }
    if (to_device) {
        HIP_TRY(hipMemcpy(dev, blk.data(), (size_t)N.size * sizeof(float), hipMemcpyHostToDevice));
        ++e->param_version;
    }
    return FRL_OK;
}

extern "C" int frl_params_get(frl_engine* e, int learner, int net, int kind, float* out_host) {
    return params_xfer(e, learner, net, kind, out_host, false);
}
extern "C" int frl_params_set(frl_engine* e, int learner, int net, int kind, const float* in_host) {
    return params_xfer(e, learner, net, kind, const_cast<float*>(in_host), true);
}

extern "C" int frl_params_pad_max(frl_engine* e, int learner, int net, int kind, float* max_abs_out) {
    ENG(e);
    if (!e->has_nets) return fail(FRL_ERR_STATE, "replay-only engine has no networks");
    if (learner < 0 || learner >= e->h.P || net < 0 || net >= e->h.n_nets || !max_abs_out) return fail(FRL_ERR_INVALID, "bad argument");
    float* base = kind_ptr(e, kind);
    if (!base) return fail(FRL_ERR_INVALID, "bad kind");
    const NetDesc& N = e->h.net[net];
    std::vector<float> blk(N.size, 0.f);
    std::vector<char> real(N.size, 0);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(blk.data(), base + (size_t)learner * e->h.learner_stride + e->h.net_off[net], (size_t)N.size * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < N.n_layers + N.n_shadow; ++i) {
        const LayerDesc& L = N.L[i];
        for (int r = 0; r < L.n; ++r) {
            for (int c = 0; c < L.k; ++c) real[L.w_off + weight_index(N, L, r, c)] = 1;
            real[L.b_off + r] = 1;
        }
    }
    for (int j = 0; j < N.extra_n; ++j) real[N.extra_off + j] = 1;
    float mx = 0.f;
    for (int i = 0; i < N.size; ++i)
        if (!real[i]) mx = std::max(mx, std::isnan(blk[i]) ? INFINITY : std::fabs(blk[i]));
    *max_abs_out = mx;
    return FRL_OK;
}

extern "C" int frl_opt_step_get(frl_engine* e, int learner, int net, int* t_out) {
    ENG(e);
    if (!e->has_nets || learner < 0 || learner >= e->h.P || net < 0 || net > kMaxNets || !t_out) return fail(FRL_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(t_out, e->h.steps + (size_t)learner * (kMaxNets + 1) + net, sizeof(int), hipMemcpyDeviceToHost));
    return FRL_OK;
}
extern "C" int frl_opt_step_set(frl_engine* e, int learner, int net, int t) {
    ENG(e);
    if (!e->has_nets || learner < 0 || learner >= e->h.P || net < 0 || net > kMaxNets) return fail(FRL_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->h.steps + (size_t)learner * (kMaxNets + 1) + net, &t, sizeof(int), hipMemcpyHostToDevice));
    return FRL_OK;
}

extern "C" int frl_alpha_get(frl_engine* e, int learner, float* vals4_out, int* step_out) {
    ENG(e);
    if (!e->has_nets || learner < 0 || learner >= e->h.P || !vals4_out) return fail(FRL_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(vals4_out, e->h.alpha + (size_t)learner * 4, 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (step_out) return frl_opt_step_get(e, learner, kMaxNets, step_out);
    return FRL_OK;
}
extern "C" int frl_alpha_set(frl_engine* e, int learner, const float* vals4, int step) {
    ENG(e);
    if (!e->has_nets || learner < 0 || learner >= e->h.P || !vals4) return fail(FRL_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->h.alpha + (size_t)learner * 4, vals4, 4 * sizeof(float), hipMemcpyHostToDevice));
    return frl_opt_step_set(e, learner, kMaxNets, step);
}

// -------------------------------------------------------------------------------------- act
// Exploration folded into an act launch (ActArgs' second half): the device pointers a collector owns.
struct ActExplore {
    frl_explore_args x;
    const float* scale_dev = nullptr;          // [P] or nullptr (x.scale for everybody)
    float* ou_state_dev = nullptr;             // [P][n_rows][A]
    const unsigned char* flags_dev = nullptr;  // [P][n_rows]
    float* env_out_dev = nullptr;              // [P][n_rows][A] (discrete: [P][n_rows])
    bool device_eps = false;
};

// build_only: fill *build_only with the launch's arguments and return without launching or consuming a Philox counter (the caller
// sets rng_counter: frl_rollout's folded step, whose act rides on the learn launches)
static int launch_act(frl_engine* e, int net, int mode_flags, int head, int use_target, int n_rows, int in_dim,
                      const float* in_dev, const float* eps_dev, float* out_dev, float* logp_dev, const ActExplore* ex = nullptr,
                      ActArgs* build_only = nullptr) {
    int mode = mode_flags & ~FRL_ACT_NO_OBSNORM;
    if (!e->has_nets) return fail(FRL_ERR_STATE, "replay-only engine has no networks");
    if (net < 0 || net >= e->h.n_nets) return fail(FRL_ERR_INVALID, "net %d out of range", net);
    const NetDesc& N = e->h.net[net];
    if (head < 0 || head >= N.heads) return fail(FRL_ERR_INVALID, "head out of range");
    const int nl = N.n_layers / N.heads;
    if (in_dim != N.L[head * nl].k) return fail(FRL_ERR_INVALID, "in_dim %d != layer input %d", in_dim, N.L[head * nl].k);
    if ((mode == FRL_ACT_SAC_SAMPLE || mode == FRL_ACT_PPO_SAMPLE) && N.extra_n == 0) return fail(FRL_ERR_INVALID, "net has no log_std");
    if (mode == FRL_ACT_CAT_SAMPLE && !eps_dev && !(ex && ex->device_eps)) return fail(FRL_ERR_INVALID, "FRL_ACT_CAT_SAMPLE needs the Exp(1) draws in eps");
    if (n_rows < 1) return fail(FRL_ERR_INVALID, "n_rows must be >= 1");
    const bool no_norm = (mode_flags & FRL_ACT_NO_OBSNORM) != 0;
    ActArgs a;
    memset(&a, 0, sizeof a);
    if (ex) {
        frl_explore_args x = ex->x;
        if (x.kind == FRL_EXPLORE_OFF) return fail(FRL_ERR_INVALID, "FRL_EXPLORE_OFF is a frl_rollout_args value; frl_explore_args.kind takes FRL_EXPLORE_NONE");
        if (x.kind < FRL_EXPLORE_NONE || x.kind > FRL_EXPLORE_OU) return fail(FRL_ERR_INVALID, "unknown exploration kind %d", x.kind);
        if (!ex->env_out_dev) return fail(FRL_ERR_INVALID, "exploration needs an env-action output");
        if (x.kind == FRL_EXPLORE_EPS_GREEDY && mode != FRL_ACT_ARGMAX) return fail(FRL_ERR_INVALID, "epsilon-greedy goes with FRL_ACT_ARGMAX");
        if ((x.kind == FRL_EXPLORE_GAUSS || x.kind == FRL_EXPLORE_OU) && !(mode == FRL_ACT_TANHHEAD || mode == FRL_ACT_SAC_SAMPLE))
            return fail(FRL_ERR_INVALID, "Gaussian / OU action noise goes with FRL_ACT_TANHHEAD or FRL_ACT_SAC_SAMPLE");
        if (x.kind == FRL_EXPLORE_OU && !ex->ou_state_dev) return fail(FRL_ERR_INVALID, "OU exploration needs a state buffer");
        a.explore = x.kind; a.device_eps = ex->device_eps ? 1 : 0;
        a.epsilon = x.epsilon; a.sigma = x.sigma; a.scale0 = x.scale; a.max_action = x.max_action != 0.f ? x.max_action : 1.f;
        a.ou_theta = x.ou_theta; a.ou_sigma = x.ou_sigma; a.ou_dt = x.ou_dt;
        a.scale = ex->scale_dev; a.ou_state = ex->ou_state_dev; a.flags = ex->flags_dev; a.env_out = ex->env_out_dev;
        if (!build_only) a.rng_counter = e->rng_counter++;
    }
    a.net = net; a.use_target = use_target; a.mode = mode; a.n_rows = n_rows; a.head = head; a.in_dim = in_dim;
    const int agent = e->h.n_agents > 1 ? net / 2 : 0;            // MADDPG: only the actors (even nets) take a single agent's obs
    a.normalize = (!no_norm && e->h.obs_norm_on && (e->h.n_agents == 1 || net % 2 == 0) && in_dim == e->h.rec.obs_dim[agent]) ? 1 : 0;
    a.in = in_dev; a.eps = eps_dev; a.out = out_dev; a.out_logp = logp_dev;
    if (build_only) { *build_only = a; return FRL_OK; }
    if (N.frag && (e->h.wide || e->h.solow)) {
        // fragment-image parameters of a shape act_frag_kernel does not take: the net is re-laid out to Wk in a scratch copy (one
        // small launch: the actor of config 4 is 67 k floats per learner) and act_kernel reads that
        // (config 4 at 512 learners: 34 M floats per net — a collector that steps its envs many times between two learn() calls
        // pays the pass once per update, not once per vector step)
        const int slot = 2 * net + (use_target ? 1 : 0);
        float* wk = e->d_act_wk + (size_t)slot * e->act_wk_slot;
        if (!e->d_act_wk || slot >= (int)e->act_wk_version.size()) return fail(FRL_ERR_STATE, "wide engine without its act scratch");
        if (e->act_wk_version[slot] != e->param_version) {
            // (enough workgroups per learner for a few thousand elements each, the chip's CUs at most twice over)
            const int per = std::max(1, std::min((N.size + 2047) / 2048, 2 * e->n_cus / e->h.P));
            hipLaunchKernelGGL(frag_to_wk_kernel, dim3(e->h.P, per), dim3(256), 0, e->stream, e->d, net, use_target, wk);
            e->act_wk_version[slot] = e->param_version;
        }
        a.theta_alt = wk;
        dim3 grid((n_rows + e->h.rc - 1) / e->h.rc, e->h.P);
        hipLaunchKernelGGL(act_kernel, grid, dim3(256), e->lds_bytes, e->stream, e->d, a);
    } else if (N.frag) {       // parameters in fragment-image order: the register-chained forward (kernels_act.hip)
        hipLaunchKernelGGL(act_frag_kernel, dim3((n_rows + 63) / 64, e->h.P), dim3(256), (size_t)critic2_lds_floats() * sizeof(float),
                           e->stream, e->d, a);
    } else {
        dim3 grid((n_rows + e->h.rc - 1) / e->h.rc, e->h.P);
        hipLaunchKernelGGL(act_kernel, grid, dim3(256), e->lds_bytes, e->stream, e->d, a);
    }
    HIP_TRY(hipGetLastError());
    return FRL_OK;
}

extern "C" int frl_act_device(frl_engine* e, int net, int mode, int head, int use_target, int n_rows, int in_dim,
                              const float* in_dev, const float* eps_dev, float* out_dev, float* logp_dev) {
    ENG(e);
    if (!in_dev || !out_dev) return fail(FRL_ERR_INVALID, "NULL device buffer");
    return launch_act(e, net, mode, head, use_target, n_rows, in_dim, in_dev, eps_dev, out_dev, logp_dev);
}

extern "C" int frl_act(frl_engine* e, int net, int mode, int head, int use_target, int n_rows, int in_dim,
                       const float* in_host, const float* eps_host, float* out_host, float* logp_host) {
    ENG(e);
    if (!e->has_nets) return fail(FRL_ERR_STATE, "replay-only engine has no networks");
    if (net < 0 || net >= e->h.n_nets) return fail(FRL_ERR_INVALID, "net %d out of range", net);
    if (!in_host || !out_host || n_rows < 1) return fail(FRL_ERR_INVALID, "bad argument");
    const NetDesc& N = e->h.net[net];
    const int nl = N.n_layers / N.heads;
    if (head < 0 || head >= N.heads) return fail(FRL_ERR_INVALID, "head out of range");
    const int nout = N.L[head * nl + nl - 1].n;
    const size_t rows = (size_t)e->h.P * n_rows;
    const size_t in_n = rows * in_dim, out_n = rows * nout;
    if (in_n > e->act_in_cap) {
        if (e->d_act_in) hipFree(e->d_act_in);
        e->d_act_in = nullptr;
        HIP_TRY(hipMalloc((void**)&e->d_act_in, in_n * sizeof(float)));
        e->act_in_cap = in_n;
    }
    if (out_n > e->act_out_cap) {
        for (float** p : {&e->d_act_eps, &e->d_act_out, &e->d_act_logp}) { if (*p) hipFree(*p); *p = nullptr; }
        HIP_TRY(hipMalloc((void**)&e->d_act_eps, out_n * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&e->d_act_out, out_n * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&e->d_act_logp, out_n * sizeof(float)));
        e->act_out_cap = out_n;
    }
    HIP_TRY(hipMemcpyAsync(e->d_act_in, in_host, in_n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (eps_host) HIP_TRY(hipMemcpyAsync(e->d_act_eps, eps_host, out_n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    int rc = launch_act(e, net, mode, head, use_target, n_rows, in_dim, e->d_act_in, eps_host ? e->d_act_eps : nullptr,
                        e->d_act_out, logp_host ? e->d_act_logp : nullptr);
    if (rc) return rc;
    const int base_mode = mode & ~FRL_ACT_NO_OBSNORM;
    const bool one_per_row = (base_mode == FRL_ACT_ARGMAX || base_mode == FRL_ACT_CAT_SAMPLE);
    const size_t got = one_per_row ? rows : out_n;
    HIP_TRY(hipMemcpyAsync(out_host, e->d_act_out, got * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (logp_host) HIP_TRY(hipMemcpyAsync(logp_host, e->d_act_logp, got * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return FRL_OK;
}

// select_action + the reference loop's exploration rule in ONE launch, host in / host out (tests, single-env callers): the
// rollout collectors use the same launch with device-resident buffers.  mode: FRL_ACT_ARGMAX (+ epsilon-greedy),
// FRL_ACT_TANHHEAD / FRL_ACT_SAC_SAMPLE (+ Gaussian or OU action noise, or none).  Sampling modes draw their own eps.
extern "C" int frl_act_explore(frl_engine* e, int mode, int n_rows, const float* obs_host, const frl_explore_args* x,
                               const uint8_t* ended_host, float* store_act_out, float* env_act_out) {
    ENG(e);
    if (!e->has_nets) return fail(FRL_ERR_STATE, "replay-only engine has no networks");
    if (!obs_host || !x || !store_act_out || !env_act_out || n_rows < 1) return fail(FRL_ERR_INVALID, "bad argument");
    const EngineDesc& h = e->h;
    if (h.n_agents != 1) return fail(FRL_ERR_STATE, "frl_act_explore: single-agent engines");
    const int O = h.rec.obs_dim[0], nout = h.net[0].L[h.net[0].n_layers / h.net[0].heads - 1].n;
    const bool disc = (mode == FRL_ACT_ARGMAX);
    const size_t rows = (size_t)h.P * n_rows, in_n = rows * O, out_n = rows * nout;
    if (in_n > e->act_in_cap) {
        if (e->d_act_in) hipFree(e->d_act_in);
        e->d_act_in = nullptr;
        HIP_TRY(hipMalloc((void**)&e->d_act_in, in_n * sizeof(float)));
        e->act_in_cap = in_n;
    }
    if (out_n > e->act_out_cap) {
        for (float** p : {&e->d_act_eps, &e->d_act_out, &e->d_act_logp}) { if (*p) hipFree(*p); *p = nullptr; }
        HIP_TRY(hipMalloc((void**)&e->d_act_eps, out_n * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&e->d_act_out, out_n * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&e->d_act_logp, out_n * sizeof(float)));
        e->act_out_cap = out_n;
    }
    if (out_n != e->ou_n) {                 // OU state of the host-facing entry point: one process per (learner, row, dimension)
        if (e->d_ou) hipFree(e->d_ou);
        e->d_ou = nullptr; e->ou_n = 0;
        if (e->d_ou_flags) hipFree(e->d_ou_flags);
        e->d_ou_flags = nullptr;
        HIP_TRY(hipMalloc((void**)&e->d_ou, out_n * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&e->d_ou_flags, rows));
        HIP_TRY(hipMemsetAsync(e->d_ou, 0, out_n * sizeof(float), e->stream));
        e->ou_n = out_n;
    }
    HIP_TRY(hipMemcpyAsync(e->d_act_in, obs_host, in_n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (ended_host) {
        std::vector<unsigned char> fl(rows);
        for (size_t i = 0; i < rows; ++i) fl[i] = ended_host[i] ? 2 : 0;
        HIP_TRY(hipMemcpy(e->d_ou_flags, fl.data(), rows, hipMemcpyHostToDevice));
    }
    ActExplore ex;
    ex.x = *x; ex.ou_state_dev = e->d_ou; ex.flags_dev = ended_host ? e->d_ou_flags : nullptr;
    ex.env_out_dev = e->d_act_eps;           // scratch of the same size, unused by a device_eps launch
    ex.device_eps = true;
    int rc = launch_act(e, 0, mode, 0, 0, n_rows, O, e->d_act_in, nullptr, e->d_act_out, nullptr, &ex);
    if (rc) return rc;
    const size_t got = disc ? rows : out_n;
    HIP_TRY(hipMemcpyAsync(store_act_out, e->d_act_out, got * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(env_act_out, e->d_act_eps, got * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return FRL_OK;
}

// ------------------------------------------------------------------------------------ learn
static int upload_idx_noise(frl_engine* e, const int64_t* idx, const float* noise, int batch, int n_sets_per_learner) {
    const EngineDesc& h = e->h;
    if (idx || noise) HIP_TRY(hipStreamSynchronize(e->stream));    // pinned scratch reuse
    if (idx) {
        for (int p = 0; p < h.P; ++p)
            for (int a = 0; a < n_sets_per_learner; ++a) {
                const int64_t* src = idx + ((size_t)p * n_sets_per_learner + a) * batch;
                int* dst = e->h_idx + ((size_t)p * h.n_agents + a) * h.batch_max;
                const int sz = e->size[p];
                for (int i = 0; i < batch; ++i) {
                    int64_t v = src[i];
                    if (v < 0) v += h.capacity;
                    if (v < 0 || v >= h.capacity) return fail(FRL_ERR_INVALID, "sample index %lld out of range", (long long)src[i]);
                    (void)sz;
                    dst[i] = (int)v;
                }
            }
        HIP_TRY(hipMemcpyAsync(h.idx, e->h_idx, e->idx_count * sizeof(int), hipMemcpyHostToDevice, e->stream));
    }
    if (noise) {
        // host [P][n_agents][noise_sets][batch][act_max] -> device [P][n_agents][noise_sets][batch_max][act_max]
        const int am = h.act_max;
        for (size_t s = 0; s < (size_t)h.P * h.n_agents * h.noise_sets; ++s)
            memcpy(e->h_noise + s * h.batch_max * am, noise + s * batch * am, (size_t)batch * am * sizeof(float));
        HIP_TRY(hipMemcpyAsync(h.noise, e->h_noise, e->noise_count * sizeof(float), hipMemcpyHostToDevice, e->stream));
    }
    return FRL_OK;
}

extern "C" int frl_stats_get(frl_engine* e, float* out_host) {
    ENG(e);
    if (!e->has_nets || !out_host) return fail(FRL_ERR_INVALID, "bad argument");
    HIP_TRY(hipMemcpyAsync(out_host, e->h.stats, (size_t)e->h.P * e->h.n_agents * ST_COUNT * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return solo_err_check(e);
}

extern "C" int frl_last_indices(frl_engine* e, int batch, int64_t* out_host) {
    ENG(e);
    if (!e->has_nets || !out_host || batch < 1 || batch > e->h.batch_max) return fail(FRL_ERR_INVALID, "bad argument");
    const size_t units = (size_t)e->h.P * e->h.n_agents;
    std::vector<int> tmp(units * e->h.batch_max);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(tmp.data(), e->h.idx, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t u = 0; u < units; ++u)
        for (int i = 0; i < batch; ++i) out_host[u * batch + i] = tmp[u * e->h.batch_max + i];
    return FRL_OK;
}

// Host noise of `n_sets` forwards -> the device layout of kernels_noisy.hip.  Host order per forward (frl_noisy_eps_size
// floats): per NoisyLinear eps_in[hidden] then eps_out[rows], V before A for a Dueling head.
static int noisy_upload(frl_engine* e, const float* eps_host, int set0, int n_sets) {
    const EngineDesc& h = e->h;
    const LayerDesc& H = h.net[0].L[h.net[0].n_layers - 1];
    const int per = e->noisy_per_set, K = H.k, kp = H.k_pad, np_ = H.n_pad;
    const int rows0 = h.dueling ? h.noisy_split : H.n, rows1 = h.dueling ? H.n - h.noisy_split : 0;
    const int host_per = K + rows0 + (rows1 ? K + rows1 : 0);
    HIP_TRY(hipStreamSynchronize(e->stream));                       // pinned staging reuse
    for (int p = 0; p < h.P; ++p)
        for (int s = 0; s < n_sets; ++s) {
            const float* src = eps_host + ((size_t)p * n_sets + s) * host_per;
            float* dst = e->h_noisy + ((size_t)p * 3 + set0 + s) * per;
            memset(dst, 0, (size_t)per * sizeof(float));
            memcpy(dst, src, (size_t)K * sizeof(float));                                   // eps_in of sub-layer 0
            memcpy(dst + kp, src + K, (size_t)rows0 * sizeof(float));                      // eps_out of sub-layer 0: rows [0, rows0)
            if (rows1) {
                memcpy(dst + kp + np_, src + K + rows0, (size_t)K * sizeof(float));        // eps_in of sub-layer 1
                memcpy(dst + kp + np_ + kp + rows0, src + K + rows0 + K, (size_t)rows1 * sizeof(float));   // rows [rows0, n)
            }
        }
    for (int p = 0; p < h.P; ++p)
        HIP_TRY(hipMemcpyAsync(h.noisy_eps + ((size_t)p * 3 + set0) * per, e->h_noisy + ((size_t)p * 3 + set0) * per,
                               (size_t)n_sets * per * sizeof(float), hipMemcpyHostToDevice, e->stream));
    return FRL_OK;
}

extern "C" int frl_noisy_eps_size(const frl_engine* e, int* n_out) {
    if (!e || !n_out) return fail(FRL_ERR_INVALID, "NULL argument");
    *n_out = 0;
    if (!e->h.noisy) return FRL_OK;
    const EngineDesc& h = e->h;
    const LayerDesc& H = h.net[0].L[h.net[0].n_layers - 1];
    const int rows0 = h.dueling ? h.noisy_split : H.n, rows1 = h.dueling ? H.n - h.noisy_split : 0;
    *n_out = H.k + rows0 + (rows1 ? H.k + rows1 : 0);
    return FRL_OK;
}

// A forward of the online net with fresh noise (select_action on a noisy net, DQN_with_tricks.py:213-216 with
// Noisy_net.py:41-44): materialises effective set 0; frl_act(..., use_target = 2, ...) then reads it.
extern "C" int frl_noisy_resample(frl_engine* e, const float* eps_host) {
    ENG(e);
    if (!e->h.noisy) return fail(FRL_ERR_STATE, "engine has no NoisyLinear head");
    if (eps_host) { int rc = noisy_upload(e, eps_host, 0, 1); if (rc) return rc; }
    else hipLaunchKernelGGL(noisy_draw_kernel, dim3(e->h.P, 1), dim3(256), 0, e->stream, e->d, 0, 1, e->rng_counter++);
    hipLaunchKernelGGL(noisy_materialise_kernel, dim3(e->h.P), dim3(256), 0, e->stream, e->d, 0, 1, 0);
    HIP_TRY(hipGetLastError());
    return FRL_OK;
}

// One stage of learn() for learners [p0, p0 + pc) on `st`: stage 0 = [draw, obsnorm,] grad(critic | Q) + reduce + adam;
// stage 1 = grad(actor) + reduce + adam; stage 2 = MADDPG's soft update.
// reduce + clip + Adam (+ soft update) of every unit's net `ad.which`: one fused launch when each net's gradient fits the
// registers of one workgroup, else the two streaming passes
static void launch_adam(frl_engine* e, hipStream_t st, const AdamArgs& ad, int units, dim3 grid_adam) {
    const EngineDesc& h = e->h;
    int max_n4 = 0;
    for (int ag = 0; ag < h.n_agents; ++ag) {
        const int net = (h.algo == ALGO_DQN) ? 0 : (ad.which == 0 ? 2 * ag + 1 : 2 * ag);
        max_n4 = std::max(max_n4, h.net[net].size / 4);
    }
    const bool two_pass = getenv("FRL_ADAM_TWO_PASS") != nullptr;
    if (max_n4 <= kFusedThreads * kFusedVec && !h.noisy && !two_pass) {
        hipLaunchKernelGGL(adam_fused_kernel, dim3(units), dim3(kFusedThreads), 0, st, e->d, ad);
    } else if (max_n4 <= kFusedThreads * kFusedVecWide && !two_pass) {       // (also every NoisyLinear head: the sigma gradients)
        hipLaunchKernelGGL(adam_fused_wide_kernel, dim3(units), dim3(kFusedThreads), 0, st, e->d, ad);
    } else {
        hipLaunchKernelGGL(reduce_kernel, grid_adam, dim3(256), 0, st, e->d, ad);
        hipLaunchKernelGGL(adam_kernel, grid_adam, dim3(256), 0, st, e->d, ad);
    }
}

// One learner per workgroup, register-chained, Adam fused (kernels_critic2.hip / kernels_actor2.hip): the reference's standard
// narrow shape at populations that give every CU a learner; everything else takes the row-chunk kernels + reduce / Adam
// launches.  The family is chosen ONCE, at frl_create (chained_shape + population, FRL_CRITIC_V2=0/1 overrides the population
// threshold — the tests run both families on the same inputs): the chained kernels keep the nets in fragment-image order in
// HBM (NetDesc::frag), which the row-chunk kernels do not read.
static bool chained_shape(const EngineDesc& h) {
    const NetDesc &NA0 = h.net[0], &NC0 = h.net[1];
    auto packed = [](const NetDesc& N) {          // every head block at the offsets the kernels hard-code (frl_desc.h: kL1w ...)
        for (int hd = 0; hd < N.heads; ++hd) {
            const LayerDesc* L = N.L + 3 * hd;
            const int b = hd * kHeadFloats;
            if (L[0].w_off != b + kL1w || L[0].b_off != b + kL1b || L[1].w_off != b + kL2w || L[1].b_off != b + kL2b ||
                L[2].w_off != b + kL3w || L[2].b_off != b + kL3b) return false;
        }
        return N.extra_n == 0 || (N.heads == 1 && N.extra_off == kHeadFloats);
    };
    if (NA0.n_layers != 3 || NC0.n_layers != 3 * NC0.heads || h.hidden != 128 || !packed(NA0) || !packed(NC0)) return false;
    // ([s | a] as one aligned slice of the record: the chained kernels read a lane's four columns of it as one dwordx4)
    if (h.rec.obs_off[0] % 4 != 0 || h.rec.act_off[0] != h.rec.obs_off[0] + h.rec.obs_dim[0] || h.rec.obs_off[0] + 16 > h.rec.stride) return false;
    return (h.algo == ALGO_DDPG || h.algo == ALGO_TD3 || h.algo == ALGO_SAC) && h.n_agents == 1 && h.hidden == 128 &&
           NA0.L[0].k_pad == 16 && NC0.L[0].k_pad == 16 && h.rec.act_dim[0] <= 4 && NA0.L[2].n_pad == 16 &&
           h.batch_max <= 256 && NA0.hidden_act == ACT_RELU && NC0.hidden_act == ACT_RELU &&
           NA0.n_layers == 3 && NC0.n_layers == 3 * NC0.heads;
}
static bool chained_path(const EngineDesc& h, int batch, int pc) {
    (void)pc;
    if (h.wide) return !h.obs_norm_on;          // any batch <= batch_max: super-chunks of 256 rows
    if (h.solow) return !h.obs_norm_on;         // any batch <= batch_max: a 16-row tile per workgroup, h.solow of them per unit
    return h.net[0].frag && h.net[1].frag && batch <= 256 && !h.obs_norm_on;
}
// The K-sliced chained family (device/chain_wide.hpp): the reference's hidden-128 ReLU actor-critic nets with first layers of up
// to 416 input columns and actor heads of up to 32 outputs that chained_shape() does not admit — SAC / TD3 / DDPG on wide
// observations (config 4: Humanoid's 376 + 17), MADDPG_simple's per-agent actors and centralised critics (config 5).
// MATD3 (MATD3_simple.py:195-262) is the same launch pair with twin critics, set j of the unit's noise on agent j's target action
// and the host's delayed actor / soft-update stages.
static bool wide_shape(const EngineDesc& h) {
    const bool single = (h.algo == ALGO_DDPG || h.algo == ALGO_TD3 || h.algo == ALGO_SAC) && h.n_agents == 1;
    const bool multi = h.algo == ALGO_MADDPG && h.n_agents >= 1;
    const int H = h.hidden;                    // 128: chain_wide.hpp; 256: chain_wide16.hpp
    if (!(single || multi) || (H != 128 && H != 256) || h.rec.act_total > kWideApitch) return false;
    // WideNet::stage_idx (chain_wide.hpp; also the hidden-256 kernels') copies the batch's ring indices into the 64 KB LDS union:
    // 2 * kWideSlice = 16384 ints.  Larger batches stay with the row-chunk family.
    if (h.batch_max > 2 * kWideSlice) return false;
    const int nt3 = h.net[0].L[2].n_pad;
    for (int j = 0; j < h.n_agents; ++j) {
        const NetDesc &NA0 = h.net[2 * j], &NC0 = h.net[2 * j + 1];
        if (NA0.n_layers != 3 || NA0.heads != 1 || NC0.n_layers != 3 * NC0.heads || NC0.heads != h.net[1].heads) return false;
        if (NA0.hidden_act != ACT_RELU || NC0.hidden_act != ACT_RELU) return false;
        if (NA0.L[0].k_pad > 16 * kWideMaxKB1 || NC0.L[0].k_pad > 16 * kWideMaxKB1) return false;
        if (NA0.L[2].n_pad != nt3 || nt3 > 32) return false;                 // one head-tile count for every agent's actor
        for (int hd = 0; hd < NC0.heads; ++hd)
            if (NC0.L[3 * hd].n_pad != H || NC0.L[3 * hd + 1].n_pad != H || NC0.L[3 * hd + 1].k_pad != H || NC0.L[3 * hd + 2].n_pad != 16) return false;
        if (NA0.L[0].n_pad != H || NA0.L[1].n_pad != H || NA0.L[1].k_pad != H) return false;
    }
    return true;
}

// kernels_dqn2.hip: the reference's Q-net (obs -> 128 -> n_actions, or the Dueling [V ; A] head) with the TD update of DQN.py and
// DQN_with_tricks.py's Double / PER-weighted variants; Noisy and Categorical heads take the row-chunk chain.  FRL_DQN_FUSED=0/1 overrides.
static bool dqn_fused_path(const EngineDesc& h, int batch, bool per_weights) {
    const NetDesc& N = h.net[0];
    (void)per_weights;          // PER's importance weights (mean or per-row) are applied in the launch
    const bool shape = h.algo == ALGO_DQN && !h.noisy && !h.c51_atoms && h.hidden == 128 && N.n_layers == 2 &&
                       N.L[0].k_pad == 16 && N.L[1].n_pad == 16 && batch <= kDqn2Batch && !h.obs_norm_on && N.hidden_act == ACT_RELU;
    const char* force = getenv("FRL_DQN_FUSED");
    return shape && (force ? atoi(force) != 0 : true);
}

static void launch_learn_stage(frl_engine* e, hipStream_t st, LearnArgs a, int stage, int p0, int pc, bool dev_rng, bool needs_noise,
                               const DqnStepArgs* step = nullptr, const SoloStepArgs* sstep = nullptr) {
    const EngineDesc& h = e->h;
    a.p0 = p0; a.p_count = pc;
    const int ns = ((a.batch + h.rc - 1) / h.rc + h.cps - 1) / h.cps;      // workgroups (= slabs) per unit
    const int units = pc * h.n_agents;
    const dim3 grid_chunks(((units + 7) / 8) * 8 * ns), grid_units(units), blk(256), grid_adam(units * h.Gmax);
    const bool sac = h.algo == ALGO_SAC, maddpg = h.algo == ALGO_MADDPG;
    AdamArgs ad;
    memset(&ad, 0, sizeof ad);
    ad.ns = ns; ad.batch = a.batch; ad.eps = a.adam_eps; ad.beta1 = a.beta1; ad.beta2 = a.beta2; ad.clip = a.clip_norm;
    ad.tau = a.tau; ad.alpha_lr = a.alpha_lr; ad.target_entropy = a.target_entropy; ad.p0 = p0; ad.G = h.Gmax;
    const bool v2 = chained_path(h, a.batch, pc);
    if (stage == 0 && dqn_fused_path(h, a.batch, a.use_isw != 0)) {
        a.dqn_split = dqn_split_for(h, a.batch, pc);
        prof_begin(e, PK_GRAD_CRITIC);
        DqnStepArgs sa;
        memset(&sa, 0, sizeof sa);
        if (step) sa = *step;
        hipLaunchKernelGGL(dqn_fused_kernel, dim3(pc * a.dqn_split), blk, (size_t)dqn2_lds_floats() * sizeof(float), st, e->d, a, sa);
        prof_end(e);
        return;
    }
    // kernels_solow.hip, MADDPG without smoothing noise: the rows may have been drawn by the previous launch's spare workgroups (one per
    // unit, the duplicate table in their own LDS) — for exactly this counter, ring size and batch, or draw_kernel runs as ever
    const int ma_pre_stride = 8 + h.batch_max;
    const bool ma_solow = v2 && h.solow && h.n_agents > 1 && stage == 0 && dev_rng && !needs_noise && pc == h.P;
    const bool ma_use_pre = ma_solow && e->ma_pre_valid && e->ma_pre_counter == a.rng_counter && e->ma_pre_size == a.size && e->ma_pre_batch == a.batch;
    if (stage == 0) {
        if (dev_rng && !ma_use_pre && !(v2 && (h.solo || (h.solow && h.n_agents == 1)))) {    // (kernels_solo.hip / single-agent kernels_solow.hip draw inside their critic stages)
            prof_begin(e, PK_DRAW);
            const char* scan = getenv("FRL_DRAW_SCAN");                        // developer / test knob: no duplicate table
            const bool table = a.batch > 256 && 4 * a.batch <= kDrawTableHost && !(scan && atoi(scan) != 0);
            const size_t draw_lds = ((size_t)2 * ((a.batch + 3) & ~3) + (table ? 2 * kDrawTableHost : 0)) * sizeof(int);
            hipLaunchKernelGGL(draw_kernel, grid_units, blk, draw_lds, st, e->d, a, (needs_noise ? 1 : 0) | (table ? 0 : 2));
            prof_end(e);
        }
        if (h.obs_norm_on && h.algo != ALGO_DQN)                         // sample(): norm(obs) updates the statistics first
            hipLaunchKernelGGL(obsnorm_kernel, dim3(pc), blk, 0, st, e->d, a.batch, 0, p0);
        if (h.noisy)      // sets: 0 online on s' (Double only), 1 target on s', 2 online on s
            hipLaunchKernelGGL(noisy_materialise_kernel, dim3(h.P), blk, 0, st, e->d, 0, 3, 0x2);
        if (v2 && h.wide) {                               // kernels_criticw.hip: one workgroup per (learner, agent)
            prof_begin(e, PK_GRAD_CRITIC);
            const bool x = h.wide == 2;
            const size_t lb = (size_t)(x ? wide16_lds_floats_host() : wide_lds_floats()) * sizeof(float);
            const bool twin = h.net[1].heads == 2, a2 = h.net[0].L[2].n_pad > 16;
            auto k = x ? (twin ? (a2 ? ac_critic_x_h2a2_kernel : ac_critic_x_h2a1_kernel) : (a2 ? ac_critic_x_h1a2_kernel : ac_critic_x_h1a1_kernel))
                       : (twin ? (a2 ? ac_critic_wide_h2a2_kernel : ac_critic_wide_h2a1_kernel) : (a2 ? ac_critic_wide_h1a2_kernel : ac_critic_wide_h1a1_kernel));
            hipLaunchKernelGGL(k, dim3(units), blk, lb, st, e->d, a);
            prof_end(e);
            return;
        }
        if (v2 && h.solow) {                              // kernels_solow.hip: sixteen workgroups per learner, W1 streamed from the block
            prof_begin(e, PK_GRAD_CRITIC);
            SoloArgs sa{e->d_solo_slab, e->d_solo_part, e->d_solo_bar, e->d_solo_err, e->solo_bar_base, e->solo_stride, nullptr, nullptr, 0ull, h.solow, e->d_solow_bar2, e->solow_row_wgs, e->solow_wgs};
            e->solo_bar_base += kSoloWG;
            // the next call's rows drawn by the learners' first helper workgroups (kernels_solo.hip's spare-workgroup scheme: two
            // alternating slots, a tag the reader checks; FRL_SOLO_PREDRAW=0 switches it off)
            if (dev_rng && e->d_solo_pre && pc == h.P && h.n_agents == 1) {
                const char* pdf = getenv("FRL_SOLO_PREDRAW");
                sa.pre_read = e->d_solo_pre + (size_t)(e->solo_pre_seq & 1) * h.P * kSoloPre;      // (stale or foreign tags fail the kernel's check)
                if (e->solow_wgs > e->solow_row_wgs && !(pdf && atoi(pdf) == 0)) {
                    sa.pre_write = e->d_solo_pre + (size_t)((e->solo_pre_seq + 1) & 1) * h.P * kSoloPre;
                    sa.pre_counter = e->rng_counter;              // what the next frl_learn takes, unless something else draws first
                }
                ++e->solo_pre_seq;
            }
            const bool twin = h.net[1].heads == 2, a2 = h.net[0].L[2].n_pad > 16;
            auto k = h.n_agents > 1 ? (twin ? (a2 ? solow_critic_ma_h2a2_kernel : solow_critic_ma_h2a1_kernel) : (a2 ? solow_critic_ma_h1a2_kernel : solow_critic_ma_h1a1_kernel))
                                    : (twin ? (a2 ? solow_critic_h2a2_kernel : solow_critic_h2a1_kernel) : (a2 ? solow_critic_h1a2_kernel : solow_critic_h1a1_kernel));
            if (a.fuse_actor) k = twin ? (a2 ? solow_step_h2a2_kernel : solow_step_h2a1_kernel) : (a2 ? solow_step_h1a2_kernel : solow_step_h1a1_kernel);
            int extra = 0;
            if (h.n_agents > 1) {
                const char* pdf = getenv("FRL_SOLO_PREDRAW");
                if (ma_use_pre) sa.pre_read = e->d_solo_pre + (size_t)(e->solo_pre_seq & 1) * h.P * h.n_agents * ma_pre_stride;
                e->ma_pre_valid = false;
                if (ma_solow && units * (e->solow_wgs + 1) <= e->n_cus && !(pdf && atoi(pdf) == 0)) {
                    sa.pre_write = e->d_solo_pre + (size_t)((e->solo_pre_seq + 1) & 1) * h.P * h.n_agents * ma_pre_stride;
                    sa.pre_counter = e->rng_counter;              // what the next frl_learn takes, unless something else draws first
                    extra = units;
                    e->ma_pre_valid = true; e->ma_pre_counter = e->rng_counter; e->ma_pre_size = a.size; e->ma_pre_batch = a.batch;
                }
                ++e->solo_pre_seq;
            }
            hipLaunchKernelGGL(k, dim3(units * e->solow_wgs + extra), blk, (size_t)solow_lds_floats() * sizeof(float), st, e->d, a, sa);
            prof_end(e);
            return;
        }
        if (v2 && h.solo) {                               // kernels_solo.hip: sixteen workgroups per learner, reduce + Adam behind grid barriers
            prof_begin(e, PK_GRAD_CRITIC);
            SoloArgs sa{e->d_solo_slab, e->d_solo_part, e->d_solo_bar, e->d_solo_err, e->solo_bar_base, e->solo_stride, nullptr, nullptr, 0ull};
            e->solo_bar_base += kSoloWG;
            SoloStepArgs ss;
            memset(&ss, 0, sizeof ss);
            if (sstep) ss = *sstep;
            const size_t lb = (size_t)std::max(solo_lds_floats(), critic2_lds_floats()) * sizeof(float);
            // the next call's rows drawn by pc spare workgroups of this launch (plain frl_learn calls with device draws; the spare ones
            // need a CU of their own — 117 KB of LDS — next to the learners' pc x 16: FRL_SOLO_PREDRAW=0/1 overrides)
            const char* pdf = getenv("FRL_SOLO_PREDRAW");
            const int W = h.solo;
            const bool predraw = dev_rng && !sstep && e->d_solo_pre && pc == h.P && pc * (W + 1) <= e->n_cus && !(pdf && atoi(pdf) == 0);
            int extra = 0;
            if (dev_rng && !sstep && e->d_solo_pre && pc == h.P) {
                sa.pre_read = e->d_solo_pre + (size_t)(e->solo_pre_seq & 1) * h.P * kSoloPre;      // (stale or foreign tags fail the kernel's check)
                if (predraw) {
                    sa.pre_write = e->d_solo_pre + (size_t)((e->solo_pre_seq + 1) & 1) * h.P * kSoloPre;
                    sa.pre_counter = e->rng_counter;              // what the next frl_learn takes, unless something else draws first
                    extra = pc;
                }
                ++e->solo_pre_seq;
            }
            const bool twin = h.net[1].heads == 2;
            auto k = W == 16 ? (twin ? solo_critic_twin_kernel : solo_critic_single_kernel) : (twin ? solo_critic_twin_w8_kernel : solo_critic_single_w8_kernel);
            hipLaunchKernelGGL(k, dim3(pc * W + extra), blk, lb, st, e->d, a, sa, ss);
            prof_end(e);
            return;
        }
        if (v2) {
            { const char* sg = getenv("FRL_STAGGER"); a.stagger = sg ? atoi(sg) : 0;         // developer knob: spread the Adam bursts of the first round
              const char* gg = getenv("FRL_STAGGER_GROUPS"); a.stagger_groups = gg ? atoi(gg) : 4; a.stagger_wgs = e->n_cus; }
            prof_begin(e, PK_GRAD_CRITIC);
            const bool w8 = e->chain_waves == 8, twin = h.net[1].heads == 2;
            const size_t lb = (size_t)(w8 ? critic8_lds_floats() : critic2_lds_floats()) * sizeof(float);
            // (_nv: next_obs 16-byte aligned with its 16 columns inside the row, (reward, done) an aligned pair)
            const RecordDesc& R = h.rec;
            const bool nv = R.nobs_off[0] % 4 == 0 && R.nobs_off[0] + 16 <= R.stride && R.rew_off % 2 == 0 && R.done_off == R.rew_off + 1 && !getenv("FRL_CRITIC2_NOVEC");
            auto k = w8 ? (nv ? (twin ? ac_critic_v2_twin_nv_kernel : ac_critic_v2_single_nv_kernel) : (twin ? ac_critic_v2_twin_kernel : ac_critic_v2_single_kernel))
                        : (twin ? ac_critic_v2w4_twin_kernel : ac_critic_v2w4_single_kernel);
            hipLaunchKernelGGL(k, dim3(pc), dim3(w8 ? 512 : 256), lb, st, e->d, a);
            prof_end(e);
            return;
        }
        prof_begin(e, PK_GRAD_CRITIC);
        if (h.algo == ALGO_DQN && h.c51_atoms) hipLaunchKernelGGL(c51_grad_kernel, grid_chunks, blk, e->lds_bytes, st, e->d, a, ns);
        else if (h.algo == ALGO_DQN) hipLaunchKernelGGL(dqn_grad_kernel, grid_chunks, blk, e->lds_bytes, st, e->d, a, ns);
        else hipLaunchKernelGGL(ac_critic_kernel, grid_chunks, blk, e->lds_bytes, st, e->d, a, ns);
        prof_end(e);
        ad.which = 0; ad.lr = a.critic_lr; ad.wd = a.critic_wd;
        ad.soft = (h.algo == ALGO_DQN) ? 1 : ((!maddpg && a.do_actor) ? 1 : 0);
        prof_begin(e, PK_ADAM_CRITIC);
        launch_adam(e, st, ad, units, grid_adam);      // (a NoisyLinear head's sigma gradients are derived in its slab sums)
        prof_end(e);
    } else if (stage == 1) {
        if (v2 && h.wide) {                               // kernels_actorw.hip
            prof_begin(e, PK_GRAD_ACTOR);
            const bool x = h.wide == 2, a2 = h.net[0].L[2].n_pad > 16;
            auto k = x ? (a2 ? ac_actor_x_a2_kernel : ac_actor_x_a1_kernel) : (a2 ? ac_actor_wide_a2_kernel : ac_actor_wide_a1_kernel);
            hipLaunchKernelGGL(k, dim3(units), blk, (size_t)(x ? wide16_lds_floats_host() : wide_lds_floats()) * sizeof(float), st, e->d, a);
            prof_end(e);
            return;
        }
        if (v2 && h.solow) {
            prof_begin(e, PK_GRAD_ACTOR);
            SoloArgs sa{e->d_solo_slab, e->d_solo_part, e->d_solo_bar, e->d_solo_err, e->solo_bar_base, e->solo_stride, nullptr, nullptr, 0ull, h.solow, e->d_solow_bar2, e->solow_row_wgs, e->solow_wgs};
            e->solo_bar_base += kSoloWG;
            const bool a2 = h.net[0].L[2].n_pad > 16;
            hipLaunchKernelGGL(h.n_agents > 1 ? (a2 ? solow_actor_ma_a2_kernel : solow_actor_ma_a1_kernel) : (a2 ? solow_actor_a2_kernel : solow_actor_a1_kernel), dim3(units * e->solow_wgs), blk,
                               (size_t)solow_lds_floats() * sizeof(float), st, e->d, a, sa);
            prof_end(e);
            return;
        }
        if (v2 && h.solo) {
            prof_begin(e, PK_GRAD_ACTOR);
            SoloArgs sa{e->d_solo_slab, e->d_solo_part, e->d_solo_bar, e->d_solo_err, e->solo_bar_base, e->solo_stride, nullptr, nullptr, 0ull};
            e->solo_bar_base += kSoloWG;
            SoloStepArgs ss;
            memset(&ss, 0, sizeof ss);
            if (sstep) ss = *sstep;
            const int W = h.solo;
            hipLaunchKernelGGL(W == 16 ? solo_actor_kernel : solo_actor_w8_kernel, dim3(pc * W), blk,
                               (size_t)std::max(solo_lds_floats(), critic2_lds_floats()) * sizeof(float), st, e->d, a, sa, ss);
            prof_end(e);
            return;
        }
        if (v2) {        // kernels_actor2.hip: the whole actor stage of DDPG / TD3 / SAC in one launch
            prof_begin(e, PK_GRAD_ACTOR);
            const bool w8 = e->chain_waves == 8;
            hipLaunchKernelGGL(w8 ? ac_actor_v2_kernel : ac_actor_v2w4_kernel, dim3(pc), dim3(w8 ? 512 : 256),
                               (size_t)(w8 ? critic8_lds_floats() : critic2_lds_floats()) * sizeof(float), st, e->d, a);
            prof_end(e);
            return;
        }
        prof_begin(e, PK_GRAD_ACTOR);
        hipLaunchKernelGGL(ac_actor_kernel, grid_chunks, blk, e->lds_bytes, st, e->d, a, ns);
        prof_end(e);
        ad.which = 1; ad.lr = a.actor_lr; ad.wd = 0.f; ad.soft = maddpg ? 0 : 1; ad.sac_alpha = sac ? 1 : 0;
        prof_begin(e, PK_ADAM_ACTOR);
        launch_adam(e, st, ad, units, grid_adam);
        prof_end(e);
    } else {                                          // MATD3_simple.py:245-246: targets move with the delayed policy step
        prof_begin(e, PK_SOFT);
        int biggest = 0;
        for (int i = 0; i < h.n_nets; ++i) biggest = std::max(biggest, h.net[i].size);
        const int per = std::max(1, std::min((biggest + 4095) / 4096, 4 * e->n_cus / std::max(1, pc * h.n_nets)));
        hipLaunchKernelGGL(soft_update_kernel, dim3(pc * h.n_nets, per), blk, 0, st, e->d, a.tau, p0);
        prof_end(e);
    }
}

// `step` (frl_rollout only, DQN engines on the fused path): the vector step's add() and the next select_action in the same launch
// size_override >= 0: the rings' common size WHEN THE LAUNCH RUNS (a pre-armed launch of frl_rollout is enqueued before the step's rows
// are counted in e->size)
static int learn_impl(frl_engine* e, const frl_learn_args* args, const DqnStepArgs* step, const SoloStepArgs* sstep = nullptr, int size_override = -1,
                      hipStream_t stream_override = nullptr) {
    ENG(e);
    if (!args) return fail(FRL_ERR_INVALID, "args is NULL");
    const EngineDesc& h = e->h;
    if (!(h.algo == ALGO_DQN || h.algo == ALGO_DDPG || h.algo == ALGO_TD3 || h.algo == ALGO_SAC || h.algo == ALGO_MADDPG))
        return fail(FRL_ERR_STATE, "frl_learn: engine algo %d has no off-policy learn (PPO: frl_ppo_learn)", h.algo);
    if (args->batch < 1 || args->batch > h.batch_max) return fail(FRL_ERR_INVALID, "batch %d outside [1,%d]", args->batch, h.batch_max);
    int min_size = h.capacity;
    for (int p = 0; p < h.P; ++p) min_size = std::min(min_size, e->size[p]);
    if (size_override >= 0) min_size = size_override;
    if (min_size < args->batch) return fail(FRL_ERR_STATE, "a ring holds %d rows < batch %d", min_size, args->batch);
    if (args->per && (h.algo != ALGO_DQN || !e->per_on)) return fail(FRL_ERR_STATE, "per = 1 needs a DQN engine with frl_per_enable");
    if (args->per && args->idx) return fail(FRL_ERR_INVALID, "per = 1 uses the rows of the last frl_per_sample; idx must be NULL");
    const bool dev_rng = (args->idx == nullptr) && !args->per;
    if (dev_rng && min_size < 2 * args->batch)
        return fail(FRL_ERR_STATE, "device index draw needs len(buffer) >= 2*batch (have %d); pass idx", min_size);
    const bool td3_like = (h.algo == ALGO_TD3 || h.algo == ALGO_MADDPG);       // MADDPG + noise/delay = MATD3_simple.py
    const bool needs_noise = (h.algo == ALGO_SAC) || (td3_like && args->use_policy_noise);
    if (!dev_rng && needs_noise && !args->noise) return fail(FRL_ERR_INVALID, "idx given without noise: both or neither");
    int rc = flush_stage(e);
    if (rc) return rc;
    rc = upload_idx_noise(e, args->idx, needs_noise ? args->noise : nullptr, args->batch, h.n_agents);
    if (rc) return rc;
    LearnArgs a;
    memset(&a, 0, sizeof a);
    a.batch = args->batch;
    a.size = min_size;
    a.device_rng = dev_rng ? 1 : 0;
    a.do_actor = td3_like ? (args->do_actor ? 1 : 0) : 1;
    a.gamma = args->gamma; a.tau = args->tau;
    a.actor_lr = args->actor_lr; a.critic_lr = args->critic_lr; a.alpha_lr = args->alpha_lr;
    a.adam_eps = args->adam_eps > 0 ? args->adam_eps : 1e-8f;
    a.beta1 = 0.9f; a.beta2 = 0.999f;
    a.critic_wd = args->critic_weight_decay;
    a.clip_norm = args->clip_norm;
    a.policy_noise = args->policy_noise; a.noise_clip = args->noise_clip;
    a.max_action = args->max_action != 0.f ? args->max_action : 1.f;
    a.policy_noise_scale = args->policy_noise_scale;
    a.use_policy_noise = (td3_like && args->use_policy_noise) ? 1 : 0;
    a.target_entropy = args->target_entropy;
    a.double_dqn = (h.algo == ALGO_DQN && args->double_dqn) ? 1 : 0;
    a.use_isw = (h.algo == ALGO_DQN && args->per) ? (args->per == 2 ? 2 : 1) : 0;
    if (args->loss_kind != FRL_LOSS_MSE && args->loss_kind != FRL_LOSS_HUBER) return fail(FRL_ERR_INVALID, "unknown loss_kind %d", args->loss_kind);
    if (args->loss_kind == FRL_LOSS_HUBER) {
        if (!(args->huber_delta > 0.f)) return fail(FRL_ERR_INVALID, "Huber loss needs huber_delta > 0");
        if (h.c51_atoms) return fail(FRL_ERR_STATE, "the Categorical head's loss is a cross-entropy: no Huber variant");
        a.huber = 1; a.huber_delta = args->huber_delta;
    }
    a.rng_counter = e->rng_counter++;
    ++e->param_version;                       // (select_action's re-laid-out copies of the nets are stale from here on)
    // One chain for the whole population.  Measured and rejected (profiles/README.md): two halves of the population on two
    // streams so that one half's HBM-bound reduce/Adam runs under the other half's MFMA-bound gradient kernel — unchained
    // +2.7 %, with the gradient kernels chained across the streams -8 %: the Adam workgroups do not get co-resident with
    // the gradient kernel's (2 x 80 KB of LDS and 448 of 512 VGPRs per SIMD are taken).
    // (kernels_solow.hip moves MADDPG's targets at the end of its actor launch)
    const bool actor_stage = (h.algo != ALGO_DQN && a.do_actor), soft_stage = (h.algo == ALGO_MADDPG && a.do_actor && !(h.solow && chained_path(h, a.batch, h.P)));
    if (h.noisy) {
        // the reference draws noise per forward in program order: [online(s') if Double,] target(s'), online(s)
        const int first = a.double_dqn ? 0 : 1;
        if (args->noisy_eps) { rc = noisy_upload(e, args->noisy_eps, first, 3 - first); if (rc) return rc; }
        else hipLaunchKernelGGL(noisy_draw_kernel, dim3(h.P, 3), dim3(256), 0, e->stream, e->d, 0, 3, e->rng_counter++);
    }
    // kernels_solow.hip, single agent, helper workgroups present: a policy step is ONE launch — the critic's update runs on the helpers
    // under the policy's forward (FRL_SOLOW_FUSE=0: two launches)
    if (actor_stage && h.solow && h.n_agents == 1 && e->solow_wgs > e->solow_row_wgs && chained_path(h, a.batch, h.P)) {
        const char* fz = getenv("FRL_SOLOW_FUSE");
        a.fuse_actor = (fz && atoi(fz) == 0) ? 0 : 1;
    }
    hipStream_t lst = stream_override ? stream_override : e->stream;      // (frl_rollout's pre-armed launches: the pool's second stream)
    if (sstep) {
        // frl_rollout on a solo engine: the step's add() rides at the head of the critic launch, its tail (obs advance + the next
        // select_action + hand-over) at the end of the step's LAST launch
        if (!(h.solo && chained_path(h, a.batch, h.P)) || !dev_rng) return fail(FRL_ERR_STATE, "step fusion needs a solo engine with device draws");
        SoloStepArgs s0 = *sstep, s1 = *sstep;
        s0.head = 1; s0.tail = actor_stage ? 0 : 1;
        s1.head = 0; s1.tail = 1;
        launch_learn_stage(e, lst, a, 0, 0, h.P, dev_rng, needs_noise, nullptr, &s0);
        if (actor_stage) launch_learn_stage(e, lst, a, 1, 0, h.P, dev_rng, needs_noise, nullptr, &s1);
        HIP_TRY(hipGetLastError());
        return FRL_OK;
    }
    launch_learn_stage(e, lst, a, 0, 0, h.P, dev_rng, needs_noise, step);
    if (actor_stage && !a.fuse_actor) launch_learn_stage(e, lst, a, 1, 0, h.P, dev_rng, needs_noise);
    if (soft_stage) launch_learn_stage(e, lst, a, 2, 0, h.P, dev_rng, needs_noise);
    HIP_TRY(hipGetLastError());
    if (args->stats_out) return frl_stats_get(e, args->stats_out);
    return FRL_OK;
}

extern "C" int frl_learn(frl_engine* e, const frl_learn_args* args) { return learn_impl(e, args, nullptr); }

// Algorithmic work of one launch (DESIGN.md "Roofline"): flops = 2*B*sum(in*out) per forward
// pass, x2 more per backward pass that needs both dX and dW, x1 for dX-only passes; bytes =
// gathered records + 24 B per trained parameter (theta, m, v read+write) + 8 B per
// soft-updated target parameter.
extern "C" int frl_learn_work(const frl_engine* e, int batch, int do_actor, double* flops_out, double* bytes_out) {
    if (!e) return fail(FRL_ERR_INVALID, "engine is NULL");
    const EngineDesc& h = e->h;
    auto macs = [](const NetDesc& N, int l0, int nl) { double s = 0; for (int i = l0; i < l0 + nl; ++i) s += (double)N.L[i].n * N.L[i].k; return s; };
    double fl = 0, by = 0;
    const double B = batch;
    const RecordDesc& R = h.rec;
    if (h.algo == ALGO_DQN) {
        const NetDesc& N = h.net[0];
        const double m = macs(N, 0, N.n_layers);
        fl = 2 * B * m * (1 + 1 + 2);                  // target fwd, online fwd, bwd (dX+dW)
        by = 4 * B * (2 * R.obs_total + R.act_total + 2) + 24.0 * N.n_params + 8.0 * N.n_params;
    } else if (h.algo == ALGO_PPO) {
        fl = 0; by = 0;
    } else {
        const int n = h.n_agents;
        for (int ag = 0; ag < n; ++ag) {
            const NetDesc& NC = h.net[2 * ag + 1];
            const NetDesc& NA = h.net[2 * ag];
            const int ql = NC.n_layers / NC.heads;
            double f = 0;
            for (int j = 0; j < n; ++j) f += macs(h.net[2 * j], 0, h.net[2 * j].n_layers);   // target actors fwd
            f += macs(NC, 0, NC.n_layers);                      // target critic heads fwd
            f += 3 * macs(NC, 0, NC.n_layers);                  // critic fwd + bwd
            double bytes = 4 * B * n * (2.0 * R.obs_total / n + R.act_total / (double)n + 2) + 24.0 * NC.n_params;
            if (do_actor) {
                const int nq = (h.algo == ALGO_SAC) ? NC.heads : 1;
                f += macs(NA, 0, NA.n_layers) * 3;              // actor fwd + bwd
                f += nq * 2 * macs(NC, 0, ql);                  // Q(s, pi(s)) fwd + dX-only bwd
                bytes += 24.0 * NA.n_params + 8.0 * (NA.n_params + NC.n_params);
            }
            fl += 2 * B * f;
            by += bytes;
        }
    }
    if (flops_out) *flops_out = fl * h.P;
    if (bytes_out) *bytes_out = by * h.P;
    return FRL_OK;
}

// Executed flops of the same launch (include/freerl_hip.h): no first-layer dX for trained nets, the agent's action columns only
// for dQ/da.  Per (learner, agent): target actors fwd + target critic fwd + critic fwd + dW (all layers) + dX (layers 2..);
// actor stage: actor fwd + dW + dX (layers 2..) + per Q head used by the policy loss fwd + dX (layers 2.. whole, layer 1 x act_dim).
extern "C" int frl_learn_work_executed(const frl_engine* e, int batch, int do_actor, double* flops_out) {
    if (!e) return fail(FRL_ERR_INVALID, "engine is NULL");
    const EngineDesc& h = e->h;
    auto macs = [](const NetDesc& N, int l0, int nl) { double s = 0; for (int i = l0; i < l0 + nl; ++i) s += (double)N.L[i].n * N.L[i].k; return s; };
    auto first = [](const NetDesc& N) {            // the first layers of all heads
        const int nl = N.n_layers / std::max(1, N.heads);
        double s = 0;
        for (int hd = 0; hd < N.heads; ++hd) s += (double)N.L[hd * nl].n * N.L[hd * nl].k;
        return s;
    };
    double fl = 0;
    const double B = batch;
    if (h.algo == ALGO_DQN) {
        const NetDesc& N = h.net[0];
        const double m = macs(N, 0, N.n_layers);
        fl = 2 * B * (m + m + m + (m - first(N)));     // target fwd, online fwd, dW, dX from the second layer up
    } else if (h.algo != ALGO_PPO) {
        const int n = h.n_agents;
        for (int ag = 0; ag < n; ++ag) {
            const NetDesc &NC = h.net[2 * ag + 1], &NA = h.net[2 * ag];
            const int ql = NC.n_layers / NC.heads;
            double f = 0;
            for (int j = 0; j < n; ++j) f += macs(h.net[2 * j], 0, h.net[2 * j].n_layers);
            const double mc = macs(NC, 0, NC.n_layers);
            f += mc + mc + mc + (mc - first(NC));
            if (do_actor) {
                const int nq = (h.algo == ALGO_SAC) ? NC.heads : 1;
                const double ma = macs(NA, 0, NA.n_layers);
                f += ma + ma + (ma - first(NA));
                f += nq * (macs(NC, 0, ql) + macs(NC, 1, ql - 1) + (double)NC.L[0].n * h.rec.act_dim[ag]);
            }
            fl += 2 * B * f;
        }
    }
    if (flops_out) *flops_out = fl * h.P;
    return FRL_OK;
}

// developer read-back (tools/solo_timing.py): the single-learner kernels' per-workgroup partial sums and, in a -DFRL_SOLO_TIMING build of
// kernels_solo.hip, their section stamps — [kSoloWG][32] floats of learner 0
extern "C" int frl_solo_debug_read(frl_engine* e, float* out_host, int n_floats) {
    ENG(e);
    if (!(e->h.solo || e->h.solow) || !e->d_solo_part || !out_host || n_floats < 0 || n_floats > kSoloWG * kSoloPartHost) return fail(FRL_ERR_STATE, "not a solo engine / bad size");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(out_host, e->d_solo_part, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return FRL_OK;
}

// ----------------------------------------------------------------------------------- timing
extern "C" int frl_timer_start(frl_engine* e) {
    ENG(e);
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    return FRL_OK;
}
extern "C" int frl_timer_stop(frl_engine* e, float* ms_out) {
    ENG(e);
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (ms_out) *ms_out = ms;
    return FRL_OK;
}

// Batch_ObsNorm (Normalization_batch_size): switch + statistics {n, mean[O], S[O], std[O]} per learner
extern "C" int frl_obsnorm_enable(frl_engine* e, int on) {
    ENG(e);
    if (!e->has_nets) return fail(FRL_ERR_STATE, "replay-only engine has no networks");
    if (on && e->h.net[0].frag) {
        // Batch_ObsNorm belongs to the row-chunk family: an engine created for the register-chained kernels (parameters in
        // fragment-image order) moves over for good — every parameter array back to Wk, through a scratch copy
        float* scratch = nullptr;
        HIP_TRY(hipMalloc((void**)&scratch, (size_t)4 * e->h.P * e->h.learner_stride * sizeof(float)));
        hipLaunchKernelGGL(relayout_to_wk_kernel, dim3(e->h.P, 4), dim3(256), 0, e->stream, e->d, scratch);
        hipError_t he = hipStreamSynchronize(e->stream);
        hipFree(scratch);
        if (he != hipSuccess) return fail(FRL_ERR_HIP, "relayout: %s", hipGetErrorString(he));
        for (int i = 0; i < e->h.n_nets; ++i) e->h.net[i].frag = 0;
        ++e->param_version;
        e->h.wide = 0;
        e->h.solo = 0;
        e->h.solow = 0;
    }
    e->h.obs_norm_on = on ? 1 : 0;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->d, &e->h, sizeof e->h, hipMemcpyHostToDevice));
    return FRL_OK;
}
// stats of one learner: n_agents blocks of (1 + 3*max obs_dim) floats, block j = {n, mean[O_j], S[O_j], std[O_j]} of agent j
// (the version the next select_action / learn starts from); single agent: one block of 1 + 3*O floats
static float* obsnorm_final(frl_engine* e, int learner) {
    const size_t n = e->h.n_agents, w = e->h.obsnorm_w;
    return e->h.obsnorm + (((size_t)learner * n + (n - 1)) * n) * w;
}
extern "C" int frl_obsnorm_get(frl_engine* e, int learner, float* stats_out) {
    ENG(e);
    if (!e->has_nets || learner < 0 || learner >= e->h.P || !stats_out) return fail(FRL_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(stats_out, obsnorm_final(e, learner), (size_t)e->h.n_agents * e->h.obsnorm_w * sizeof(float), hipMemcpyDeviceToHost));
    return FRL_OK;
}
extern "C" int frl_obsnorm_set(frl_engine* e, int learner, const float* stats) {
    ENG(e);
    if (!e->has_nets || learner < 0 || learner >= e->h.P || !stats) return fail(FRL_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(obsnorm_final(e, learner), stats, (size_t)e->h.n_agents * e->h.obsnorm_w * sizeof(float), hipMemcpyHostToDevice));
    return FRL_OK;
}

extern "C" int frl_profile_enable(frl_engine* e, int on) {
    ENG(e);
    prof_collect(e);
    e->profile = on != 0;
    if (on) { for (int k = 0; k < 8; ++k) { e->prof_ms[k] = 0; e->prof_n[k] = 0; } }
    return FRL_OK;
}
extern "C" int frl_profile_read(frl_engine* e, double* ms_sum8, long long* count8) {
    ENG(e);
    prof_collect(e);
    for (int k = 0; k < 8; ++k) { if (ms_sum8) ms_sum8[k] = e->prof_ms[k]; if (count8) count8[k] = e->prof_n[k]; }
    return FRL_OK;
}

// ------------------------------------------------------------------------- prioritised replay
extern "C" int frl_per_enable(frl_engine* e, double alpha, double beta, double beta_increment, double epsilon) {
    ENG(e);
    if (e->h.n_agents != 1) return fail(FRL_ERR_INVALID, "PER is a single-agent buffer (DQN_file/Buffer.py:66)");
    if (e->per_on) return fail(FRL_ERR_STATE, "PER already enabled");
    int rc = flush_stage(e);
    if (rc) return rc;
    for (int p = 0; p < e->h.P; ++p)
        if (e->size[p] != 0) return fail(FRL_ERR_STATE, "enable PER on an empty buffer (priorities are assigned by add)");
    const size_t nn = 2 * (size_t)e->h.capacity - 1, P = e->h.P;
    HIP_TRY(hipMalloc((void**)&e->d_per_sum, P * nn * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&e->d_per_max, P * nn * sizeof(double)));
    HIP_TRY(hipMemsetAsync(e->d_per_sum, 0, P * nn * sizeof(double), e->stream));
    HIP_TRY(hipMemsetAsync(e->d_per_max, 0, P * nn * sizeof(double), e->stream));
    HIP_TRY(hipMalloc((void**)&e->d_size, 2 * P * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&e->d_stage_bucket, (2 * P + 1 + (size_t)e->stage_cap) * sizeof(int)));
    HIP_TRY(hipHostMalloc((void**)&e->stage_bucket, (2 * P + 1 + (size_t)e->stage_cap) * sizeof(int)));
    const size_t bm = (size_t)std::max(e->h.batch_max, 1);
    HIP_TRY(hipMalloc((void**)&e->d_per_prio, P * bm * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&e->d_uniforms, P * bm * sizeof(double)));
    if (!e->h.isw) {          // replay-only engine (Buffer.py's PER_Buffer on its own): sample rows, weights and TD errors still need a home
        HIP_TRY(hipMalloc((void**)&e->h.isw, P * bm * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&e->h.td_err, P * bm * sizeof(float)));
        e->idx_count = P * e->h.n_agents * bm;
        HIP_TRY(hipMalloc((void**)&e->h.idx, e->idx_count * sizeof(int)));
        HIP_TRY(hipHostMalloc((void**)&e->h_idx, e->idx_count * sizeof(int)));
        HIP_TRY(hipMemcpy(e->d, &e->h, sizeof(EngineDesc), hipMemcpyHostToDevice));
    }
    e->per_alpha = (float)alpha; e->per_beta = beta; e->per_beta_inc = beta_increment; e->per_eps = (float)epsilon;
    e->size_flushed.assign(P, 0);
    e->per_on = true;
    return FRL_OK;
}

extern "C" int frl_per_sample(frl_engine* e, int batch, const double* uniforms, int64_t* idx_out, float* is_weight_out) {
    ENG(e);
    if (!e->per_on) return fail(FRL_ERR_STATE, "frl_per_enable first");
    if (batch < 1 || batch > e->h.batch_max) return fail(FRL_ERR_INVALID, "batch %d outside [1,%d]", batch, e->h.batch_max);
    int rc = flush_stage(e);
    if (rc) return rc;
    const size_t P = e->h.P;
    for (size_t p = 0; p < P; ++p)
        if (e->size[p] < 1) return fail(FRL_ERR_STATE, "learner %zu's buffer is empty", p);
    e->per_beta = std::min(1.0, e->per_beta + e->per_beta_inc);             // Buffer.py:105 (before the weights are computed)
    HIP_TRY(hipMemcpyAsync(e->d_size + P, e->size.data(), P * sizeof(int), hipMemcpyHostToDevice, e->stream));
    if (uniforms) HIP_TRY(hipMemcpyAsync(e->d_uniforms, uniforms, P * batch * sizeof(double), hipMemcpyHostToDevice, e->stream));
    PerArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.sum_tree = e->d_per_sum; pa.max_tree = e->d_per_max; pa.cap = e->h.capacity; pa.n = batch;
    pa.size = e->d_size + P; pa.uniforms = uniforms ? e->d_uniforms : nullptr; pa.isw = e->h.isw; pa.prio_out = e->d_per_prio;
    pa.beta = e->per_beta; pa.rng_counter = e->rng_counter++;
    hipLaunchKernelGGL(per_sample_kernel, dim3((unsigned)P), dim3(256), 0, e->stream, e->d, pa);
    HIP_TRY(hipGetLastError());
    if (!idx_out && !is_weight_out) return FRL_OK;            // device-resident use (frl_learn with per = 1): asynchronous
    HIP_TRY(hipStreamSynchronize(e->stream));
    const size_t bm = e->h.batch_max;
    if (idx_out) {
        std::vector<int> tmp(bm * P);
        HIP_TRY(hipMemcpy(tmp.data(), e->h.idx, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t p = 0; p < P; ++p)
            for (int i = 0; i < batch; ++i) idx_out[p * batch + i] = tmp[p * bm + i];
    }
    if (is_weight_out) {
        std::vector<float> tmp(bm * P);
        HIP_TRY(hipMemcpy(tmp.data(), e->h.isw, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t p = 0; p < P; ++p) memcpy(is_weight_out + p * batch, tmp.data() + p * bm, (size_t)batch * sizeof(float));
    }
    return FRL_OK;
}

extern "C" int frl_per_update(frl_engine* e, int batch, const int64_t* idx, const float* td_error) {
    ENG(e);
    if (!e->per_on) return fail(FRL_ERR_STATE, "frl_per_enable first");
    if (batch < 1 || batch > e->h.batch_max) return fail(FRL_ERR_INVALID, "batch %d outside [1,%d]", batch, e->h.batch_max);
    if (batch > kPerSetMax) return fail(FRL_ERR_INVALID, "priority update of %d rows: at most %d per call (split the batch)", batch, kPerSetMax);
    int rc = flush_stage(e);
    if (rc) return rc;
    const size_t P = e->h.P, bm = e->h.batch_max;
    if (idx) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (size_t p = 0; p < P; ++p)
            for (int i = 0; i < batch; ++i) {
                const int64_t v = idx[p * batch + i];
                if (v < 0 || v >= e->h.capacity) return fail(FRL_ERR_INVALID, "priority index %lld out of range", (long long)v);
                e->h_idx[p * bm + i] = (int)v;
            }
        HIP_TRY(hipMemcpyAsync(e->h.idx, e->h_idx, P * bm * sizeof(int), hipMemcpyHostToDevice, e->stream));
    }
    if (td_error)
        for (size_t p = 0; p < P; ++p)
            HIP_TRY(hipMemcpyAsync(e->h.td_err + p * bm, td_error + p * batch, (size_t)batch * sizeof(float), hipMemcpyHostToDevice, e->stream));
    PerArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.sum_tree = e->d_per_sum; pa.max_tree = e->d_per_max; pa.cap = e->h.capacity; pa.n = batch;
    pa.leaf = e->h.idx; pa.n_pitch = (int)bm; pa.td = e->h.td_err; pa.alpha = e->per_alpha; pa.eps = e->per_eps;
    hipLaunchKernelGGL(per_set_kernel, dim3((unsigned)P), dim3(256), 0, e->stream, e->d, pa);
    HIP_TRY(hipGetLastError());
    return FRL_OK;
}

extern "C" int frl_per_state(frl_engine* e, int learner, double* sum_out, double* max_out, double* beta_out) {
    ENG(e);
    if (!e->per_on) return fail(FRL_ERR_STATE, "frl_per_enable first");
    if (learner < 0 || learner >= e->h.P) return fail(FRL_ERR_INVALID, "learner out of range");
    int rc = flush_stage(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    const size_t nn = 2 * (size_t)e->h.capacity - 1;
    if (sum_out) HIP_TRY(hipMemcpy(sum_out, e->d_per_sum + learner * nn, sizeof(double), hipMemcpyDeviceToHost));
    if (max_out) HIP_TRY(hipMemcpy(max_out, e->d_per_max + learner * nn, sizeof(double), hipMemcpyDeviceToHost));
    if (beta_out) *beta_out = e->per_beta;
    return FRL_OK;
}

#ifdef FRL_PHASE_TIMING
// developer build only (tools/phase_timing.py): not part of include/freerl_hip.h
extern "C" int frl_debug_phase_clocks(int* out, int stride) {
    if (stride < 0) { const int k = -stride - 1; return (int)hipMemcpyToSymbol(HIP_SYMBOL(frl::g_phase_kernel), &k, sizeof(int), 0, hipMemcpyHostToDevice); }
    if (stride > 0) return (int)hipMemcpyToSymbol(HIP_SYMBOL(frl::g_phase_stride), &stride, sizeof(int), 0, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(frl::g_phase_clock), sizeof(int) * frl::kPhaseBlocks * frl::kPhaseWords, 0,
                                    hipMemcpyDeviceToHost);
}
#endif
#include "frl_api_ppo.inc"
#include "frl_api_rollout.inc"
#include "frl_api_comm.inc"
