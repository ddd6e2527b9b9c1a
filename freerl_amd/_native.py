"""ctypes binding of the C ABI declared in include/freerl_hip.h.

There is NO CPU fallback: if `libfreerl_hip.so` is missing or no HIP device is visible, the
product raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "_lib")
# FRL_HIP_VARIANT=<name> selects a developer build (tools/phase_timing.py): lib<...>_<name>.so compiled with
# the extra flags in FRL_HIPCC_FLAGS.  Unset = the product library.
_VARIANT = os.environ.get("FRL_HIP_VARIANT", "")
# (developer variants live under tools/_bin/, next to the other developer binaries: _lib/ holds the product library only)
LIB_PATH = os.path.join(ROOT, "tools", "_bin", "libfreerl_hip_%s.so" % _VARIANT) if _VARIANT else os.path.join(LIB_DIR, "libfreerl_hip.so")
HEADER = os.path.join(ROOT, "include", "freerl_hip.h")

FRL_MAX_AGENTS = 8
FRL_STAT_COUNT = 8
FRL_COMM_ID_BYTES = 128
FRL_COMM_MAX_VALUES = 64

# enum frl_algo
ALGO_REPLAY_ONLY, ALGO_DQN, ALGO_DDPG, ALGO_TD3, ALGO_SAC, ALGO_MADDPG, ALGO_PPO = -1, 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
PARAM_ONLINE, PARAM_TARGET, PARAM_ADAM_M, PARAM_ADAM_V, PARAM_GRAD = 0, 1, 2, 3, 4
ACT_RAW, ACT_ARGMAX, ACT_TANHHEAD, ACT_SAC_SAMPLE, ACT_PPO_SAMPLE, ACT_CAT_SAMPLE = 0, 1, 2, 3, 4, 5
ACT_NO_OBSNORM = 0x100
STAT_CRITIC_LOSS, STAT_ACTOR_LOSS, STAT_ALPHA_LOSS, STAT_ALPHA, STAT_CRITIC_GNORM, STAT_ACTOR_GNORM, STAT_ENTROPY = range(7)


class FrlError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("algo", C.c_int), ("n_learners", C.c_int), ("n_agents", C.c_int),
                ("obs_dim", C.c_int * FRL_MAX_AGENTS), ("act_dim", C.c_int * FRL_MAX_AGENTS),
                ("discrete", C.c_int), ("hidden", C.c_int), ("hidden_act", C.c_int), ("twin_critic", C.c_int),
                ("capacity", C.c_int), ("batch_max", C.c_int), ("extra_cols", C.c_int), ("actor_dist", C.c_int), ("dueling", C.c_int), ("noisy", C.c_int),
                ("c51_atoms", C.c_int), ("c51_vmin", C.c_float), ("c51_vmax", C.c_float),
                ("device_id", C.c_int),
                ("seed", C.c_uint64)]


class RecordLayout(C.Structure):
    _fields_ = [("n_agents", C.c_int), ("width", C.c_int), ("stride", C.c_int),
                ("obs_off", C.c_int * FRL_MAX_AGENTS), ("obs_dim", C.c_int * FRL_MAX_AGENTS),
                ("act_off", C.c_int * FRL_MAX_AGENTS), ("act_dim", C.c_int * FRL_MAX_AGENTS),
                ("rew_off", C.c_int), ("done_off", C.c_int), ("next_obs_off", C.c_int * FRL_MAX_AGENTS),
                ("extra_off", C.c_int), ("extra", C.c_int)]


class LearnArgs(C.Structure):
    _fields_ = [("batch", C.c_int), ("do_actor", C.c_int), ("use_policy_noise", C.c_int),
                ("gamma", C.c_float), ("tau", C.c_float), ("actor_lr", C.c_float), ("critic_lr", C.c_float),
                ("alpha_lr", C.c_float), ("adam_eps", C.c_float), ("critic_weight_decay", C.c_float),
                ("clip_norm", C.c_float), ("policy_noise", C.c_float), ("noise_clip", C.c_float),
                ("max_action", C.c_float), ("policy_noise_scale", C.c_float), ("target_entropy", C.c_float),
                ("double_dqn", C.c_int), ("per", C.c_int), ("noisy_eps", C.POINTER(C.c_float)),
                ("idx", C.POINTER(C.c_int64)), ("noise", C.POINTER(C.c_float)), ("stats_out", C.POINTER(C.c_float)),
                ("loss_kind", C.c_int), ("huber_delta", C.c_float)]


class PpoArgs(C.Structure):
    _fields_ = [("horizon", C.c_int), ("minibatch", C.c_int), ("k_epochs", C.c_int), ("adv_norm", C.c_int),
                ("gamma", C.c_float), ("lmbda", C.c_float), ("clip", C.c_float), ("ent_coef", C.c_float),
                ("actor_lr", C.c_float), ("critic_lr", C.c_float), ("adam_eps", C.c_float), ("clip_norm", C.c_float),
                ("optimizer", C.c_int), ("perms", C.POINTER(C.c_int64)), ("loss_trace_out", C.POINTER(C.c_float)),
                ("adv_out", C.POINTER(C.c_float)), ("vtarget_out", C.POINTER(C.c_float)),
                ("gae_mode", C.c_int), ("last_value", C.POINTER(C.c_float)), ("gae_gamma", C.c_double), ("gae_lmbda", C.c_double)]


class ExploreArgs(C.Structure):
    _fields_ = [("kind", C.c_int), ("epsilon", C.c_float), ("sigma", C.c_float), ("scale", C.c_float), ("max_action", C.c_float),
                ("ou_theta", C.c_float), ("ou_sigma", C.c_float), ("ou_dt", C.c_float)]


class RolloutArgs(C.Structure):
    _fields_ = [("n_steps", C.c_int), ("envs_per_learner", C.c_int), ("start_steps", C.c_int), ("learn_every", C.c_int),
                ("policy_freq", C.c_int), ("epsilon", C.c_float), ("explore_sigma", C.c_float), ("learn", LearnArgs),
                ("host_explore", C.c_int), ("explore_kind", C.c_int), ("gauss_init_scale", C.c_float),
                ("gauss_final_scale", C.c_float), ("max_episodes", C.c_int), ("ou_theta", C.c_float), ("ou_sigma", C.c_float),
                ("ou_dt", C.c_float)]


EXPLORE_NONE, EXPLORE_EPS_GREEDY, EXPLORE_GAUSS, EXPLORE_OU, EXPLORE_OFF = range(5)
# the vectorised callbacks of a pool over caller-supplied envs (frl_envpool_create_callback)
ENV_STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                          C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_float))
ENV_RESET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float))


class PpoRolloutArgs(C.Structure):
    _fields_ = [("n_iters", C.c_int), ("envs_per_learner", C.c_int), ("steps_per_env", C.c_int), ("learn", PpoArgs)]


class RolloutStats(C.Structure):
    _fields_ = [("env_steps", C.c_longlong), ("updates", C.c_longlong), ("episodes", C.c_longlong),
                ("return_sum", C.c_double), ("seconds", C.c_double)]


ENV_PENDULUM, ENV_CARTPOLE, ENV_SYNLINEAR, ENV_SYNLINEAR_DISCRETE, ENV_PENDULUM_SHORT, ENV_SYNBAND_WIDE = range(6)

_P = C.POINTER
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_fp, _ip, _i64p = _P(C.c_float), _P(C.c_int), _P(C.c_int64)

# name -> (restype, argtypes); every symbol include/freerl_hip.h declares
SIGNATURES = {
    "frl_last_error": (C.c_char_p, []),
    "frl_version": (_i, []),
    "frl_device_count": (_i, [_ip]),
    "frl_create": (_i, [_P(Config), _P(_vp)]),
    "frl_destroy": (_i, [_vp]),
    "frl_sync": (_i, [_vp]),
    "frl_lds_bytes": (_i, [_vp, _ip, _ip]),
    "frl_learn_path": (_i, [_vp, _i, _ip, _ip, _ip]),
    "frl_record_layout_get": (_i, [_vp, _P(RecordLayout)]),
    "frl_buffer_add": (_i, [_vp, _i, _fp]),
    "frl_buffer_add_batch": (_i, [_vp, _i, _ip, _fp]),
    "frl_buffer_flush": (_i, [_vp]),
    "frl_buffer_cursor_get": (_i, [_vp, _i, _ip, _ip]),
    "frl_buffer_cursor_set": (_i, [_vp, _i, _i, _i]),
    "frl_buffer_sample": (_i, [_vp, _i, _i64p, _i, _i, _ip, _ip, _P(_vp)]),
    "frl_buffer_read": (_i, [_vp, _i, _i, _i, _fp]),
    "frl_buffer_fill_synthetic": (_i, [_vp, _i, C.c_uint64]),
    "frl_net_count": (_i, [_vp, _ip]),
    "frl_net_num_params": (_i, [_vp, _i, _ip]),
    "frl_params_get": (_i, [_vp, _i, _i, _i, _fp]),
    "frl_params_set": (_i, [_vp, _i, _i, _i, _fp]),
    "frl_params_pad_max": (_i, [_vp, _i, _i, _i, _fp]),
    "frl_opt_step_get": (_i, [_vp, _i, _i, _ip]),
    "frl_opt_step_set": (_i, [_vp, _i, _i, _i]),
    "frl_alpha_get": (_i, [_vp, _i, _fp, _ip]),
    "frl_alpha_set": (_i, [_vp, _i, _fp, _i]),
    "frl_obsnorm_enable": (_i, [_vp, _i]),
    "frl_obsnorm_get": (_i, [_vp, _i, _fp]),
    "frl_obsnorm_set": (_i, [_vp, _i, _fp]),
    "frl_act": (_i, [_vp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp]),
    "frl_act_device": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "frl_act_explore": (_i, [_vp, _i, _i, _fp, _P(ExploreArgs), _P(C.c_uint8), _fp, _fp]),
    "frl_learn": (_i, [_vp, _P(LearnArgs)]),
    "frl_stats_get": (_i, [_vp, _fp]),
    "frl_last_indices": (_i, [_vp, _i, _i64p]),
    "frl_noisy_eps_size": (_i, [_vp, _ip]),
    "frl_noisy_resample": (_i, [_vp, _fp]),
    "frl_per_enable": (_i, [_vp, C.c_double, C.c_double, C.c_double, C.c_double]),
    "frl_per_sample": (_i, [_vp, _i, _P(C.c_double), _i64p, _fp]),
    "frl_per_update": (_i, [_vp, _i, _i64p, _fp]),
    "frl_per_state": (_i, [_vp, _i, _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "frl_learn_work": (_i, [_vp, _i, _i, _P(C.c_double), _P(C.c_double)]),
    "frl_solo_debug_read": (_i, [_vp, _fp, _i]),
    "frl_learn_work_executed": (_i, [_vp, _i, _i, _P(C.c_double)]),
    "frl_ppo_learn": (_i, [_vp, _P(PpoArgs)]),
    "frl_ppo_work": (_i, [_vp, _i, _i, _P(C.c_double), _P(C.c_double)]),
    "frl_gae": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp]),
    "frl_envpool_create": (_i, [_i, _i, _i, C.c_uint64, _P(C.c_double), _i, _P(_vp)]),
    "frl_envpool_create_callback": (_i, [_i, _i, _i, _i, _f, ENV_STEP_FN, ENV_RESET_FN, _vp, _P(_vp)]),
    "frl_envpool_destroy": (_i, [_vp]),
    "frl_envpool_dims": (_i, [_vp, _ip, _ip, _ip, _ip, _fp, _ip]),
    "frl_envpool_reset": (_i, [_vp, _fp]),
    "frl_envpool_set_state": (_i, [_vp, _i, _P(C.c_double)]),
    "frl_envpool_step": (_i, [_vp, _fp, _fp, _fp, _P(C.c_uint8), _P(C.c_uint8), _fp]),
    "frl_rollout": (_i, [_vp, _vp, _P(RolloutArgs), _P(RolloutStats)]),
    "frl_ppo_rollout": (_i, [_vp, _vp, _P(PpoRolloutArgs), _P(RolloutStats)]),
    "frl_comm_unique_id": (_i, [_P(C.c_uint8)]),
    "frl_comm_create": (_i, [_P(C.c_uint8), _i, _i, _i, _P(_vp)]),
    "frl_comm_destroy": (_i, [_vp]),
    "frl_comm_info": (_i, [_vp, _ip, _ip]),
    "frl_metrics_allreduce": (_i, [_vp, _P(C.c_double), _i, _P(C.c_double), _i]),
    "frl_timer_start": (_i, [_vp]),
    "frl_timer_stop": (_i, [_vp, _fp]),
    "frl_profile_enable": (_i, [_vp, _i]),
    "frl_profile_read": (_i, [_vp, _P(C.c_double), _P(C.c_longlong)]),
}

_lib = None


def _headers():
    dev = os.path.join(CSRC, "device")
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hpp", ".inc"))] + \
           [os.path.join(dev, f) for f in sorted(os.listdir(dev))] + [HEADER]


def units():
    """The translation units of the library: the host API and one unit per kernels_*.hip (a kernel's device code is
    complete in the unit that defines it, so no relocatable device code is needed)."""
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]


def sources():
    return units() + _headers()


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in sources())


_HIPCC = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]


def build(force=False, verbose=False, jobs=None):
    """Compile libfreerl_hip.so for gfx950 with hipcc (cross-compiles without a GPU): every unit to an object under
    _lib/obj/ in parallel (only the stale ones), then one link.  Developer variants (FRL_HIP_VARIANT) are one unity
    unit built with the extra flags in FRL_HIPCC_FLAGS."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    if _VARIANT:
        cmd = _HIPCC + ["-shared", "-DFRL_UNITY", "-o", LIB_PATH, os.path.join(CSRC, "frl_api.hip")] + \
            os.environ.get("FRL_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise FrlError("hipcc failed:\n" + r.stdout + r.stderr)
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    todo, objs = [], []
    for src in units():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(src)):
            todo.append((src, obj))

    def compile_one(so):
        cmd = _HIPCC + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        return so[0], subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=jobs or min(len(todo) or 1, os.cpu_count() or 4)) as ex:
        for src, r in ex.map(compile_one, todo):
            if r.returncode != 0:
                raise FrlError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise FrlError("link failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def lib():
    """The loaded library; raises FrlError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FrlError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the engine has no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise FrlError("freerl_hip error %d: %s" % (rc, lib().frl_last_error().decode("utf-8", "replace")))


def device_count():
    n = C.c_int(0)
    rc = lib().frl_device_count(C.byref(n))
    return n.value if rc == 0 else 0
