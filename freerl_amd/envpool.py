"""Python face of the native vectorised env pool and the rollout collector (include/freerl_hip.h,
`frl_envpool_*`, `frl_rollout`).  The pool itself is host C++ (worker threads + pinned staging) and
needs no GPU; `rollout()` drives a GPU engine with it."""
import ctypes as C

import numpy as np

from . import _native as N

KINDS = {"Pendulum-v1": N.ENV_PENDULUM, "PendulumShort-v1": N.ENV_PENDULUM_SHORT, "CartPole-v1": N.ENV_CARTPOLE,
         "SynLinear-v0": N.ENV_SYNLINEAR, "SynLinearDiscrete-v0": N.ENV_SYNLINEAR_DISCRETE, "SynBandWide-v0": N.ENV_SYNBAND_WIDE}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class EnvPool:
    def __init__(self, env_name, n_envs, n_threads=1, seed=0):
        self._L = N.lib()
        kind = KINDS[env_name]
        params, n_params = None, 0
        if env_name.startswith("SynLinear"):        # same A, B as freerl_amd.envs.LinearGaussianEnv
            from .envs import LinearGaussianEnv
            ref = LinearGaussianEnv(env_name.endswith("Discrete-v0"))
            flat = np.ascontiguousarray(np.concatenate([ref.A.reshape(-1), ref.B.reshape(-1)]), dtype=np.float64)
            self._params = flat
            params, n_params = flat.ctypes.data_as(C.POINTER(C.c_double)), flat.size
        h = C.c_void_p()
        N.check(self._L.frl_envpool_create(kind, int(n_envs), int(n_threads), int(seed), params, n_params, C.byref(h)))
        self._h = h
        n, o, a, na, ms = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        ma = C.c_float()
        N.check(self._L.frl_envpool_dims(h, C.byref(n), C.byref(o), C.byref(a), C.byref(na), C.byref(ma), C.byref(ms)))
        self.n, self.obs_dim, self.act_dim, self.n_actions = n.value, o.value, a.value, na.value
        self.max_action, self.max_steps = ma.value, ms.value
        self.env_name = env_name

    def close(self):
        if getattr(self, "_h", None):
            self._L.frl_envpool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        obs = np.empty((self.n, self.obs_dim), np.float32)
        N.check(self._L.frl_envpool_reset(self._h, _fp(obs)))
        return obs

    def set_state(self, env, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        N.check(self._L.frl_envpool_set_state(self._h, int(env), s.ctypes.data_as(C.POINTER(C.c_double))))

    def step(self, actions):
        """actions [n, act_dim] (env units; discrete: indices) -> (next_obs, reward, terminated,
        truncated, obs_next): obs_next is the reset observation where the episode ended."""
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, self.act_dim)
        nobs = np.empty((self.n, self.obs_dim), np.float32)
        onext = np.empty((self.n, self.obs_dim), np.float32)
        rew = np.empty(self.n, np.float32)
        term = np.empty(self.n, np.uint8)
        trunc = np.empty(self.n, np.uint8)
        u8 = C.POINTER(C.c_uint8)
        N.check(self._L.frl_envpool_step(self._h, _fp(a), _fp(nobs), _fp(rew), term.ctypes.data_as(u8),
                                         trunc.ctypes.data_as(u8), _fp(onext)))
        return nobs, rew, term.astype(bool), trunc.astype(bool), onext


class CallbackEnvPool(EnvPool):
    """A pool over caller-supplied environments speaking the gymnasium protocol the reference's loops drive
    (`reset(seed=...) -> (obs, info)`, `step(a) -> (obs, reward, terminated, truncated, info)`; DQN.py:292,316):
    `frl_envpool_create_callback` with one vectorised callback that steps every env into the pool's pinned staging and
    resets the finished ones.  `frl_rollout` / `frl_ppo_rollout` drive it exactly like the built-in kinds, so LunarLander /
    HalfCheetah / ... instances can sit behind the batched, device-side select_action wherever gymnasium is installed.
    Discrete action spaces (`.n`) pass action indices, Box spaces env-unit vectors."""

    def __init__(self, envs, seed=None):
        self._L = N.lib()
        self.envs = list(envs)
        e0 = self.envs[0]
        osp, asp = e0.observation_space, e0.action_space
        self.obs_dim = int(osp.shape[0]) if getattr(osp, "shape", None) else 1
        disc = hasattr(asp, "n")
        self.n_actions = int(asp.n) if disc else 0
        self.act_dim = 1 if disc else int(asp.shape[0])
        self.max_action = 0.0 if disc else float(np.max(asp.high))
        self.n, self.max_steps, self.env_name = len(self.envs), 0, type(e0).__name__
        self._seed = seed
        self.error = None
        O, A, n = self.obs_dim, self.act_dim, self.n

        def on_reset(_user, obs_out):
            try:
                out = np.ctypeslib.as_array(obs_out, shape=(n, O))
                for i, env in enumerate(self.envs):
                    o, _ = env.reset(seed=self._seed) if self._seed is not None else env.reset()
                    out[i] = np.asarray(o, np.float32).reshape(-1)
                return 0
            except Exception as ex:          # an exception cannot cross the C ABI: report through the status code
                self.error = ex
                return 1

        def on_step(_user, actions, next_obs, reward, term, trunc, obs_next):
            try:
                a = np.ctypeslib.as_array(actions, shape=(n, A))
                no = np.ctypeslib.as_array(next_obs, shape=(n, O))
                on = np.ctypeslib.as_array(obs_next, shape=(n, O))
                rw = np.ctypeslib.as_array(reward, shape=(n,))
                te = np.ctypeslib.as_array(term, shape=(n,))
                tr = np.ctypeslib.as_array(trunc, shape=(n,))
                for i, env in enumerate(self.envs):
                    act = int(a[i, 0]) if disc else a[i].copy()
                    o, r, t, u, _ = env.step(act)
                    no[i] = np.asarray(o, np.float32).reshape(-1)
                    rw[i], te[i], tr[i] = r, bool(t), bool(u)
                    if t or u:               # the reference resets inline at the episode's end (DQN.py:323-335)
                        o2, _ = env.reset(seed=self._seed) if self._seed is not None else env.reset()
                        on[i] = np.asarray(o2, np.float32).reshape(-1)
                    else:
                        on[i] = no[i]
                return 0
            except Exception as ex:
                self.error = ex
                return 1

        self._step_cb, self._reset_cb = N.ENV_STEP_FN(on_step), N.ENV_RESET_FN(on_reset)      # keep the thunks alive
        h = C.c_void_p()
        N.check(self._L.frl_envpool_create_callback(n, O, A, self.n_actions, self.max_action, self._step_cb, self._reset_cb, None,
                                                    C.byref(h)))
        self._h = h


def rollout(engine, pool, n_steps, *, envs_per_learner=1, start_steps=500, learn_every=1, policy_freq=2, epsilon=0.1,
            explore_sigma=0.1, batch=256, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, alpha_lr=1e-4,
            clip_norm=None, use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, target_entropy=None, host_explore=False,
            explore=None, gauss_init_scale=1.0, gauss_final_scale=0.0, max_episodes=0, ou_theta=0.15, ou_sigma=0.2, ou_dt=1e-2,
            double_dqn=False, per=0):
    """Run `n_steps` vector steps of the rollout-and-update loop; returns a dict of counters.
    explore: None = the algorithm's loop default (DQN epsilon-greedy, DDPG / TD3 Gaussian, SAC none) or "gauss" / "ou" /
    "none"; max_episodes > 0 decays a learner's noise multiplier with its finished episodes (TD3.py:425-427).
    clip_norm: None = the reference's rule per algorithm (0.5, none for DQN — DQN.py:56-59 has no clip_grad_norm_).
    per: 1 / 2 on a PER-enabled DQN engine (frl_per_sample -> learn -> frl_per_update per step, DQN_with_tricks.py:242-284)."""
    a = N.RolloutArgs()
    a.n_steps, a.envs_per_learner, a.start_steps, a.learn_every = int(n_steps), int(envs_per_learner), int(start_steps), int(learn_every)
    a.policy_freq, a.epsilon, a.explore_sigma = int(policy_freq), epsilon, explore_sigma
    a.host_explore = int(bool(host_explore))
    a.explore_kind = 0 if explore is None else {"none": N.EXPLORE_OFF, "eps": N.EXPLORE_EPS_GREEDY, "gauss": N.EXPLORE_GAUSS,
                                                 "ou": N.EXPLORE_OU}[explore]
    a.gauss_init_scale, a.gauss_final_scale, a.max_episodes = gauss_init_scale, gauss_final_scale, int(max_episodes)
    a.ou_theta, a.ou_sigma, a.ou_dt = ou_theta, ou_sigma, ou_dt
    la = a.learn
    la.batch, la.do_actor, la.use_policy_noise = int(batch), 1, int(bool(use_policy_noise))
    la.gamma, la.tau, la.actor_lr, la.critic_lr, la.alpha_lr = gamma, tau, actor_lr, critic_lr, alpha_lr
    if clip_norm is None:
        clip_norm = 0.0 if engine.algo == N.ALGO_DQN else 0.5
    la.adam_eps, la.clip_norm = 1e-8, clip_norm
    la.double_dqn, la.per = int(bool(double_dqn)), int(per)
    la.policy_noise, la.noise_clip, la.max_action, la.policy_noise_scale = policy_noise, noise_clip, pool.max_action or 1.0, 1.0
    la.target_entropy = float(-pool.act_dim if target_entropy is None else target_entropy)
    st = N.RolloutStats()
    if hasattr(pool, "error"):
        pool.error = None                      # a stale exception of an earlier call must not be re-raised
    rc = engine._L.frl_rollout(engine._h, pool._h, C.byref(a), C.byref(st))
    if rc and getattr(pool, "error", None) is not None:
        raise pool.error
    N.check(rc)
    return dict(env_steps=st.env_steps, updates=st.updates, episodes=st.episodes, return_sum=st.return_sum,
                seconds=st.seconds)


def ppo_rollout(engine, pool, n_iters, *, envs_per_learner, steps_per_env, minibatch=64, k_epochs=10, gamma=0.99, lmbda=0.95,
                clip=0.2, ent_coef=0.01, actor_lr=3e-4, critic_lr=3e-4, adam_eps=1e-8, clip_norm=0.5, adv_norm=False,
                optimizer=0):
    """`n_iters` cycles of (collect envs_per_learner x steps_per_env transitions per learner, then PPO.learn) on the device:
    the on-policy counterpart of `rollout` (frl_ppo_rollout)."""
    a = N.PpoRolloutArgs()
    a.n_iters, a.envs_per_learner, a.steps_per_env = int(n_iters), int(envs_per_learner), int(steps_per_env)
    la = a.learn
    la.horizon, la.minibatch, la.k_epochs = int(envs_per_learner) * int(steps_per_env), int(minibatch), int(k_epochs)
    la.adv_norm, la.optimizer = int(bool(adv_norm)), int(optimizer)
    la.gamma, la.lmbda, la.clip, la.ent_coef = gamma, lmbda, clip, ent_coef
    la.actor_lr, la.critic_lr, la.adam_eps, la.clip_norm = actor_lr, critic_lr, adam_eps, clip_norm
    st = N.RolloutStats()
    if hasattr(pool, "error"):
        pool.error = None
    rc = engine._L.frl_ppo_rollout(engine._h, pool._h, C.byref(a), C.byref(st))
    if rc and getattr(pool, "error", None) is not None:
        raise pool.error
    N.check(rc)
    return dict(env_steps=st.env_steps, updates=st.updates, episodes=st.episodes, return_sum=st.return_sum,
                seconds=st.seconds)
