"""Training loops: the counterpart of the `__main__` blocks of the reference scripts
(`DQN_file/DQN.py:227-349`, `DDPG_file/DDPG_simple.py:238-362`, `TD3_file/TD3.py:315-456`,
`SAC_file/SAC.py:429-586`, `PPO_file/PPO_with_tricks.py:435-584`,
`MADDPG_file/MADDPG_simple.py:268-395`), driving freerl_amd's GPU-backed policies.

    python -m freerl_amd.train td3 --env_name Pendulum-v1 --seed 0 --max_episodes 500

Same flags and defaults as the reference script of each algorithm, same results layout
(`results/<env>/<policy>_<n>/{<ALGO>.pt, <policy>_seed_<s>.npy}`), and the same per-step order of
operations — action rule, legacy-NumPy / torch RNG draws, add before learn, learn trigger
`step > start_steps`, per-episode noise schedules, `env.reset(seed=args.seed)` at every episode —
so a seeded run follows the reference's trajectory (tests/test_gpu_loops.py replays golden runs of
the reference's own loops).  One generic loop with per-algorithm hooks replaces the six copies.
"""
import argparse
import os
import re
import sys
import time

import numpy as np
import torch

from . import envs as _envs
from . import normalization as _norm

# --------------------------------------------------------------------------------- flag tables
_COMMON = [("seed", int, 0), ("max_episodes", int, 500), ("save_freq", int, 500 // 4), ("start_steps", int, 500),
           ("random_steps", int, 0), ("learn_steps_interval", int, 1), ("gamma", float, 0.99), ("tau", float, 0.01)]
_AC = [("actor_lr", float, 1e-3), ("critic_lr", float, 1e-3)]
_REPLAY = [("buffer_size", int, int(1e6)), ("batch_size", int, 256)]
_GAUSS = [("gauss_sigma", float, 0.1), ("gauss_scale", float, 1), ("gauss_init_scale", float, None),
          ("gauss_final_scale", float, 0.0)]

FLAGS = {
    "dqn": dict(env_name="BipedalWalker-v3", policy_name="DQN", device="cuda", is_dis_to_con=True,
                flags=_COMMON + _REPLAY + [("Qnet_lr", float, 1e-3), ("epsilon", float, 0.1)],
                trick={"Double": False, "Dueling": False, "PER": False, "Noisy": False, "N_Step": False, "Categorical": False}),
    "ddpg": dict(env_name="Pendulum-v1", policy_name="DDPG_simple", device="cuda", is_dis_to_con=False,
                 flags=_COMMON + _AC + _REPLAY + _GAUSS, trick=None),
    "td3": dict(env_name="MountainCarContinuous-v0", policy_name="TD3", device="cpu", is_dis_to_con=False,
                flags=_COMMON + _AC + _REPLAY + _GAUSS + [("policy_noise", float, 0.1), ("noise_clip", float, 0.5),
                                                           ("policy_freq", int, 2), ("policy_noise_scale", float, 1),
                                                           ("policy_noise_init_scale", float, None)],
                overrides=dict(seed=100, batch_size=64, gauss_sigma=1, gauss_init_scale=1), trick=None),
    "sac": dict(env_name="MountainCarContinuous-v0", policy_name="SAC", device="cpu", is_dis_to_con=False,
                flags=_COMMON + _AC + _REPLAY + _GAUSS + [("ou_sigma", float, 1), ("ou_dt", float, 1),
                                                           ("init_scale", float, 1), ("final_scale", float, 0.0)],
                overrides=dict(random_steps=500, batch_size=64, gauss_sigma=1, gauss_init_scale=1),
                trick={"ObsNorm": False, "Batch_ObsNorm": False, "OUNoise": True, "GaussNoise": False}),
    "ppo": dict(env_name="CartPole-v1", policy_name="PPO", device="cpu", is_dis_to_con=False,
                flags=_COMMON + _AC + [("horizon", int, 2048), ("clip_param", float, 0.2), ("K_epochs", int, 10),
                                       ("entropy_coefficient", float, 0.01), ("minibatch_size", int, 64),
                                       ("lmbda", float, 0.95)],
                overrides=dict(start_steps=0, learn_steps_interval=0),
                trick={"adv_norm": False, "ObsNorm": False, "Batch_ObsNorm": False, "reward_norm": False,
                       "reward_scaling": False, "lr_decay": False, "orthogonal_init": False, "adam_eps": False,
                       "tanh": False}),
    "maddpg": dict(env_name="simple_spread_v3", policy_name="MADDPG_simple", device="cpu", is_dis_to_con=False,
                   flags=_COMMON + _AC + _REPLAY + _GAUSS + [("N", int, 5)],
                   overrides=dict(seed=100, max_episodes=600, save_freq=600 // 4, gamma=0.95, gauss_sigma=1,
                                  gauss_init_scale=1), trick=None),
}


def build_parser(algo):
    spec = FLAGS[algo]
    p = argparse.ArgumentParser(prog="freerl_amd.train " + algo)
    p.add_argument("--env_name", type=str, default=spec["env_name"])
    over = spec.get("overrides", {})
    for name, typ, default in spec["flags"]:
        p.add_argument("--" + name, type=typ, default=over.get(name, default))
    p.add_argument("--is_dis_to_con", type=bool, default=spec["is_dis_to_con"])
    p.add_argument("--policy_name", type=str, default=spec["policy_name"])
    p.add_argument("--trick", type=dict, default=spec["trick"])
    if algo == "td3":
        p.add_argument("--realize", type=dict, default={"clip_double": True, "policy_noise": True, "twin_delay": True})
    if algo == "ppo":
        p.add_argument("--beta", type=bool, default=False)
    p.add_argument("--device", type=str, default=spec["device"])
    # additions of this build (not in the reference)
    p.add_argument("--results_root", type=str, default=os.path.join(os.getcwd(), "results"))
    p.add_argument("--rng", type=str, default="auto", choices=["auto", "host", "device"])
    return p


# -------------------------------------------------------------------------------- bookkeeping
def make_dir(results_root, env_name, policy_name="DQN", trick=None):
    """results/<env>/<prefix><n+1>: prefix = policy name + the names of the enabled tricks
    (DQN.py:173-192)."""
    env_dir = os.path.join(results_root, env_name)
    os.makedirs(env_dir, exist_ok=True)
    prefix = policy_name + "_" + "".join(k + "_" for k, v in (trick or {}).items() if v)
    taken = [int(d.split("_")[-1]) for d in os.listdir(env_dir)
             if re.match("^" + re.escape(prefix) + r"\d+", d) and d.split("_")[-1].isdigit()]
    model_dir = os.path.join(env_dir, prefix + str(max(taken, default=0) + 1))
    os.makedirs(model_dir)
    return model_dir


class ScalarWriter:
    """`SummaryWriter(model_dir).add_scalar(tag, value, step)` (DQN.py:276,330): TensorBoard when the
    package exists, else a CSV with the same three columns."""

    def __init__(self, model_dir):
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(model_dir)
        except Exception:
            self._f = open(os.path.join(model_dir, "scalars.csv"), "w")
            self._f.write("tag,step,value\n")

    def add_scalar(self, tag, value, step):
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)
        else:
            self._f.write("%s,%d,%r\n" % (tag, step, float(value)))

    def close(self):
        if self._tb is not None:
            self._tb.close()
        else:
            self._f.close()


class ActionDiscretizer:
    """`dis_to_con` (DQN.py:195-217): 1-D Box -> `n` evenly spaced torques; A-D Box -> n = per^A grid
    points enumerated little-endian in base `per`."""

    def __init__(self, space, n):
        self.low, self.high, self.n = np.asarray(space.low, dtype=np.float64), np.asarray(space.high, dtype=np.float64), n
        self.dims = space.shape[0]
        self.per = n if self.dims == 1 else int(n ** (1 / self.dims))

    def __call__(self, k):
        if self.dims == 1:
            return np.array([self.low[0] + (k / (self.n - 1)) * (self.high[0] - self.low[0])])
        digits = [(k // self.per ** i) % self.per for i in range(self.dims)]
        return np.array([self.low[i] + digits[i] / (self.per - 1) * (self.high[i] - self.low[i]) for i in range(self.dims)])


class OUNoise:
    """Ornstein-Uhlenbeck exploration noise (SAC.py:334-356), NumPy legacy stream."""

    def __init__(self, action_dim, mu=0, theta=0.15, sigma=0.1, dt=1e-2, scale=None):
        self.action_dim, self.mu, self.theta, self.sigma, self.dt, self.scale = action_dim, mu, theta, sigma, dt, scale
        self.reset()

    def reset(self):
        self.state = np.ones(self.action_dim) * self.mu

    def noise(self):
        self.state = self.state + self.theta * (self.mu - self.state) + np.sqrt(self.dt) * self.sigma * np.random.randn(self.action_dim)
        return self.state if self.scale is None else self.state * self.scale


def get_env(env_name, is_dis_to_con=False):
    """(env, dim_info, max_action, is_continue) like the reference's get_env (DQN.py:142-170)."""
    env = _envs.make(env_name)
    obs_space, act_space = env.observation_space, env.action_space
    obs_dim = obs_space.shape[0] if hasattr(obs_space, "low") else 1
    if hasattr(act_space, "low"):
        action_dim = act_space.shape[0]
        dim_info, max_action, is_continue = [obs_dim, action_dim], act_space.high[0], True
        if is_dis_to_con:
            dim_info, is_continue = [obs_dim, 16 if action_dim == 1 else 2 ** action_dim], False
    else:
        dim_info, max_action, is_continue = [obs_dim, act_space.n], None, False
    return env, dim_info, max_action, is_continue


def _remaining(args, episode_num):
    return max(0, args.max_episodes - (episode_num + 1)) / args.max_episodes


# ------------------------------------------------------------------------- per-algorithm hooks
class _Hooks:
    """What differs between the reference loops; the loop itself is `_run_loop`."""

    def __init__(self, args, env, policy, dim_info, max_action):
        self.args, self.env, self.policy, self.dim_info, self.max_action = args, env, policy, dim_info, max_action
        self.action_dim = dim_info[1]

    def begin(self, obs):
        return obs

    def act(self, step, obs):           # -> (stored action, env action, extra stored fields)
        raise NotImplementedError

    def observe(self, next_obs, reward):  # -> (next_obs to store/use, reward to store)
        return next_obs, reward

    def store(self, obs, action, reward, next_obs, terminated, done, extra):
        self.policy.add(obs, action, reward, next_obs, terminated)

    def episode_end(self, episode_num):
        pass

    def after_reset(self, obs):
        return obs

    def learn_due(self, step):
        a = self.args
        return step > a.start_steps and step % a.learn_steps_interval == 0

    def learn(self, episode_num):
        raise NotImplementedError

    def finish(self, model_dir):
        pass


class _DQNHooks(_Hooks):
    def begin(self, obs):
        box = hasattr(self.env.action_space, "low")
        self.disc = ActionDiscretizer(self.env.action_space, self.action_dim) if (self.args.is_dis_to_con and box) else None
        return obs

    def act(self, step, obs):
        a = self.args
        if step < a.random_steps:                                   # DQN.py:297-306
            action = self.env.action_space.sample()
            if self.max_action is not None:
                action = action / self.max_action
                action = np.clip(np.digitize(action[0], np.linspace(-1, 1, self.action_dim + 1)) - 1, 0, self.action_dim - 1)
        elif np.random.rand() < a.epsilon:                          # DQN.py:307-310
            action = np.random.randint(self.action_dim)
        else:
            action = self.policy.select_action(obs)
        return action, (self.disc(action) if self.disc is not None else action), None

    def learn(self, episode_num):
        self.policy.learn(self.args.batch_size, self.args.gamma, self.args.tau)


class _GaussHooks(_Hooks):
    """DDPG_simple / TD3: Gaussian action noise with optional per-episode linear decay."""

    def begin(self, obs):
        a = self.args
        if a.gauss_init_scale is not None:
            a.gauss_scale = a.gauss_init_scale
        if getattr(a, "policy_noise_init_scale", None) is not None:
            a.policy_noise_scale = a.policy_noise_init_scale
        return obs

    def act(self, step, obs):
        a, m = self.args, self.max_action
        if step < a.random_steps:
            env_action = self.env.action_space.sample()
            return env_action / m, env_action, None
        action = self.policy.select_action(obs)
        noise = a.gauss_scale * np.random.normal(scale=a.gauss_sigma * m, size=self.action_dim)
        return action, np.clip(action * m + noise, -m, m), None

    def episode_end(self, episode_num):
        a = self.args
        if a.gauss_init_scale is not None:
            a.gauss_scale = a.gauss_final_scale + (a.gauss_init_scale - a.gauss_final_scale) * _remaining(a, episode_num)
        if getattr(a, "policy_noise_scale", None) is not None and hasattr(a, "policy_freq"):
            a.policy_noise_scale = a.policy_noise_scale * _remaining(a, episode_num)      # TD3.py:428-430

    def learn(self, episode_num):
        a = self.args
        if hasattr(a, "policy_freq"):
            self.policy.learn(a.batch_size, a.gamma, a.tau, a.policy_noise, a.noise_clip, self.max_action, a.policy_freq,
                              a.policy_noise_scale)
        else:
            self.policy.learn(a.batch_size, a.gamma, a.tau)


class _SACHooks(_Hooks):
    def begin(self, obs):
        a = self.args
        self.obs_norm = _norm.Normalization(shape=self.dim_info[0]) if a.trick["ObsNorm"] else None
        if self.obs_norm is not None:
            obs = self.obs_norm(obs)
        self.ou = OUNoise(self.action_dim, sigma=a.ou_sigma, dt=a.ou_dt, scale=a.init_scale) if a.trick["OUNoise"] else None
        if a.trick["GaussNoise"] and a.gauss_init_scale is not None:
            a.gauss_scale = a.gauss_init_scale
        return obs

    def act(self, step, obs):
        a, m = self.args, self.max_action
        if step < a.random_steps:                                    # SAC.py:522-525
            env_action = self.env.action_space.sample()
            return env_action / m, env_action, None
        action = self.policy.select_action(obs)
        if self.ou is not None:
            env_action = np.clip(action * m + self.ou.noise() * m, -m, m)
        elif a.trick["GaussNoise"]:
            env_action = np.clip(action * m + a.gauss_scale * np.random.normal(scale=a.gauss_sigma * m, size=self.action_dim), -m, m)
        else:
            env_action = np.clip(action * m, -m, m)
        return action, env_action, None

    def observe(self, next_obs, reward):
        return (self.obs_norm(next_obs) if self.obs_norm is not None else next_obs), reward

    def episode_end(self, episode_num):
        a = self.args
        if self.ou is not None:
            self.ou.reset()
            if a.init_scale is not None:
                self.ou.scale = a.final_scale + (a.init_scale - a.final_scale) * _remaining(a, episode_num)
        elif a.trick["GaussNoise"] and a.gauss_init_scale is not None:
            a.gauss_scale = a.gauss_final_scale + (a.gauss_init_scale - a.gauss_final_scale) * _remaining(a, episode_num)

    def after_reset(self, obs):
        return self.obs_norm(obs) if self.obs_norm is not None else obs

    def learn(self, episode_num):
        self.policy.learn(self.args.batch_size, self.args.gamma, self.args.tau)

    def finish(self, model_dir):
        if self.obs_norm is not None:                                # SAC.py:583-584
            np.save(os.path.join(model_dir, "%s_running_mean_std.npy" % self.args.policy_name),
                    np.array([self.obs_norm.running_ms.mean, self.obs_norm.running_ms.std]))


class _PPOHooks(_Hooks):
    def begin(self, obs):
        t = self.args.trick
        self.obs_norm = _norm.Normalization(shape=self.dim_info[0]) if t["ObsNorm"] else None
        self.reward_norm = _norm.Normalization(shape=1) if t["reward_norm"] else None
        self.reward_scaling = _norm.RewardScaling(shape=1, gamma=self.args.gamma) if (t["reward_scaling"] and not t["reward_norm"]) else None
        return self.obs_norm(obs) if self.obs_norm is not None else obs

    def act(self, step, obs):
        action, log_pi = self.policy.select_action(obs)
        m = self.max_action
        env_action = np.clip(action * m, -m, m) if m is not None else action
        return action, env_action, log_pi

    def observe(self, next_obs, reward):
        if self.obs_norm is not None:
            next_obs = self.obs_norm(next_obs)
        if self.reward_norm is not None:
            reward = self.reward_norm(reward)
        elif self.reward_scaling is not None:
            reward = self.reward_scaling(reward)
        return next_obs, reward

    def store(self, obs, action, reward, next_obs, terminated, done, log_pi):
        self.policy.add(obs, action, reward, next_obs, terminated, log_pi, done)       # :542-546

    def after_reset(self, obs):
        if self.reward_scaling is not None:
            self.reward_scaling.reset()
        return self.obs_norm(obs) if self.obs_norm is not None else obs

    def learn_due(self, step):
        return step % self.args.horizon == 0

    def learn(self, episode_num):
        a = self.args
        self.policy.learn(a.minibatch_size, a.gamma, a.lmbda, a.clip_param, a.K_epochs, a.entropy_coefficient)
        if a.trick["lr_decay"]:
            self.policy.lr_decay(episode_num, max_episodes=a.max_episodes)

    def finish(self, model_dir):
        if self.obs_norm is not None:
            np.save(os.path.join(model_dir, "%s_running_mean_std.npy" % self.args.policy_name),
                    np.array([self.obs_norm.running_ms.mean, self.obs_norm.running_ms.std]))


def _run_loop(args, env, policy, hooks, model_dir, writer, log=print):
    """One env step per iteration (the reference's `while episode_num < max_episodes`)."""
    episode_num, step, episode_reward, returns = 0, 0, 0, []
    obs, _ = env.reset(seed=args.seed)
    if args.random_steps > 0:
        env.action_space.seed(seed=args.seed)
    obs = hooks.begin(obs)
    t0 = time.time()
    while episode_num < args.max_episodes:
        step += 1
        action, env_action, extra = hooks.act(step, obs)
        next_obs, reward, terminated, truncated, _ = env.step(env_action)
        next_obs, stored_reward = hooks.observe(next_obs, reward)
        done = terminated or truncated
        hooks.store(obs, action, stored_reward, next_obs, terminated, done, extra)   # done flag = terminated only
        episode_reward += reward
        obs = next_obs
        if done:
            hooks.episode_end(episode_num)
            if (episode_num + 1) % 100 == 0:
                log("episode: {}, reward: {}".format(episode_num + 1, episode_reward))
            if (episode_num + 1) % args.save_freq == 0 and not isinstance(hooks, (_SACHooks, _PPOHooks)):
                policy.save(model_dir)
            writer.add_scalar("reward", episode_reward, episode_num + 1)
            returns.append(episode_reward)
            episode_num += 1
            obs, _ = env.reset(seed=args.seed)          # same seed at every episode start (DQN.py:334)
            obs = hooks.after_reset(obs)
            episode_reward = 0
        if hooks.learn_due(step):
            hooks.learn(episode_num)
        if episode_num % args.save_freq == 0:           # fires on every step of those episodes (DQN.py:342-343)
            policy.save(model_dir)
    log("total_time:", time.time() - t0)
    policy.save(model_dir)
    np.save(os.path.join(model_dir, "%s_seed_%d.npy" % (args.policy_name, args.seed)), np.array(returns))
    hooks.finish(model_dir)
    return dict(returns=np.array(returns), steps=step, seconds=time.time() - t0)


def _run_maddpg(args, env, policy, dim_info, max_action, model_dir, writer, log=print):
    """MADDPG_simple.py:336-395: dict-in/dict-out parallel env, per-agent returns."""
    agents = list(env.agents)
    episode_num, step = 0, 0
    episode_reward = {a: 0 for a in agents}
    returns = {a: [] for a in agents}
    obs, _ = env.reset(seed=args.seed)
    for a in agents:
        env.action_space(a).seed(seed=args.seed)
    if args.gauss_init_scale is not None:
        args.gauss_scale = args.gauss_init_scale
    t0 = time.time()
    while episode_num < args.max_episodes:
        step += 1
        if step < args.random_steps:
            env_action = {a: env.action_space(a).sample() for a in agents}
            action = {a: (env_action[a] * 2 - 1) * max_action for a in agents}
        else:
            action = policy.select_action(obs)
            env_action = {}
            for a in agents:        # one NumPy draw per agent, in env.agents order (:350)
                noisy = action[a] * max_action + args.gauss_scale * np.random.normal(scale=args.gauss_sigma * max_action, size=dim_info[a][1])
                env_action[a] = (np.clip(noisy, -max_action, max_action).astype(np.float32) + 1) / 2
        next_obs, reward, terminated, truncated, _ = env.step(env_action)
        done = {a: terminated[a] or truncated[a] for a in agents}
        policy.add(obs, action, reward, next_obs, {a: terminated[a] for a in agents})
        for a, r in reward.items():
            episode_reward[a] += r
        obs = next_obs
        if any(done.values()):
            if args.gauss_init_scale is not None:
                args.gauss_scale = args.gauss_final_scale + (args.gauss_init_scale - args.gauss_final_scale) * _remaining(args, episode_num)
            if (episode_num + 1) % 100 == 0:
                log("episode: {}, reward: {}".format(episode_num + 1, episode_reward))
            for a, r in episode_reward.items():
                writer.add_scalar("reward_%s" % a, r, episode_num + 1)
                returns[a].append(r)
            episode_num += 1
            obs, _ = env.reset(seed=args.seed)
            episode_reward = {a: 0 for a in agents}
        if step > args.start_steps and step % args.learn_steps_interval == 0:
            policy.learn(args.batch_size, args.gamma, args.tau)
        if episode_num % args.save_freq == 0:
            policy.save(model_dir)
    log("total_time:", time.time() - t0)
    policy.save(model_dir)
    arr = np.array([returns[a] for a in agents])
    suffix = "" if args.N is None else "_N_%d" % len(agents)
    np.save(os.path.join(model_dir, "%s_seed_%d%s.npy" % (args.policy_name, args.seed, suffix)), arr)
    return dict(returns=arr, steps=step, seconds=time.time() - t0)


# ------------------------------------------------------------------------------------ drivers
def run(algo, argv=None, env=None, log=print):
    """Parse the reference's flags for `algo`, build env + policy, run the loop, return a dict
    (model_dir, returns, steps, seconds, policy)."""
    args = build_parser(algo).parse_args(argv)
    if algo == "td3" and args.policy_name == "TD3":
        args.realize = {"clip_double": True, "policy_noise": True, "twin_delay": True}      # TD3.py:357-358
    if algo == "ppo" and args.trick["reward_norm"] and args.trick["reward_scaling"]:
        raise ValueError("reward_norm and reward_scaling are mutually exclusive")
    log(args)
    log("-" * 50)
    log("Algorithm:", args.policy_name)
    device = torch.device(args.device) if torch.cuda.is_available() else torch.device("cpu")
    if algo == "maddpg":
        env = env if env is not None else _envs.make_parallel(args.env_name, args.N)
        env.reset()
        dim_info = {a: [env.observation_space(a).shape[0], env.action_space(a).shape[0]] for a in env.agents}
        max_action, is_continue = 1, True
    else:
        if env is None:
            env, dim_info, max_action, is_continue = get_env(args.env_name, args.is_dis_to_con)
        else:
            _, dim_info, max_action, is_continue = _space_info(env, args.is_dis_to_con)
    np.random.seed(args.seed)                       # seeds AFTER env creation, BEFORE the policy (DQN.py:260-266)
    torch.manual_seed(args.seed)
    model_dir = make_dir(args.results_root, args.env_name, policy_name=args.policy_name, trick=args.trick)
    log("model_dir:", model_dir)
    writer = ScalarWriter(model_dir)
    # the engine's Philox key follows --seed too: with rng="auto" the index / noise draws move from the seeded NumPy / torch
    # streams to the device generator once the buffer holds _core.AUTO_DEVICE_MIN_ROWS rows, and two --seed values must not
    # share that stream
    kw = dict(rng=args.rng, seed=args.seed)
    if args.rng == "auto":
        from ._core import AUTO_DEVICE_MIN_ROWS
        log("rng=auto: index / noise draws follow the reference's host streams below %d buffer rows, the device "
            "generator (keyed by --seed) from there on; --rng host keeps the reference's streams throughout" % AUTO_DEVICE_MIN_ROWS)
    if algo == "dqn":
        from .DQN import DQN
        policy = DQN(dim_info, is_continue, Qnet_lr=args.Qnet_lr, buffer_size=args.buffer_size, device=device, **kw)
        hooks = _DQNHooks(args, env, policy, dim_info, max_action)
    elif algo == "ddpg":
        from .DDPG import DDPG
        policy = DDPG(dim_info, is_continue, args.actor_lr, args.critic_lr, args.buffer_size, device, trick=args.trick, **kw)
        hooks = _GaussHooks(args, env, policy, dim_info, max_action)
    elif algo == "td3":
        from .TD3 import TD3
        policy = TD3(dim_info, is_continue, args.actor_lr, args.critic_lr, args.buffer_size, device, trick=args.trick,
                     realize=args.realize, **kw)
        hooks = _GaussHooks(args, env, policy, dim_info, max_action)
    elif algo == "sac":
        from .SAC import SAC
        policy = SAC(dim_info, is_continue, args.actor_lr, args.critic_lr, args.buffer_size, device, trick=args.trick, **kw)
        hooks = _SACHooks(args, env, policy, dim_info, max_action)
    elif algo == "ppo":
        from .PPO import PPO
        policy = PPO(dim_info, is_continue, args.actor_lr, args.critic_lr, args.horizon, device, trick=args.trick,
                     beta=args.beta, **kw)
        hooks = _PPOHooks(args, env, policy, dim_info, max_action)
    elif algo == "maddpg":
        from .MADDPG import MADDPG
        policy = MADDPG(dim_info, is_continue, args.actor_lr, args.critic_lr, args.buffer_size, device, args.trick, **kw)
        out = _run_maddpg(args, env, policy, dim_info, max_action, model_dir, writer, log)
        writer.close()
        return dict(out, model_dir=model_dir, policy=policy, args=args)
    else:
        raise ValueError(algo)
    out = _run_loop(args, env, policy, hooks, model_dir, writer, log)
    writer.close()
    return dict(out, model_dir=model_dir, policy=policy, args=args)


def _space_info(env, is_dis_to_con):
    obs_space, act_space = env.observation_space, env.action_space
    obs_dim = obs_space.shape[0] if hasattr(obs_space, "low") else 1
    if hasattr(act_space, "low"):
        ad = act_space.shape[0]
        if is_dis_to_con:
            return env, [obs_dim, 16 if ad == 1 else 2 ** ad], act_space.high[0], False
        return env, [obs_dim, ad], act_space.high[0], True
    return env, [obs_dim, act_space.n], None, False


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in FLAGS:
        raise SystemExit("usage: python -m freerl_amd.train {%s} [reference flags]" % "|".join(FLAGS))
    run(argv[0], argv[1:])


if __name__ == "__main__":
    main()
