"""In-repo environments behind the gymnasium protocol the reference's loops rely on (SURVEY.md
§3.5): `make(name)`, `env.reset(seed=) -> (obs, info)`, `env.step(a) -> (obs, reward, terminated,
truncated, info)`, `observation_space` / `action_space` (`Box.shape/.low/.high`, `Discrete.n`,
`.seed()`, `.sample()`), `spec(name).reward_threshold`.

`gymnasium` is not installed in the build image nor on the GPU box, so the classic-control tasks
are restated from their published equations (Gymnasium docs: Pendulum-v1, CartPole-v1) and a
synthetic linear-Gaussian task provides the north_star's obs_dim 8 / act_dim 2|4 shape.  When the
real `gymnasium` is importable, `make()` prefers it.  Env-level parity with gymnasium cannot be
executed in this environment and is unpinned; the learner path does not depend
on it.
"""
import math

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape if shape else np.shape(low)).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.low.shape).copy()
        self.shape = self.low.shape
        self.dtype = dtype
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(self.n))


class _Env:
    max_episode_steps = None
    reward_threshold = None

    def _seed(self, seed):
        if seed is not None or not hasattr(self, "np_random"):
            self.np_random = np.random.default_rng(seed)

    def close(self):
        pass


class PendulumEnv(_Env):
    """Pendulum-v1: theta'' = 3g/(2l) sin(theta) + 3/(m l^2) u, dt 0.05, |u| <= 2, |theta'| <= 8,
    reward -(angle_normalize(theta)^2 + 0.1 theta'^2 + 0.001 u^2), time limit 200, never terminates."""
    max_episode_steps = 200
    max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0

    def __init__(self):
        high = np.array([1.0, 1.0, self.max_speed], dtype=np.float32)
        self.observation_space = Box(-high, high)
        self.action_space = Box(-self.max_torque, self.max_torque, shape=(1,))
        self._t = 0

    def _obs(self):
        th, thdot = self.state
        return np.array([math.cos(th), math.sin(th), thdot], dtype=np.float32)

    def reset(self, seed=None, options=None):
        self._seed(seed)
        self.state = self.np_random.uniform(low=[-math.pi, -1.0], high=[math.pi, 1.0])
        self._t = 0
        return self._obs(), {}

    def step(self, u):
        th, thdot = self.state
        u = float(np.clip(np.asarray(u, dtype=np.float64).reshape(-1)[0], -self.max_torque, self.max_torque))
        ang = ((th + math.pi) % (2 * math.pi)) - math.pi
        cost = ang ** 2 + 0.1 * thdot ** 2 + 0.001 * u ** 2
        thdot = thdot + (3 * self.g / (2 * self.l) * math.sin(th) + 3.0 / (self.m * self.l ** 2) * u) * self.dt
        thdot = float(np.clip(thdot, -self.max_speed, self.max_speed))
        th = th + thdot * self.dt
        self.state = np.array([th, thdot])
        self._t += 1
        return self._obs(), -cost, False, self._t >= self.max_episode_steps, {}


class CartPoleEnv(_Env):
    """CartPole-v1: Euler integration (tau 0.02) of the cart-pole equations, force +-10, reward 1 per
    step, terminated outside |x| <= 2.4 or |theta| <= 12 deg, time limit 500."""
    max_episode_steps = 500
    reward_threshold = 475.0
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02

    def __init__(self):
        self.theta_threshold = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        high = np.array([self.x_threshold * 2, np.finfo(np.float32).max, self.theta_threshold * 2,
                         np.finfo(np.float32).max], dtype=np.float32)
        self.observation_space = Box(-high, high)
        self.action_space = Discrete(2)
        self._t = 0

    def reset(self, seed=None, options=None):
        self._seed(seed)
        self.state = self.np_random.uniform(low=-0.05, high=0.05, size=(4,))
        self._t = 0
        return np.array(self.state, dtype=np.float32), {}

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if int(action) == 1 else -self.force_mag
        total_mass = self.masspole + self.masscart
        pml = self.masspole * self.length
        costh, sinth = math.cos(theta), math.sin(theta)
        temp = (force + pml * theta_dot ** 2 * sinth) / total_mass
        thetaacc = (self.gravity * sinth - costh * temp) / (self.length * (4.0 / 3.0 - self.masspole * costh ** 2 / total_mass))
        xacc = temp - pml * thetaacc * costh / total_mass
        x, x_dot = x + self.tau * x_dot, x_dot + self.tau * xacc
        theta, theta_dot = theta + self.tau * theta_dot, theta_dot + self.tau * thetaacc
        self.state = np.array([x, x_dot, theta, theta_dot])
        self._t += 1
        terminated = bool(x < -self.x_threshold or x > self.x_threshold or theta < -self.theta_threshold
                          or theta > self.theta_threshold)
        return np.array(self.state, dtype=np.float32), 1.0, terminated, self._t >= self.max_episode_steps, {}


class LinearGaussianEnv(_Env):
    """Synthetic task of SURVEY.md §8(d): s' = A s + B a + eps, obs_dim 8; continuous act_dim 2
    (`SynLinear-v0`) or 4 discrete pushes (`SynLinearDiscrete-v0`); reward -(|s|^2 + 0.1|a|^2)/O;
    terminates when |s|_inf > 6 (leaving the basin), time limit 200.  A, B are fixed (seed 2024)."""
    max_episode_steps = 200

    def __init__(self, discrete=False, obs_dim=8, act_dim=2):
        g = np.random.default_rng(2024)
        q, _ = np.linalg.qr(g.standard_normal((obs_dim, obs_dim)))
        self.A = 0.95 * q
        self.B = g.standard_normal((obs_dim, act_dim)) * 0.5
        self.discrete = discrete
        self.obs_dim, self.act_dim = obs_dim, act_dim
        high = np.full(obs_dim, 6.0, dtype=np.float32)
        self.observation_space = Box(-high, high)
        self.action_space = Discrete(2 * act_dim) if discrete else Box(-1.0, 1.0, shape=(act_dim,))
        self._t = 0

    def reset(self, seed=None, options=None):
        self._seed(seed)
        self.state = self.np_random.standard_normal(self.obs_dim)
        self._t = 0
        return self.state.astype(np.float32), {}

    def step(self, action):
        if self.discrete:
            a = np.zeros(self.act_dim)
            k = int(action)
            a[k // 2] = 1.0 if k % 2 == 0 else -1.0
        else:
            a = np.clip(np.asarray(action, dtype=np.float64).reshape(-1), -1.0, 1.0)
        s = self.A @ self.state + self.B @ a + 0.05 * self.np_random.standard_normal(self.obs_dim)
        reward = -(float(s @ s) + 0.1 * float(a @ a)) / self.obs_dim
        self.state = s
        self._t += 1
        terminated = bool(np.max(np.abs(s)) > 6.0)
        return s.astype(np.float32), reward, terminated, self._t >= self.max_episode_steps, {}


class PendulumShortEnv(PendulumEnv):
    """Pendulum-v1 dynamics with a 40-step time limit: short episodes for loop tests (several
    episode boundaries within few learner updates)."""
    max_episode_steps = 40


_REGISTRY = {
    "Pendulum-v1": PendulumEnv,
    "PendulumShort-v1": PendulumShortEnv,
    "CartPole-v1": CartPoleEnv,
    "SynLinear-v0": lambda: LinearGaussianEnv(False),
    "SynLinearDiscrete-v0": lambda: LinearGaussianEnv(True),
}


class _Spec:
    def __init__(self, env_id, reward_threshold, max_episode_steps):
        self.id, self.reward_threshold, self.max_episode_steps = env_id, reward_threshold, max_episode_steps


def spec(name):
    if name not in _REGISTRY:
        raise KeyError(name)
    e = _REGISTRY[name]()
    return _Spec(name, e.reward_threshold, e.max_episode_steps)


def make(name, prefer_gymnasium=True, **kwargs):
    """gym.make: the real gymnasium env when the package is importable, else the in-repo restatement."""
    if prefer_gymnasium:
        try:
            import gymnasium
            return gymnasium.make(name, **kwargs)
        except Exception:
            pass
    if name not in _REGISTRY:
        raise KeyError("%r is not available without gymnasium; in-repo envs: %s" % (name, sorted(_REGISTRY)))
    return _REGISTRY[name]()


class spaces:           # `gym.spaces.Box` / `gym.spaces.Discrete` for isinstance checks (DQN.py:144-165)
    Box = Box
    Discrete = Discrete


class SpreadEnv:
    """Cooperative navigation after PettingZoo MPE `simple_spread_v3` (parallel API, continuous
    actions): N agents, N landmarks, 2-D point masses with damping 0.25, dt 0.1, action sensitivity
    5, soft contact forces; observation = [vel(2), pos(2), landmark offsets(2N), other agents'
    offsets(2(N-1)), comm(2(N-1)) zeros] (18 floats at N = 3); action = 5 floats in [0,1]
    (no-op, left, right, down, up); reward = 0.5*(-sum over landmarks of the closest agent distance)
    + 0.5*(-collisions of the agent); episode truncates after `max_cycles` (25) steps."""

    def __init__(self, N=3, max_cycles=25, local_ratio=0.5):
        self.N, self.max_cycles, self.local_ratio = int(N), int(max_cycles), local_ratio
        self.possible_agents = ["agent_%d" % i for i in range(self.N)]
        self.agents = list(self.possible_agents)
        od = 4 + 2 * self.N + 4 * (self.N - 1)
        self._obs_space = {a: Box(-np.inf, np.inf, shape=(od,)) for a in self.agents}
        self._act_space = {a: Box(0.0, 1.0, shape=(5,)) for a in self.agents}
        self.size, self.damping, self.dt, self.sens = 0.15, 0.25, 0.1, 5.0
        self.contact_force, self.contact_margin = 1e2, 1e-3
        self.np_random = np.random.default_rng()

    def observation_space(self, agent):
        return self._obs_space[agent]

    def action_space(self, agent):
        return self._act_space[agent]

    def _observe(self):
        obs = {}
        for i, a in enumerate(self.agents):
            others = [self.pos[j] - self.pos[i] for j in range(self.N) if j != i]
            parts = [self.vel[i], self.pos[i]] + [self.lm[k] - self.pos[i] for k in range(self.N)] + others
            parts.append(np.zeros(2 * (self.N - 1)))
            obs[a] = np.concatenate(parts).astype(np.float32)
        return obs

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.np_random = np.random.default_rng(seed)
        self.agents = list(self.possible_agents)
        self.pos = self.np_random.uniform(-1, 1, (self.N, 2))
        self.vel = np.zeros((self.N, 2))
        self.lm = self.np_random.uniform(-1, 1, (self.N, 2))
        self._t = 0
        return self._observe(), {a: {} for a in self.agents}

    def step(self, actions):
        force = np.zeros((self.N, 2))
        for i, a in enumerate(self.agents):
            u = np.asarray(actions[a], dtype=np.float64)
            force[i] = self.sens * np.array([u[2] - u[1], u[4] - u[3]])
        dmin = 2 * self.size
        for i in range(self.N):
            for j in range(i + 1, self.N):
                delta = self.pos[i] - self.pos[j]
                dist = max(float(np.sqrt(delta @ delta)), 1e-9)
                k = self.contact_margin
                pen = np.logaddexp(0, -(dist - dmin) / k) * k
                f = self.contact_force * delta / dist * pen
                force[i] += f
                force[j] -= f
        self.vel = self.vel * (1 - self.damping) + force * self.dt
        self.pos = self.pos + self.vel * self.dt
        self._t += 1
        glob = 0.0
        for k in range(self.N):
            glob -= min(float(np.linalg.norm(self.pos[i] - self.lm[k])) for i in range(self.N))
        rewards = {}
        for i, a in enumerate(self.agents):
            coll = sum(1 for j in range(self.N) if j != i and np.linalg.norm(self.pos[i] - self.pos[j]) < dmin)
            rewards[a] = glob * (1 - self.local_ratio) + (-float(coll)) * self.local_ratio
        trunc = self._t >= self.max_cycles
        return (self._observe(), rewards, {a: False for a in self.agents}, {a: trunc for a in self.agents},
                {a: {} for a in self.agents})


def make_parallel(env_name, N=None, max_cycles=25, prefer_pettingzoo=True):
    """`importlib.import_module(f'pettingzoo.mpe.{env_name}').parallel_env(...)` (MADDPG_simple.py:214-225)
    when pettingzoo is importable, else the in-repo restatement of simple_spread."""
    if prefer_pettingzoo:
        try:
            import importlib
            mod = importlib.import_module("pettingzoo.mpe.%s" % env_name)
            kw = dict(max_cycles=max_cycles, continuous_actions=True)
            if N is not None:
                kw["N"] = N
            return mod.parallel_env(**kw)
        except Exception:
            pass
    if not env_name.startswith("simple_spread"):
        raise KeyError("only simple_spread is restated in-repo; %r needs pettingzoo" % env_name)
    return SpreadEnv(N if N is not None else 3, max_cycles)
