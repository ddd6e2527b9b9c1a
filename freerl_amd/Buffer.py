"""Drop-in for the reference's `Buffer.py`: `from Buffer import Buffer` ->
`from freerl_amd.Buffer import Buffer` (same constructor, methods and public attributes).

class Buffer           <- TD3_file/Buffer.py:11-61 (= DQN_file/Buffer.py:12-62, DDPG/SAC/MADDPG copies)
class Buffer_for_PPO   <- PPO_file/Buffer.py:266-323

Storage is a GPU-resident ring of fp32 transition records (one 128-byte-aligned row per
transition) instead of five float64 host arrays; `add` stages into pinned host memory,
`sample(indices)` is ONE gather launch that writes the five dense fp32 tensors the reference
returns.  float32 storage is lossless for what the reference stores: observations and actions
arrive as float32, rewards are cast to float32 by `sample` anyway (Buffer.py:52), dones are
0/1.
"""
import numpy as np
import torch

from . import _native as N
from ._core import Engine, F32, resolve_device


class _RingView:
    """Shared implementation: a (learner, agent) view of an engine's replay ring."""

    def _attach(self, engine, learner, agent, device):
        self._e, self._learner, self._agent = engine, learner, agent
        self._own = False
        self.device = device
        lay = engine.layout
        self._obs = (lay.obs_off[agent], lay.obs_dim[agent])
        self._act = (lay.act_off[agent], lay.act_dim[agent])
        self._rew = (lay.rew_off + agent, 1)
        self._nobs = (lay.next_obs_off[agent], lay.obs_dim[agent])
        self._done = (lay.done_off + agent, 1)
        self._extra = (lay.extra_off, lay.extra)

    # -- cursor attributes the reference exposes (PER reads buffer._index, DQN_file/Buffer.py:96)
    @property
    def _index(self):
        return self._e.cursor(self._learner)[0]

    @_index.setter
    def _index(self, v):
        self._e.set_cursor(self._learner, int(v), self._e.cursor(self._learner)[1])

    @property
    def _size(self):
        return self._e.cursor(self._learner)[1]

    @_size.setter
    def _size(self, v):
        self._e.set_cursor(self._learner, self._e.cursor(self._learner)[0], int(v))

    def __len__(self):
        return self._size

    def _gather(self, indices, fields):
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        outs = [torch.empty((idx.size, w), dtype=torch.float32, device="cuda:%d" % self._e.cfg.device_id)
                for _, w in fields]
        self._e.sample_into(self._learner, idx, fields, [t.data_ptr() for t in outs])
        if self.device.type != "cuda":
            outs = [t.to(self.device) for t in outs]
        return outs

    def _column(self, field, squeeze=False, dtype=None):
        rows = self._e.read_rows(self._learner, 0, self.capacity)
        a = rows[:, field[0]:field[0] + field[1]]
        a = a[:, 0] if squeeze else a
        return a.astype(dtype) if dtype is not None else a

    # -- the reference's public ndarray fields, as read-only copies of the ring content
    @property
    def obs(self):
        return self._column(self._obs)

    @property
    def actions(self):
        return self._column(self._act)

    @property
    def rewards(self):
        return self._column(self._rew, squeeze=True)

    @property
    def next_obs(self):
        return self._column(self._nobs)

    @property
    def dones(self):
        return self._column(self._done, squeeze=True, dtype=bool)


class Buffer(_RingView):
    """replay buffer for each agent (TD3_file/Buffer.py:11-61)."""

    def __init__(self, capacity, obs_dim, act_dim, device, *, batch_max=1024, _engine=None, _learner=0, _agent=0):
        self.capacity = capacity = int(capacity)
        if _engine is None:
            hip_id, dev = resolve_device(device)
            eng = Engine(N.ALGO_REPLAY_ONLY, int(obs_dim), int(act_dim), max(capacity, 1), device_id=hip_id,
                         batch_max=batch_max)
            self._attach(eng, 0, 0, dev)
            self._own = True
        else:
            self._attach(_engine, _learner, _agent, device)
        self._rec = np.zeros(self._e.width, dtype=F32)

    def add(self, obs, action, reward, next_obs, done):
        """add an experience to the memory (Buffer.py:28-38)."""
        if self._own or self._e.n_agents == 1:
            r = self._rec
            r[self._obs[0]:self._obs[0] + self._obs[1]] = np.asarray(obs, dtype=F32).reshape(-1)
            r[self._act[0]:self._act[0] + self._act[1]] = np.asarray(action, dtype=F32).reshape(-1)
            r[self._rew[0]] = reward
            r[self._nobs[0]:self._nobs[0] + self._nobs[1]] = np.asarray(next_obs, dtype=F32).reshape(-1)
            r[self._done[0]] = float(done)
            self._e.add(self._learner, r)
        else:
            raise RuntimeError("per-agent views of a joint MADDPG ring are written through MADDPG.add")

    def sample(self, indices):
        """(obs[B,O], actions[B,A'], rewards[B,1], next_obs[B,O], dones[B,1]) float32 on `device`
        (Buffer.py:40-57)."""
        return tuple(self._gather(indices, [self._obs, self._act, self._rew, self._nobs, self._done]))


class Buffer_for_PPO(_RingView):
    """PPO rollout storage (PPO_file/Buffer.py:266-323): transition + per-dimension old
    log-probs + adv_done; `all()` returns the WHOLE arrays, `clear()` resets the counters only."""

    _tail_cols = 1           # columns after the log-probs: adv_done

    def __init__(self, capacity, obs_dim, act_dim, device, trick=None, *, batch_max=256, _engine=None, _learner=0):
        self.capacity = capacity = int(capacity)
        # trick['decaystd'] (Buffer.py:277-278): ONE scalar log-prob per step, `all()` returns it as a 1-D tensor
        self._scalar_logp = bool(trick is not None and trick.get("decaystd"))
        if self._scalar_logp and _engine is not None:
            raise NotImplementedError("decaystd storage is a stand-alone buffer: PPO_std_decay.py's learn() is not ported")
        self._act_dim = int(act_dim)
        self._logp_dim = 1 if self._scalar_logp else int(act_dim)
        if _engine is None:
            hip_id, dev = resolve_device(device)
            eng = Engine(N.ALGO_REPLAY_ONLY, int(obs_dim), int(act_dim), max(capacity, 1), device_id=hip_id,
                         batch_max=batch_max, extra_cols=self._logp_dim + self._tail_cols)
            self._attach(eng, 0, 0, dev)
            self._own = True
        else:
            self._attach(_engine, _learner, 0, device)
        self._rec = np.zeros(self._e.width, dtype=F32)

    def add(self, obs, action, reward, next_obs, done, action_log_probs, adv_done):
        r = self._rec
        r[self._obs[0]:self._obs[0] + self._obs[1]] = np.asarray(obs, dtype=F32).reshape(-1)
        r[self._act[0]:self._act[0] + self._act[1]] = np.asarray(action, dtype=F32).reshape(-1)
        r[self._rew[0]] = reward
        r[self._nobs[0]:self._nobs[0] + self._nobs[1]] = np.asarray(next_obs, dtype=F32).reshape(-1)
        r[self._done[0]] = float(done)
        x0 = self._extra[0]
        r[x0:x0 + self._logp_dim] = np.asarray(action_log_probs, dtype=F32).reshape(-1)
        r[x0 + self._logp_dim] = float(adv_done)
        self._e.add(self._learner, r)

    def clear(self):
        self._e.set_cursor(self._learner, 0, 0)

    @property
    def action_log_probs(self):
        return self._column((self._extra[0], self._logp_dim), squeeze=self._scalar_logp)

    @property
    def adv_dones(self):
        return self._column((self._extra[0] + self._logp_dim, 1), squeeze=True, dtype=bool)

    def all(self):
        idx = np.arange(self.capacity, dtype=np.int64)
        fields = [self._obs, self._act, self._rew, self._nobs, self._done, (self._extra[0], self._logp_dim),
                  (self._extra[0] + self._logp_dim, 1)]
        # one gather per <= 16*batch_max rows (C ABI limit)
        step = 16 * self._e.batch_max
        chunks = [self._gather(idx[s:s + step], fields) for s in range(0, self.capacity, step)]
        out = [torch.cat([c[f] for c in chunks], dim=0) for f in range(len(fields))]
        if self._scalar_logp:
            out[5] = out[5].reshape(-1)                     # np.zeros((capacity)) in the reference: shape [N]
        return tuple(out)


class Buffer_for_PPO_2(Buffer_for_PPO):
    """PPO_advance/Buffer.py:435-533: Buffer_for_PPO + the critic's value of every step, stored at rollout time
    (`add(..., value)`, `all()` returns it as an eighth tensor).  `compute_returns_and_advantage` (:480-507) is not a
    host loop here: `PPO.learn(..., last_value)` runs it on the device over the ring (frl_ppo_learn, gae_mode 1) and leaves
    its float32 results in `.advantages` / `.returns`."""
    _tail_cols = 2           # value, adv_done (adv_done stays the last extra column)

    def __init__(self, capacity, obs_dim, act_dim, device, trick=None, **kw):
        super().__init__(capacity, obs_dim, act_dim, device, trick, **kw)
        self.advantages = np.zeros(self.capacity, dtype=F32)
        self.returns = np.zeros(self.capacity, dtype=F32)

    def add(self, obs, action, reward, next_obs, done, action_log_probs, adv_done, value):
        r = self._rec
        r[self._obs[0]:self._obs[0] + self._obs[1]] = np.asarray(obs, dtype=F32).reshape(-1)
        r[self._act[0]:self._act[0] + self._act[1]] = np.asarray(action, dtype=F32).reshape(-1)
        r[self._rew[0]] = reward
        r[self._nobs[0]:self._nobs[0] + self._nobs[1]] = np.asarray(next_obs, dtype=F32).reshape(-1)
        r[self._done[0]] = float(done)
        x0 = self._extra[0]
        r[x0:x0 + self._logp_dim] = np.asarray(action_log_probs, dtype=F32).reshape(-1)
        r[x0 + self._logp_dim] = float(np.asarray(value).reshape(-1)[0])
        r[x0 + self._logp_dim + 1] = float(adv_done)
        self._e.add(self._learner, r)

    @property
    def values(self):
        return self._column((self._extra[0] + self._logp_dim, 1), squeeze=True)

    @property
    def adv_dones(self):
        return self._column((self._extra[0] + self._logp_dim + 1, 1), squeeze=True, dtype=bool)

    def all(self):
        idx = np.arange(self.capacity, dtype=np.int64)
        x0 = self._extra[0]
        fields = [self._obs, self._act, self._rew, self._nobs, self._done, (x0, self._logp_dim),
                  (x0 + self._logp_dim + 1, 1), (x0 + self._logp_dim, 1)]
        step = 16 * self._e.batch_max
        chunks = [self._gather(idx[s:s + step], fields) for s in range(0, self.capacity, step)]
        out = [torch.cat([c[f] for c in chunks], dim=0) for f in range(len(fields))]
        if self._scalar_logp:
            out[5] = out[5].reshape(-1)
        return tuple(out)


# ------------------------------------------------------------------------------------ PER / N-step (DQN_file/Buffer.py:66-399)
class SumTreeView:
    """What callers read on `PER_Buffer.sumtree` (DQN_file/Buffer.py:187-194): `sum()` and `max()`."""

    def __init__(self, engine, learner=0):
        self._e, self._learner = engine, learner

    def sum(self):
        return self._e.per_state(self._learner)["sum"]

    def max(self):
        return self._e.per_state(self._learner)["max"]


class PER_Buffer:
    """PER_Buffer (DQN_file/Buffer.py:66-129): the sum-tree lives next to the ring in HBM; `add` gives a new row the
    current maximum priority, `sample(batch_size)` -> (indices int64 [B], is_weight float32 tensor [B]),
    `update_priorities(indices, td_error)`.  `.buffer` is the plain ring view (`.buffer.sample(indices)`, :230)."""

    def __init__(self, capacity, obs_dim, act_dim, device, alpha=0.5, beta=0.4, beta_increment=0.001, epsilon=0.01, *,
                 batch_max=1024, _engine=None):
        self.capacity, self.alpha, self.beta_increment, self.epsilon = int(capacity), alpha, beta_increment, epsilon
        self.buffer = Buffer(capacity, obs_dim, act_dim, device, batch_max=batch_max, _engine=_engine)
        self.device = self.buffer.device
        self._e = self.buffer._e
        self._e.per_enable(alpha, beta, beta_increment, epsilon)
        self.sumtree = SumTreeView(self._e)

    @property
    def beta(self):
        return self._e.per_state(0)["beta"]

    def add(self, obs, action, reward, next_obs, done):
        self.buffer.add(obs, action, reward, next_obs, done)

    def sample(self, batch_size):
        """The stratified draws are np.random.uniform's own arithmetic on np.random.random_sample() (a + (b-a)*u), so a
        seeded run consumes NumPy's global stream exactly like the reference's loop (:107-114)."""
        u = np.array([np.random.random_sample() for _ in range(batch_size)])
        if getattr(self, "_device_only", False):       # learn(): rows and weights are consumed on the device, nothing read back
            self._e.per_sample(batch_size, uniforms=u, want_outputs=False)
            return None, None
        idx, w = self._e.per_sample(batch_size, uniforms=u)
        return idx[0], torch.as_tensor(w[0], dtype=torch.float32).to(self.device)

    def update_priorities(self, indices, td_error):
        td = np.asarray(torch.as_tensor(td_error).detach().cpu().numpy(), dtype=F32).reshape(-1)
        self._e.per_update(td.size, idx=np.asarray(indices, dtype=np.int64).reshape(1, -1), td_error=td.reshape(1, -1))

    def __len__(self):
        return len(self.buffer)


def _n_step_info(window, gamma):
    """_get_n_step_info (DQN_file/Buffer.py:240-275): the oldest (obs, action); the return folded from the newest entry
    back; next_obs / done of the earliest terminal inside the window.  Host arithmetic in Python floats, like the reference."""
    obs, action = window[0][0], window[0][1]
    _, _, reward, next_obs, done = window[-1]
    for i in range(len(window) - 2, -1, -1):
        _, _, r, n_o, d = window[i]
        reward = r + gamma * reward * (1 - d)
        if d:
            next_obs, done = n_o, d
    return obs, action, reward, next_obs, done


class N_Step_Buffer(Buffer):
    """N_Step_Buffer (DQN_file/Buffer.py:199-296): a deque of the last n transitions; once full, every add stores the
    n-step transition of the oldest entry.  `n_step_gamma` = gamma ** n_step is what learn() bootstraps with."""

    def __init__(self, capacity, obs_dim, act_dim, device, gamma, n_step=2, **kw):
        from collections import deque
        super().__init__(capacity, obs_dim, act_dim, device, **kw)
        self.n_step, self.gamma, self.n_step_gamma = n_step, gamma, gamma ** n_step
        self.n_step_deque = deque(maxlen=n_step)

    def add(self, obs, action, reward, next_obs, done):
        self.n_step_deque.append((obs, action, reward, next_obs, done))
        if len(self.n_step_deque) == self.n_step:
            super().add(*_n_step_info(list(self.n_step_deque), self.gamma))


class N_Step_PER_Buffer(PER_Buffer):
    """N_Step_PER_Buffer (DQN_file/Buffer.py:333-399): the n-step fold in front of the prioritised ring."""

    def __init__(self, capacity, obs_dim, act_dim, device, alpha=0.5, beta=0.4, beta_increment=0.001, epsilon=0.01,
                 gamma=None, n_step=3, **kw):
        from collections import deque
        super().__init__(capacity, obs_dim, act_dim, device, alpha, beta, beta_increment, epsilon, **kw)
        self.n_step, self.gamma, self.n_step_gamma = n_step, gamma, gamma ** n_step
        self.n_step_deque = deque(maxlen=n_step)

    def add(self, obs, action, reward, next_obs, done):
        self.n_step_deque.append((obs, action, reward, next_obs, done))
        if len(self.n_step_deque) == self.n_step:
            super().add(*_n_step_info(list(self.n_step_deque), self.gamma))
