"""Oracle replay storage: restates TD3_file/Buffer.py:11-61 (= DQN_file/Buffer.py:12-62,
DDPG/SAC/MADDPG identical) and PPO_file/Buffer.py:266-323.  Test infrastructure.

Host float64 / bool ring arrays, write cursor `_index`, fill count `_size`; `sample` casts
to float32 and reshapes rewards/dones to [B,1] exactly like the reference (minus the torch
tensor wrapping / `.to(device)` — the oracle returns NumPy arrays).
"""
import numpy as np

F32 = np.float32


class Buffer:
    def __init__(self, capacity, obs_dim, act_dim):
        self.capacity = capacity = int(capacity)          # Buffer.py:15
        self.obs = np.zeros((capacity, obs_dim))
        self.actions = np.zeros((capacity, act_dim))
        self.rewards = np.zeros(capacity)
        self.next_obs = np.zeros((capacity, obs_dim))
        self.dones = np.zeros(capacity, dtype=bool)
        self._index = 0
        self._size = 0

    def add(self, obs, action, reward, next_obs, done):    # Buffer.py:28-38
        i = self._index
        self.obs[i] = obs
        self.actions[i] = action
        self.rewards[i] = reward
        self.next_obs[i] = next_obs
        self.dones[i] = done
        self._index = (self._index + 1) % self.capacity
        if self._size < self.capacity:
            self._size += 1

    def sample(self, indices):                             # Buffer.py:40-57
        return (self.obs[indices].astype(F32), self.actions[indices].astype(F32),
                self.rewards[indices].astype(F32).reshape(-1, 1), self.next_obs[indices].astype(F32),
                self.dones[indices].astype(F32).reshape(-1, 1))

    def __len__(self):
        return self._size


class BufferForPPO(Buffer):
    """PPO_file/Buffer.py:266-323: + per-dimension old log-probs and adv_dones; `all()` returns
    the WHOLE arrays (capacity rows) as float32; `clear()` resets the counters only."""

    def __init__(self, capacity, obs_dim, act_dim, decaystd=False):
        super().__init__(capacity, obs_dim, act_dim)
        # trick['decaystd'] (:277-278): one scalar log-prob per step instead of one per action dimension
        self.action_log_probs = np.zeros(self.capacity) if decaystd else np.zeros((self.capacity, act_dim))
        self.adv_dones = np.zeros(self.capacity, dtype=bool)

    def add(self, obs, action, reward, next_obs, done, action_log_probs, adv_done):
        i = self._index
        self.action_log_probs[i] = action_log_probs
        self.adv_dones[i] = adv_done
        super().add(obs, action, reward, next_obs, done)

    def clear(self):
        self._index = 0
        self._size = 0

    def all(self):
        return (self.obs.astype(F32), self.actions.astype(F32), self.rewards.astype(F32).reshape(-1, 1),
                self.next_obs.astype(F32), self.dones.astype(F32).reshape(-1, 1),
                self.action_log_probs.astype(F32), self.adv_dones.astype(F32).reshape(-1, 1))


class BufferForPPO2(BufferForPPO):
    """PPO_advance/Buffer.py:435-533 `Buffer_for_PPO_2`: + the critic's value of every stored step, written at rollout time,
    and stable-baselines3's `compute_returns_and_advantage` (:480-507) — a float64 scan over the stored rows."""

    def __init__(self, capacity, obs_dim, act_dim):
        super().__init__(capacity, obs_dim, act_dim)
        self.values = np.zeros(self.capacity)
        self.advantages = np.zeros(self.capacity)
        self.returns = np.zeros(self.capacity)

    def add(self, obs, action, reward, next_obs, done, action_log_probs, adv_done, value):
        self.values[self._index] = value
        super().add(obs, action, reward, next_obs, done, action_log_probs, adv_done)

    def compute_returns_and_advantage(self, gamma, gae_lambda, last_value):
        last = 0
        for step in reversed(range(self._size)):
            next_value = last_value if step == self._size - 1 else self.values[step + 1]
            delta = self.rewards[step] + gamma * next_value * (1.0 - self.dones[step]) - self.values[step]
            last = delta + gamma * gae_lambda * (1.0 - self.adv_dones[step]) * last
            self.advantages[step] = last
        self.returns = self.advantages + self.values


# --------------------------------------------------------------------------------------------- PER / N-step
class SumTree:
    """DQN_file/Buffer.py:131-194: float64 array heap of 2*capacity - 1 nodes, leaves last; `update` walks to the root
    adding the change (:155-164); `get` descends with `s <= left` (:166-185); `max` is np.max over the leaves (:193-194)."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.tree = np.zeros(2 * self.capacity - 1)

    def add(self, buffer_index, priority):
        idx = int(buffer_index + self.capacity - 1)
        change = priority - self.tree[idx]
        self.tree[idx] = priority
        while idx != 0:
            idx = (idx - 1) // 2
            self.tree[idx] += change

    def get(self, s):
        idx = 0
        while True:
            left = 2 * idx + 1
            if left >= len(self.tree):
                break
            if s <= self.tree[left]:
                idx = left
            else:
                s -= self.tree[left]
                idx = left + 1
        return self.tree[idx], idx - self.capacity + 1

    def sum(self):
        return self.tree[0]

    def max(self):
        return np.max(self.tree[-self.capacity:])


class PERBuffer:
    """PER_Buffer (DQN_file/Buffer.py:66-129).  `sample_with(u)` takes the random_sample() draws behind the reference's
    np.random.uniform(a, b) = a + (b - a) * u calls (:111)."""

    def __init__(self, capacity, obs_dim, act_dim, alpha=0.5, beta=0.4, beta_increment=0.001, epsilon=0.01):
        self.alpha, self.beta, self.beta_increment, self.epsilon = alpha, beta, beta_increment, epsilon
        self.sumtree = SumTree(capacity)
        self.buffer = Buffer(capacity, obs_dim, act_dim)

    def add(self, obs, action, reward, next_obs, done):
        max_priority = 1.0 if len(self.buffer) == 0 else self.sumtree.max()       # :96
        self.sumtree.add(self.buffer._index, max_priority)
        self.buffer.add(obs, action, reward, next_obs, done)

    def sample_with(self, u):
        batch_size = len(u)
        idx = np.zeros(batch_size, dtype=np.int64)
        pri = np.zeros(batch_size, dtype=F32)
        segment = self.sumtree.sum() / batch_size
        self.beta = np.min([1., self.beta + self.beta_increment])
        for i in range(batch_size):
            a, b = segment * i, segment * (i + 1)
            p, bi = self.sumtree.get(a + (b - a) * u[i])
            pri[i], idx[i] = p, bi
        prob = np.clip(pri / self.sumtree.sum(), 1e-7, None)
        w = (len(self.buffer) * prob) ** (-self.beta)
        w /= w.max()
        return idx, w.astype(F32)

    def update_priorities(self, indices, td_error):
        pr = (np.abs(np.asarray(td_error, dtype=F32)) + self.epsilon) ** self.alpha   # float32 (the reference's td is a float32 array)
        for i, p in zip(indices, pr.reshape(-1)):
            self.sumtree.add(i, p)

    def __len__(self):
        return len(self.buffer)


def n_step_fold(window, gamma):
    """_get_n_step_info (DQN_file/Buffer.py:240-275): (obs, action) of the oldest entry, the n-step return folded from the
    newest back, next_obs/done of the earliest terminal inside the window."""
    obs, action = window[0][0], window[0][1]
    _, _, reward, next_obs, done = window[-1]
    for i in range(len(window) - 2, -1, -1):
        _, _, r, n_o, d = window[i]
        reward = r + gamma * reward * (1 - d)
        if d:
            next_obs, done = n_o, d
    return obs, action, reward, next_obs, done


class NStepWrapper:
    """N_Step_Buffer / N_Step_PER_Buffer's add() (:222-238, :351-359) in front of a Buffer or PERBuffer."""

    def __init__(self, inner, gamma, n_step):
        from collections import deque
        self.inner, self.gamma, self.n_step = inner, gamma, n_step
        self.n_step_gamma = gamma ** n_step
        self.window = deque(maxlen=n_step)

    def add(self, obs, action, reward, next_obs, done):
        self.window.append((obs, action, reward, next_obs, done))
        if len(self.window) == self.n_step:
            self.inner.add(*n_step_fold(list(self.window), self.gamma))

    def __len__(self):
        return len(self.inner)
