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

    def __init__(self, capacity, obs_dim, act_dim):
        super().__init__(capacity, obs_dim, act_dim)
        self.action_log_probs = np.zeros((self.capacity, act_dim))
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
