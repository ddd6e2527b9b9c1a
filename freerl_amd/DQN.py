"""`DQN` with the reference's class surface (DQN_file/DQN.py:48-138), backed by the HIP engine.

    policy = DQN(dim_info, is_continue, Qnet_lr, buffer_size, device, trick=None)
    policy.select_action(obs) / evaluate_action(obs) / add(...) / sample(B) / learn(B, gamma, tau)
    policy.update_target(tau) / save(model_dir) / DQN.load(dim_info, is_continue, model_dir)
    policy.agent.Qnet / .Qnet_target / .Qnet_optimizer, policy.buffer

`rng="host"` draws the sample indices exactly like the reference (`np.random.choice(size, B,
replace=False)`, DQN.py:97 — a permutation of the whole buffer per learn()) so a seeded run consumes
the same RNG stream; `rng="device"` draws them with the engine's Philox generator (no host work per
learn); `rng="auto"` (default) is "host" while the buffer is small enough for that permutation to be
cheap (< _core.AUTO_DEVICE_MIN_ROWS rows) and "device" beyond.
"""
import os

import numpy as np
import torch

from . import _native as N
from ._core import DeviceNet, Engine, OptimizerView, draw_indices, host_draw, init_layers, resolve_device
from .Buffer import Buffer


class Agent:
    """Agent (DQN.py:47-59): Qnet, Qnet_target (deepcopy), Adam(Qnet.parameters(), lr)."""

    def __init__(self, engine, obs_dim, action_dim, Qnet_lr, hidden):
        layers = [("l1", hidden, obs_dim), ("l2", action_dim, hidden)]     # MLP (DQN.py:32-45)
        flat = init_layers(layers)                                          # torch RNG: l1 then l2
        engine.set_params(0, flat, N.PARAM_ONLINE)
        engine.set_params(0, flat, N.PARAM_TARGET)                          # deepcopy: no RNG draw
        self.Qnet = DeviceNet(engine, 0, layers)
        self.Qnet_target = DeviceNet(engine, 0, layers, kind=N.PARAM_TARGET)
        self.Qnet_optimizer = OptimizerView(engine, 0, Qnet_lr)

    def update_Qnet(self, loss):
        raise NotImplementedError("zero_grad/backward/step are fused into DQN.learn() on the GPU")


class DQN:
    def __init__(self, dim_info, is_continue, Qnet_lr, buffer_size, device, trick=None, *, rng="auto", hidden=128,
                 batch_max=1024, seed=0):
        obs_dim, action_dim = dim_info
        if is_continue:
            raise ValueError("DQN is not suitable for continuous action spaces; use the dis_to_con wrapper "
                             "(DQN.py:78-81)")
        hip_id, self.device = resolve_device(device)
        self._e = Engine(N.ALGO_DQN, obs_dim, action_dim, max(int(buffer_size), 1), discrete=True, hidden=hidden,
                         batch_max=batch_max, device_id=hip_id, seed=seed)
        self.agent = Agent(self._e, obs_dim, action_dim, Qnet_lr, hidden)
        self.buffer = Buffer(buffer_size, obs_dim, act_dim=1, device=self.device, _engine=self._e)   # DQN.py:66
        self.is_continue = is_continue
        self._rng = rng
        self.last_loss = None

    def select_action(self, obs):
        """argmax_a Q(obs, a) as a NumPy integer scalar (DQN.py:70-84)."""
        a = self._e.act(0, N.ACT_ARGMAX, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1))
        return np.int64(a[0, 0, 0])

    def evaluate_action(self, obs):
        return self.select_action(obs)

    def add(self, obs, action, reward, next_obs, done):
        self.buffer.add(obs, action, reward, next_obs, done)

    def sample(self, batch_size):
        indices = draw_indices(len(self.buffer), batch_size)
        return self.buffer.sample(indices)

    def learn(self, batch_size, gamma, tau):
        """One fused update (DQN.py:104-118); returns None like the reference, the loss is kept
        in `last_loss` only when `track_loss` was requested (it costs a device sync)."""
        total = len(self.buffer)
        batch = min(total, batch_size)
        idx = draw_indices(total, batch_size) if host_draw(self._rng, total, batch_size) else None
        st = self._e.learn(batch, gamma=gamma, tau=tau, critic_lr=self.agent.Qnet_optimizer.lr, clip_norm=0.0,
                           idx=idx, want_stats=getattr(self, "track_loss", False))
        if st is not None:
            self.last_loss = float(st[0, 0, N.STAT_CRITIC_LOSS])

    def update_target(self, tau):
        """theta_t <- theta_t*(1-tau) + theta*tau (DQN.py:120-128); learn() already does it — this
        is the stand-alone form for callers that invoke it directly."""
        q, t = self._e.get_params(0, N.PARAM_ONLINE), self._e.get_params(0, N.PARAM_TARGET)
        self._e.set_params(0, t * np.float32(1.0 - tau) + q * np.float32(tau), N.PARAM_TARGET)

    def save(self, model_dir):
        torch.save(self.agent.Qnet.state_dict(), os.path.join(model_dir, "DQN.pt"))

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None):
        policy = DQN(dim_info, is_continue, 0, 0, device=torch.device("cpu"), trick=trick)
        policy.agent.Qnet.load_state_dict(torch.load(os.path.join(model_dir, "DQN.pt")))
        return policy
