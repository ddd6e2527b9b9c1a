"""`DQN` of DQN_file/DQN_with_tricks.py (:160-308) with the tricks that live on the replay path: Double
(:263-265), PER (PER_Buffer / N_Step_PER_Buffer, :276-279), N_Step (:269-270), plus the Dueling head (:60-79).  Noisy
(Noisy_net.py) and Categorical/C51 (:82-158) are not ported and raise NotImplementedError.

    policy = DQN(dim_info, is_continue, Qnet_lr, buffer_size, device, trick=..., gamma=..., batch_size=...)

One learn() = PER sample (stratified sum-tree descents) -> fused DQN update with the importance weights -> priority
update, all on the GPU; the TD errors never leave the device.
"""
import os

import numpy as np
import torch

from . import _native as N
from ._core import DeviceNet, Engine, OptimizerView, draw_indices, init_layers, resolve_device
from .Buffer import Buffer, N_Step_Buffer, N_Step_PER_Buffer, PER_Buffer
from .DQN import Agent

_NETWORK_TRICKS = ("Noisy", "Categorical")


class DuelingNet(DeviceNet):
    """`agent.Qnet` of Dueling (DQN_with_tricks.py:60-79): state_dict keys l1 / V / A; the engine keeps V and A as one
    (1 + n_actions)-wide head [V ; A].  Calling it returns Q = V + A - mean(A)."""

    def __init__(self, engine, hidden, obs_dim, action_dim, kind=N.PARAM_ONLINE):
        super().__init__(engine, 0, [("l1", hidden, obs_dim), ("head", 1 + action_dim, hidden)], kind=kind, act_mode=N.ACT_RAW)
        self._nA = action_dim

    def keys(self):
        return ["l1.weight", "l1.bias", "V.weight", "V.bias", "A.weight", "A.bias"]

    def _split(self, flat):
        parts = super()._split(flat)
        w, b = parts.pop("head.weight"), parts.pop("head.bias")
        parts.update({"V.weight": w[:1], "V.bias": b[:1], "A.weight": w[1:], "A.bias": b[1:]})
        return parts

    def load_state_dict(self, sd, strict=True):
        if strict and set(sd.keys()) != set(self.keys()):
            raise RuntimeError("Error(s) in loading state_dict: expected keys %s, got %s" % (self.keys(), list(sd.keys())))
        t = lambda k: np.asarray(torch.as_tensor(sd[k]).detach().cpu().numpy(), dtype=np.float32).reshape(-1)
        flat = [t(k) for k in ("l1.weight", "l1.bias", "V.weight", "A.weight", "V.bias", "A.bias")]
        self._e.set_params(self._net, np.concatenate(flat), self._kind, self._learner)

    def __call__(self, obs):
        h = super().__call__(obs)
        return h[:, :1] + h[:, 1:] - h[:, 1:].mean(dim=1, keepdim=True)


class DuelingAgent:
    def __init__(self, engine, obs_dim, action_dim, Qnet_lr, hidden):
        # torch RNG order of Dueling.__init__: l1, V, A (:67-74); the engine's flat order is l1, [V.w ; A.w], [V.b ; A.b]
        f = init_layers([("l1", hidden, obs_dim), ("V", 1, hidden), ("A", action_dim, hidden)])
        o = hidden * obs_dim + hidden
        vw, vb = f[o:o + hidden], f[o + hidden:o + hidden + 1]
        aw, ab = f[o + hidden + 1:o + hidden + 1 + action_dim * hidden], f[o + hidden + 1 + action_dim * hidden:]
        flat = np.concatenate([f[:o], vw, aw, vb, ab])
        engine.set_params(0, flat, N.PARAM_ONLINE)
        engine.set_params(0, flat, N.PARAM_TARGET)
        self.Qnet = DuelingNet(engine, hidden, obs_dim, action_dim)
        self.Qnet_target = DuelingNet(engine, hidden, obs_dim, action_dim, kind=N.PARAM_TARGET)
        self.Qnet_optimizer = OptimizerView(engine, 0, Qnet_lr)


class DQN:
    def __init__(self, dim_info, is_continue, Qnet_lr, buffer_size, device, trick=None, gamma=None, batch_size=None, *,
                 hidden=128, batch_max=1024, seed=0):
        obs_dim, action_dim = dim_info
        if is_continue:
            raise ValueError("DQN is not suitable for continuous action spaces (DQN_with_tricks.py:207-210)")
        for k in _NETWORK_TRICKS:
            if trick[k]:
                raise NotImplementedError("trick['%s'] (DQN_with_tricks.py) is not ported" % k)
        hip_id, self.device = resolve_device(device)
        self._e = Engine(N.ALGO_DQN, obs_dim, action_dim, max(int(buffer_size), 1), discrete=True, hidden=hidden,
                         batch_max=batch_max, device_id=hip_id, seed=seed, dueling=bool(trick["Dueling"]))
        self.agent = (DuelingAgent if trick["Dueling"] else Agent)(self._e, obs_dim, action_dim, Qnet_lr, hidden)
        kw = dict(_engine=self._e)
        if trick["PER"] and trick["N_Step"]:                                   # :176-183
            self.buffer = N_Step_PER_Buffer(buffer_size, obs_dim, 1, self.device, gamma=gamma, **kw)
        elif trick["PER"]:
            self.buffer = PER_Buffer(buffer_size, obs_dim, 1, self.device, **kw)
        elif trick["N_Step"]:
            self.buffer = N_Step_Buffer(buffer_size, obs_dim, 1, self.device, gamma, **kw)
        else:
            self.buffer = Buffer(buffer_size, obs_dim, 1, self.device, **kw)
        self.is_continue, self.trick = is_continue, trick
        self.last_loss = None

    def select_action(self, obs):
        a = self._e.act(0, N.ACT_ARGMAX, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1))
        return np.int64(a[0, 0, 0])

    def evaluate_action(self, obs):
        return self.select_action(obs)

    def add(self, obs, action, reward, next_obs, done):
        self.buffer.add(obs, action, reward, next_obs, done)

    def sample(self, batch_size):                                              # :226-239
        batch_size = min(batch_size, len(self.buffer))
        if self.trick["PER"]:
            indices, is_weight = self.buffer.sample(batch_size)
            return (*self.buffer.buffer.sample(indices), is_weight, indices)
        return self.buffer.sample(draw_indices(len(self.buffer), batch_size))

    def learn(self, batch_size, gamma, tau):                                   # :242-284
        batch = min(batch_size, len(self.buffer))
        if self.trick["N_Step"]:
            gamma = self.buffer.n_step_gamma                                   # :269-270
        idx = None
        if self.trick["PER"]:
            self.buffer.sample(batch)                 # the rows and weights stay on the device for the update below
        else:
            idx = draw_indices(len(self.buffer), batch_size)
        st = self._e.learn(batch, gamma=gamma, tau=tau, critic_lr=self.agent.Qnet_optimizer.lr, clip_norm=0.0, idx=idx,
                           double_dqn=bool(self.trick["Double"]), per=1 if self.trick["PER"] else 0,
                           want_stats=getattr(self, "track_loss", False))
        if self.trick["PER"]:
            self._e.per_update(batch)                 # priorities from the TD errors the kernel left behind (:279)
        if st is not None:
            self.last_loss = float(st[0, 0, N.STAT_CRITIC_LOSS])

    def update_target(self, tau):
        q, t = self._e.get_params(0, N.PARAM_ONLINE), self._e.get_params(0, N.PARAM_TARGET)
        self._e.set_params(0, t * np.float32(1.0 - tau) + q * np.float32(tau), N.PARAM_TARGET)

    def save(self, model_dir):
        torch.save(self.agent.Qnet.state_dict(), os.path.join(model_dir, "DQN.pt"))

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None):
        policy = DQN(dim_info, is_continue, 0, 0, device=torch.device("cpu"), trick=trick, gamma=0.99)
        policy.agent.Qnet.load_state_dict(torch.load(os.path.join(model_dir, "DQN.pt")))
        return policy
