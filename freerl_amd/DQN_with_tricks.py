"""`DQN` of DQN_file/DQN_with_tricks.py (:160-308) with the tricks that live on the replay path: Double
(:263-265), PER (PER_Buffer / N_Step_PER_Buffer, :276-279), N_Step (:269-270), plus the Dueling head (:60-79) and NoisyLinear
heads (Noisy_net.py:17-76) and the Categorical / C51 distributional head (:82-158): all six tricks, i.e. the reference's
default configuration (:416), run as one fused update.

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

ATOMS, V_MIN, V_MAX = 51, -100.0, 100.0       # Categorical's defaults (DQN_with_tricks.py:88)
SIGMA_INIT = 0.05            # NoisyLinear's default (Noisy_net.py:18)


def _scale_noise(size):
    """NoisyLinear.scale_noise (Noisy_net.py:72-76) on torch's global generator."""
    x = torch.randn(size)
    return (x.sign() * torch.sqrt(abs(x))).numpy()


class NoisyQNet(DeviceNet):
    """`agent.Qnet` with NoisyLinear heads: MLP.l2 (DQN_with_tricks.py:50-51) or Dueling's V and A (:68-70).  The engine keeps
    the heads' mu as its head layer [V ; A] and their sigma as a parameter-only shadow layer; the epsilon buffers of the
    state_dict are the last noise this object drew."""

    def __init__(self, engine, hidden, obs_dim, action_dim, dueling, kind=N.PARAM_ONLINE, per_out=1):
        rows = ((1 + action_dim) if dueling else action_dim) * per_out
        super().__init__(engine, 0, [("l1", hidden, obs_dim), ("head", rows, hidden), ("sigma", rows, hidden)], kind=kind,
                         act_mode=N.ACT_RAW)
        self._nA, self._H, self._dueling = action_dim, hidden, dueling
        self.heads = [("V", 0, per_out), ("A", per_out, rows)] if dueling else [("l2", 0, rows)]
        self.eps = {h: (np.zeros(hidden, np.float32), np.zeros(b - a, np.float32)) for h, a, b in self.heads}
        self.is_train = True

    def keys(self):
        ks = ["l1.weight", "l1.bias"]
        for h, _, _ in self.heads:
            ks += [h + s for s in (".weight_mu", ".weight_sigma", ".bias_mu", ".bias_sigma", ".weight_epsilon", ".bias_epsilon")]
        return ks

    def _split(self, flat):
        parts = super()._split(flat)
        out = {"l1.weight": parts["l1.weight"], "l1.bias": parts["l1.bias"]}
        for h, a, b in self.heads:
            out[h + ".weight_mu"], out[h + ".bias_mu"] = parts["head.weight"][a:b], parts["head.bias"][a:b]
            out[h + ".weight_sigma"], out[h + ".bias_sigma"] = parts["sigma.weight"][a:b], parts["sigma.bias"][a:b]
            ei, eo = self.eps[h]
            out[h + ".weight_epsilon"], out[h + ".bias_epsilon"] = np.outer(eo, ei).astype(np.float32), eo.copy()
        return out

    def load_state_dict(self, sd, strict=True):
        if strict and set(sd.keys()) != set(self.keys()):
            raise RuntimeError("Error(s) in loading state_dict: expected keys %s, got %s" % (self.keys(), list(sd.keys())))
        t = lambda k: np.asarray(torch.as_tensor(sd[k]).detach().cpu().numpy(), dtype=np.float32)
        cat = lambda sfx: np.concatenate([t(h + sfx).reshape(b - a, -1) for h, a, b in self.heads]).reshape(-1)
        flat = [t("l1.weight").reshape(-1), t("l1.bias").reshape(-1), cat(".weight_mu"), cat(".bias_mu"), cat(".weight_sigma"),
                cat(".bias_sigma")]
        self._e.set_params(self._net, np.concatenate(flat), self._kind, self._learner)

    def draw(self):
        """One forward's noise in the reference's order (V then A; eps_in then eps_out each), as the engine's flat array."""
        flat = []
        for h, a, b in self.heads:
            ei, eo = _scale_noise(self._H), _scale_noise(b - a)
            self.eps[h] = (ei, eo)
            flat += [ei, eo]
        return np.concatenate(flat).astype(np.float32)

    def __call__(self, obs):
        raise NotImplementedError("use policy.select_action / the engine's act for a noisy forward")


class NoisyAgent:
    def __init__(self, engine, obs_dim, action_dim, Qnet_lr, hidden, dueling, per_out=1):
        # torch RNG order: l1 (nn.Linear default), then per NoisyLinear: weight_mu.uniform_, bias_mu.uniform_, reset_noise's two
        # randn, and torch.manual_seed(100) (Noisy_net.py:30-33) — the seed reset is part of the reference's behaviour
        from ._core import linear_init
        l1w, l1b = linear_init(hidden, obs_dim)
        heads = [("V", per_out), ("A", action_dim * per_out)] if dueling else [("l2", action_dim * per_out)]
        mu_w, mu_b, sg_w, sg_b, eps = [], [], [], [], {}
        for name, rows in heads:
            r = 1.0 / np.sqrt(hidden)
            mu_w.append(torch.empty(rows, hidden).uniform_(-r, r).numpy())
            mu_b.append(torch.empty(rows).uniform_(-r, r).numpy())
            sg_w.append(np.full((rows, hidden), SIGMA_INIT / np.sqrt(hidden), np.float32))
            sg_b.append(np.full(rows, SIGMA_INIT / np.sqrt(rows), np.float32))
            eps[name] = (_scale_noise(hidden), _scale_noise(rows))
            torch.manual_seed(100)
        flat = np.concatenate([l1w.reshape(-1), l1b] + [np.concatenate(mu_w).reshape(-1), np.concatenate(mu_b)] +
                              [np.concatenate(sg_w).reshape(-1), np.concatenate(sg_b)]).astype(np.float32)
        engine.set_params(0, flat, N.PARAM_ONLINE)
        engine.set_params(0, flat, N.PARAM_TARGET)
        self.Qnet = NoisyQNet(engine, hidden, obs_dim, action_dim, dueling, per_out=per_out)
        self.Qnet_target = NoisyQNet(engine, hidden, obs_dim, action_dim, dueling, kind=N.PARAM_TARGET, per_out=per_out)
        self.Qnet.eps = dict(eps)
        self.Qnet_target.eps = dict(eps)              # deepcopy copies the buffers too
        self.Qnet_optimizer = OptimizerView(engine, 0, Qnet_lr)


class DuelingNet(DeviceNet):
    """`agent.Qnet` of Dueling (DQN_with_tricks.py:60-79): state_dict keys l1 / V / A; the engine keeps V and A as one
    (1 + n_actions)-wide head [V ; A].  Calling it returns Q = V + A - mean(A)."""

    def __init__(self, engine, hidden, obs_dim, action_dim, kind=N.PARAM_ONLINE, per_out=1):
        super().__init__(engine, 0, [("l1", hidden, obs_dim), ("head", (1 + action_dim) * per_out, hidden)], kind=kind,
                         act_mode=N.ACT_RAW)
        self._nA, self._v = action_dim, per_out            # per_out = atoms for the Categorical head (:96-97)

    def keys(self):
        return ["l1.weight", "l1.bias", "V.weight", "V.bias", "A.weight", "A.bias"]

    def _split(self, flat):
        parts = super()._split(flat)
        w, b = parts.pop("head.weight"), parts.pop("head.bias")
        v = self._v
        parts.update({"V.weight": w[:v], "V.bias": b[:v], "A.weight": w[v:], "A.bias": b[v:]})
        return parts

    def load_state_dict(self, sd, strict=True):
        if strict and set(sd.keys()) != set(self.keys()):
            raise RuntimeError("Error(s) in loading state_dict: expected keys %s, got %s" % (self.keys(), list(sd.keys())))
        t = lambda k: np.asarray(torch.as_tensor(sd[k]).detach().cpu().numpy(), dtype=np.float32).reshape(-1)
        flat = [t(k) for k in ("l1.weight", "l1.bias", "V.weight", "A.weight", "V.bias", "A.bias")]
        self._e.set_params(self._net, np.concatenate(flat), self._kind, self._learner)

    def __call__(self, obs):
        h = super().__call__(obs)
        if self._v > 1:
            raise NotImplementedError("the Categorical net returns (action, dist); use select_action")
        return h[:, :1] + h[:, 1:] - h[:, 1:].mean(dim=1, keepdim=True)


class DuelingAgent:
    def __init__(self, engine, obs_dim, action_dim, Qnet_lr, hidden, per_out=1):
        # torch RNG order of Dueling.__init__ / Categorical.__init__: l1, V, A (:67-74, :90-97); the engine's flat order is
        # l1, [V.w ; A.w], [V.b ; A.b]
        rv, ra = per_out, action_dim * per_out
        f = init_layers([("l1", hidden, obs_dim), ("V", rv, hidden), ("A", ra, hidden)])
        o = hidden * obs_dim + hidden
        vw, vb = f[o:o + rv * hidden], f[o + rv * hidden:o + rv * hidden + rv]
        o2 = o + rv * hidden + rv
        aw, ab = f[o2:o2 + ra * hidden], f[o2 + ra * hidden:]
        flat = np.concatenate([f[:o], vw, aw, vb, ab])
        engine.set_params(0, flat, N.PARAM_ONLINE)
        engine.set_params(0, flat, N.PARAM_TARGET)
        self.Qnet = DuelingNet(engine, hidden, obs_dim, action_dim, per_out=per_out)
        self.Qnet_target = DuelingNet(engine, hidden, obs_dim, action_dim, kind=N.PARAM_TARGET, per_out=per_out)
        self.Qnet_optimizer = OptimizerView(engine, 0, Qnet_lr)


class PlainC51Agent:
    """Categorical with a plain nn.Linear head l2: hidden -> action_dim * atoms (:98-99)."""

    def __init__(self, engine, obs_dim, action_dim, Qnet_lr, hidden, atoms):
        layers = [("l1", hidden, obs_dim), ("l2", action_dim * atoms, hidden)]
        flat = init_layers(layers)
        engine.set_params(0, flat, N.PARAM_ONLINE)
        engine.set_params(0, flat, N.PARAM_TARGET)
        self.Qnet = DeviceNet(engine, 0, layers)
        self.Qnet_target = DeviceNet(engine, 0, layers, kind=N.PARAM_TARGET)
        self.Qnet_optimizer = OptimizerView(engine, 0, Qnet_lr)


class DQN:
    def __init__(self, dim_info, is_continue, Qnet_lr, buffer_size, device, trick=None, gamma=None, batch_size=None, *,
                 hidden=128, batch_max=1024, seed=0):
        obs_dim, action_dim = dim_info
        if is_continue:
            raise ValueError("DQN is not suitable for continuous action spaces (DQN_with_tricks.py:207-210)")
        cat = bool(trick["Categorical"])
        if cat and trick["Noisy"] and not trick["Dueling"]:
            # the reference builds NoisyLinear(hidden, action_dim) there (:94-95) and fails reshaping it to [B, nA, atoms] (:122)
            raise RuntimeError("shape '[-1, %d, %d]' is invalid for input of size %d" % (action_dim, ATOMS, action_dim))
        hip_id, self.device = resolve_device(device)
        self._e = Engine(N.ALGO_DQN, obs_dim, action_dim, max(int(buffer_size), 1), discrete=True, hidden=hidden,
                         batch_max=batch_max, device_id=hip_id, seed=seed, dueling=bool(trick["Dueling"]), noisy=bool(trick["Noisy"]),
                         c51=(ATOMS, V_MIN, V_MAX) if cat else None)
        per_out = ATOMS if cat else 1
        if trick["Noisy"]:
            self.agent = NoisyAgent(self._e, obs_dim, action_dim, Qnet_lr, hidden, bool(trick["Dueling"]), per_out=per_out)
        elif trick["Dueling"]:
            self.agent = DuelingAgent(self._e, obs_dim, action_dim, Qnet_lr, hidden, per_out=per_out)
        elif cat:
            self.agent = PlainC51Agent(self._e, obs_dim, action_dim, Qnet_lr, hidden, ATOMS)
        else:
            self.agent = Agent(self._e, obs_dim, action_dim, Qnet_lr, hidden)
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
        use = False
        if self.trick["Noisy"] and self.agent.Qnet.is_train:      # every forward of a training NoisyLinear redraws its noise
            self._e.noisy_resample(self.agent.Qnet.draw())
            use = 2
        a = self._e.act(0, N.ACT_ARGMAX, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), use_target=use)
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
            self.buffer._device_only = True           # the rows and weights stay on the device for the update below
            try:
                self.buffer.sample(batch)
            finally:
                self.buffer._device_only = False
        else:
            idx = draw_indices(len(self.buffer), batch_size)
        noisy_eps = None
        if self.trick["Noisy"]:        # the reference's forwards in program order: [Qnet(s') if Double,] Qnet_target(s'), Qnet(s)
            draws = ([self.agent.Qnet.draw()] if self.trick["Double"] else []) + [self.agent.Qnet_target.draw(), self.agent.Qnet.draw()]
            noisy_eps = np.stack(draws)[None]
        st = self._e.learn(batch, gamma=gamma, tau=tau, critic_lr=self.agent.Qnet_optimizer.lr, clip_norm=0.0, idx=idx,
                           noisy_eps=noisy_eps, double_dqn=bool(self.trick["Double"]), per=1 if self.trick["PER"] else 0,
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
        if trick["Noisy"]:
            policy.agent.Qnet.is_train = False                                     # :304-307
        return policy
