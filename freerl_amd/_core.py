"""Shared plumbing of the reference-shaped classes: device resolution, the `DeviceNet` /
`OptimizerView` stand-ins for `nn.Module` / `torch.optim.Adam` attributes that callers touch
(`policy.agent.Qnet.load_state_dict`, `optimizer.param_groups[0]['lr']`), parameter init with the
reference's torch-RNG draw order, and record packing.

PyTorch is used here only as plumbing: default-init draws (RNG parity with the reference),
`state_dict` containers for checkpoints, and device tensors handed back by `Buffer.sample`.
All arithmetic of the hot path happens in libfreerl_hip.so.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import _native as N
from .engine import Engine  # noqa: F401  (re-export)

F32 = np.float32


def resolve_device(device):
    """-> (hip_device_id, output torch.device).  The reference takes `device` = cpu|cuda and
    silently falls back to CPU without a GPU (DQN.py:279); this engine has no CPU path: the
    compute device is always a HIP GPU, `device` only says where returned tensors live."""
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if N.device_count() == 0:
        raise N.FrlError("no HIP device visible: freerl_amd has no CPU fallback")
    hip_id = dev.index if (dev.type == "cuda" and dev.index is not None) else 0
    return hip_id, dev


class OptimizerView:
    """What callers read/write on `agent.*_optimizer`: `param_groups[i]['lr']` (PPO.lr_decay,
    PPO_with_tricks.py:357-363) and `state_dict()` for inspection."""

    def __init__(self, engine, net, lr, eps=1e-8, weight_decay=0.0):
        self._e, self._net = engine, net
        self.param_groups = [dict(lr=float(lr), betas=(0.9, 0.999), eps=float(eps), weight_decay=float(weight_decay))]

    @property
    def lr(self):
        return float(self.param_groups[0]["lr"])

    def zero_grad(self):       # gradients live in the fused kernels; nothing to clear
        pass

    def step(self):
        raise NotImplementedError("the Adam step is fused into learn() on the GPU")

    def state_dict(self):
        return dict(step=self._e.opt_step(self._net), exp_avg=self._e.get_params(self._net, N.PARAM_ADAM_M),
                    exp_avg_sq=self._e.get_params(self._net, N.PARAM_ADAM_V), param_groups=self.param_groups)


class DeviceNet:
    """Stands where the reference has an `nn.Module` (`agent.Qnet`, `agent.actor`, ...): the
    parameters live in the engine; `state_dict()/load_state_dict()` keep the reference's
    checkpoint layout (DQN.py:131-138: same keys, shapes, fp32 CPU tensors)."""

    def __init__(self, engine, net, layers, extra=None, kind=N.PARAM_ONLINE, act_mode=N.ACT_RAW, learner=0):
        self._e, self._net, self._kind, self._learner = engine, net, kind, learner
        self._layers = list(layers)            # [(name, out, in)]
        self._extra = extra                    # (name, shape) of log_std or None
        self._act_mode = act_mode

    # ---- state_dict order of the reference: own Parameters (log_std) first, then sub-modules
    def keys(self):
        ks = [self._extra[0]] if self._extra else []
        for n, _, _ in self._layers:
            ks += [n + ".weight", n + ".bias"]
        return ks

    def _split(self, flat):
        out, o = {}, 0
        for n, od, idim in self._layers:
            out[n + ".weight"] = flat[o:o + od * idim].reshape(od, idim); o += od * idim
            out[n + ".bias"] = flat[o:o + od]; o += od
        if self._extra:
            shp = self._extra[1]
            sz = int(np.prod(shp))
            out[self._extra[0]] = flat[o:o + sz].reshape(shp); o += sz
        assert o == flat.size
        return out

    def state_dict(self):
        parts = self._split(self._e.get_params(self._net, self._kind, self._learner))
        return OrderedDict((k, torch.from_numpy(np.array(parts[k], dtype=F32, copy=True))) for k in self.keys())

    def load_state_dict(self, sd, strict=True):
        want = self.keys()
        if strict and set(sd.keys()) != set(want):
            raise RuntimeError("Error(s) in loading state_dict: expected keys %s, got %s" % (want, list(sd.keys())))
        flat = []
        for n, od, idim in self._layers:
            w = np.asarray(torch.as_tensor(sd[n + ".weight"]).detach().cpu().numpy(), dtype=F32)
            b = np.asarray(torch.as_tensor(sd[n + ".bias"]).detach().cpu().numpy(), dtype=F32)
            if w.shape != (od, idim) or b.shape != (od,):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (n, w.shape, (od, idim)))
            flat += [w.reshape(-1), b.reshape(-1)]
        if self._extra:
            flat.append(np.asarray(torch.as_tensor(sd[self._extra[0]]).detach().cpu().numpy(), dtype=F32).reshape(-1))
        self._e.set_params(self._net, np.concatenate(flat), self._kind, self._learner)

    def parameters(self):
        return iter(self.state_dict().values())

    def named_parameters(self):
        return iter(self.state_dict().items())

    def __call__(self, *inputs):
        """Forward on the GPU (head 0): torch/NumPy [rows, in] -> torch tensor [rows, out] on CPU."""
        x = np.concatenate([np.asarray(torch.as_tensor(t).detach().cpu().numpy(), dtype=F32) for t in inputs], axis=1)
        out_dim = self._layers[-1][1] if len(self._layers) <= 3 else self._layers[2][1]
        y = self._e.act(self._net, self._act_mode, x[None] if self._e.P == 1 else x, out_dim=out_dim,
                        use_target=(self._kind == N.PARAM_TARGET))
        return torch.from_numpy(y[0])

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self


def linear_init(out_dim, in_dim):
    """One `nn.Linear(in, out)` default init: consumes torch's global CPU generator exactly like
    the reference's module constructors do (kaiming_uniform a=sqrt(5) weight, then bias)."""
    lin = torch.nn.Linear(in_dim, out_dim)
    return lin.weight.detach().numpy().astype(F32).copy(), lin.bias.detach().numpy().astype(F32).copy()


def init_layers(layers, orthogonal=None):
    """Draw default inits for [(name, out, in)] in order; `orthogonal` = list of gains applied
    afterwards in the same order (PPO_with_tricks.py:71-77,92-95: orthogonal_ weight, zero bias)."""
    ws = [linear_init(o, i) for _, o, i in layers]
    if orthogonal is not None:
        new = []
        for (w, b), gain in zip(ws, orthogonal):
            t = torch.empty(w.shape)
            torch.nn.init.orthogonal_(t, gain=gain)
            new.append((t.numpy().astype(F32).copy(), np.zeros_like(b)))
        ws = new
    return np.concatenate([np.concatenate([w.reshape(-1), b.reshape(-1)]) for w, b in ws])


def init_layers_ddpg(layers, final=3e-3):
    """DDPG.py net_init (DDPG.py:57-68,78-86): default nn.Linear draws for every layer first, then
    `other_net_init` on all but the last layer (U(+-1/sqrt(weight.size(0))) — the reference takes
    size(0), i.e. out_features) and `final_net_init` U(+-3e-3) on the last, weight then bias each."""
    shapes = [(o, i) for _, o, i in layers]
    for o, i in shapes:
        linear_init(o, i)                                   # consumed, then overwritten like the reference does
    parts = []
    for j, (o, i) in enumerate(shapes):
        lim = final if j == len(shapes) - 1 else 1.0 / (o ** 0.5)
        w = torch.empty(o, i).uniform_(-lim, lim).numpy().astype(F32)
        b = torch.empty(o).uniform_(-lim, lim).numpy().astype(F32)
        parts += [w.reshape(-1), b.reshape(-1)]
    return np.concatenate(parts)


class BatchObsNormView:
    """`policy.batch_size_obs_norm.running_ms.{mean,std}` (SAC.py:586, DDPG.py:160): reads the engine's
    device-side Normalization_batch_size statistics."""

    def __init__(self, engine, agent=0):
        self._e, self._agent = engine, agent
        self.running_ms = self

    @property
    def n(self):
        return self._e.obsnorm_stats(agent=self._agent)["n"]

    @property
    def mean(self):
        return torch.from_numpy(self._e.obsnorm_stats(agent=self._agent)["mean"].reshape(1, -1))

    @property
    def std(self):
        return torch.from_numpy(self._e.obsnorm_stats(agent=self._agent)["std"].reshape(1, -1))


def as_f32(x, n):
    a = np.asarray(x, dtype=F32).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    return a


# rng="auto" (the classes' default): the reference's index draw is np.random.choice(len(buffer), B, replace=False), a full
# permutation of the buffer per learn() — 0.3 ms at 1e4 rows, 30 ms at 1e6, against ~0.1 ms for the fused
# update itself.  Below AUTO_DEVICE_MIN_ROWS the reference's legacy NumPy / torch streams are consumed exactly like the
# reference does (bit-parity with a seeded reference run is only checkable at such sizes); from there on indices and
# noise come from the engine's Philox generator (uniform over subsets, validated statistically).  rng="host" keeps the
# reference's streams at every size, rng="device" never touches them.
AUTO_DEVICE_MIN_ROWS = 32768


def host_draw(mode, total_size, batch_size):
    """True: draw indices / noise from the reference's host streams; False: leave them to the device."""
    if mode == "host":
        return True
    if total_size < 2 * min(total_size, batch_size):      # the device's rejection sampler needs len(buffer) >= 2*batch
        return True
    if mode == "device":
        return False
    if mode != "auto":
        raise ValueError("rng must be 'auto', 'host' or 'device', got %r" % (mode,))
    return total_size < AUTO_DEVICE_MIN_ROWS


def draw_indices(total_size, batch_size):
    """`<ALGO>.sample` (DQN.py:94-97): batch = min(size, batch); the legacy global NumPy stream."""
    batch = min(total_size, batch_size)
    return np.random.choice(total_size, batch, replace=False)
