"""Oracle building blocks: linear layers, MLP forward/backward, Adam, grad-norm clip, soft update.

Test infrastructure (see oracle/__init__.py).  Parameters are dicts keyed like the
reference's `state_dict()` ("l1.weight" [out,in], "l1.bias" [out]).
"""
import math

import numpy as np

F32 = np.float32


def f32(x):
    return np.asarray(x, dtype=F32)


def linear(x, p, name):
    """nn.Linear forward: x @ W^T + b (DQN_file/DQN.py:40-45)."""
    return x @ p[name + ".weight"].T + p[name + ".bias"]


def act_fwd(z, kind):
    if kind == "relu":
        return np.maximum(z, F32(0))
    if kind == "tanh":
        return np.tanh(z)
    if kind is None:
        return z
    raise ValueError(kind)


def act_bwd(h, dh, kind):
    """Backward through the activation given its OUTPUT h."""
    if kind == "relu":
        return dh * (h > 0)
    if kind == "tanh":
        return dh * (F32(1) - h * h)
    if kind is None:
        return dh
    raise ValueError(kind)


class MLP:
    """Sequential Linear layers; `hidden_act` after every layer but the last, `out_act` after
    the last.  Covers MLP (DQN.py:32-45: l1,l2), Actor (TD3.py:52-64: l1,l2,l3 + tanh),
    Critic (DDPG_simple.py:58-74: l1,l2,l3) and one half of the twin critics."""

    def __init__(self, names, hidden_act="relu", out_act=None):
        self.names = list(names)
        self.hidden_act = hidden_act
        self.out_act = out_act

    def forward(self, p, x):
        acts = [x]
        h = x
        for i, n in enumerate(self.names):
            z = linear(h, p, n)
            h = act_fwd(z, self.hidden_act if i < len(self.names) - 1 else self.out_act)
            acts.append(h)
        return h, acts

    def backward(self, p, acts, dy, need_dx=True):
        """Returns (dx, grads).  dy is d(loss)/d(output after out_act)."""
        grads = {}
        d = dy
        L = len(self.names)
        for i in reversed(range(L)):
            n = self.names[i]
            d = act_bwd(acts[i + 1], d, self.hidden_act if i < L - 1 else self.out_act)
            grads[n + ".weight"] = d.T @ acts[i]
            grads[n + ".bias"] = d.sum(axis=0)
            if i > 0 or need_dx:
                d = d @ p[n + ".weight"]
        return (d if need_dx else None), grads


class CautiousAdamW:
    """PPO_file/c_adamw.py:80-127 (the AdamW PPO.py:121 builds over actor + critic parameters): eps 1e-6, weight_decay 0,
    bias correction folded into the step size, and the "cautious" mask (exp_avg*grad > 0) rescaled by its per-tensor mean."""

    def __init__(self, params, lr, eps=1e-6, betas=(0.9, 0.999)):
        self.lr, self.eps = float(lr), float(eps)
        self.b1, self.b2 = betas
        self.t = 0
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = self.lr * math.sqrt(bc2) / bc1
        for k in params:
            g = grads[k]
            m, v = self.m[k], self.v[k]
            m *= F32(self.b1)
            m += F32(1.0 - self.b1) * g
            v *= F32(self.b2)
            v += F32(1.0 - self.b2) * g * g
            denom = np.sqrt(v) + F32(self.eps)
            mask = (m * g > 0).astype(F32)
            mask /= max(F32(mask.mean(dtype=F32)), F32(1e-3))
            params[k] += F32(-step_size) * ((m * mask) / denom)


class Adam:
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay, no amsgrad), the
    single-tensor CPU implementation's operation order:
        m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2)
        bc1 = 1-b1^t; bc2 = 1-b2^t; denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom
    Call sites: DQN.py:54,57-59; TD3.py:133-147; SAC.py:136-151,158."""

    def __init__(self, params, lr, eps=1e-8, weight_decay=0.0, betas=(0.9, 0.999)):
        self.lr, self.eps, self.wd = float(lr), float(eps), float(weight_decay)
        self.b1, self.b2 = betas
        self.t = 0
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for k in params:
            g = grads[k]
            if self.wd != 0.0:           # L2-in-grad form (DDPG.py:131-134)
                g = g + F32(self.wd) * params[k]
            m, v = self.m[k], self.v[k]
            m += (g - m) * F32(1.0 - self.b1)
            v *= F32(self.b2)
            v += F32(1.0 - self.b2) * g * g
            denom = np.sqrt(v) / F32(bc2_sqrt) + F32(self.eps)
            params[k] -= F32(step_size) * (m / denom)


def clip_grad_norm(grads, max_norm=0.5):
    """torch.nn.utils.clip_grad_norm_(params, 0.5) (TD3.py:140,146): L2 norm of the per-tensor
    L2 norms, coef = max_norm/(total+1e-6) clamped to 1, grads scaled in place."""
    norms = np.array([np.sqrt(np.sum(g.astype(F32) ** 2, dtype=F32)) for g in grads.values()], dtype=F32)
    total = np.sqrt(np.sum(norms ** 2, dtype=F32))
    coef = F32(max_norm) / (total + F32(1e-6))
    coef = min(coef, F32(1.0))
    for k in grads:
        grads[k] = grads[k] * F32(coef)
    return float(total)


def soft_update(target, source, tau):
    """theta_t <- theta_t*(1-tau) + theta*tau per parameter (DQN.py:120-128)."""
    for k in target:
        target[k] = target[k] * F32(1.0 - tau) + source[k] * F32(tau)


def mse(a, b):
    """F.mse_loss, mean reduction (DQN.py:116).  Returns (loss, d loss / d a)."""
    diff = a - b
    n = diff.size
    return F32(np.mean(diff * diff, dtype=F32)), diff * F32(2.0 / n)


def huber(delta):
    """The TD-loss option north_star names; the reference's only Huber is `huber_loss(e, d) = e**2/2 if |e| <= d else
    d*(|e| - d/2)`, mean-reduced (MAPPO_file/MAPPO.py:273-276, MAPPO_attention.py:389-397, d = 10).  Returns a function with
    mse()'s interface: (a, b) -> (loss, d loss / d a)."""
    d = F32(delta)

    def f(a, b):
        e = (a - b).astype(F32)
        n = e.size
        small = np.abs(e) <= d
        per = np.where(small, e * e * F32(0.5), d * (np.abs(e) - d * F32(0.5))).astype(F32)
        grad = np.where(small, e, d * np.sign(e)).astype(F32) * F32(1.0 / n)
        return F32(np.mean(per, dtype=F32)), grad
    return f


def copy_params(p):
    return {k: np.array(v, dtype=F32, copy=True) for k, v in p.items()}
