"""Oracle off-policy learners: DQN, DDPG, TD3, SAC, MADDPG `learn()` restated in NumPy fp32
with hand-written backward passes.  Test infrastructure (see oracle/__init__.py).

Every learner takes the sample indices (and the Gaussian noise the reference draws from
torch's generator) as explicit inputs, `learn_with(...)`, so the oracle, the reference and
the HIP engine can be fed identical draws; `learn(...)` keeps the reference's signature and
draws `np.random.choice(size, B, replace=False)` like `<ALGO>.sample` does.
"""
import math

import numpy as np

from . import nn
from .buffer import Buffer
from .nn import F32, MLP, Adam
from .normalization import NormalizationBatch

LOG2 = math.log(2.0)
LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


def _choice(size, batch):
    """<ALGO>.sample: batch=min(size,batch); np.random.choice(size,batch,replace=False)
    (DQN.py:94-97, TD3.py:180-183, SAC.py:210-213)."""
    batch = min(size, batch)
    return np.random.choice(size, batch, replace=False)


# ---------------------------------------------------------------------------------------- DQN
class DuelingNet:
    """Dueling (DQN_file/DQN_with_tricks.py:60-79): a = relu(l1 s); Q = V(a) + A(a) - mean_j A_j(a).  Same forward /
    backward interface as nn.MLP."""

    def forward(self, p, x):
        h = np.maximum(x @ p["l1.weight"].T + p["l1.bias"], 0).astype(F32)
        V = h @ p["V.weight"].T + p["V.bias"]
        A = h @ p["A.weight"].T + p["A.bias"]
        q = ((V + A) - A.mean(axis=1, keepdims=True, dtype=F32)).astype(F32)
        return q, [x, h]

    def backward(self, p, acts, dq, need_dx=False):
        x, h = acts
        dV = dq.sum(axis=1, keepdims=True).astype(F32)
        dA = (dq - dq.mean(axis=1, keepdims=True, dtype=F32)).astype(F32)
        g = {"V.weight": dV.T @ h, "V.bias": dV.sum(axis=0), "A.weight": dA.T @ h, "A.bias": dA.sum(axis=0)}
        dh = (dV @ p["V.weight"] + dA @ p["A.weight"]) * (h > 0)
        g["l1.weight"] = dh.T @ x
        g["l1.bias"] = dh.sum(axis=0)
        return None, {k: g[k].astype(F32) for k in p}


class NoisyNet:
    """MLP / Dueling of DQN_with_tricks.py (:40-79) with NoisyLinear heads (Noisy_net.py:17-76): head weight = weight_mu +
    weight_sigma * (eps_out (x) eps_in), bias = bias_mu + bias_sigma * eps_out, fresh eps at every forward.  `eps` = {head
    name: (eps_in, eps_out)} for heads "l2" or "V" and "A"; None = is_train False (mu only)."""

    def __init__(self, dueling):
        self.heads = ["V", "A"] if dueling else ["l2"]
        self.dueling = dueling

    def _eff(self, p, name, eps):
        w, b = p[name + ".weight_mu"], p[name + ".bias_mu"]
        if eps is None:
            return w, b
        ei, eo = nn.f32(eps[name][0]), nn.f32(eps[name][1])
        return (w + p[name + ".weight_sigma"] * np.outer(eo, ei)).astype(F32), (b + p[name + ".bias_sigma"] * eo).astype(F32)

    def forward(self, p, x, eps=None):
        h = np.maximum(x @ p["l1.weight"].T + p["l1.bias"], 0).astype(F32)
        outs, effs = {}, {}
        for name in self.heads:
            w, b = self._eff(p, name, eps)
            effs[name] = w
            outs[name] = (h @ w.T + b).astype(F32)
        q = ((outs["V"] + outs["A"]) - outs["A"].mean(axis=1, keepdims=True, dtype=F32)).astype(F32) if self.dueling else outs["l2"]
        return q, [x, h, effs, eps]

    def backward(self, p, acts, dq, need_dx=False):
        x, h, effs, eps = acts
        d = {"l2": dq} if not self.dueling else {"V": dq.sum(axis=1, keepdims=True).astype(F32),
                                                 "A": (dq - dq.mean(axis=1, keepdims=True, dtype=F32)).astype(F32)}
        g, dh = {}, 0
        for name in self.heads:
            dW, db = d[name].T @ h, d[name].sum(axis=0)
            g[name + ".weight_mu"], g[name + ".bias_mu"] = dW, db
            ei, eo = nn.f32(eps[name][0]), nn.f32(eps[name][1])
            g[name + ".weight_sigma"], g[name + ".bias_sigma"] = dW * np.outer(eo, ei), db * eo
            dh = dh + d[name] @ effs[name]
        dh = dh * (h > 0)
        g["l1.weight"], g["l1.bias"] = dh.T @ x, dh.sum(axis=0)
        return None, {k: g[k].astype(F32) for k in p}


class C51Net:
    """Categorical (DQN_with_tricks.py:82-132): logits [B, nA, atoms] from l2 (plain) or V + A - mean_a A (Dueling, V: atoms
    rows, A: nA*atoms rows); dist = softmax over atoms; q = sum_i z_i dist_i.  Heads are nn.Linear or NoisyLinear."""

    def __init__(self, n_actions, atoms, vmin, vmax, dueling, noisy):
        self.nA, self.atoms, self.vmin, self.vmax = n_actions, atoms, vmin, vmax
        self.dueling, self.noisy = dueling, noisy
        self.heads = ["V", "A"] if dueling else ["l2"]
        self.z = np.linspace(vmin, vmax, atoms, dtype=np.float64).astype(F32)       # torch.linspace in float32
        self.delta_z = (vmax - vmin) / (atoms - 1)

    def _eff(self, p, name, eps):
        if not self.noisy:
            return p[name + ".weight"], p[name + ".bias"]
        w, b = p[name + ".weight_mu"], p[name + ".bias_mu"]
        if eps is None:
            return w, b
        ei, eo = nn.f32(eps[name][0]), nn.f32(eps[name][1])
        return (w + p[name + ".weight_sigma"] * np.outer(eo, ei)).astype(F32), (b + p[name + ".bias_sigma"] * eo).astype(F32)

    def forward(self, p, x, eps=None):
        h = np.maximum(x @ p["l1.weight"].T + p["l1.bias"], 0).astype(F32)
        outs, effs = {}, {}
        for name in self.heads:
            w, b = self._eff(p, name, eps)
            effs[name] = w
            outs[name] = (h @ w.T + b).astype(F32)
        B = x.shape[0]
        if self.dueling:
            V = outs["V"].reshape(B, 1, self.atoms)
            A = outs["A"].reshape(B, self.nA, self.atoms)
            logits = ((V + A) - A.mean(axis=1, keepdims=True, dtype=F32)).astype(F32)
        else:
            logits = outs["l2"].reshape(B, self.nA, self.atoms)
        e = np.exp(logits - logits.max(axis=2, keepdims=True))
        dist = (e / e.sum(axis=2, keepdims=True)).astype(F32)
        q = (dist * self.z).sum(axis=2).astype(F32)
        return dist, q, [x, h, effs, eps]

    def backward(self, p, acts, dlogits):
        x, h, effs, eps = acts
        B = x.shape[0]
        if self.dueling:
            d = {"V": dlogits.sum(axis=1).astype(F32),
                 "A": (dlogits - dlogits.mean(axis=1, keepdims=True, dtype=F32)).reshape(B, -1).astype(F32)}
        else:
            d = {"l2": dlogits.reshape(B, -1)}
        g, dh = {}, 0
        for name in self.heads:
            dW, db = d[name].T @ h, d[name].sum(axis=0)
            if self.noisy:
                ei, eo = nn.f32(eps[name][0]), nn.f32(eps[name][1])
                g[name + ".weight_mu"], g[name + ".bias_mu"] = dW, db
                g[name + ".weight_sigma"], g[name + ".bias_sigma"] = dW * np.outer(eo, ei), db * eo
            else:
                g[name + ".weight"], g[name + ".bias"] = dW, db
            dh = dh + d[name] @ effs[name]
        dh = dh * (h > 0)
        g["l1.weight"], g["l1.bias"] = dh.T @ x, dh.sum(axis=0)
        return {k: g[k].astype(F32) for k in p}


class C51DQN:
    """DQN_with_tricks.learn, Categorical branch (:248-260) with projection_dist (:134-158); buffer/PER handled by the caller."""

    def __init__(self, params, obs_dim, n_actions, lr, capacity, atoms=51, vmin=-100, vmax=100, dueling=False, noisy=False):
        self.q = nn.copy_params(params)
        self.q_t = nn.copy_params(params)
        self.net = C51Net(n_actions, atoms, vmin, vmax, dueling, noisy)
        self.opt = Adam(self.q, lr)
        self.buffer = Buffer(capacity, obs_dim, 1)
        self.losses = []

    def add(self, *a):
        self.buffer.add(*a)

    def select_action(self, obs, eps=None):
        return int(np.argmax(self.net.forward(self.q, nn.f32(obs).reshape(1, -1), eps)[1], axis=1)[0])

    def learn_with(self, idx, gamma, tau, double=False, is_weight=None, noisy_eps=(None, None, None)):
        net = self.net
        obs, act, rew, nobs, done = self.buffer.sample(idx)
        B, atoms = obs.shape[0], net.atoms
        if double:
            next_a = np.argmax(net.forward(self.q, nobs, noisy_eps[0])[1], axis=1)
            dist_t, _, _ = net.forward(self.q_t, nobs, noisy_eps[1])
        else:
            dist_t, q_t, _ = net.forward(self.q_t, nobs, noisy_eps[1])
            next_a = np.argmax(q_t, axis=1)
        next_dist = dist_t[np.arange(B), next_a]                                     # [B, atoms]
        t_z = np.clip(rew + F32(gamma) * net.z * (F32(1) - done), net.vmin, net.vmax).astype(F32)
        b = ((t_z - F32(net.vmin)) / F32(net.delta_z)).astype(F32)
        l, u = np.floor(b).astype(np.int64), np.ceil(b).astype(np.int64)
        dl = ((u + (l == u) - b) * next_dist).astype(F32)
        du = ((b - l) * next_dist).astype(F32)
        m = np.zeros((B, atoms), dtype=F32)
        for r in range(B):                               # index_add_: sequential in source order, lower bins first
            for i in range(atoms):
                m[r, l[r, i]] += dl[r, i]
            for i in range(atoms):
                m[r, u[r, i]] += du[r, i]
        dist, _, acts = net.forward(self.q, obs, noisy_eps[2])
        a = act.astype(np.int64).reshape(-1)
        p = dist[np.arange(B), a]
        logp = np.log(np.clip(p, 1e-5, 1 - 1e-5)).astype(F32)
        w = np.ones((B, 1), F32) if is_weight is None else nn.f32(is_weight).reshape(-1, 1)
        loss = F32(np.mean(-(m * logp * w).sum(axis=1), dtype=F32))
        self.last_td = (m * logp).sum(axis=1).astype(F32)                            # `error` (:255)
        inside = (p > 1e-5) & (p < 1 - 1e-5)
        gp_ = np.where(inside, -(m * w / F32(B)) / p, F32(0)).astype(F32)            # d loss / d p
        dlog_a = (p * (gp_ - (gp_ * p).sum(axis=1, keepdims=True))).astype(F32)      # softmax backward
        dlogits = np.zeros((B, net.nA, atoms), dtype=F32)
        dlogits[np.arange(B), a] = dlog_a
        g = net.backward(self.q, acts, dlogits)
        self.opt.step(self.q, g)
        nn.soft_update(self.q_t, self.q, tau)
        self.losses.append(loss)
        return loss


class DQN:
    """DQN_file/DQN.py:62-128.  Q-net MLP obs->128->n_actions, target copy, Adam(lr).  dueling=True: DQN_with_tricks'
    Dueling net (params l1, V, A); noisy=True: NoisyLinear heads (learn_with takes the per-forward noise)."""

    def __init__(self, params, obs_dim, n_actions, lr, capacity, dueling=False, noisy=False):
        self.q = nn.copy_params(params)
        self.q_t = nn.copy_params(params)
        self.noisy = noisy
        self.net = NoisyNet(dueling) if noisy else (DuelingNet() if dueling else MLP(["l1", "l2"]))
        self.opt = Adam(self.q, lr)
        self.buffer = Buffer(capacity, obs_dim, 1)
        self.losses = []

    def q_values(self, obs):
        return self.net.forward(self.q, nn.f32(obs).reshape(-1, obs.shape[-1]))[0]

    def select_action(self, obs):                       # DQN.py:70-84: argmax_a Q(s,a)
        return int(np.argmax(self.q_values(nn.f32(obs).reshape(1, -1)), axis=1)[0])

    def add(self, *a):
        self.buffer.add(*a)

    def learn(self, batch_size, gamma, tau):
        return self.learn_with(_choice(len(self.buffer), batch_size), gamma, tau)

    def learn_with(self, idx, gamma, tau, double=False, is_weight=None, noisy_eps=None):
        """DQN.py:104-118; `double` / `is_weight`: DQN_with_tricks.py:263-265 / :276-279 (returns the TD errors too).
        noisy_eps = [eps of Qnet(next_obs) (Double only, else None), eps of Qnet_target(next_obs), eps of Qnet(obs)]."""
        obs, act, rew, nobs, done = self.buffer.sample(idx)
        B = obs.shape[0]
        fwd = (lambda p, x, j: self.net.forward(p, x, noisy_eps[j])) if self.noisy else (lambda p, x, j: self.net.forward(p, x))
        a_star = np.argmax(fwd(self.q, nobs, 0)[0], axis=1) if double else None       # program order of the reference
        qt = fwd(self.q_t, nobs, 1)[0]
        if double:
            next_q = qt[np.arange(B), a_star].reshape(-1, 1)
        else:
            next_q = qt.max(axis=1).reshape(-1, 1)
        y = rew + F32(gamma) * next_q * (F32(1) - done)
        q, acts = fwd(self.q, obs, 2)
        a = act.astype(np.int64).reshape(-1)
        cur = q[np.arange(B), a].reshape(-1, 1)
        if is_weight is None:
            loss, dcur = getattr(self, "td_loss", nn.mse)(cur, y)      # `td_loss = nn.huber(delta)`: the Huber option
        else:
            # DQN_with_tricks.py:277-278: `is_weight` is a 1-D [B] tensor and `td_error ** 2` is [B,1], so their product
            # broadcasts to [B,B] and `.mean()` = mean(w) * mean(td^2): every sample is weighted by the MEAN weight
            w = nn.f32(is_weight).reshape(1, -1)
            td = cur - y
            loss = F32(np.mean(w * (td * td), dtype=F32))
            dcur = (F32(2.0 / B) * F32(np.mean(w, dtype=F32)) * td).astype(F32)
        dq = np.zeros_like(q)
        dq[np.arange(B), a] = dcur.reshape(-1)
        _, g = self.net.backward(self.q, acts, dq, need_dx=False)
        if getattr(self, "clip", 0.0) > 0.0:            # not in DQN.py (Agent.update_Qnet, :56-59, has no clipping): the engine's
            nn.clip_grad_norm(g, self.clip)             # clip_norm argument, checked with torch's clip_grad_norm_ arithmetic
        self.opt.step(self.q, g)
        nn.soft_update(self.q_t, self.q, tau)           # DQN.py:120-128
        self.losses.append(loss)
        self.last_td = (cur - y).reshape(-1)
        return loss


# ---------------------------------------------------------------------------------------- critics
class QNet:
    """Critic on cat([o,a],1): l1,l2,l3 (DDPG_simple.py:58-74); twin adds l4,l5,l6 sharing the
    input (TD3.py:85-121, SAC.py:103-127)."""

    def __init__(self, twin):
        self.q1 = MLP(["l1", "l2", "l3"])
        self.q2 = MLP(["l4", "l5", "l6"]) if twin else None

    def forward(self, p, oa):
        out1 = self.q1.forward(p, oa)
        out2 = self.q2.forward(p, oa) if self.q2 else None
        return out1, out2


def _critic_step(qnet, p, opt, oa, y, clip=True, td_loss=nn.mse):
    """critic_loss = mse(Q1,y) [+ mse(Q2,y)]; zero_grad/backward/clip 0.5/Adam
    (TD3.py:210-213,142-147; DDPG_simple.py:146-149)."""
    (q1, a1), two = qnet.forward(p, oa)
    l1, d1 = td_loss(q1, y)
    _, g = qnet.q1.backward(p, a1, d1, need_dx=False)
    loss = l1
    if two is not None:
        q2, a2 = two
        l2, d2 = td_loss(q2, y)
        _, g2 = qnet.q2.backward(p, a2, d2, need_dx=False)
        g.update(g2)
        loss = F32(l1 + l2)
    g = {k: g[k] for k in p}          # parameter order of the module
    norm = nn.clip_grad_norm(g, 0.5) if clip else 0.0
    opt.step(p, g)
    return loss, norm


# ---------------------------------------------------------------------------------------- DDPG / TD3
class TD3:
    """TD3_file/TD3.py:150-256; with twin=False, policy_noise off and policy_freq 1 it is
    DDPG_simple (DDPG_file/DDPG_simple.py:100-179)."""

    def __init__(self, actor_p, critic_p, obs_dim, act_dim, actor_lr, critic_lr, capacity, twin=True,
                 use_policy_noise=True, twin_delay=True, critic_weight_decay=0.0, batch_obs_norm=False):
        # DDPG.py:160-161: Normalization_batch_size over the sampled observations
        self.bn = NormalizationBatch(obs_dim) if batch_obs_norm else None
        self.actor, self.actor_t = nn.copy_params(actor_p), nn.copy_params(actor_p)
        self.critic, self.critic_t = nn.copy_params(critic_p), nn.copy_params(critic_p)
        self.pi = MLP(["l1", "l2", "l3"], out_act="tanh")
        self.qnet = QNet(twin)
        self.twin, self.use_policy_noise, self.twin_delay = twin, use_policy_noise, twin_delay
        self.actor_opt = Adam(self.actor, actor_lr)
        self.critic_opt = Adam(self.critic, critic_lr, weight_decay=critic_weight_decay)
        self.buffer = Buffer(capacity, obs_dim, act_dim)
        self.total_it = 0
        self.critic_losses, self.actor_losses = [], []

    def select_action(self, obs):                       # TD3.py:163-170 (DDPG.py:165-166: norm, update=False)
        o = nn.f32(obs).reshape(1, -1)
        if self.bn is not None:
            o = self.bn(o, update=False)
        return self.pi.forward(self.actor, o)[0][0]

    def add(self, *a):
        self.buffer.add(*a)

    def learn(self, batch_size, gamma, tau, policy_noise=0.0, noise_clip=0.0, max_action=1.0, policy_freq=1,
              policy_noise_scale=1.0, noise=None):
        idx = _choice(len(self.buffer), batch_size)
        if noise is None and self.use_policy_noise:
            raise ValueError("the oracle takes the torch.randn_like draw as an input")
        return self.learn_with(idx, noise, gamma, tau, policy_noise, noise_clip, max_action, policy_freq,
                               policy_noise_scale)

    def learn_with(self, idx, noise, gamma, tau, policy_noise=0.0, noise_clip=0.0, max_action=1.0,
                   policy_freq=1, policy_noise_scale=1.0):
        self.total_it += 1                              # TD3.py:191
        obs, act, rew, nobs, done = self.buffer.sample(idx)
        if self.bn is not None:                         # DDPG.py:190-192: update on obs only
            obs = self.bn(obs)
            nobs = self.bn(nobs, update=False)
        a_next = self.pi.forward(self.actor_t, nobs)[0]
        if self.use_policy_noise:                       # TD3.py:196-198
            n = np.clip(F32(policy_noise_scale) * (nn.f32(noise) * F32(policy_noise)), -noise_clip, noise_clip).astype(F32)
            a_next = (np.clip(a_next * F32(max_action) + n, -max_action, max_action) / F32(max_action)).astype(F32)
        oa_next = np.concatenate([nobs, a_next], axis=1)
        (q1t, _), two = self.qnet.forward(self.critic_t, oa_next)
        next_q = np.minimum(q1t, two[0]) if self.twin else q1t     # TD3.py:203-206
        y = rew + F32(gamma) * next_q * (F32(1) - done)            # TD3.py:209
        oa = np.concatenate([obs, act], axis=1)
        closs, _ = _critic_step(self.qnet, self.critic, self.critic_opt, oa, y, td_loss=getattr(self, "td_loss", nn.mse))
        self.critic_losses.append(closs)
        if not self.twin_delay:
            policy_freq = 1                             # TD3.py:219-222
        aloss = None
        if self.total_it % policy_freq == 0:            # TD3.py:224-233
            a_new, pacts = self.pi.forward(self.actor, obs)
            oa_new = np.concatenate([obs, a_new], axis=1)
            q, qacts = self.qnet.q1.forward(self.critic, oa_new)
            aloss = F32(-np.mean(q, dtype=F32))
            dq = np.full_like(q, F32(-1.0 / q.shape[0]))
            doa, _ = self.qnet.q1.backward(self.critic, qacts, dq, need_dx=True)
            da = doa[:, obs.shape[1]:]
            _, g = self.pi.backward(self.actor, pacts, da, need_dx=False)
            g = {k: g[k] for k in self.actor}
            nn.clip_grad_norm(g, 0.5)
            self.actor_opt.step(self.actor, g)
            self.actor_losses.append(aloss)
            nn.soft_update(self.critic_t, self.critic, tau)        # TD3.py:243-244 (critic first)
            nn.soft_update(self.actor_t, self.actor, tau)
        return closs, aloss


def DDPG(actor_p, critic_p, obs_dim, act_dim, actor_lr, critic_lr, capacity, critic_weight_decay=0.0,
         batch_obs_norm=False):
    """DDPG_simple.learn (DDPG_simple.py:137-156; DDPG.py:203-222 with its supplements) = the TD3
    skeleton with a single critic, no target-policy noise and an actor/target update on every call."""
    return TD3(actor_p, critic_p, obs_dim, act_dim, actor_lr, critic_lr, capacity, twin=False,
               use_policy_noise=False, twin_delay=False, critic_weight_decay=critic_weight_decay,
               batch_obs_norm=batch_obs_norm)


# ---------------------------------------------------------------------------------------- SAC
class GaussianActor:
    """SAC_file/SAC.py:60-97: mean = mean_layer(relu(l2(relu(l1 s)))); state-independent
    log_std [1,A] clamped to [-20,2]; u = mean + std*eps; a = tanh(u);
    log_pi = sum_j N(u_j;mean_j,std_j).log_prob - sum_j 2(log2 - u_j - softplus(-2u_j))."""

    def __init__(self):
        self.body = MLP(["l1", "l2", "mean_layer"])

    @staticmethod
    def _softplus(x):
        return np.logaddexp(F32(0), x).astype(F32)

    def forward(self, p, obs, eps):
        mean, acts = self.body.forward(p, obs)
        log_std = np.clip(np.broadcast_to(p["log_std"], mean.shape), -20, 2).astype(F32)
        std = np.exp(log_std)
        u = mean + std * eps if eps is not None else mean
        var = std * std
        lp = -((u - mean) ** 2) / (F32(2) * var) - log_std - F32(LOG_SQRT_2PI)
        log_pi = lp.sum(axis=1, keepdims=True)
        log_pi = log_pi - (F32(2) * (F32(LOG2) - u - self._softplus(F32(-2) * u))).sum(axis=1, keepdims=True)
        a = np.tanh(u)
        return a, log_pi.astype(F32), dict(acts=acts, std=std, u=u, eps=eps, a=a)

    def backward(self, p, cache, da, dlogpi):
        """da: dL/da [B,A]; dlogpi: dL/dlog_pi [B,1].  Analytic reparameterised gradient:
        the Normal term contributes 0 to d/dmean and -1 to d/dlog_std; the tanh correction
        contributes 2 tanh(u) to d/du; du/dmean = 1, du/dlog_std = std*eps."""
        a, u, std, eps = cache["a"], cache["u"], cache["std"], cache["eps"]
        du = da * (F32(1) - a * a) + dlogpi * (F32(2) * np.tanh(u))
        _, g = self.body.backward(p, cache["acts"], du, need_dx=False)
        raw = p["log_std"]
        inside = ((raw >= -20) & (raw <= 2)).astype(F32)
        dls = (du * std * eps - dlogpi).sum(axis=0, keepdims=True) * inside
        g["log_std"] = dls.astype(F32)
        return g


class SAC:
    """SAC_file/SAC.py:171-282 with adaptive alpha (Alpha, SAC.py:154-169: log_alpha scalar,
    Adam lr 1e-4, alpha0 = 0.01, target_entropy = -act_dim)."""

    def __init__(self, actor_p, critic_p, obs_dim, act_dim, actor_lr, critic_lr, capacity, alpha0=0.01,
                 alpha_lr=1e-4, batch_obs_norm=False):
        self.bn = NormalizationBatch(obs_dim) if batch_obs_norm else None      # SAC.py:181-182
        self.actor, self.actor_t = nn.copy_params(actor_p), nn.copy_params(actor_p)
        self.critic, self.critic_t = nn.copy_params(critic_p), nn.copy_params(critic_p)
        self.pi = GaussianActor()
        self.qnet = QNet(True)
        self.actor_opt = Adam(self.actor, actor_lr)
        self.critic_opt = Adam(self.critic, critic_lr)
        self.alpha_p = {"log_alpha": np.array(np.log(alpha0), dtype=F32)}
        self.alpha_opt = Adam(self.alpha_p, alpha_lr)
        self.alpha = F32(np.exp(self.alpha_p["log_alpha"]))
        self.target_entropy = -act_dim
        self.buffer = Buffer(capacity, obs_dim, act_dim)
        self.critic_losses, self.actor_losses, self.alpha_losses, self.alphas = [], [], [], []

    def select_action(self, obs, eps):                  # SAC.py:192-198
        o = nn.f32(obs).reshape(1, -1)
        if self.bn is not None:
            o = self.bn(o, update=False)
        return self.pi.forward(self.actor, o, nn.f32(eps).reshape(1, -1))[0][0]

    def evaluate_action(self, obs):                     # SAC.py:200-204: tanh(mean)
        return self.pi.forward(self.actor, nn.f32(obs).reshape(1, -1), None)[0][0]

    def add(self, *a):
        self.buffer.add(*a)

    def learn_with(self, idx, eps_next, eps_new, gamma, tau):       # SAC.py:222-260
        obs, act, rew, nobs, done = self.buffer.sample(idx)
        if self.bn is not None:                         # SAC.py:215-217
            obs = self.bn(obs)
            nobs = self.bn(nobs, update=False)
        B = obs.shape[0]
        a_next, logpi_next, _ = self.pi.forward(self.actor_t, nobs, nn.f32(eps_next))
        (q1t, _), (q2t, _) = self.qnet.forward(self.critic_t, np.concatenate([nobs, a_next], axis=1))
        next_q = np.minimum(q1t, q2t)
        y = rew + F32(gamma) * (F32(1) - done) * (next_q + self.alpha * (-logpi_next))
        closs, _ = _critic_step(self.qnet, self.critic, self.critic_opt, np.concatenate([obs, act], axis=1), y,
                                td_loss=getattr(self, "td_loss", nn.mse))
        a_new, logpi, cache = self.pi.forward(self.actor, obs, nn.f32(eps_new))
        oa_new = np.concatenate([obs, a_new], axis=1)
        (q1, acts1), (q2, acts2) = self.qnet.forward(self.critic, oa_new)
        q_pi = (q1 + q2) / F32(2)                       # mean of the twins (SAC.py:250)
        entropy = -logpi
        aloss = F32(np.mean(-q_pi - self.alpha * entropy, dtype=F32))
        dq = np.full_like(q1, F32(-0.5 / B))
        doa1, _ = self.qnet.q1.backward(self.critic, acts1, dq, need_dx=True)
        doa2, _ = self.qnet.q2.backward(self.critic, acts2, dq, need_dx=True)
        da = (doa1 + doa2)[:, obs.shape[1]:]
        dlogpi = np.full_like(logpi, self.alpha / F32(B))
        g = self.pi.backward(self.actor, cache, da, dlogpi)
        g = {k: g[k] for k in self.actor}
        nn.clip_grad_norm(g, 0.5)
        self.actor_opt.step(self.actor, g)
        nn.soft_update(self.critic_t, self.critic, tau)
        nn.soft_update(self.actor_t, self.actor, tau)
        # alpha_loss = (alpha * (entropy - target_entropy).detach()).mean() (SAC.py:259)
        mean_term = F32(np.mean(entropy - F32(self.target_entropy), dtype=F32))
        alpha_loss = F32(self.alpha * mean_term)
        self.alpha_opt.step(self.alpha_p, {"log_alpha": np.array(self.alpha * mean_term, dtype=F32)})
        self.alpha = F32(np.exp(self.alpha_p["log_alpha"]))
        self.critic_losses.append(closs)
        self.actor_losses.append(aloss)
        self.alpha_losses.append(alpha_loss)
        self.alphas.append(self.alpha)
        return closs, aloss, alpha_loss


# ---------------------------------------------------------------------------------------- MADDPG
class MADDPG:
    """MADDPG_file/MADDPG_simple.py:107-210.  dims: {agent_id: [obs_dim, act_dim]} (ordered).
    Per-agent buffers written in lock-step; centralised critic on cat(all obs + all acts);
    `learn` re-samples per agent, steps critic then actor, one soft update of every agent's
    actor then critic at the end."""

    def __init__(self, params, dims, actor_lr, critic_lr, capacity, critic_weight_decay=0.0, batch_obs_norm=False):
        """critic_weight_decay / batch_obs_norm: MADDPG.py's supplements (:118-121, :155-156,162-163,194-196)."""
        self.ids = list(dims.keys())
        self.dims = dims
        self.bn = {a: NormalizationBatch(dims[a][0]) for a in self.ids} if batch_obs_norm else None
        self._wd = critic_weight_decay
        self.pi = MLP(["l1", "l2", "l3"], out_act="tanh")
        self.q = MLP(["l1", "l2", "l3"])
        self.actor = {a: nn.copy_params(params[a]["actor"]) for a in self.ids}
        self.actor_t = {a: nn.copy_params(params[a]["actor"]) for a in self.ids}
        self.critic = {a: nn.copy_params(params[a]["critic"]) for a in self.ids}
        self.critic_t = {a: nn.copy_params(params[a]["critic"]) for a in self.ids}
        self.actor_opt = {a: Adam(self.actor[a], actor_lr) for a in self.ids}
        self.critic_opt = {a: Adam(self.critic[a], critic_lr, weight_decay=critic_weight_decay) for a in self.ids}
        self.buffers = {a: Buffer(capacity, dims[a][0], dims[a][1]) for a in self.ids}
        self.critic_losses = {a: [] for a in self.ids}
        self.actor_losses = {a: [] for a in self.ids}

    def select_action(self, obs):                       # MADDPG_simple.py:122-132; MADDPG.py:162-163 normalises, no update
        out = {}
        for a in self.ids:
            o = nn.f32(obs[a]).reshape(1, -1)
            if self.bn is not None:
                o = self.bn[a](o, update=False)
            out[a] = self.pi.forward(self.actor[a], o)[0][0]
        return out

    def _sample(self, idx):
        """MADDPG.py:189-197: every agent's rows; obs normalised with an update, next_obs without."""
        batch = {a: list(self.buffers[a].sample(idx)) for a in self.ids}
        if self.bn is not None:
            for a in self.ids:
                batch[a][0] = self.bn[a](batch[a][0])
                batch[a][3] = self.bn[a](batch[a][3], update=False)
        return batch

    def add(self, obs, action, reward, next_obs, done):
        for a in self.ids:
            self.buffers[a].add(obs[a], action[a], reward[a], next_obs[a], done[a])

    def learn_with(self, idx_per_agent, gamma, tau):    # MADDPG_simple.py:165-186
        for j, aid in enumerate(self.ids):
            idx = idx_per_agent[j]
            batch = self._sample(idx)
            next_act = {a: self.pi.forward(self.actor_t[a], batch[a][3])[0] for a in self.ids}
            x_next = np.concatenate([batch[a][3] for a in self.ids] + [next_act[a] for a in self.ids], axis=1)
            next_q = self.q.forward(self.critic_t[aid], x_next)[0]
            y = batch[aid][2] + F32(gamma) * next_q * (F32(1) - batch[aid][4])
            x = np.concatenate([batch[a][0] for a in self.ids] + [batch[a][1] for a in self.ids], axis=1)
            qv, qa = self.q.forward(self.critic[aid], x)
            closs, dq = nn.mse(qv, y)
            _, g = self.q.backward(self.critic[aid], qa, dq, need_dx=False)
            g = {k: g[k] for k in self.critic[aid]}
            nn.clip_grad_norm(g, 0.5)
            self.critic_opt[aid].step(self.critic[aid], g)
            # actor step through this agent's own action slot (MADDPG_simple.py:178-183)
            a_new, pacts = self.pi.forward(self.actor[aid], batch[aid][0])
            acts = {a: batch[a][1] for a in self.ids}
            acts[aid] = a_new
            x2 = np.concatenate([batch[a][0] for a in self.ids] + [acts[a] for a in self.ids], axis=1)
            q2, q2a = self.q.forward(self.critic[aid], x2)
            aloss = F32(-np.mean(q2, dtype=F32))
            dx, _ = self.q.backward(self.critic[aid], q2a, np.full_like(q2, F32(-1.0 / q2.shape[0])), need_dx=True)
            off = sum(self.dims[a][0] for a in self.ids) + sum(self.dims[a][1] for a in self.ids[:j])
            da = dx[:, off:off + self.dims[aid][1]]
            _, ga = self.pi.backward(self.actor[aid], pacts, da, need_dx=False)
            ga = {k: ga[k] for k in self.actor[aid]}
            nn.clip_grad_norm(ga, 0.5)
            self.actor_opt[aid].step(self.actor[aid], ga)
            self.critic_losses[aid].append(closs)
            self.actor_losses[aid].append(aloss)
        for a in self.ids:                              # MADDPG_simple.py:188-195: actor then critic
            nn.soft_update(self.actor_t[a], self.actor[a], tau)
            nn.soft_update(self.critic_t[a], self.critic[a], tau)


class MATD3(MADDPG):
    """MADDPG_file/MATD3_simple.py:151-262: MADDPG with twin centralised critics (Critic_TD3 :88-124, min of the
    targets :222-224, loss on both heads :229-231, actor through Q1 only :240-241), target policy smoothing on EVERY
    agent's target action (:199-201, one randn_like per agent inside each agent's sample()) and the delayed actor
    step + target update (:236,245-246)."""

    def __init__(self, params, dims, actor_lr, critic_lr, capacity):
        super().__init__(params, dims, actor_lr, critic_lr, capacity)
        self.q = QNet(True)
        self.total_it = 0

    def learn_with(self, idx_per_agent, noise_per_agent, gamma, tau, policy_noise_scale, policy_noise, noise_clip,
                   max_action, policy_freq):
        """noise_per_agent[i][j]: the [B, A_j] draw for agent j's target action inside agent i's sample()."""
        self.total_it += 1
        do_actor = (self.total_it % policy_freq == 0)
        for i, aid in enumerate(self.ids):
            idx = idx_per_agent[i]
            batch = {a: self.buffers[a].sample(idx) for a in self.ids}
            next_act = {}
            for j, a in enumerate(self.ids):
                n = np.clip(F32(policy_noise_scale) * (nn.f32(noise_per_agent[i][j]) * F32(policy_noise)), -noise_clip,
                            noise_clip).astype(F32)
                at = self.pi.forward(self.actor_t[a], batch[a][3])[0]
                next_act[a] = (np.clip(at * F32(max_action) + n, -max_action, max_action) / F32(max_action)).astype(F32)
            x_next = np.concatenate([batch[a][3] for a in self.ids] + [next_act[a] for a in self.ids], axis=1)
            (q1t, _), two = self.q.forward(self.critic_t[aid], x_next)
            y = batch[aid][2] + F32(gamma) * np.minimum(q1t, two[0]) * (F32(1) - batch[aid][4])
            x = np.concatenate([batch[a][0] for a in self.ids] + [batch[a][1] for a in self.ids], axis=1)
            closs, _ = _critic_step(self.q, self.critic[aid], self.critic_opt[aid], x, y)
            self.critic_losses[aid].append(closs)
            if do_actor:
                a_new, pacts = self.pi.forward(self.actor[aid], batch[aid][0])
                acts = {a: batch[a][1] for a in self.ids}
                acts[aid] = a_new
                x2 = np.concatenate([batch[a][0] for a in self.ids] + [acts[a] for a in self.ids], axis=1)
                q2, q2a = self.q.q1.forward(self.critic[aid], x2)
                aloss = F32(-np.mean(q2, dtype=F32))
                dx, _ = self.q.q1.backward(self.critic[aid], q2a, np.full_like(q2, F32(-1.0 / q2.shape[0])), need_dx=True)
                off = sum(self.dims[a][0] for a in self.ids) + sum(self.dims[a][1] for a in self.ids[:i])
                _, ga = self.pi.backward(self.actor[aid], pacts, dx[:, off:off + self.dims[aid][1]], need_dx=False)
                ga = {k: ga[k] for k in self.actor[aid]}
                nn.clip_grad_norm(ga, 0.5)
                self.actor_opt[aid].step(self.actor[aid], ga)
                self.actor_losses[aid].append(aloss)
        if do_actor:
            for a in self.ids:                          # MATD3_simple.py:248-255: actor then critic per agent
                nn.soft_update(self.actor_t[a], self.actor[a], tau)
                nn.soft_update(self.critic_t[a], self.critic[a], tau)
