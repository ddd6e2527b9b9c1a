"""Oracle PPO: GAE + clipped-surrogate minibatch updates, restating
PPO_file/PPO_with_tricks.py:290-354 (Gaussian actor :79-108, critic :158-177, Agent :179-209).
Test infrastructure (see oracle/__init__.py).

Reference defect handled explicitly: PPO_with_tricks.py:302 `np.zeros(h, dtype=torch.float32)`
raises TypeError as committed; the intent (float32 advantage array, as PPO.py:222 modulo
precision) is what is restated here and what the golden generator patches in.
"""
import math

import numpy as np

from . import nn
from .buffer import BufferForPPO, BufferForPPO2
from .nn import F32, MLP, Adam

HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)
LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


def gae(td_delta, adv_dones, gamma, lmbda):
    """Sequential reverse scan, PPO_with_tricks.py:308-311.  Under NumPy >= 2 the recurrence
    runs in float32: `gamma*lmbda` is a Python float (weak scalar), `td_delta[i]` and
    `adv_dones[i]` are float32 scalars, so every product/sum rounds to float32."""
    T = td_delta.shape[0]
    adv = np.zeros(T, dtype=F32)
    g = 0
    c = gamma * lmbda
    for i in reversed(range(T)):
        g = td_delta[i] + c * g * (1.0 - adv_dones[i])
        adv[i] = g
    return adv


F32_EPS = float(np.finfo(np.float32).eps)


class PPO:
    def __init__(self, actor_p, critic_p, obs_dim, act_dim, actor_lr, critic_lr, horizon, trick, discrete=False,
                 optimizer="adam", beta=False, rollout_values=False, cat_logits=False):
        """rollout_values=True: PPO_advance/PPO_2.py:152-292 — values stored at rollout time, advantages and returns from the
        buffer's stable-baselines3-style float64 scan, `learn_with(..., last_value=...)`; otherwise that file's learn is
        PPO_file/PPO.py's with two torch Adams.
        optimizer="c_adamw": PPO_file/PPO.py:109-152,213-286 — the same clipped-surrogate learn with no tricks and ONE
        cautious AdamW (lr = actor_lr) over actor + critic parameters, each net's gradients clipped to 0.5 on its own."""
        self.trick = trick
        # cat_logits: PPO_file/PPO.py's discrete actor returns raw logits (:78-90) into Categorical(logits=...) (:176,257):
        # log-probs are the log-softmax itself, where PPO_with_tricks.py's Categorical(probs=softmax) clamps at float eps
        self.cat_logits = bool(cat_logits)
        self.discrete = discrete
        self.beta = beta
        act = "tanh" if trick.get("tanh") else "relu"
        if beta:        # Actor_Beta (:120-151): alpha_layer and beta_layer on the shared trunk
            self.pi = None
            self.trunk = MLP(["l1", "l2"], hidden_act=act, out_act=act)
        elif discrete:    # Actor_discrete (:110-121): ReLU hidden layers regardless of trick['tanh'], softmax head
            self.pi = MLP(["l1", "l2", "l3"], hidden_act="relu", out_act=None)
        else:
            self.pi = MLP(["l1", "l2", "mean_layer"], hidden_act=act, out_act="tanh")   # mean = tanh(...) (:99)
        self.v = MLP(["l1", "l2", "l3"], hidden_act=act)
        self.actor = nn.copy_params(actor_p)
        self.critic = nn.copy_params(critic_p)
        eps = 1e-5 if trick.get("adam_eps") else 1e-8                               # :191-196
        if optimizer == "c_adamw":        # per-tensor state: one optimiser over both nets == one per net with the same lr
            self.actor_opt = nn.CautiousAdamW(self.actor, actor_lr)
            self.critic_opt = nn.CautiousAdamW(self.critic, actor_lr)
        else:
            self.actor_opt = Adam(self.actor, actor_lr, eps=eps)
            self.critic_opt = Adam(self.critic, critic_lr, eps=eps)
        self.horizon = int(horizon)
        self.rollout_values = rollout_values
        self.buffer = (BufferForPPO2 if rollout_values else BufferForPPO)(horizon, obs_dim, 1 if discrete else act_dim)
        self.actor_losses, self.critic_losses = [], []
        self.adv_raw = self.v_target = None

    def _dist(self, obs):
        mean, acts = self.pi.forward({k: v for k, v in self.actor.items() if k != "log_std"}, obs)
        log_std = np.clip(np.broadcast_to(self.actor["log_std"], mean.shape), -20, 2).astype(F32)   # :102
        return mean, log_std, acts

    def probs(self, obs):
        z = self.pi.forward(self.actor, obs)[0]
        z = z - z.max(axis=1, keepdims=True)
        e = np.exp(z)
        return (e / e.sum(axis=1, keepdims=True)).astype(F32)

    # ---- Beta policy -------------------------------------------------------------------------------------------
    @staticmethod
    def _softplus(z):
        return np.where(z > 20, z, np.log1p(np.exp(np.minimum(z, 20)))).astype(F32)

    def _alpha_beta(self, obs):
        """-> alpha, beta, (trunk activations, z_alpha, z_beta)"""
        tp = {k: v for k, v in self.actor.items() if k.startswith(("l1.", "l2."))}
        h, acts = self.trunk.forward(tp, obs)
        za = h @ self.actor["alpha_layer.weight"].T + self.actor["alpha_layer.bias"]
        zb = h @ self.actor["beta_layer.weight"].T + self.actor["beta_layer.bias"]
        return self._softplus(za) + F32(1), self._softplus(zb) + F32(1), (h, acts, za.astype(F32), zb.astype(F32))

    def beta_log_prob(self, obs, a):                     # Beta(alpha, beta).log_prob(a), per dimension (:241-251)
        from scipy.special import gammaln
        al, be, _ = self._alpha_beta(nn.f32(obs).reshape(1, -1))
        a = nn.f32(a).reshape(1, -1).astype(np.float64)
        al, be = al.astype(np.float64), be.astype(np.float64)
        return ((al - 1) * np.log(a) + (be - 1) * np.log1p(-a) - (gammaln(al) + gammaln(be) - gammaln(al + be)))[0].astype(F32)

    def _beta_actor_step(self, obs, action, logp_old, A, clip_param, ent_coef):
        """Beta branch of the actor update (:325-332,337-346); special functions in float64 (scipy), the rest fp32."""
        from scipy.special import digamma, gammaln, polygamma
        mb = obs.shape[0]
        al32, be32, (h, tacts, za, zb) = self._alpha_beta(obs)
        al, be, x = al32.astype(np.float64), be32.astype(np.float64), action.astype(np.float64)
        lB = gammaln(al) + gammaln(be) - gammaln(al + be)
        logp = (al - 1) * np.log(x) + (be - 1) * np.log1p(-x) - lB
        ent = (lB - (al - 1) * digamma(al) - (be - 1) * digamma(be) + (al + be - 2) * digamma(al + be)).sum(axis=1, keepdims=True)
        ratio = np.exp((logp.sum(axis=1, keepdims=True) - logp_old.astype(np.float64).sum(axis=1, keepdims=True))).astype(F32)
        surr1 = ratio * A
        surr2 = np.clip(ratio, 1 - clip_param, 1 + clip_param).astype(F32) * A
        aloss = F32(-np.mean(np.minimum(surr1, surr2), dtype=F32) - F32(ent_coef) * F32(np.mean(ent)))
        coef = (np.where(surr1 <= surr2, A, F32(0)) * F32(-1.0 / mb) * ratio).astype(np.float64)
        ce = ent_coef / mb
        psi_ab, tri_ab = digamma(al + be), polygamma(1, al + be)
        d_al = coef * (np.log(x) - digamma(al) + psi_ab) - ce * (-(al - 1) * polygamma(1, al) + (al + be - 2) * tri_ab)
        d_be = coef * (np.log1p(-x) - digamma(be) + psi_ab) - ce * (-(be - 1) * polygamma(1, be) + (al + be - 2) * tri_ab)
        sig = lambda z: np.where(z > 20, 1.0, 1.0 / (1.0 + np.exp(-z.astype(np.float64))))
        dza, dzb = (d_al * sig(za)).astype(F32), (d_be * sig(zb)).astype(F32)
        g = {"alpha_layer.weight": dza.T @ h, "alpha_layer.bias": dza.sum(axis=0),
             "beta_layer.weight": dzb.T @ h, "beta_layer.bias": dzb.sum(axis=0)}
        dh = dza @ self.actor["alpha_layer.weight"] + dzb @ self.actor["beta_layer.weight"]
        tp = {k: v for k, v in self.actor.items() if k.startswith(("l1.", "l2."))}
        _, gt = self.trunk.backward(tp, tacts, dh.astype(F32), need_dx=False)      # applies the trunk's last activation's derivative
        g.update(gt)
        g = {kk: g[kk].astype(F32) for kk in self.actor}
        nn.clip_grad_norm(g, 0.5)
        self.actor_opt.step(self.actor, g)
        self.actor_losses.append(aloss)

    def evaluate_action(self, obs):                      # :257-270: the mean / argmax of the probabilities
        if self.beta:                                    # Actor_Beta.mean (:145-151) mapped [0,1] -> [-1,1] (:259-261)
            al, be, _ = self._alpha_beta(nn.f32(obs).reshape(1, -1))
            return (F32(2) * (al / (al + be) - F32(0.5)))[0]
        if self.discrete:
            return int(np.argmax(self.probs(nn.f32(obs).reshape(1, -1))[0]))
        return self._dist(nn.f32(obs).reshape(1, -1))[0][0]

    def select_action_discrete(self, obs, q):
        """Categorical(probs).sample() == argmax(probs / q), q ~ Exp(1) per class (torch's
        single-draw multinomial); log_prob = log(clamp(p_a)) (:249-251)."""
        p = self.probs(nn.f32(obs).reshape(1, -1))[0]
        a = int(np.argmax(p / nn.f32(q).reshape(-1)))
        if self.cat_logits:
            z = self.pi.forward(self.actor, nn.f32(obs).reshape(1, -1))[0][0]
            zs = z - z.max()
            return a, F32(zs[a] - np.log(np.exp(zs).sum(dtype=F32)))
        return a, F32(np.log(np.clip(p[a] / p.sum(dtype=F32), F32_EPS, 1 - F32_EPS)))

    def select_action(self, obs, eps):                   # :234-255: a ~ N(mean,std); per-dim log-prob
        mean, log_std, _ = self._dist(nn.f32(obs).reshape(1, -1))
        std = np.exp(log_std)
        a = mean + std * nn.f32(eps).reshape(1, -1)
        logp = -((a - mean) ** 2) / (F32(2) * std * std) - log_std - F32(LOG_SQRT_2PI)
        return a[0], logp[0]

    def add(self, *a):
        self.buffer.add(*a)

    def learn_with(self, perms, minibatch_size, gamma, lmbda, clip_param, k_epochs, ent_coef, last_value=None):
        obs, action, reward, nobs, done, logp_old, adv_dones = self.buffer.all()
        T = self.horizon
        if self.rollout_values:                                                      # PPO_2.py:214,223-224
            self.buffer.compute_returns_and_advantage(gamma, lmbda, last_value)
            adv = self.buffer.advantages.astype(F32).reshape(-1, 1)
            v_target = self.buffer.returns.astype(F32).reshape(-1, 1)
        else:
            vs = self.v.forward(self.critic, obs)[0]
            vs_ = self.v.forward(self.critic, nobs)[0]
            td = reward + F32(gamma) * (F32(1.0) - done) * vs_ - vs                  # :306
            adv = gae(td.reshape(-1), adv_dones.reshape(-1), gamma, lmbda).reshape(-1, 1)
            v_target = adv + vs                                                      # :313
        self.adv_raw, self.v_target = adv.copy(), v_target.copy()
        if self.trick.get("adv_norm"):                                               # :314-315
            std = np.sqrt(np.sum((adv - adv.mean(dtype=F32)) ** 2, dtype=F32) / F32(T - 1))   # unbiased
            adv = (adv - adv.mean(dtype=F32)) / (std + F32(1e-8))
        for k in range(k_epochs):
            perm = perms[k]
            for s in range(0, T, minibatch_size):
                ix = perm[s:s + minibatch_size]
                mb = len(ix)
                if self.beta:
                    self._beta_actor_step(obs[ix], action[ix], logp_old[ix], adv[ix], clip_param, ent_coef)
                    self._critic_step(obs[ix], v_target[ix])
                    continue
                if self.discrete:
                    self._discrete_actor_step(obs[ix], action[ix], logp_old[ix], adv[ix], clip_param, ent_coef)
                    self._critic_step(obs[ix], v_target[ix])
                    continue
                # ---- actor (:324-346)
                body = {kk: vv for kk, vv in self.actor.items() if kk != "log_std"}
                mean, acts = self.pi.forward(body, obs[ix])
                raw = self.actor["log_std"]
                log_std = np.clip(np.broadcast_to(raw, mean.shape), -20, 2).astype(F32)
                std = np.exp(log_std)
                var = std * std
                a = action[ix]
                logp = -((a - mean) ** 2) / (F32(2) * var) - log_std - F32(LOG_SQRT_2PI)
                ent = (F32(0.5 + HALF_LOG_2PI) + log_std).sum(axis=1, keepdims=True)
                ratio = np.exp(logp.sum(axis=1, keepdims=True) - logp_old[ix].sum(axis=1, keepdims=True))
                A = adv[ix]
                surr1 = ratio * A
                surr2 = np.clip(ratio, 1 - clip_param, 1 + clip_param).astype(F32) * A
                aloss = F32(-np.mean(np.minimum(surr1, surr2), dtype=F32) - F32(ent_coef) * np.mean(ent, dtype=F32))
                # backward: d min / d ratio = A where surr1 <= surr2 (ties: both branches pass
                # the clamp, half each, summing to A), else 0 (clamped branch has zero slope)
                dratio = np.where(surr1 <= surr2, A, F32(0)) * F32(-1.0 / mb)
                dlogp_sum = dratio * ratio
                dmean = dlogp_sum * (a - mean) / var
                inside = ((raw >= -20) & (raw <= 2)).astype(F32)
                dls = dlogp_sum * ((a - mean) ** 2 / var - F32(1)) - F32(ent_coef / mb)
                _, g = self.pi.backward(body, acts, dmean.astype(F32), need_dx=False)
                g["log_std"] = (dls.sum(axis=0, keepdims=True) * inside).astype(F32)
                g = {kk: g[kk] for kk in self.actor}
                nn.clip_grad_norm(g, 0.5)
                self.actor_opt.step(self.actor, g)
                self.actor_losses.append(aloss)
                self._critic_step(obs[ix], v_target[ix])
        self.buffer.clear()                                                          # :354

    def _critic_step(self, obs, v_target):
        """critic (:349-351): mse(v_target[idx], V(obs[idx]))"""
        v_s, vacts = self.v.forward(self.critic, obs)
        closs, dv = nn.mse(v_s, v_target)
        _, gc = self.v.backward(self.critic, vacts, dv, need_dx=False)
        gc = {kk: gc[kk] for kk in self.critic}
        nn.clip_grad_norm(gc, 0.5)
        self.critic_opt.step(self.critic, gc)
        self.critic_losses.append(closs)

    def _discrete_actor_step(self, obs, action, logp_old, A, clip_param, ent_coef):
        """Categorical branch (:333-336): dist = Categorical(probs=softmax(l3(.))); entropy and
        log-prob from the clamped, renormalised probabilities."""
        mb = obs.shape[0]
        z, acts = self.pi.forward(self.actor, obs)
        zs = z - z.max(axis=1, keepdims=True)
        e = np.exp(zs)
        p = (e / e.sum(axis=1, keepdims=True)).astype(F32)
        logit = np.log(np.clip(p / p.sum(axis=1, keepdims=True, dtype=F32), F32_EPS, 1 - F32_EPS)).astype(F32)
        if self.cat_logits:                       # Categorical(logits=z): logits - logsumexp, unclamped (PPO.py:257-259)
            logit = (zs - np.log(e.sum(axis=1, keepdims=True, dtype=F32))).astype(F32)
        a = action.astype(np.int64).reshape(-1)
        lp = logit[np.arange(mb), a].reshape(-1, 1)
        ent = -(logit * p).sum(axis=1, keepdims=True)
        ratio = np.exp(lp - logp_old.sum(axis=1, keepdims=True))
        surr1 = ratio * A
        surr2 = np.clip(ratio, 1 - clip_param, 1 + clip_param).astype(F32) * A
        aloss = F32(-np.mean(np.minimum(surr1, surr2), dtype=F32) - F32(ent_coef) * np.mean(ent, dtype=F32))
        coef = np.where(surr1 <= surr2, A, F32(0)) * F32(-1.0 / mb) * ratio
        onehot = np.zeros_like(p)
        onehot[np.arange(mb), a] = 1
        dz = coef * (onehot - p) + F32(ent_coef / mb) * p * (logit + ent)
        _, g = self.pi.backward(self.actor, acts, dz.astype(F32), need_dx=False)
        g = {kk: g[kk] for kk in self.actor}
        nn.clip_grad_norm(g, 0.5)
        self.actor_opt.step(self.actor, g)
        self.actor_losses.append(aloss)
