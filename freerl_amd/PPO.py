"""`PPO` with the reference's class surface (PPO_file/PPO_with_tricks.py:179-374), Gaussian actor.

    PPO(dim_info, is_continue, actor_lr, critic_lr, horizon, device, trick=None, beta=False)

The whole `learn()` (value pass, GAE scan, v_target, advantage normalisation, K_epochs x
minibatch actor/critic steps) runs in three GPU launches.  Reference defect handled: the
committed `learn` raises TypeError at :302 (`np.zeros(..., dtype=torch.float32)`); the evident
intent (float32 advantages) is what runs here.  Gaussian (`Actor`), Categorical (`Actor_discrete`) and
Beta (`Actor_Beta`) policies are ported.  ObsNorm / reward tricks that live in the caller's loop use
`freerl_amd.normalization`.  `trick=None` builds PPO_file/PPO.py's class (one cautious AdamW).
"""
import os

import numpy as np
import torch

from . import _native as N
from ._core import BatchObsNormView, DeviceNet, Engine, OptimizerView, init_layers, resolve_device
from .Buffer import Buffer_for_PPO

_TRICK_DEFAULT = dict(adv_norm=False, ObsNorm=False, reward_norm=False, reward_scaling=False, orthogonal_init=False,
                      adam_eps=False, lr_decay=False, tanh=False, Batch_ObsNorm=False)


class BetaActorNet(DeviceNet):
    """`agent.actor` of Actor_Beta (PPO_with_tricks.py:120-151): the reference keeps `alpha_layer` and `beta_layer`
    as two nn.Linear on the shared trunk; the engine holds them as one 2A-wide head [alpha_layer ; beta_layer]."""

    def __init__(self, engine, hidden, obs_dim, action_dim):
        super().__init__(engine, 0, [("l1", hidden, obs_dim), ("l2", hidden, hidden), ("head", 2 * action_dim, hidden)],
                         act_mode=N.ACT_RAW)
        self._A, self._H = action_dim, hidden

    def keys(self):
        return ["l1.weight", "l1.bias", "l2.weight", "l2.bias", "alpha_layer.weight", "alpha_layer.bias",
                "beta_layer.weight", "beta_layer.bias"]

    def _split(self, flat):
        parts = super()._split(flat)
        A = self._A
        w, b = parts.pop("head.weight"), parts.pop("head.bias")
        parts.update({"alpha_layer.weight": w[:A], "beta_layer.weight": w[A:], "alpha_layer.bias": b[:A], "beta_layer.bias": b[A:]})
        return parts

    def load_state_dict(self, sd, strict=True):
        if strict and set(sd.keys()) != set(self.keys()):
            raise RuntimeError("Error(s) in loading state_dict: expected keys %s, got %s" % (self.keys(), list(sd.keys())))
        t = lambda k: np.asarray(torch.as_tensor(sd[k]).detach().cpu().numpy(), dtype=np.float32).reshape(-1)
        flat = [t(k) for k in ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "alpha_layer.weight", "beta_layer.weight",
                               "alpha_layer.bias", "beta_layer.bias")]
        self._e.set_params(self._net, np.concatenate(flat), self._kind, self._learner)

    def alpha_beta(self, obs):
        """-> (alpha, beta) torch tensors [rows, A] = softplus(head) + 1 (:140-143)."""
        z = self(obs)
        sp = torch.nn.functional.softplus(z) + 1.0
        return sp[:, :self._A], sp[:, self._A:]


class Agent:
    def __init__(self, engine, obs_dim, action_dim, actor_lr, critic_lr, trick, hidden, is_continue=True, beta=False):
        if beta:
            self._init_beta(engine, obs_dim, action_dim, actor_lr, critic_lr, trick, hidden)
            return
        head = "mean_layer" if is_continue else "l3"
        al = [("l1", hidden, obs_dim), ("l2", hidden, hidden), (head, action_dim, hidden)]
        cl = [("l1", hidden, obs_dim), ("l2", hidden, hidden), ("l3", 1, hidden)]
        ortho = bool(trick["orthogonal_init"])
        # Actor (:79-108) applies orthogonal_init, Actor_discrete (:110-121) does not
        fa = init_layers(al, orthogonal=[1.0, 1.0, 0.01] if (ortho and is_continue) else None)     # :92-95
        fc = init_layers(cl, orthogonal=[1.0, 1.0, 1.0] if ortho else None)      # :165-168
        if is_continue:
            engine.set_params(0, np.concatenate([fa, np.zeros(action_dim, np.float32)]))   # log_std zeros (:85)
        else:
            engine.set_params(0, fa)
        engine.set_params(1, fc)
        eps = 1e-5 if trick["adam_eps"] else 1e-8                                 # :191-196
        self.actor = DeviceNet(engine, 0, al, extra=("log_std", (1, action_dim)) if is_continue else None,
                               act_mode=N.ACT_TANHHEAD if is_continue else N.ACT_RAW)
        self.critic = DeviceNet(engine, 1, cl)
        self.actor_optimizer = OptimizerView(engine, 0, actor_lr, eps=eps)
        self.critic_optimizer = OptimizerView(engine, 1, critic_lr, eps=eps)


    def _init_beta(self, engine, obs_dim, action_dim, actor_lr, critic_lr, trick, hidden):
        # torch RNG order: l1, l2, alpha_layer, beta_layer default draws, then (orthogonal) the same four in order (:123-133)
        al = [("l1", hidden, obs_dim), ("l2", hidden, hidden), ("alpha_layer", action_dim, hidden), ("beta_layer", action_dim, hidden)]
        cl = [("l1", hidden, obs_dim), ("l2", hidden, hidden), ("l3", 1, hidden)]
        ortho = bool(trick["orthogonal_init"])
        fa = init_layers(al, orthogonal=[1.0, 1.0, 0.01, 0.01] if ortho else None)
        fc = init_layers(cl, orthogonal=[1.0, 1.0, 1.0] if ortho else None)
        self.actor = BetaActorNet(engine, hidden, obs_dim, action_dim)
        # init_layers returns [l1.w l1.b l2.w l2.b alpha.w alpha.b beta.w beta.b]; the engine wants [.. alpha.w beta.w alpha.b beta.b]
        o = hidden * obs_dim + hidden + hidden * hidden + hidden
        hw, A = hidden * action_dim, action_dim
        aw, ab, bw, bb = fa[o:o + hw], fa[o + hw:o + hw + A], fa[o + hw + A:o + 2 * hw + A], fa[o + 2 * hw + A:]
        engine.set_params(0, np.concatenate([fa[:o], aw, bw, ab, bb]))
        engine.set_params(1, fc)
        eps = 1e-5 if trick["adam_eps"] else 1e-8
        self.critic = DeviceNet(engine, 1, cl)
        self.actor_optimizer = OptimizerView(engine, 0, actor_lr, eps=eps)
        self.critic_optimizer = OptimizerView(engine, 1, critic_lr, eps=eps)


class PPO:
    _buffer_cls = Buffer_for_PPO          # PPO_2 swaps in Buffer_for_PPO_2 (one more stored column: the rollout-time value)

    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, horizon, device, trick=None, beta=False, *,
                 rng="host", hidden=128, minibatch_max=256, seed=0):
        obs_dim, action_dim = dim_info
        if beta and not is_continue:
            raise ValueError("Actor_Beta is a continuous-action policy (PPO_with_tricks.py:182-186)")
        # `trick=None` is how PPO_file/PPO.py's class is built (PPO.py:156, it has no tricks and PPO_with_tricks.py
        # would fail on trick[...]): that file trains with ONE cautious AdamW (c_adamw.py) over actor + critic at
        # lr = actor_lr, eps 1e-6 (PPO.py:121,145-152)
        self._cautious = trick is None
        self.trick = dict(_TRICK_DEFAULT, **(trick or {}))
        self.actor_dist = {"Beta": bool(beta)}
        hip_id, self.device = resolve_device(device)
        self.horizon = int(horizon)
        stored = action_dim if is_continue else 1            # Buffer act_dim (Buffer.py:3-9)
        self._e = Engine(N.ALGO_PPO, obs_dim, action_dim, max(self.horizon, 2), hidden=hidden, batch_max=minibatch_max,
                         extra_cols=stored + self._buffer_cls._tail_cols, hidden_act=N.ACT_TANH if self.trick["tanh"] else N.ACT_RELU,
                         discrete=not is_continue, device_id=hip_id, seed=seed,
                         actor_dist=1 if beta else (2 if (self._cautious and not is_continue) else 0))   # PPO.py: Categorical(logits=)
        self.agent = Agent(self._e, obs_dim, action_dim, actor_lr, critic_lr, self.trick, hidden, is_continue, beta=bool(beta))
        self.buffer = self._buffer_cls(self.horizon, obs_dim, stored, self.device, _engine=self._e)
        if self.trick["Batch_ObsNorm"]:                                   # PPO_with_tricks.py:225-226
            self._e.obsnorm_enable(True)
            self.batch_size_obs_norm = BatchObsNormView(self._e)
        self.is_continue = is_continue
        self.actor_lr, self.critic_lr = actor_lr, critic_lr
        self._rng = rng
        self._act_dim = action_dim
        self.last_trace = None

    def select_action(self, obs):
        """a ~ N(mean, std), per-dimension log-prob; eps from torch's generator like
        `Normal.sample()` (PPO_with_tricks.py:234-255).  Returns (action[A], log_pi[A])."""
        if not self.is_continue:
            # Categorical(probs).sample() draws `empty(1, nA).exponential_(1)` and takes argmax(p / q)
            q = torch.empty(1, self._act_dim).exponential_(1).numpy()
            a, lp = self._e.act(0, N.ACT_CAT_SAMPLE, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), eps=q,
                                want_logp=True)
            return np.int64(a[0, 0, 0]), np.float32(lp[0, 0, 0])
        if self.actor_dist["Beta"]:
            # alpha, beta come from the GPU forward; the Beta variate itself is torch's own sampler on the host (its gamma
            # rejection sampler consumes the generator in C++ and cannot be fed injected draws): one [1, A] draw per env step
            al, be = self.agent.actor.alpha_beta(np.asarray(obs, dtype=np.float32).reshape(1, -1))
            dist = torch.distributions.Beta(al, be)
            act = dist.sample()
            return act.numpy().squeeze(0), dist.log_prob(act).numpy().squeeze(0)      # action in [0,1] (:243-247)
        eps = torch.randn(1, self._act_dim).numpy()
        a, lp = self._e.act(0, N.ACT_PPO_SAMPLE, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), eps=eps,
                            out_dim=self._act_dim, want_logp=True)
        return a[0, 0], lp[0, 0]

    def evaluate_action(self, obs):                                       # the mean / argmax prob (:257-270)
        if not self.is_continue:
            return np.int64(self._e.act(0, N.ACT_ARGMAX, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1),
                                        normalize=False)[0, 0, 0])
        if self.actor_dist["Beta"]:                                       # mean alpha/(alpha+beta) mapped to [-1,1] (:259-261)
            al, be = self.agent.actor.alpha_beta(np.asarray(obs, dtype=np.float32).reshape(1, -1))
            return (2 * (al / (al + be) - 0.5)).numpy().squeeze(0)
        return self._e.act(0, N.ACT_TANHHEAD, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), out_dim=self._act_dim,
                           normalize=False)[0, 0]          # no Batch_ObsNorm in evaluate_action (:257-270)

    def add(self, obs, action, reward, next_obs, done, action_log_pi, adv_dones):
        self.buffer.add(obs, action, reward, next_obs, done, action_log_pi, adv_dones)

    def learn(self, minibatch_size, gamma, lmbda, clip_param, K_epochs, entropy_coefficient):
        perms = None
        if self._rng != "device":                                           # np.random.permutation per epoch (:320)
            perms = np.stack([np.random.permutation(self.horizon) for _ in range(K_epochs)])[None]
        out = self._e.ppo_learn(self.horizon, minibatch_size, K_epochs, gamma=gamma, lmbda=lmbda, clip=clip_param,
                                ent_coef=entropy_coefficient, actor_lr=self.agent.actor_optimizer.lr,
                                critic_lr=self.agent.critic_optimizer.lr,
                                adam_eps=1e-6 if self._cautious else self.agent.actor_optimizer.param_groups[0]["eps"],
                                optimizer=1 if self._cautious else 0,
                                adv_norm=self.trick["adv_norm"], perms=perms,
                                want_trace=getattr(self, "track_loss", False))
        self.last_trace = out.get("trace")

    def lr_decay(self, episode_num, max_episodes):                        # :357-363
        lr_a_now = self.actor_lr * (1 - episode_num / max_episodes)
        lr_c_now = self.critic_lr * (1 - episode_num / max_episodes)
        for p in self.agent.actor_optimizer.param_groups:
            p["lr"] = lr_a_now
        for p in self.agent.critic_optimizer.param_groups:
            p["lr"] = lr_c_now

    def save(self, model_dir):
        torch.save(self.agent.actor.state_dict(), os.path.join(model_dir, "PPO.pt"))

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick, beta):
        policy = PPO(dim_info, is_continue, 0, 0, 2, device=torch.device("cpu"), trick=trick, beta=beta)
        policy.agent.actor.load_state_dict(torch.load(os.path.join(model_dir, "PPO.pt")))
        return policy
