"""`TD3` and `DDPG` with the reference's class surface, backed by the HIP engine.

TD3  <- TD3_file/TD3.py:123-256   TD3(dim_info, is_continue, actor_lr, critic_lr, buffer_size, device,
                                      trick=None, realize=None)
DDPG <- DDPG_file/DDPG_simple.py:76-179 (+ DDPG.py's `supplement['weight_decay']`, DDPG.py:131-134)
"""
import os

import numpy as np
import torch

from . import _native as N
from ._core import (BatchObsNormView, DeviceNet, Engine, OptimizerView, draw_indices, host_draw, init_layers, init_layers_ddpg,
                    resolve_device)
from .Buffer import Buffer


def actor_layers(obs_dim, action_dim, hidden):
    return [("l1", hidden, obs_dim), ("l2", hidden, hidden), ("l3", action_dim, hidden)]


def critic_layers(in_dim, hidden, twin):
    ls = [("l1", hidden, in_dim), ("l2", hidden, hidden), ("l3", 1, hidden)]
    if twin:
        ls += [("l4", hidden, in_dim), ("l5", hidden, hidden), ("l6", 1, hidden)]
    return ls


class Agent:
    """Agent (TD3.py:123-147): actor, critic (Critic_TD3 when clip_double), Adam each, targets = deepcopy."""

    def __init__(self, engine, obs_dim, action_dim, dim_info, actor_lr, critic_lr, twin, hidden, critic_wd=0.0,
                 net_init=False):
        al, cl = actor_layers(obs_dim, action_dim, hidden), critic_layers(sum(dim_info), hidden, twin)
        init = init_layers_ddpg if net_init else init_layers
        fa = init(al)                   # torch RNG order: actor l1..l3, then critic l1..l3[,l4..l6] (TD3.py:125-129)
        fc = init(cl)
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            engine.set_params(0, fa, kind)
            engine.set_params(1, fc, kind)
        self.actor = DeviceNet(engine, 0, al, act_mode=N.ACT_TANHHEAD)
        self.critic = DeviceNet(engine, 1, cl)
        self.actor_target = DeviceNet(engine, 0, al, kind=N.PARAM_TARGET, act_mode=N.ACT_TANHHEAD)
        self.critic_target = DeviceNet(engine, 1, cl, kind=N.PARAM_TARGET)
        self.actor_optimizer = OptimizerView(engine, 0, actor_lr)
        self.critic_optimizer = OptimizerView(engine, 1, critic_lr, weight_decay=critic_wd)

    def update_actor(self, loss):
        raise NotImplementedError("zero_grad/backward/clip/step are fused into learn() on the GPU")

    update_critic = update_actor


class TD3:
    _ALGO, _FILE = N.ALGO_TD3, "TD3.pt"

    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick=None, realize=None, *,
                 rng="auto", hidden=128, batch_max=1024, seed=0, critic_weight_decay=0.0, net_init=False,
                 batch_obs_norm=False):
        obs_dim, action_dim = dim_info
        if not is_continue:
            raise ValueError("the discrete branch of TD3.select_action is dead code in the reference (TD3.py:169)")
        self.realize = realize if realize is not None else {"clip_double": True, "policy_noise": True, "twin_delay": True}
        hip_id, self.device = resolve_device(device)
        twin = bool(self.realize["clip_double"])
        self._e = Engine(self._ALGO, obs_dim, action_dim, max(int(buffer_size), 1), twin_critic=twin, hidden=hidden,
                         batch_max=batch_max, device_id=hip_id, seed=seed)
        self.agent = Agent(self._e, obs_dim, action_dim, dim_info, actor_lr, critic_lr, twin, hidden, critic_weight_decay,
                           net_init)
        if batch_obs_norm:                                              # DDPG.py:160-161
            self._e.obsnorm_enable(True)
            self.batch_size_obs_norm = BatchObsNormView(self._e)
        self.buffer = Buffer(buffer_size, obs_dim, act_dim=action_dim, device=self.device, _engine=self._e)
        self.is_continue = is_continue
        self.trick = trick
        self.total_it = 0
        self._rng = rng
        self._act_dim = action_dim
        self.last_losses = None

    def select_action(self, obs):
        """actor(obs) in (-1,1), float32 [action_dim] (TD3.py:163-170)."""
        return self._e.act(0, N.ACT_TANHHEAD, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), out_dim=self._act_dim)[0, 0]

    def evaluate_action(self, obs):
        """deterministic policy; DDPG.py:173-181 does NOT apply Batch_ObsNorm here (select_action does)."""
        return self._e.act(0, N.ACT_TANHHEAD, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), out_dim=self._act_dim,
                           normalize=False)[0, 0]

    def add(self, obs, action, reward, next_obs, done):
        self.buffer.add(obs, action, reward, next_obs, done)

    def sample(self, batch_size):
        return self.buffer.sample(draw_indices(len(self.buffer), batch_size))

    def learn(self, batch_size, gamma, tau, policy_noise, noise_clip, max_action, policy_freq, policy_noise_scale):
        self.total_it += 1                                              # TD3.py:191
        total = len(self.buffer)
        batch = min(total, batch_size)
        use_noise = bool(self.realize["policy_noise"])
        idx = noise = None
        if host_draw(self._rng, total, batch_size):
            idx = draw_indices(total, batch_size)                       # np.random.choice (TD3.py:183)
            if use_noise:                                               # torch.randn_like(actions) (TD3.py:197)
                noise = np.zeros((1, 1, 2, batch, self._act_dim), np.float32)
                noise[0, 0, 0] = torch.randn(batch, self._act_dim).numpy()
        if not self.realize["twin_delay"]:
            policy_freq = 1                                             # TD3.py:219-222
        do_actor = self.total_it % policy_freq == 0
        st = self._e.learn(batch, gamma=gamma, tau=tau, actor_lr=self.agent.actor_optimizer.lr,
                           critic_lr=self.agent.critic_optimizer.lr,
                           critic_weight_decay=self.agent.critic_optimizer.param_groups[0]["weight_decay"],
                           do_actor=do_actor, use_policy_noise=use_noise, policy_noise=policy_noise,
                           noise_clip=noise_clip, max_action=max_action, policy_noise_scale=policy_noise_scale,
                           idx=idx, noise=noise, want_stats=getattr(self, "track_loss", False))
        if st is not None:
            self.last_losses = (float(st[0, 0, N.STAT_CRITIC_LOSS]), float(st[0, 0, N.STAT_ACTOR_LOSS]) if do_actor else None)

    def update_target(self, tau):
        for net in (1, 0):                                              # critic then actor (TD3.py:243-244)
            q, t = self._e.get_params(net, N.PARAM_ONLINE), self._e.get_params(net, N.PARAM_TARGET)
            self._e.set_params(net, t * np.float32(1.0 - tau) + q * np.float32(tau), N.PARAM_TARGET)

    def save(self, model_dir):
        torch.save(self.agent.actor.state_dict(), os.path.join(model_dir, self._FILE))

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None, realize=None):
        policy = TD3(dim_info, is_continue, 0, 0, 0, device=torch.device("cpu"), trick=trick, realize=realize)
        policy.agent.actor.load_state_dict(torch.load(os.path.join(model_dir, "TD3.pt")))
        return policy


class DDPG(TD3):
    """DDPG_simple (DDPG_simple.py:100-179): single critic, no target-policy noise, actor and
    targets updated on every call.  `supplement={'weight_decay': True}` adds DDPG.py's critic
    Adam weight_decay 1e-3."""
    _ALGO, _FILE = N.ALGO_DDPG, "DDPG.pt"

    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick=None, supplement=None, **kw):
        sup = supplement or {}
        self.supplement = supplement
        # DDPG.py supplements: weight_decay (critic Adam, :131-134), net_init (:78-86), Batch_ObsNorm
        # (:160-161,190-192); OUNoise / ObsNorm live in the caller's loop (freerl_amd.train)
        super().__init__(dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick=trick,
                         realize={"clip_double": False, "policy_noise": False, "twin_delay": False},
                         critic_weight_decay=1e-3 if sup.get("weight_decay") else 0.0,
                         net_init=bool(sup.get("net_init")), batch_obs_norm=bool(sup.get("Batch_ObsNorm")), **kw)

    def learn(self, batch_size, gamma, tau):                            # DDPG_simple.py:137-156
        super().learn(batch_size, gamma, tau, 0.0, 0.0, 1.0, 1, 1.0)

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None):
        policy = DDPG(dim_info, is_continue, 0, 0, 0, device=torch.device("cpu"), trick=trick)
        policy.agent.actor.load_state_dict(torch.load(os.path.join(model_dir, "DDPG.pt")))
        return policy
