"""`PPO` of PPO_advance/PPO_2.py:152-303: the critic's value of every step is stored at rollout time and the advantages /
returns come from stable-baselines3's `compute_returns_and_advantage` (PPO_advance/Buffer.py:480-507).

    PPO(dim_info, is_continue, actor_lr, critic_lr, horizon, device, trick=None)
    action, log_pi, value = policy.select_action(obs)
    policy.add(obs, action, reward, next_obs, terminated, log_pi, done, value)
    policy.learn(minibatch_size, gamma, lmbda, clip_param, K_epochs, entropy_coefficient, last_value)

Nets, optimisers (two torch Adams, eps 1e-8, clip 0.5) and the minibatch steps are PPO_file/PPO_with_tricks.py's with every
trick off, so the engine's persistent update kernel is reused unchanged; only the front of `learn()` differs: no value
pass, a float64 scan over the stored values on the device (frl_ppo_learn, gae_mode 1).
"""
import numpy as np
import torch

from . import _native as N
from .Buffer import Buffer_for_PPO_2
from .PPO import PPO as _PPO


class PPO(_PPO):
    _buffer_cls = Buffer_for_PPO_2

    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, horizon, device, trick=None, **kw):
        super().__init__(dim_info, is_continue, actor_lr, critic_lr, horizon, device, trick={}, beta=False, **kw)
        self.trick = trick                                # stored, never read by the class (PPO_2.py:163)

    def _value(self, obs):
        return self._e.act(1, N.ACT_RAW, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), out_dim=1)[0, 0]

    def select_action(self, obs):
        """-> (action, log_pi, value): the critic runs first (PPO_2.py:167), then the actor's sample."""
        value = self._value(obs)                          # shape (1,), like `value.detach().cpu().numpy().squeeze(0)` (:180)
        action, log_pi = super().select_action(obs)
        return action, log_pi, value

    def add(self, obs, action, reward, next_obs, terminated, action_log_pi, dones, value):
        self.buffer.add(obs, action, reward, next_obs, terminated, action_log_pi, dones, value)

    def learn(self, minibatch_size, gamma, lmbda, clip_param, K_epochs, entropy_coefficient, last_value):
        perms = None
        if self._rng != "device":                           # np.random.permutation per epoch (:255)
            perms = np.stack([np.random.permutation(self.horizon) for _ in range(K_epochs)])[None]
        out = self._e.ppo_learn(self.horizon, minibatch_size, K_epochs, gamma=gamma, lmbda=lmbda, clip=clip_param,
                                ent_coef=entropy_coefficient, actor_lr=self.agent.actor_optimizer.lr,
                                critic_lr=self.agent.critic_optimizer.lr, adam_eps=1e-8, optimizer=0, adv_norm=False,
                                perms=perms, want_trace=getattr(self, "track_loss", False), want_adv=True,
                                last_value=float(np.asarray(last_value).reshape(-1)[0]))
        self.last_trace = out.get("trace")
        self.buffer.advantages, self.buffer.returns = out["adv"][0], out["v_target"][0]

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None):
        import os
        policy = PPO(dim_info, is_continue, 0, 0, 2, device=torch.device("cpu"), trick=trick)
        policy.agent.actor.load_state_dict(torch.load(os.path.join(model_dir, "PPO.pt")))
        return policy
