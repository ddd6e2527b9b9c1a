"""`SAC` with the reference's class surface (SAC_file/SAC.py:129-282), backed by the HIP engine.

    SAC(dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick)   # trick dict required (SAC.py:181)
"""
import os

import numpy as np
import torch

from . import _native as N
from ._core import BatchObsNormView, DeviceNet, Engine, OptimizerView, draw_indices, host_draw, init_layers, resolve_device
from .Buffer import Buffer
from .TD3 import critic_layers


class Agent:
    def __init__(self, engine, obs_dim, action_dim, dim_info, actor_lr, critic_lr, hidden):
        al = [("l1", hidden, obs_dim), ("l2", hidden, hidden), ("mean_layer", action_dim, hidden)]
        cl = critic_layers(sum(dim_info), hidden, True)
        fa = np.concatenate([init_layers(al), np.zeros(action_dim, np.float32)])   # log_std = zeros (SAC.py:66)
        fc = init_layers(cl)
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            engine.set_params(0, fa, kind)
            engine.set_params(1, fc, kind)
        extra = ("log_std", (1, action_dim))
        self.actor = DeviceNet(engine, 0, al, extra=extra, act_mode=N.ACT_TANHHEAD)
        self.critic = DeviceNet(engine, 1, cl)
        self.actor_target = DeviceNet(engine, 0, al, extra=extra, kind=N.PARAM_TARGET, act_mode=N.ACT_TANHHEAD)
        self.critic_target = DeviceNet(engine, 1, cl, kind=N.PARAM_TARGET)
        self.actor_optimizer = OptimizerView(engine, 0, actor_lr)
        self.critic_optimizer = OptimizerView(engine, 1, critic_lr)

    def update_actor(self, loss):
        raise NotImplementedError("zero_grad/backward/clip/step are fused into learn() on the GPU")

    update_critic = update_actor


class Alpha:
    """Alpha (SAC.py:154-169): log_alpha scalar + Adam(lr 1e-4); `.alpha` reads the engine's value."""

    def __init__(self, engine, action_dim, alpha_lr=0.0001, alpha=0.2):
        self._e = engine
        self.alpha_lr = alpha_lr
        self.target_entropy = -action_dim                                # SAC.py:160
        engine.set_alpha_state([np.log(alpha), 0.0, 0.0, alpha], 0)

    @property
    def log_alpha(self):
        return torch.tensor(float(self._e.alpha_state()[0][0]))

    @property
    def alpha(self):
        return torch.tensor(float(self._e.alpha_state()[0][3]))


class SAC:
    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick=None, *, rng="auto",
                 hidden=128, batch_max=1024, seed=0):
        obs_dim, action_dim = dim_info
        if not is_continue:
            raise NotImplementedError("SAC_add_discrete.py is out of scope (SURVEY.md §2.1)")
        if trick is None:
            raise TypeError("SAC needs the `trick` dict (the reference indexes it, SAC.py:181)")
        hip_id, self.device = resolve_device(device)
        self._e = Engine(N.ALGO_SAC, obs_dim, action_dim, max(int(buffer_size), 1), twin_critic=True, hidden=hidden,
                         batch_max=batch_max, device_id=hip_id, seed=seed)
        self.agent = Agent(self._e, obs_dim, action_dim, dim_info, actor_lr, critic_lr, hidden)
        self.buffer = Buffer(buffer_size, obs_dim, act_dim=action_dim, device=self.device, _engine=self._e)
        self.is_continue = is_continue
        self.trick = trick
        if trick.get("Batch_ObsNorm"):                                   # SAC.py:181-182
            self._e.obsnorm_enable(True)
            self.batch_size_obs_norm = BatchObsNormView(self._e)
        self.adaptive_alpha = True                                       # SAC.py:185
        self.alphas = Alpha(self._e, action_dim, alpha=0.01)             # SAC.py:188
        self._rng = rng
        self._act_dim = action_dim
        self.last_losses = None

    def select_action(self, obs):
        """tanh(mean + std*eps), eps from torch's generator like Normal.rsample (SAC.py:192-198)."""
        eps = torch.randn(1, self._act_dim).numpy() if self._rng != "device" else \
            np.random.default_rng().standard_normal((1, self._act_dim)).astype(np.float32)
        return self._e.act(0, N.ACT_SAC_SAMPLE, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), eps=eps,
                           out_dim=self._act_dim)[0, 0]

    def evaluate_action(self, obs):                                      # tanh(mean) (SAC.py:200-204)
        return self._e.act(0, N.ACT_TANHHEAD, np.asarray(obs, dtype=np.float32).reshape(1, 1, -1), out_dim=self._act_dim,
                           normalize=False)[0, 0]        # the reference does not apply Batch_ObsNorm here

    def add(self, obs, action, reward, next_obs, done):
        self.buffer.add(obs, action, reward, next_obs, done)

    def sample(self, batch_size):
        return self.buffer.sample(draw_indices(len(self.buffer), batch_size))

    def learn(self, batch_size, gamma, tau):                             # SAC.py:222-260
        total = len(self.buffer)
        batch = min(total, batch_size)
        idx = noise = None
        if host_draw(self._rng, total, batch_size):
            idx = draw_indices(total, batch_size)
            noise = np.zeros((1, 1, 2, batch, self._act_dim), np.float32)
            noise[0, 0, 0] = torch.randn(batch, self._act_dim).numpy()   # actor_target rsample (SAC.py:227)
            noise[0, 0, 1] = torch.randn(batch, self._act_dim).numpy()   # actor rsample (SAC.py:244)
        st = self._e.learn(batch, gamma=gamma, tau=tau, actor_lr=self.agent.actor_optimizer.lr,
                           critic_lr=self.agent.critic_optimizer.lr, alpha_lr=self.alphas.alpha_lr,
                           target_entropy=float(self.alphas.target_entropy), idx=idx, noise=noise,
                           want_stats=getattr(self, "track_loss", False))
        if st is not None:
            self.last_losses = tuple(float(st[0, 0, k]) for k in (N.STAT_CRITIC_LOSS, N.STAT_ACTOR_LOSS, N.STAT_ALPHA_LOSS))

    def update_target(self, tau):
        for net in (1, 0):
            q, t = self._e.get_params(net, N.PARAM_ONLINE), self._e.get_params(net, N.PARAM_TARGET)
            self._e.set_params(net, t * np.float32(1.0 - tau) + q * np.float32(tau), N.PARAM_TARGET)

    def save(self, model_dir):
        torch.save(self.agent.actor.state_dict(), os.path.join(model_dir, "SAC.pt"))

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None):
        policy = SAC(dim_info, is_continue, 0, 0, 0, device=torch.device("cpu"), trick=trick)
        policy.agent.actor.load_state_dict(torch.load(os.path.join(model_dir, "SAC.pt")))
        return policy
