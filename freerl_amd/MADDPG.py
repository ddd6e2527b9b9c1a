"""`MADDPG` with the reference's class surface (MADDPG_file/MADDPG_simple.py:82-210), backed by
the HIP engine: all agents' nets and ONE joint replay ring live in one engine; `learn` updates
every agent's critic and actor in one launch chain (the per-agent updates are independent given
the pre-update target nets), then the soft updates."""
import os

import numpy as np
import torch

from . import _native as N
from ._core import DeviceNet, Engine, OptimizerView, F32, init_layers, resolve_device
from .Buffer import Buffer
from .TD3 import actor_layers, critic_layers


class Agent:
    def __init__(self, engine, j, obs_dim, action_dim, total_dim, actor_lr, critic_lr, hidden):
        al, cl = actor_layers(obs_dim, action_dim, hidden), critic_layers(total_dim, hidden, False)
        fa, fc = init_layers(al), init_layers(cl)        # torch RNG: this agent's actor then critic (:111-112)
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            engine.set_params(2 * j, fa, kind)
            engine.set_params(2 * j + 1, fc, kind)
        self.actor = DeviceNet(engine, 2 * j, al, act_mode=N.ACT_TANHHEAD)
        self.critic = DeviceNet(engine, 2 * j + 1, cl)
        self.actor_target = DeviceNet(engine, 2 * j, al, kind=N.PARAM_TARGET, act_mode=N.ACT_TANHHEAD)
        self.critic_target = DeviceNet(engine, 2 * j + 1, cl, kind=N.PARAM_TARGET)
        self.actor_optimizer = OptimizerView(engine, 2 * j, actor_lr)
        self.critic_optimizer = OptimizerView(engine, 2 * j + 1, critic_lr)


class MADDPG:
    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick=None, *, rng="host",
                 hidden=128, batch_max=1024, seed=0):
        if not is_continue:
            raise ValueError("only continuous actions are implemented in the reference (MADDPG_simple.py:126)")
        self.agent_ids = list(dim_info.keys())
        od = [dim_info[a][0] for a in self.agent_ids]
        ad = [dim_info[a][1] for a in self.agent_ids]
        hip_id, self.device = resolve_device(device)
        self._e = Engine(N.ALGO_MADDPG, od, ad, max(int(buffer_size), 1), hidden=hidden, batch_max=batch_max,
                         device_id=hip_id, seed=seed)
        total = sum(od) + sum(ad)
        self.agents, self.buffers = {}, {}
        for j, aid in enumerate(self.agent_ids):
            self.agents[aid] = Agent(self._e, j, od[j], ad[j], total, actor_lr, critic_lr, hidden)
            self.buffers[aid] = Buffer(buffer_size, od[j], ad[j], self.device, _engine=self._e, _agent=j)
        self.is_continue = is_continue
        self.agent_x = self.agent_ids[0]
        self.regular = False
        self._rng = rng
        self._ad = ad
        self._rec = np.zeros(self._e.width, dtype=F32)
        self.last_losses = None

    def select_action(self, obs):
        actions = {}
        for j, aid in enumerate(self.agent_ids):
            o = np.asarray(obs[aid], dtype=np.float32).reshape(1, 1, -1)
            actions[aid] = self._e.act(2 * j, N.ACT_TANHHEAD, o, out_dim=self._ad[j])[0, 0]
        return actions

    def evaluate_action(self, obs):
        return self.select_action(obs)

    def add(self, obs, action, reward, next_obs, done):
        """Every agent's buffer is written in lock-step (MADDPG_simple.py:143-145): one joint record."""
        lay, r = self._e.layout, self._rec
        for j, aid in enumerate(self.agent_ids):
            r[lay.obs_off[j]:lay.obs_off[j] + lay.obs_dim[j]] = np.asarray(obs[aid], dtype=F32).reshape(-1)
            r[lay.act_off[j]:lay.act_off[j] + lay.act_dim[j]] = np.asarray(action[aid], dtype=F32).reshape(-1)
            r[lay.rew_off + j] = reward[aid]
            r[lay.done_off + j] = float(done[aid])
            r[lay.next_obs_off[j]:lay.next_obs_off[j] + lay.obs_dim[j]] = np.asarray(next_obs[aid], dtype=F32).reshape(-1)
        self._e.add(0, r)

    def sample(self, batch_size):
        """(obs, action, reward, next_obs, done, next_action) dicts for one index draw (:147-157)."""
        total = len(self.buffers[self.agent_x])
        indices = np.random.choice(total, batch_size, replace=False)
        obs, action, reward, next_obs, done, next_action = {}, {}, {}, {}, {}, {}
        for aid in self.agent_ids:
            obs[aid], action[aid], reward[aid], next_obs[aid], done[aid] = self.buffers[aid].sample(indices)
            next_action[aid] = self.agents[aid].actor_target(next_obs[aid])
        return obs, action, reward, next_obs, done, next_action

    def learn(self, batch_size, gamma, tau):
        total = len(self.buffers[self.agent_x])
        idx = None
        if self._rng == "host":        # one np.random.choice PER AGENT, in agent order (:169, :149)
            idx = np.stack([np.random.choice(total, batch_size, replace=False) for _ in self.agent_ids])[None]
        a0 = self.agents[self.agent_x]
        st = self._e.learn(batch_size, gamma=gamma, tau=tau, actor_lr=a0.actor_optimizer.lr,
                           critic_lr=a0.critic_optimizer.lr, idx=idx, want_stats=getattr(self, "track_loss", False))
        if st is not None:
            self.last_losses = {aid: (float(st[0, j, N.STAT_CRITIC_LOSS]), float(st[0, j, N.STAT_ACTOR_LOSS]))
                                for j, aid in enumerate(self.agent_ids)}

    def update_target(self, tau):
        for net in range(self._e.n_nets):
            q, t = self._e.get_params(net, N.PARAM_ONLINE), self._e.get_params(net, N.PARAM_TARGET)
            self._e.set_params(net, t * np.float32(1.0 - tau) + q * np.float32(tau), N.PARAM_TARGET)

    def save(self, model_dir):
        torch.save({name: agent.actor.state_dict() for name, agent in self.agents.items()},
                   os.path.join(model_dir, "MADDPG.pth"))

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None):
        policy = MADDPG(dim_info, is_continue=is_continue, actor_lr=0, critic_lr=0, buffer_size=0, device="cpu")
        data = torch.load(os.path.join(model_dir, "MADDPG.pth"))
        for agent_id, agent in policy.agents.items():
            agent.actor.load_state_dict(data[agent_id])
        return policy
