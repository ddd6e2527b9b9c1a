"""`MADDPG` with the reference's class surface (MADDPG_file/MADDPG_simple.py:82-210; with `supplement`
MADDPG_file/MADDPG.py:60-290), backed by the HIP engine: all agents' nets and ONE joint replay ring live in one
engine; `learn` updates every agent's critic and actor in one launch chain (the per-agent updates are
independent given the pre-update target nets), then the soft updates.  `MATD3` (MADDPG_file/MATD3_simple.py:151-262)
adds twin critics, target policy smoothing and the delayed actor / target update."""
import os

import numpy as np
import torch

from . import _native as N
from ._core import BatchObsNormView, DeviceNet, Engine, OptimizerView, F32, host_draw, init_layers, init_layers_ddpg, resolve_device
from .Buffer import Buffer
from .TD3 import actor_layers, critic_layers


class Agent:
    def __init__(self, engine, j, obs_dim, action_dim, total_dim, actor_lr, critic_lr, hidden, twin=False, net_init=False,
                 weight_decay=0.0):
        al, cl = actor_layers(obs_dim, action_dim, hidden), critic_layers(total_dim, hidden, twin)
        init = init_layers_ddpg if net_init else init_layers      # MADDPG.py:66-71,87-93 (same rule as DDPG.py)
        fa, fc = init(al), init(cl)                      # torch RNG: this agent's actor then critic (:111-112)
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            engine.set_params(2 * j, fa, kind)
            engine.set_params(2 * j + 1, fc, kind)
        self.actor = DeviceNet(engine, 2 * j, al, act_mode=N.ACT_TANHHEAD)
        self.critic = DeviceNet(engine, 2 * j + 1, cl)
        self.actor_target = DeviceNet(engine, 2 * j, al, kind=N.PARAM_TARGET, act_mode=N.ACT_TANHHEAD)
        self.critic_target = DeviceNet(engine, 2 * j + 1, cl, kind=N.PARAM_TARGET)
        self.actor_optimizer = OptimizerView(engine, 2 * j, actor_lr)
        self.critic_optimizer = OptimizerView(engine, 2 * j + 1, critic_lr, weight_decay=weight_decay)


class MADDPG:
    _twin = False

    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick=None, supplement=None, *,
                 rng="auto", hidden=128, batch_max=1024, seed=0):
        if not is_continue:
            raise ValueError("only continuous actions are implemented in the reference (MADDPG_simple.py:126)")
        sup = dict(supplement or {})
        self.supplement = supplement
        self._wd = 1e-3 if sup.get("weight_decay") else 0.0      # MADDPG.py:118-121: critic Adam weight_decay
        self.agent_ids = list(dim_info.keys())
        od = [dim_info[a][0] for a in self.agent_ids]
        ad = [dim_info[a][1] for a in self.agent_ids]
        hip_id, self.device = resolve_device(device)
        self._e = Engine(N.ALGO_MADDPG, od, ad, max(int(buffer_size), 1), hidden=hidden, batch_max=batch_max,
                         device_id=hip_id, seed=seed, twin_critic=self._twin)
        total = sum(od) + sum(ad)
        self.agents, self.buffers = {}, {}
        for j, aid in enumerate(self.agent_ids):
            self.agents[aid] = Agent(self._e, j, od[j], ad[j], total, actor_lr, critic_lr, hidden, twin=self._twin,
                                     net_init=bool(sup.get("net_init")), weight_decay=self._wd)
            self.buffers[aid] = Buffer(buffer_size, od[j], ad[j], self.device, _engine=self._e, _agent=j)
        if sup.get("Batch_ObsNorm"):                       # MADDPG.py:155-156: one Normalization_batch_size per agent
            self._e.obsnorm_enable(True)
            self.batch_size_obs_norm = {aid: BatchObsNormView(self._e, j) for j, aid in enumerate(self.agent_ids)}
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

    def evaluate_action(self, obs):                     # no Batch_ObsNorm here (MADDPG.py:173-180), unlike select_action (:162-163)
        actions = {}
        for j, aid in enumerate(self.agent_ids):
            o = np.asarray(obs[aid], dtype=np.float32).reshape(1, 1, -1)
            actions[aid] = self._e.act(2 * j, N.ACT_TANHHEAD, o, out_dim=self._ad[j], normalize=False)[0, 0]
        return actions

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
        if host_draw(self._rng, total, batch_size):        # one np.random.choice PER AGENT, in agent order (:169, :149)
            idx = np.stack([np.random.choice(total, batch_size, replace=False) for _ in self.agent_ids])[None]
        a0 = self.agents[self.agent_x]
        st = self._e.learn(batch_size, gamma=gamma, tau=tau, actor_lr=a0.actor_optimizer.lr,
                           critic_lr=a0.critic_optimizer.lr, critic_weight_decay=self._wd, idx=idx,
                           want_stats=getattr(self, "track_loss", False))
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


class MATD3(MADDPG):
    """MADDPG_file/MATD3_simple.py:151-262.  `realize` = {'clip_double','policy_noise','twin_delay'}; the reference's
    Agent always builds Critic_TD3 (:136), so `clip_double` False fails there on the tuple it returns (:226) — same here."""
    _twin = True

    def __init__(self, dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick=None, realize=None, *,
                 rng="auto", hidden=128, batch_max=1024, seed=0):
        super().__init__(dim_info, is_continue, actor_lr, critic_lr, buffer_size, device, trick, rng=rng, hidden=hidden,
                         batch_max=batch_max, seed=seed)
        self.realize = realize
        self.total_it = 0

    def sample(self, batch_size, policy_noise_scale, policy_noise, noise_clip, max_action):       # :193-207
        total = len(self.buffers[self.agent_x])
        indices = np.random.choice(total, batch_size, replace=False)
        obs, action, reward, next_obs, done, next_action = {}, {}, {}, {}, {}, {}
        for aid in self.agent_ids:
            obs[aid], action[aid], reward[aid], next_obs[aid], done[aid] = self.buffers[aid].sample(indices)
            na = self.agents[aid].actor_target(next_obs[aid])
            if self.realize["policy_noise"]:
                noise = (policy_noise_scale * (torch.randn_like(action[aid].cpu()) * policy_noise)).clamp(-noise_clip, noise_clip)
                na = (na * max_action + noise).clamp(-max_action, max_action) / max_action
            next_action[aid] = na
        return obs, action, reward, next_obs, done, next_action

    def learn(self, batch_size, gamma, tau, policy_noise_scale, policy_noise, noise_clip, max_action, policy_freq):
        if not self.realize["clip_double"]:
            raise TypeError("unsupported operand: the reference multiplies the (Q1, Q2) tuple of Critic_TD3 (MATD3_simple.py:226-229)")
        self.total_it += 1
        if not self.realize["twin_delay"]:
            policy_freq = 1
        n, total = len(self.agent_ids), len(self.buffers[self.agent_x])
        idx = noise = None
        if host_draw(self._rng, total, batch_size):
            # the reference's draw order: per updating agent i one np.random.choice, then one randn_like per agent j
            am = max(self._ad)
            idx = np.zeros((1, n, batch_size), np.int64)
            noise = np.zeros((1, n, max(2, n), batch_size, am), F32)
            for i in range(n):
                idx[0, i] = np.random.choice(total, batch_size, replace=False)
                if self.realize["policy_noise"]:
                    for j in range(n):
                        noise[0, i, j, :, :self._ad[j]] = torch.randn(batch_size, self._ad[j]).numpy()
        a0 = self.agents[self.agent_x]
        st = self._e.learn(batch_size, gamma=gamma, tau=tau, actor_lr=a0.actor_optimizer.lr, critic_lr=a0.critic_optimizer.lr,
                           use_policy_noise=bool(self.realize["policy_noise"]), policy_noise=policy_noise,
                           noise_clip=noise_clip, max_action=max_action, policy_noise_scale=policy_noise_scale,
                           do_actor=(self.total_it % policy_freq == 0), idx=idx, noise=noise,
                           want_stats=getattr(self, "track_loss", False))
        if st is not None:
            self.last_losses = {aid: (float(st[0, j, N.STAT_CRITIC_LOSS]), float(st[0, j, N.STAT_ACTOR_LOSS]))
                                for j, aid in enumerate(self.agent_ids)}

    @staticmethod
    def load(dim_info, is_continue, model_dir, trick=None):          # the file name stays 'MADDPG.pth' (MATD3_simple.py:271)
        policy = MATD3(dim_info, is_continue=is_continue, actor_lr=0, critic_lr=0, buffer_size=0, device="cpu")
        data = torch.load(os.path.join(model_dir, "MADDPG.pth"))
        for agent_id, agent in policy.agents.items():
            agent.actor.load_state_dict(data[agent_id])
        return policy
