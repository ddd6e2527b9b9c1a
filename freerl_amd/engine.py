"""Thin object wrapper over the C ABI (include/freerl_hip.h): one `Engine` = one `frl_engine`.

Host-side plumbing only (NumPy arrays in, NumPy arrays out); all compute runs in the HIP
library.  The reference-shaped classes (`freerl_amd.DQN`, `.TD3`, ... `.Buffer`) sit on top.
"""
import ctypes as C

import numpy as np

from . import _native as N

F32 = np.float32


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Engine:
    def __init__(self, algo, obs_dim, act_dim, capacity, *, n_learners=1, discrete=False, hidden=128,
                 hidden_act=N.ACT_RELU, twin_critic=False, batch_max=256, extra_cols=0, device_id=0, seed=0, actor_dist=0, dueling=False, noisy=False, c51=None):
        obs_dim = list(obs_dim) if isinstance(obs_dim, (list, tuple)) else [int(obs_dim)]
        act_dim = list(act_dim) if isinstance(act_dim, (list, tuple)) else [int(act_dim)]
        assert len(obs_dim) == len(act_dim)
        cfg = N.Config()
        cfg.algo, cfg.n_learners, cfg.n_agents = int(algo), int(n_learners), len(obs_dim)
        for j, (o, a) in enumerate(zip(obs_dim, act_dim)):
            cfg.obs_dim[j], cfg.act_dim[j] = int(o), int(a)
        cfg.discrete, cfg.hidden, cfg.hidden_act = int(bool(discrete)), int(hidden), int(hidden_act)
        cfg.twin_critic, cfg.capacity, cfg.batch_max = int(bool(twin_critic)), int(capacity), int(batch_max)
        cfg.extra_cols, cfg.device_id, cfg.seed = int(extra_cols), int(device_id), int(seed) & (2 ** 64 - 1)
        cfg.actor_dist, cfg.dueling, cfg.noisy = int(actor_dist), int(bool(dueling)), int(bool(noisy))
        if c51 is not None:                     # (atoms, v_min, v_max)
            cfg.c51_atoms, cfg.c51_vmin, cfg.c51_vmax = int(c51[0]), float(c51[1]), float(c51[2])
        self._L = N.lib()
        h = C.c_void_p()
        N.check(self._L.frl_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.cfg = cfg
        self.algo = int(algo)
        self.P = int(n_learners)
        self.n_agents = len(obs_dim)
        self.capacity = int(capacity)
        self.batch_max = int(batch_max)
        lay = N.RecordLayout()
        N.check(self._L.frl_record_layout_get(self._h, C.byref(lay)))
        self.layout = lay
        self.width = lay.width
        self.act_max = max(lay.act_dim[j] for j in range(self.n_agents))
        n = C.c_int(0)
        N.check(self._L.frl_net_count(self._h, C.byref(n)))
        self.n_nets = n.value
        self._nparams = {}

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.frl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        N.check(self._L.frl_sync(self._h))

    def lds_bytes(self):
        b, r = C.c_int(0), C.c_int(0)
        N.check(self._L.frl_lds_bytes(self._h, C.byref(b), C.byref(r)))
        return b.value, r.value

    def learn_path(self, batch):
        """(chained, lds_bytes, rows_per_workgroup) of the kernel family learn() launches at this batch size."""
        c, b, r = C.c_int(0), C.c_int(0), C.c_int(0)
        N.check(self._L.frl_learn_path(self._h, int(batch), C.byref(c), C.byref(b), C.byref(r)))
        return bool(c.value), b.value, r.value

    # ------------------------------------------------------------------ replay
    def add(self, learner, record):
        rec = np.ascontiguousarray(record, dtype=F32)
        assert rec.size == self.width, (rec.size, self.width)
        N.check(self._L.frl_buffer_add(self._h, int(learner), _fp(rec)))

    def add_batch(self, records, learners=None):
        recs = np.ascontiguousarray(records, dtype=F32).reshape(-1, self.width)
        lp = None
        if learners is not None:
            ln = np.ascontiguousarray(learners, dtype=np.int32)
            assert ln.size == recs.shape[0]
            lp = ln.ctypes.data_as(C.POINTER(C.c_int))
        N.check(self._L.frl_buffer_add_batch(self._h, recs.shape[0], lp, _fp(recs)))

    def flush(self):
        N.check(self._L.frl_buffer_flush(self._h))

    def cursor(self, learner=0):
        i, s = C.c_int(0), C.c_int(0)
        N.check(self._L.frl_buffer_cursor_get(self._h, int(learner), C.byref(i), C.byref(s)))
        return i.value, s.value

    def set_cursor(self, learner, index, size):
        N.check(self._L.frl_buffer_cursor_set(self._h, int(learner), int(index), int(size)))

    def sample_into(self, learner, idx, fields, out_ptrs):
        """fields: [(col0, ncols)], out_ptrs: device addresses of dense [B][ncols] fp32 tensors."""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        nf = len(fields)
        col0 = (C.c_int * nf)(*[f[0] for f in fields])
        ncols = (C.c_int * nf)(*[f[1] for f in fields])
        outs = (C.c_void_p * nf)(*out_ptrs)
        N.check(self._L.frl_buffer_sample(self._h, int(learner), idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                          idx.size, nf, col0, ncols, outs))

    def read_rows(self, learner, row0, n):
        out = np.empty((n, self.width), dtype=F32)
        if n:
            N.check(self._L.frl_buffer_read(self._h, int(learner), int(row0), int(n), _fp(out)))
        return out

    def fill_synthetic(self, rows, seed=0):
        N.check(self._L.frl_buffer_fill_synthetic(self._h, int(rows), int(seed)))

    # ------------------------------------------------------------------ parameters
    def num_params(self, net):
        if net not in self._nparams:
            n = C.c_int(0)
            N.check(self._L.frl_net_num_params(self._h, int(net), C.byref(n)))
            self._nparams[net] = n.value
        return self._nparams[net]

    def get_params(self, net, kind=N.PARAM_ONLINE, learner=0):
        out = np.empty(self.num_params(net), dtype=F32)
        N.check(self._L.frl_params_get(self._h, int(learner), int(net), int(kind), _fp(out)))
        return out

    def set_params(self, net, flat, kind=N.PARAM_ONLINE, learner=0):
        flat = np.ascontiguousarray(flat, dtype=F32).reshape(-1)
        assert flat.size == self.num_params(net), (flat.size, self.num_params(net))
        N.check(self._L.frl_params_set(self._h, int(learner), int(net), int(kind), _fp(flat)))

    def pad_max(self, net, kind=N.PARAM_ONLINE, learner=0):
        """max |x| over the padding slots of a net's block (must stay 0: frl_params_pad_max)."""
        v = C.c_float(0)
        N.check(self._L.frl_params_pad_max(self._h, int(learner), int(net), int(kind), C.byref(v)))
        return v.value

    def opt_step(self, net, learner=0):
        t = C.c_int(0)
        N.check(self._L.frl_opt_step_get(self._h, int(learner), int(net), C.byref(t)))
        return t.value

    def set_opt_step(self, net, t, learner=0):
        N.check(self._L.frl_opt_step_set(self._h, int(learner), int(net), int(t)))

    def alpha_state(self, learner=0):
        v = np.zeros(4, dtype=F32)
        t = C.c_int(0)
        N.check(self._L.frl_alpha_get(self._h, int(learner), _fp(v), C.byref(t)))
        return v, t.value

    def set_alpha_state(self, vals4, step=0, learner=0):
        v = np.ascontiguousarray(vals4, dtype=F32)
        N.check(self._L.frl_alpha_set(self._h, int(learner), _fp(v), int(step)))

    # ------------------------------------------------------------------ forward
    def act(self, net, mode, obs, *, eps=None, head=0, use_target=False, out_dim=None, want_logp=False, normalize=True):
        """obs [P, n_rows, in_dim] (or [n_rows, in_dim] when P == 1) -> out [P, n_rows, out_dim]."""
        obs = np.ascontiguousarray(obs, dtype=F32)
        if obs.ndim == 2:
            obs = obs[None]
        P, n_rows, in_dim = obs.shape
        assert P == self.P
        od = 1 if mode in (N.ACT_ARGMAX, N.ACT_CAT_SAMPLE) else int(out_dim)
        out = np.empty((P, n_rows, od), dtype=F32)
        logp = np.empty((P, n_rows, od), dtype=F32) if want_logp else None
        ep = None
        if eps is not None:
            eps = np.ascontiguousarray(eps, dtype=F32).reshape(P, n_rows, -1)
            ep = _fp(eps)
        flags = int(mode) | (0 if normalize else N.ACT_NO_OBSNORM)
        N.check(self._L.frl_act(self._h, int(net), flags, int(head), int(use_target), n_rows, in_dim,      # 2: noisy effective set 0
                                _fp(obs), ep, _fp(out), _fp(logp) if want_logp else None))
        return (out, logp) if want_logp else out

    # ------------------------------------------------------------------ learn
    def act_explore(self, mode, obs, *, kind, epsilon=0.0, sigma=0.0, scale=1.0, max_action=1.0, ou_theta=0.15, ou_sigma=0.2,
                    ou_dt=1e-2, ended=None, out_dim=None):
        """select_action + the reference loop's exploration rule in one launch (frl_act_explore): obs [P, rows, obs_dim] ->
        (stored action, env-unit action), each [P, rows, out_dim] ([P, rows] for ACT_ARGMAX).  Draws come from the engine's
        Philox stream; `ended` [P, rows] resets those rows' OU state first."""
        obs = np.ascontiguousarray(obs, dtype=F32)
        assert obs.ndim == 3 and obs.shape[0] == self.P
        rows = obs.shape[1]
        x = N.ExploreArgs()
        x.kind, x.epsilon, x.sigma, x.scale, x.max_action = int(kind), epsilon, sigma, scale, max_action
        x.ou_theta, x.ou_sigma, x.ou_dt = ou_theta, ou_sigma, ou_dt
        disc = mode == N.ACT_ARGMAX
        shape = (self.P, rows) if disc else (self.P, rows, int(out_dim if out_dim is not None else self.act_max))
        store, env = np.empty(shape, F32), np.empty(shape, F32)
        ep = None
        if ended is not None:
            en = np.ascontiguousarray(ended, dtype=np.uint8).reshape(self.P, rows)
            ep = en.ctypes.data_as(C.POINTER(C.c_uint8))
        N.check(self._L.frl_act_explore(self._h, int(mode), rows, _fp(obs), C.byref(x), ep, _fp(store), _fp(env)))
        return store, env

    def learn(self, batch, *, gamma, tau, actor_lr=0.0, critic_lr=0.0, alpha_lr=1e-4, adam_eps=1e-8,
              critic_weight_decay=0.0, clip_norm=0.5, do_actor=True, use_policy_noise=False, policy_noise=0.0,
              noise_clip=0.0, max_action=1.0, policy_noise_scale=1.0, target_entropy=0.0, double_dqn=False, per=False,
              noisy_eps=None, idx=None, noise=None,
              want_stats=False, huber_delta=None):
        a = N.LearnArgs()
        a.batch, a.do_actor, a.use_policy_noise = int(batch), int(bool(do_actor)), int(bool(use_policy_noise))
        a.double_dqn, a.per = int(bool(double_dqn)), int(per)
        a.gamma, a.tau = gamma, tau
        a.actor_lr, a.critic_lr, a.alpha_lr, a.adam_eps = actor_lr, critic_lr, alpha_lr, adam_eps
        a.critic_weight_decay, a.clip_norm = critic_weight_decay, clip_norm
        a.policy_noise, a.noise_clip, a.max_action, a.policy_noise_scale = policy_noise, noise_clip, max_action, policy_noise_scale
        a.target_entropy = target_entropy
        if huber_delta is not None:             # TD loss = huber_loss(e, delta).mean() instead of F.mse_loss
            a.loss_kind, a.huber_delta = 1, float(huber_delta)
        keep = []
        if noisy_eps is not None:
            ne = np.ascontiguousarray(noisy_eps, dtype=F32).reshape(self.P, -1)
            keep.append(ne)
            a.noisy_eps = _fp(ne)
        if idx is not None:
            ix = np.ascontiguousarray(idx, dtype=np.int64).reshape(self.P, self.n_agents, int(batch))
            keep.append(ix)
            a.idx = ix.ctypes.data_as(C.POINTER(C.c_int64))
        if noise is not None:
            nz = np.ascontiguousarray(noise, dtype=F32).reshape(self.P, self.n_agents, max(2, self.n_agents), int(batch), self.act_max)
            keep.append(nz)
            a.noise = _fp(nz)
        stats = None
        if want_stats:
            stats = np.zeros((self.P, self.n_agents, N.FRL_STAT_COUNT), dtype=F32)
            a.stats_out = _fp(stats)
        N.check(self._L.frl_learn(self._h, C.byref(a)))
        return stats

    def stats(self):
        out = np.zeros((self.P, self.n_agents, N.FRL_STAT_COUNT), dtype=F32)
        N.check(self._L.frl_stats_get(self._h, _fp(out)))
        return out

    def last_indices(self, batch):
        """int64 [P][n_agents][batch]: the ring rows the last learn() trained on."""
        out = np.zeros((self.P, self.n_agents, int(batch)), dtype=np.int64)
        N.check(self._L.frl_last_indices(self._h, int(batch), out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def learn_work(self, batch, do_actor=True):
        fl, by = C.c_double(0), C.c_double(0)
        N.check(self._L.frl_learn_work(self._h, int(batch), int(bool(do_actor)), C.byref(fl), C.byref(by)))
        return fl.value, by.value

    def learn_work_executed(self, batch, do_actor=True):
        """Flops the launch executes (no first-layer dX of trained nets, action columns only for dQ/da): frl_learn_work_executed."""
        fl = C.c_double(0)
        N.check(self._L.frl_learn_work_executed(self._h, int(batch), int(bool(do_actor)), C.byref(fl)))
        return fl.value

    # ------------------------------------------------------------------ noisy head
    def noisy_eps_size(self):
        n = C.c_int(0)
        N.check(self._L.frl_noisy_eps_size(self._h, C.byref(n)))
        return n.value

    def noisy_resample(self, eps=None):
        if eps is None:
            N.check(self._L.frl_noisy_resample(self._h, None))
        else:
            ne = np.ascontiguousarray(eps, dtype=F32).reshape(self.P, -1)
            N.check(self._L.frl_noisy_resample(self._h, _fp(ne)))

    # ------------------------------------------------------------------ prioritised replay
    def per_enable(self, alpha=0.5, beta=0.4, beta_increment=0.001, epsilon=0.01):
        N.check(self._L.frl_per_enable(self._h, alpha, beta, beta_increment, epsilon))

    def per_sample(self, batch, uniforms=None, want_outputs=True):
        """-> (idx int64 [P][batch], is_weight f32 [P][batch]); the rows become the current sample for learn(per=True).
        want_outputs=False: nothing is read back and the call is asynchronous (returns None)."""
        up = None
        if uniforms is not None:
            u = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(self.P, int(batch))
            up = u.ctypes.data_as(C.POINTER(C.c_double))
        if not want_outputs:
            N.check(self._L.frl_per_sample(self._h, int(batch), up, None, None))
            return None
        idx = np.zeros((self.P, int(batch)), np.int64)
        w = np.zeros((self.P, int(batch)), F32)
        N.check(self._L.frl_per_sample(self._h, int(batch), up, idx.ctypes.data_as(C.POINTER(C.c_int64)), _fp(w)))
        return idx, w

    def per_update(self, batch, idx=None, td_error=None):
        ip = tp = None
        if idx is not None:
            ix = np.ascontiguousarray(idx, dtype=np.int64).reshape(self.P, int(batch))
            ip = ix.ctypes.data_as(C.POINTER(C.c_int64))
        if td_error is not None:
            td = np.ascontiguousarray(td_error, dtype=F32).reshape(self.P, int(batch))
            tp = _fp(td)
        N.check(self._L.frl_per_update(self._h, int(batch), ip, tp))

    def per_state(self, learner=0):
        s, m, b = C.c_double(0), C.c_double(0), C.c_double(0)
        N.check(self._L.frl_per_state(self._h, int(learner), C.byref(s), C.byref(m), C.byref(b)))
        return dict(sum=s.value, max=m.value, beta=b.value)

    def ppo_learn(self, horizon, minibatch, k_epochs, *, gamma, lmbda, clip, ent_coef, actor_lr, critic_lr,
                  adam_eps=1e-8, clip_norm=0.5, adv_norm=False, perms=None, want_trace=False, want_adv=False,
                  optimizer=0, last_value=None):
        """last_value (one float per learner): PPO_advance/PPO_2.py's learn — advantages and returns from the values the
        ring stored at rollout time (extra column before adv_done), no value pass."""
        a = N.PpoArgs()
        a.horizon, a.minibatch, a.k_epochs, a.adv_norm = int(horizon), int(minibatch), int(k_epochs), int(bool(adv_norm))
        a.gamma, a.lmbda, a.clip, a.ent_coef = gamma, lmbda, clip, ent_coef
        a.actor_lr, a.critic_lr, a.adam_eps, a.clip_norm = actor_lr, critic_lr, adam_eps, clip_norm
        a.optimizer = int(optimizer)
        keep = []
        if perms is not None:
            pm = np.ascontiguousarray(perms, dtype=np.int64).reshape(self.P, int(k_epochs), int(horizon))
            keep.append(pm)
            a.perms = pm.ctypes.data_as(C.POINTER(C.c_int64))
        if last_value is not None:
            lv = np.ascontiguousarray(np.broadcast_to(np.asarray(last_value, dtype=F32).reshape(-1), (self.P,)))
            keep.append(lv)
            a.gae_mode, a.last_value, a.gae_gamma, a.gae_lmbda = 1, _fp(lv), float(gamma), float(lmbda)
        n_mb = (int(horizon) + int(minibatch) - 1) // int(minibatch)
        out = {}
        if want_trace:
            out["trace"] = np.zeros((self.P, int(k_epochs) * n_mb, 2), dtype=F32)
            a.loss_trace_out = _fp(out["trace"])
        if want_adv:
            out["adv"] = np.zeros((self.P, int(horizon)), dtype=F32)
            out["v_target"] = np.zeros((self.P, int(horizon)), dtype=F32)
            a.adv_out, a.vtarget_out = _fp(out["adv"]), _fp(out["v_target"])
        N.check(self._L.frl_ppo_learn(self._h, C.byref(a)))
        return out

    # ------------------------------------------------------------------ timing
    def ppo_work(self, horizon, k_epochs):
        fl, by = C.c_double(0), C.c_double(0)
        N.check(self._L.frl_ppo_work(self._h, int(horizon), int(k_epochs), C.byref(fl), C.byref(by)))
        return fl.value, by.value

    def timer_start(self):
        N.check(self._L.frl_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        N.check(self._L.frl_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def profile(self, on):
        N.check(self._L.frl_profile_enable(self._h, int(bool(on))))

    def profile_read(self):
        """-> {slot name: (total ms, launches)} of frl_learn's kernels since profile(True)."""
        ms = (C.c_double * 8)()
        cnt = (C.c_longlong * 8)()
        N.check(self._L.frl_profile_read(self._h, ms, cnt))
        names = ["draw", "grad_critic", "adam_critic", "grad_actor", "adam_actor", "soft_update", "ppo", "_"]
        return {names[k]: (ms[k], cnt[k]) for k in range(8) if cnt[k]}

    def obsnorm_enable(self, on=True):
        N.check(self._L.frl_obsnorm_enable(self._h, int(bool(on))))

    def obsnorm_stats(self, learner=0, agent=0):
        """-> dict(n, mean[O], S[O], std[O]) of the Batch_ObsNorm running statistics (of one agent of a MADDPG engine)."""
        w = 1 + 3 * max(self.layout.obs_dim[j] for j in range(self.n_agents))
        buf = np.zeros(self.n_agents * w, dtype=F32)
        N.check(self._L.frl_obsnorm_get(self._h, int(learner), _fp(buf)))
        O, b = self.layout.obs_dim[agent], buf[agent * w:(agent + 1) * w]
        return dict(n=int(b[0]), mean=b[1:1 + O].copy(), S=b[1 + O:1 + 2 * O].copy(), std=b[1 + 2 * O:1 + 3 * O].copy())
