"""Device-side exploration and the callback env pool.

The reference's loops draw exploration on the host per env step: epsilon-greedy (DQN.py:307-310), Gaussian action noise with
a per-episode decayed scale (TD3.py:412,425-427), an Ornstein-Uhlenbeck process (SAC.py:334-356,529,546-547).  Here the
act launch applies the same rules from the engine's Philox stream, so they are validated statistically: the rule's
parameters are recovered from many rows, and the stored action is exactly the policy's own output.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    _native.build()
    assert _native.device_count() > 0, "no HIP device: the engine has no CPU fallback"
    return _native


def _rand_params(e, N, scale=0.1, seed=0):
    g = np.random.default_rng(seed)
    for p in range(e.P):
        for net in range(e.n_nets):
            flat = (g.standard_normal(e.num_params(net)) * scale).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
            e.set_params(net, flat, N.PARAM_TARGET, learner=p)


def test_epsilon_greedy_rule(N):
    """P(random) = epsilon, the random action uniform over the n_actions (so it differs from the greedy one with
    probability epsilon * (1 - 1/nA)), env action == stored action, epsilon 0 == argmax, another call draws again."""
    from freerl_amd.engine import Engine
    P, R, O, nA, eps = 2, 8192, 8, 4, 0.3
    e = Engine(N.ALGO_DQN, O, nA, 64, discrete=True, batch_max=32, n_learners=P, seed=11)
    _rand_params(e, N, 0.3)
    obs = np.random.default_rng(1).standard_normal((P, R, O)).astype(np.float32)
    greedy = e.act(0, N.ACT_ARGMAX, obs)
    s0, v0 = e.act_explore(N.ACT_ARGMAX, obs, kind=N.EXPLORE_EPS_GREEDY, epsilon=0.0)
    np.testing.assert_array_equal(s0, greedy.reshape(P, R))
    np.testing.assert_array_equal(v0, s0)
    s1, v1 = e.act_explore(N.ACT_ARGMAX, obs, kind=N.EXPLORE_EPS_GREEDY, epsilon=eps)
    np.testing.assert_array_equal(v1, s1)
    assert set(np.unique(s1)) <= set(float(k) for k in range(nA))
    changed = (s1 != s0)
    want = eps * (1 - 1 / nA)
    sd = np.sqrt(want * (1 - want) / (P * R))
    assert abs(changed.mean() - want) < 5 * sd, (changed.mean(), want)
    # where the action changed it is uniform over the other nA - 1 actions: every action is hit
    for p in range(P):
        cnt = np.bincount(s1[p][changed[p]].astype(int), minlength=nA)
        assert cnt.min() > 0.5 * cnt.mean()
    s2, _ = e.act_explore(N.ACT_ARGMAX, obs, kind=N.EXPLORE_EPS_GREEDY, epsilon=eps)
    assert not np.array_equal(s1, s2)                                  # a new counter value per launch
    assert not np.array_equal(changed[0], changed[1])                  # and a key per learner
    e.close()


def test_gaussian_action_noise_rule(N):
    """action_ = clip(a*max_action + scale * N(0, sigma*max_action), +-max_action) (TD3.py:412); the stored action is the
    actor's own tanh output."""
    from freerl_amd.engine import Engine
    P, R, O, A, ma, sigma, scale = 2, 8192, 5, 3, 2.0, 0.1, 0.7
    e = Engine(N.ALGO_TD3, O, A, 64, twin_critic=True, batch_max=32, n_learners=P, seed=3)
    _rand_params(e, N, 0.1)
    obs = np.random.default_rng(2).standard_normal((P, R, O)).astype(np.float32)
    a = e.act(0, N.ACT_TANHHEAD, obs, out_dim=A)
    store, env = e.act_explore(N.ACT_TANHHEAD, obs, kind=N.EXPLORE_GAUSS, sigma=sigma, scale=scale, max_action=ma, out_dim=A)
    np.testing.assert_array_equal(store, a)
    assert np.all(np.abs(env) <= ma)
    inside = np.abs(env) < ma
    noise = (env - store * ma)[inside]
    want_sd = scale * sigma * ma
    assert abs(noise.mean()) < 5 * want_sd / np.sqrt(noise.size)
    assert abs(noise.std() / want_sd - 1) < 0.03
    k = np.mean(((noise - noise.mean()) / noise.std()) ** 4)
    assert abs(k - 3.0) < 0.2                                          # Gaussian kurtosis
    # none: action_ = clip(a * max_action)
    s2, env2 = e.act_explore(N.ACT_TANHHEAD, obs, kind=N.EXPLORE_NONE, max_action=ma, out_dim=A)
    np.testing.assert_allclose(env2, np.clip(a * ma, -ma, ma), rtol=0, atol=1e-7)
    e.close()


def test_ou_noise_rule(N):
    """x += theta*(0 - x) + sqrt(dt)*sigma*N(0,1); action_ = clip(a*max_action + x*scale*max_action) (SAC.py:334-356,529):
    the state's variance follows the recursion, consecutive states are correlated by (1 - theta), `ended` rows restart
    from zero (SAC.py:546-547)."""
    from freerl_amd.engine import Engine
    P, R, O, A, ma, th, sg, dt, scale = 1, 16384, 4, 2, 1.0, 0.15, 0.2, 1e-2, 0.5
    e = Engine(N.ALGO_DDPG, O, A, 64, batch_max=32, n_learners=P, seed=5)
    _rand_params(e, N, 0.02)                                           # |a| small: nothing clips
    obs = np.random.default_rng(4).standard_normal((P, R, O)).astype(np.float32)
    kw = dict(kind=N.EXPLORE_OU, scale=scale, max_action=ma, ou_theta=th, ou_sigma=sg, ou_dt=dt, out_dim=A)
    s1, e1 = e.act_explore(N.ACT_TANHHEAD, obs, **kw)
    x1 = (e1 - s1 * ma) / (scale * ma)
    v = dt * sg * sg
    assert abs(x1.std() ** 2 / v - 1) < 0.04 and abs(x1.mean()) < 5 * np.sqrt(v / x1.size)
    s2, e2 = e.act_explore(N.ACT_TANHHEAD, obs, **kw)
    x2 = (e2 - s2 * ma) / (scale * ma)
    assert abs(x2.std() ** 2 / (((1 - th) ** 2 + 1) * v) - 1) < 0.04
    innov = x2 - (1 - th) * x1                                         # = sqrt(dt)*sigma*N, independent of x1
    assert abs(innov.std() ** 2 / v - 1) < 0.04
    assert abs(np.corrcoef(innov.reshape(-1), x1.reshape(-1))[0, 1]) < 0.03
    ended = np.zeros((P, R), np.uint8)
    ended[:, ::2] = 1
    s3, e3 = e.act_explore(N.ACT_TANHHEAD, obs, ended=ended, **kw)
    x3 = (e3 - s3 * ma) / (scale * ma)
    assert abs(x3[:, ::2].std() ** 2 / v - 1) < 0.06                    # restarted from 0
    assert x3[:, 1::2].std() ** 2 > 2.0 * v                            # third step of the running process
    e.close()


def test_sac_sample_draws_its_own_eps(N):
    """FRL_ACT_SAC_SAMPLE on the device path: a = tanh(mean + std*eps), eps ~ N(0,1) from the launch's stream."""
    from freerl_amd.engine import Engine
    P, R, O, A = 1, 8192, 6, 2
    e = Engine(N.ALGO_SAC, O, A, 64, twin_critic=True, batch_max=32, n_learners=P, seed=8)
    _rand_params(e, N, 0.05)
    fa = e.get_params(0)
    fa[-A:] = -0.5                                                     # log_std
    e.set_params(0, fa)
    obs = np.random.default_rng(6).standard_normal((P, R, O)).astype(np.float32)
    mean = e.act(0, N.ACT_RAW, obs, out_dim=A)
    store, env = e.act_explore(N.ACT_SAC_SAMPLE, obs, kind=N.EXPLORE_NONE, max_action=1.5, out_dim=A)
    eps = (np.arctanh(np.clip(store, -0.999999, 0.999999)) - mean) / np.exp(-0.5)
    assert abs(eps.mean()) < 0.04 and abs(eps.std() - 1) < 0.03
    np.testing.assert_allclose(env, np.clip(store * 1.5, -1.5, 1.5), atol=1e-7)
    e.close()


class _Recorder:
    """Wraps an in-repo env (gymnasium protocol) and records what the pool asked of it."""

    def __init__(self, env):
        self.env, self.log = env, []
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.last = None

    def reset(self, seed=None):
        o, info = self.env.reset(seed=seed)
        self.last = np.asarray(o, np.float32).copy()
        return o, info

    def step(self, a):
        o, r, t, u, info = self.env.step(a)
        self.log.append((self.last.copy(), np.array(a, np.float32).reshape(-1).copy(), float(r), np.asarray(o, np.float32).copy(), bool(t), bool(u)))
        self.last = np.asarray(o, np.float32).copy()
        return o, r, t, u, info


@pytest.mark.parametrize("host_explore", [False, True])
def test_rollout_over_callback_pool_matches_what_the_envs_saw(N, host_explore):
    """frl_rollout over caller-supplied Python envs: every ring row is the transition its env recorded (obs, reward,
    next_obs, done), the env-unit action is clip(stored*max_action + noise), and the Gaussian noise decays to exactly zero
    once a learner has finished max_episodes episodes (TD3.py:425-427)."""
    from freerl_amd import envs as E
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import CallbackEnvPool, rollout
    P, Ev, steps = 2, 2, 100
    envs = [_Recorder(E.make("PendulumShort-v1", prefer_gymnasium=False)) for _ in range(P * Ev)]
    pool = CallbackEnvPool(envs)
    assert (pool.obs_dim, pool.act_dim, pool.n_actions, pool.max_action) == (3, 1, 0, 2.0)
    e = Engine(N.ALGO_TD3, 3, 1, 400, twin_critic=True, batch_max=32, n_learners=P, seed=2)
    _rand_params(e, N, 0.03, seed=3)                                   # small |a|: the env-action clip stays out of the statistics
    out = rollout(e, pool, steps, envs_per_learner=Ev, learn_every=0, explore_sigma=0.2, batch=32, host_explore=host_explore,
                  gauss_init_scale=1.0, gauss_final_scale=0.0, max_episodes=4)
    assert out["env_steps"] == steps * P * Ev and out["episodes"] == 2 * P * Ev          # 40-step episodes
    lay, ma = e.layout, 2.0
    for p in range(P):
        rows = e.read_rows(p, 0, steps * Ev)
        assert e.cursor(p) == (steps * Ev, steps * Ev)
        noises = []
        for j in range(Ev):
            log = envs[p * Ev + j].log
            tr = rows[j::Ev]
            assert len(log) == steps
            for t in range(steps):
                o, a_env, r, o2, term, trunc = log[t]
                np.testing.assert_allclose(tr[t, lay.obs_off[0]:lay.obs_off[0] + 3], o, atol=1e-7)
                np.testing.assert_allclose(tr[t, lay.next_obs_off[0]:lay.next_obs_off[0] + 3], o2, atol=1e-7)
                assert abs(tr[t, lay.rew_off] - r) < 1e-6 * max(1, abs(r)) and tr[t, lay.done_off] == float(term)
                assert abs(a_env[0]) <= ma + 1e-6 and abs(tr[t, lay.act_off[0]]) <= 1.0
            stored = tr[:, lay.act_off[0]]
            a_env = np.array([l[1][0] for l in log])
            noises.append(a_env - np.clip(stored * ma, -ma, ma))
        # the learner's two envs finish episodes together at steps 40 and 80: scale 1 -> 0.5 -> 0 (sigma 0.2 * max_action 2)
        nz = np.stack(noises)
        s_a, s_b = nz[:, :40].std(), nz[:, 40:80].std()
        assert 0.3 < s_a < 0.5 and 0.14 < s_b < 0.26 and s_a > 1.4 * s_b, (s_a, s_b)
        np.testing.assert_allclose(nz[:, 80:], 0, atol=1e-6)
    pool.close(); e.close()


def test_rollout_over_callback_pool_discrete_and_learning(N):
    """DQN over Python CartPole instances: explored indices are stored and stepped, learning runs on the device draws."""
    from freerl_amd import envs as E
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import CallbackEnvPool, rollout
    P, Ev = 2, 3
    envs = [_Recorder(E.make("CartPole-v1", prefer_gymnasium=False)) for _ in range(P * Ev)]
    pool = CallbackEnvPool(envs)
    assert (pool.obs_dim, pool.act_dim, pool.n_actions) == (4, 1, 2)
    e = Engine(N.ALGO_DQN, 4, 2, 1000, discrete=True, batch_max=32, n_learners=P, seed=4)
    _rand_params(e, N, 0.2, seed=5)
    before = e.get_params(0, learner=1).copy()
    out = rollout(e, pool, 60, envs_per_learner=Ev, start_steps=64, learn_every=1, epsilon=0.5, batch=32, critic_lr=1e-3)
    assert out["env_steps"] == 60 * P * Ev and out["updates"] > 0 and out["episodes"] > 0
    lay = e.layout
    for p in range(P):
        rows = e.read_rows(p, 0, 60 * Ev)
        for j in range(Ev):
            log, tr = envs[p * Ev + j].log, rows[j::Ev]
            np.testing.assert_array_equal(tr[:, lay.act_off[0]], np.array([l[1][0] for l in log]))
            np.testing.assert_array_equal(tr[:, lay.done_off], np.array([float(l[4]) for l in log]))
            np.testing.assert_allclose(tr[:, lay.obs_off[0]:lay.obs_off[0] + 4], np.stack([l[0] for l in log]), atol=1e-7)
        assert 0.1 < rows[:, lay.act_off[0]].mean() < 0.9
    assert not np.allclose(before, e.get_params(0, learner=1)) and np.all(np.isfinite(e.stats()))
    pool.close(); e.close()


def test_rollout_on_a_per_engine_samples_and_updates_priorities(N):
    """A PER-enabled DQN engine in frl_rollout: the loop of DQN_with_tricks.py (sample by priority -> learn with the
    importance weights -> update the priorities); without learn.per the call is refused instead of training on stale trees."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout
    P, Ev = 2, 2
    e = Engine(N.ALGO_DQN, 8, 4, 512, discrete=True, batch_max=32, n_learners=P, seed=6)
    _rand_params(e, N, 0.2, seed=7)
    e.per_enable(0.6, 0.4, 0.001, 0.01)
    pool = EnvPool("SynLinearDiscrete-v0", P * Ev, n_threads=1, seed=3)
    with pytest.raises(N.FrlError):
        rollout(e, pool, 40, envs_per_learner=Ev, start_steps=0, batch=32)
    out = rollout(e, pool, 40, envs_per_learner=Ev, learn_every=0, batch=32)          # collect only: every row at priority 1
    st0 = e.per_state(0)
    assert abs(st0["sum"] - 80.0) < 1e-9 and st0["max"] == 1.0
    out = rollout(e, pool, 30, envs_per_learner=Ev, start_steps=0, batch=32, per=1, double_dqn=True, critic_lr=1e-3)
    assert out["updates"] == 30 * P
    st1 = e.per_state(0)
    assert st1["sum"] != st0["sum"] + 60.0 and st1["beta"] > 0.4 and np.all(np.isfinite(e.stats()))   # priorities rewritten from TD errors
    pool.close(); e.close()


@pytest.mark.parametrize("P,Ev,split,cap,every,handover", [
    (3, 2, "1", 16384, 1, "flag"), (2, 5, "4", 16384, 1, "flag"), (1, 70, "2", 16384, 1, "flag"), (2, 3, "4", 300, 1, "flag"),
    (2, 2, "4", 16384, 3, "flag"), (3, 2, "4", 16384, 1, "sync"), (2, 4, "1", 16384, 2, "memcpy")])
def test_dqn_rollout_one_launch_per_step_matches_the_separate_launches(N, monkeypatch, P, Ev, split, cap, every, handover):
    """frl_rollout on a plain DQN engine folds add(), learn() and the next select_action + epsilon-greedy into one launch per
    vector step (kernels_dqn2.hip).  Same engine seed, same pool seed: the separate commit / learn / act launches
    (FRL_DQN_STEP_FUSE=0) consume the same Philox counters, so the rings, the parameters and the returns must agree — the only
    arithmetic that differs is the Q forward behind the argmax (MFMA chain against act_kernel's row-chunk layers)."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout
    monkeypatch.setenv("FRL_DQN_SPLIT", split)
    # how the block reaches the launch and the actions the host: pinned memory in place + a flagged word (default for small steps),
    # in place + stream synchronisation, or a hipMemcpyAsync each way (what large vector steps get)
    if handover == "sync":
        monkeypatch.setenv("FRL_ROLLOUT_POLL", "0")
    if handover == "memcpy":
        monkeypatch.setenv("FRL_ROLLOUT_ZEROCOPY", "0")
    res = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("FRL_DQN_STEP_FUSE", fuse)
        e = Engine(N.ALGO_DQN, 8, 4, cap, discrete=True, batch_max=64, n_learners=P, seed=11)       # cap 300: the ring wraps
        _rand_params(e, N, 0.3, seed=12)
        pool = EnvPool("SynLinearDiscrete-v0", P * Ev, n_threads=1, seed=5)
        kw = dict(envs_per_learner=Ev, start_steps=128 // Ev, learn_every=every, epsilon=0.2, batch=64, critic_lr=1e-3, tau=0.05)
        o1 = rollout(e, pool, 90, **kw)
        o2 = rollout(e, pool, 35, **kw)                     # a second call: starts with a separate act launch again
        rows = [e.read_rows(p, 0, min(cap, 125 * Ev)) for p in range(P)]
        res.append((o1, o2, rows, [e.get_params(0, learner=p) for p in range(P)], [e.get_params(0, N.PARAM_TARGET, learner=p) for p in range(P)],
                    [e.opt_step(0, learner=p) for p in range(P)]))
        pool.close(); e.close()
    a, b = res
    assert a[0]["updates"] == b[0]["updates"] > 0 and a[1]["updates"] == b[1]["updates"] == (35 // every) * P
    for p in range(P):
        np.testing.assert_array_equal(a[2][p], b[2][p])
        np.testing.assert_array_equal(a[3][p], b[3][p])
        np.testing.assert_array_equal(a[4][p], b[4][p])
        assert a[5][p] == b[5][p]
    assert a[0]["return_sum"] == b[0]["return_sum"] and a[1]["episodes"] == b[1]["episodes"]


@pytest.mark.parametrize("algo,P,Ev,every,freq,handover", [
    ("td3", 1, 1, 1, 2, "flag"), ("td3", 3, 2, 1, 2, "flag"), ("td3", 2, 5, 1, 1, "sync"), ("ddpg", 1, 70, 1, 1, "flag"), ("sac", 2, 3, 1, 1, "flag"),
    ("td3", 2, 2, 3, 2, "flag"), ("sac", 1, 4, 2, 1, "memcpy"), ("td3", 8, 1, 1, 2, "flag"),
    ("td3", 20, 2, 1, 2, "flag"), ("sac", 31, 1, 1, 1, "flag")])       # (17 .. 32 learners: eight workgroups per learner, two row tiles each)
def test_solo_rollout_folded_step_matches_the_separate_launches(N, monkeypatch, algo, P, Ev, every, freq, handover):
    """frl_rollout on a single-learner engine (kernels_solo.hip) folds add() into the head of the critic launch and the next
    select_action + exploration into the tail of the step's last launch (the actor launch on policy steps, behind a second flag
    hand-over; the critic launch otherwise).  FRL_SOLO_STEP_FUSE=0 runs the separate commit / learn / act launches on the same Philox
    counters; the tail IS act_frag_kernel's body, so rings, parameters, targets and returns must agree bit for bit."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout
    monkeypatch.delenv("FRL_CRITIC_V2", raising=False)
    if handover == "sync":
        monkeypatch.setenv("FRL_ROLLOUT_POLL", "0")
    if handover == "memcpy":
        monkeypatch.setenv("FRL_ROLLOUT_ZEROCOPY", "0")
    aid = dict(td3=N.ALGO_TD3, ddpg=N.ALGO_DDPG, sac=N.ALGO_SAC)[algo]
    res = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("FRL_SOLO_STEP_FUSE", fuse)
        e = Engine(aid, 8, 2, 16384, twin_critic=algo != "ddpg", batch_max=32, n_learners=P, seed=11)
        assert e.learn_path(32) == (True, 117376, 16 if P <= 16 else 32)
        _rand_params(e, N, 0.3, seed=12)
        if algo == "sac":
            for p in range(P):
                e.set_alpha_state([np.log(0.05), 0, 0, 0.05], learner=p)
        pool = EnvPool("SynLinear-v0", P * Ev, n_threads=1, seed=5)
        kw = dict(envs_per_learner=Ev, start_steps=64, learn_every=every, batch=32, actor_lr=1e-3, critic_lr=1e-3, tau=0.05, policy_freq=freq)     # (learn from 65 rows on)
        o1 = rollout(e, pool, 90, **kw)
        o2 = rollout(e, pool, 35, **kw)                     # a second call: starts with a separate act launch again
        rows = [e.read_rows(p, 0, min(16384, 125 * Ev)) for p in range(P)]
        par = [np.concatenate([e.get_params(net, kind, learner=p) for net in (0, 1) for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M)]) for p in range(P)]
        res.append((o1, o2, rows, par, [(e.opt_step(0, learner=p), e.opt_step(1, learner=p)) for p in range(P)]))
        pool.close(); e.close()
    a, b = res
    assert a[0]["updates"] == b[0]["updates"] > 0 and a[1]["updates"] == b[1]["updates"] == (35 // every) * P
    for p in range(P):
        np.testing.assert_array_equal(a[2][p], b[2][p])
        np.testing.assert_array_equal(a[3][p], b[3][p])
        assert a[4][p] == b[4][p]
    assert a[0]["return_sum"] == b[0]["return_sum"] and a[1]["episodes"] == b[1]["episodes"]


@pytest.mark.parametrize("kind,P,Ev,every", [("dqn", 5, 2, 1), ("dqn", 1, 1, 2), ("td3", 3, 2, 1), ("sac", 1, 1, 3)])
def test_prearmed_launches_on_and_off_agree(N, monkeypatch, kind, P, Ev, every):
    """The folded rollout step with the next step's launch enqueued a step ahead on the pool's second stream (doorbell + device word,
    frl_api_rollout.inc) against the same loop launching each step when its block is filled: FRL_ROLLOUT_PREARM=1 forces the first
    (also past the population sizes it is the default for), =0 the second.  Same Philox counters, same predicted ring sizes: rings,
    parameters, targets, Adam moments and step counts bit for bit — with learn_every > 1 armed and unarmed steps alternate."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout
    monkeypatch.delenv("FRL_CRITIC_V2", raising=False)
    res = []
    for arm in ("0", "1", "cancel"):
        # "cancel": armed, and the host's patience with an armed launch set to zero — every one of them is cancelled (what a 1 s env step
        # does: the launch's own 2 s bound must not cost the update) and the step replayed the plain way with the counters put back
        monkeypatch.setenv("FRL_ROLLOUT_PREARM", "0" if arm == "0" else "1")
        if arm == "cancel":
            monkeypatch.setenv("FRL_ROLLOUT_ARM_PATIENCE_MS", "0")
        else:
            monkeypatch.delenv("FRL_ROLLOUT_ARM_PATIENCE_MS", raising=False)
        if kind == "dqn":
            e = Engine(N.ALGO_DQN, 8, 4, 400, discrete=True, batch_max=64, n_learners=P, seed=11)           # (the ring wraps inside the run)
            pool = EnvPool("SynLinearDiscrete-v0", P * Ev, n_threads=1, seed=5)
            kw = dict(envs_per_learner=Ev, start_steps=128 // Ev, learn_every=every, epsilon=0.2, batch=64, critic_lr=1e-3, tau=0.05)
            nets = (0,)
        else:
            aid = dict(td3=N.ALGO_TD3, sac=N.ALGO_SAC)[kind]
            e = Engine(aid, 8, 2, 400, twin_critic=True, batch_max=32, n_learners=P, seed=11)
            assert e.learn_path(32) == (True, 117376, 16)
            if kind == "sac":
                for p in range(P):
                    e.set_alpha_state([np.log(0.05), 0, 0, 0.05], learner=p)
            pool = EnvPool("SynLinear-v0", P * Ev, n_threads=1, seed=5)
            kw = dict(envs_per_learner=Ev, start_steps=64, learn_every=every, batch=32, actor_lr=1e-3, critic_lr=1e-3, tau=0.05, policy_freq=2)
            nets = (0, 1)
        _rand_params(e, N, 0.3, seed=12)
        outs = [rollout(e, pool, n, **kw) for n in (150, 1, 60)]                 # (a one-step call in the middle: nothing to arm)
        rows = [e.read_rows(p, 0, 400) for p in range(P)]
        par = [np.concatenate([e.get_params(net, k, learner=p) for net in nets for k in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)])
               for p in range(P)]
        res.append((outs, rows, par, [[e.opt_step(net, learner=p) for net in nets] for p in range(P)], e.last_indices(kw["batch"]) if hasattr(e, "last_indices") else None))
        pool.close(); e.close()
    a = res[0]
    for b in res[1:]:
        assert [o["updates"] for o in a[0]] == [o["updates"] for o in b[0]] and a[0][0]["updates"] > 0
        assert [o["return_sum"] for o in a[0]] == [o["return_sum"] for o in b[0]]
        for p in range(P):
            np.testing.assert_array_equal(a[1][p], b[1][p])
            np.testing.assert_array_equal(a[2][p], b[2][p])
            assert a[3][p] == b[3][p]
        if a[4] is not None:
            np.testing.assert_array_equal(a[4], b[4])


class _SlowOnce:
    """An in-repo env whose k-th step takes `seconds` — longer than the host's patience with a pre-armed launch (1 s) and, with the
    default launch bound, most of the way to the launch's own 2 s."""

    def __init__(self, env, k, seconds):
        self.env, self.k, self.seconds, self.n = env, k, seconds, 0
        self.observation_space, self.action_space = env.observation_space, env.action_space

    def reset(self, seed=None):
        return self.env.reset(seed=seed)

    def step(self, a):
        self.n += 1
        if self.n == self.k:
            import time
            time.sleep(self.seconds)
        return self.env.step(a)


@pytest.mark.parametrize("seconds", [0.0, 1.3])
def test_a_slow_env_step_does_not_cost_the_prearmed_update(N, monkeypatch, seconds):
    """One env.step of 1.3 s in the middle of a DQN run over Python CartPole: the launch that was armed for that step is cancelled and
    the step replayed the plain way — same ring, same net and optimiser state as the run that never arms."""
    from freerl_amd import envs as E
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import CallbackEnvPool, rollout
    monkeypatch.delenv("FRL_ROLLOUT_ARM_PATIENCE_MS", raising=False)
    res = []
    for arm in ("0", "1"):
        monkeypatch.setenv("FRL_ROLLOUT_PREARM", arm)
        pool = CallbackEnvPool([_SlowOnce(E.make("CartPole-v1", prefer_gymnasium=False), 90, seconds)], seed=7)        # (seeded resets: the two runs see the same env)
        e = Engine(N.ALGO_DQN, 4, 2, 1000, discrete=True, batch_max=32, n_learners=1, seed=4)
        _rand_params(e, N, 0.2, seed=5)
        out = rollout(e, pool, 120, envs_per_learner=1, start_steps=64, learn_every=1, epsilon=0.3, batch=32, critic_lr=1e-3, tau=0.05)
        res.append((out["updates"], out["return_sum"], e.read_rows(0, 0, 120), e.get_params(0), e.get_params(0, N.PARAM_TARGET),
                    e.get_params(0, N.PARAM_ADAM_V), e.opt_step(0)))
        pool.close(); e.close()
    a, b = res
    assert a[0] == b[0] > 40 and a[1] == b[1] and a[6] == b[6]
    for k in (2, 3, 4, 5):
        np.testing.assert_array_equal(a[k], b[k])
