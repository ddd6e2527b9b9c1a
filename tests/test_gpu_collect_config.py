"""The collection legs of BASELINE configs 3 and 4 at CONFIG SIZE (the learn() legs of those configs are in
test_gpu_config_shapes.py): `frl_ppo_rollout` with 64 vectorised envs x 32 steps at HalfCheetah-v4's dims (obs 17, act 6;
PPO_file/PPO_with_tricks.py:524-574) and `frl_rollout` with SAC at Humanoid-v4's dims (obs 376, act 17) over 256 vectorised envs
(SAC_file/SAC.py:519-576).  The envs are Python objects behind the callback pool that RECORD what they were asked, so every ring
row is compared with the transition its env saw: a ring-layout or pinned-block sizing bug at 64 x 2048-row segments or at
256 x 393-float observations cannot pass.  MuJoCo itself is not in the image: the dynamics are synthetic, the dims are the configs'."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    assert _native.device_count() > 0
    return _native


class _Box:
    def __init__(self, high, n):
        self.shape, self.high, self.low = (n,), np.full(n, high, np.float32), np.full(n, -high, np.float32)


class _BandEnv:
    """s'[r] = 0.6 s[r] + 0.25 s[r-1] + 0.1 s[r+1] + 0.5 a[r mod A] / max_action + noise; terminates when max |s| > bound;
    records every transition it is stepped through (gymnasium protocol)."""

    def __init__(self, O, A, max_action, seed, limit=1000, bound=8.0):
        self.O, self.A, self.ma, self.limit, self.bound = O, A, max_action, limit, bound
        self.observation_space, self.action_space = _Box(10.0, O), _Box(max_action, A)
        self.g = np.random.default_rng(seed)
        self.log, self.t, self.s = [], 0, None
        self.resets = 0

    def reset(self, seed=None):
        self.s = self.g.standard_normal(self.O).astype(np.float32)
        self.t = 0
        self.resets += 1
        return self.s.copy(), {}

    def step(self, a):
        a = np.asarray(a, np.float32).reshape(-1).copy()
        assert a.shape == (self.A,)
        av = np.clip(a / self.ma, -1, 1)
        s = self.s
        nxt = (0.6 * s + 0.25 * np.roll(s, 1) + 0.1 * np.roll(s, -1) + 0.5 * av[np.arange(self.O) % self.A] +
               0.05 * self.g.standard_normal(self.O)).astype(np.float32)
        r = float(-(np.mean(nxt * nxt) + 0.1 * np.mean(av * av)))
        self.t += 1
        term, trunc = bool(np.max(np.abs(nxt)) > self.bound), self.t >= self.limit
        self.log.append((s.copy(), a, r, nxt.copy(), term, trunc))
        self.s = nxt
        return nxt.copy(), r, term, trunc, {}


def _rand_params(e, N, scale, seed):
    g = np.random.default_rng(seed)
    for p in range(e.P):
        for net in range(e.n_nets):
            flat = (g.standard_normal(e.num_params(net)) * scale).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
            e.set_params(net, flat, N.PARAM_TARGET, learner=p)


def test_c3_ppo_collector_64_envs_at_halfcheetah_dims(N):
    """Config 3's collection leg: 64 envs x 32 steps = the 2048-row horizon per learner, obs 17 / act 6.  Every env's steps
    are one contiguous time-ordered segment of the ring and equal what the env recorded; the env-unit action is
    clip(stored * max_action); the segment ends (and only they, plus true episode ends) carry adv_done; the stored per-dimension
    log-probs are those of the stored actions under the collecting policy; a cycle runs K_epochs x 32 minibatch steps."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import CallbackEnvPool, ppo_rollout
    P, E, Tseg, O, A, ma = 2, 64, 32, 17, 6, 1.0
    T = E * Tseg
    envs = [_BandEnv(O, A, ma, 100 + i, limit=50 if i % 7 == 3 else 1000) for i in range(P * E)]      # a few envs truncate inside their segment
    pool = CallbackEnvPool(envs)
    assert (pool.obs_dim, pool.act_dim, pool.n_actions, pool.max_action) == (O, A, 0, ma)
    e = Engine(N.ALGO_PPO, O, A, T, batch_max=64, n_learners=P, extra_cols=A + 1, seed=3)
    rng = np.random.default_rng(2)
    ls = np.linspace(-0.6, -0.1, A).astype(np.float32)
    for p in range(P):
        fa = (rng.standard_normal(e.num_params(0)) * 0.1).astype(np.float32)
        fa[-A:] = ls                                             # log_std
        e.set_params(0, fa, learner=p)
        e.set_params(1, (rng.standard_normal(e.num_params(1)) * 0.1).astype(np.float32), learner=p)
    out = ppo_rollout(e, pool, 1, envs_per_learner=E, steps_per_env=Tseg, minibatch=64, k_epochs=1, actor_lr=0.0, critic_lr=0.0)
    assert out["env_steps"] == P * T and out["updates"] == P * (T // 64)
    lay = e.layout
    assert lay.extra == A + 1
    for p in range(P):
        assert e.cursor(p) == (0, 0)                             # learn() cleared the buffer (PPO_with_tricks.py:354)
        rows = e.read_rows(p, 0, T)
        obs = rows[:, lay.obs_off[0]:lay.obs_off[0] + O]
        nobs = rows[:, lay.next_obs_off[0]:lay.next_obs_off[0] + O]
        act = rows[:, lay.act_off[0]:lay.act_off[0] + A]
        logp, adv_done = rows[:, lay.extra_off:lay.extra_off + A], rows[:, lay.extra_off + A]
        for j in range(E):
            env = envs[p * E + j]
            seg = slice(j * Tseg, (j + 1) * Tseg)
            assert len(env.log) == Tseg
            lo = np.stack([l[0] for l in env.log]); ln = np.stack([l[3] for l in env.log])
            np.testing.assert_array_equal(obs[seg], lo)
            np.testing.assert_array_equal(nobs[seg], ln)
            np.testing.assert_allclose(rows[seg, lay.rew_off], [l[2] for l in env.log], rtol=1e-6, atol=1e-7)
            term = np.array([l[4] for l in env.log]); trunc = np.array([l[5] for l in env.log])
            np.testing.assert_array_equal(rows[seg, lay.done_off], term.astype(np.float32))
            want_adv = (term | trunc).astype(np.float32)
            want_adv[-1] = 1.0                                   # the segment end: nothing flows in from the next env's segment
            np.testing.assert_array_equal(adv_done[seg], want_adv)
            if env.limit == 50:
                assert trunc.sum() == 0 and env.resets == 1      # 32 steps: below every limit in the first cycle
            a_env = np.stack([l[1] for l in env.log])
            np.testing.assert_allclose(a_env, np.clip(act[seg] * ma, -ma, ma), rtol=0, atol=1e-6)
        full = np.zeros((P, T, O), np.float32)
        full[p] = obs
        mean = e.act(0, N.ACT_TANHHEAD, full, out_dim=A)[p]
        want = -((act - mean) ** 2) / (2 * np.exp(2 * ls)) - ls - 0.9189385332
        np.testing.assert_allclose(logp, want, rtol=2e-4, atol=3e-5)
    # second cycle with the config's K = 10 epochs: the 50-step envs truncate INSIDE their segment now (adv_done mid-segment, the
    # reset observation in the next row)
    before = e.get_params(0, learner=1).copy()
    for env in envs:
        env.log.clear()
    out = ppo_rollout(e, pool, 1, envs_per_learner=E, steps_per_env=Tseg, minibatch=64, k_epochs=10, actor_lr=3e-4, critic_lr=3e-4)
    assert out["env_steps"] == P * T and out["updates"] == P * 10 * (T // 64)
    assert e.opt_step(0, learner=0) == (T // 64) + 10 * (T // 64)
    assert not np.allclose(before, e.get_params(0, learner=1)) and np.all(np.isfinite(e.get_params(1, learner=0)))
    n_mid = 0
    for p in range(P):
        rows = e.read_rows(p, 0, T)                              # learn() resets the cursor, the rollout is still in the ring
        for j in range(E):
            env = envs[p * E + j]
            seg = slice(j * Tseg, (j + 1) * Tseg)
            assert len(env.log) == Tseg
            np.testing.assert_array_equal(rows[seg, lay.obs_off[0]:lay.obs_off[0] + O], np.stack([l[0] for l in env.log]))
            np.testing.assert_array_equal(rows[seg, lay.next_obs_off[0]:lay.next_obs_off[0] + O], np.stack([l[3] for l in env.log]))
            ended = np.array([l[5] or l[4] for l in env.log]).astype(np.float32)
            if env.limit == 50:
                assert ended.sum() == 1 and ended[17] == 1       # episode step 50 = step 18 of the second 32-step segment
                n_mid += 1
            ended[-1] = 1.0
            np.testing.assert_array_equal(rows[seg, lay.extra_off + A], ended)
    assert n_mid == len([i for i in range(P * E) if i % 7 == 3])
    pool.close(); e.close()


@pytest.mark.parametrize("P", [1, 2])
def test_c4_sac_collector_256_envs_at_humanoid_dims(N, P):
    """Config 4's collection leg: SAC at obs 376 / act 17 with 256 vectorised envs per learner, one learn() per vector step
    (batch 256) once the ring holds 2 x batch rows.  Every ring row equals what its env recorded (3.1 KB records through the
    pinned block), the stored action is the policy's tanh sample, the env action clip(stored * max_action) (SAC.py:529-533), the
    update counters advance, and a second call continues where the first stopped (ring cursor, device-resident observations)."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import CallbackEnvPool, rollout
    E, O, A, ma, steps = 256, 376, 17, 0.4, 5
    envs = [_BandEnv(O, A, ma, 500 + i, limit=3 if (i % E) % 50 == 7 else 1000) for i in range(P * E)]       # some envs truncate and reset inside the run
    pool = CallbackEnvPool(envs)
    assert (pool.obs_dim, pool.act_dim, pool.n_actions) == (O, A, 0) and abs(pool.max_action - ma) < 1e-7
    e = Engine(N.ALGO_SAC, O, A, 4096, twin_critic=True, batch_max=256, n_learners=P, seed=9)
    _rand_params(e, N, 0.03, seed=4)
    for p in range(P):
        e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0, learner=p)
    lay = e.layout
    assert lay.width >= 2 * O + A + 2
    before = e.get_params(1, learner=P - 1).copy()
    kw = dict(envs_per_learner=E, start_steps=2 * 256 - 1, learn_every=1, batch=256, alpha_lr=1e-4, target_entropy=-float(A))
    out = rollout(e, pool, steps, **kw)
    assert out["env_steps"] == steps * P * E
    assert out["updates"] == (steps - 1) * P                     # learn() from the step after which len(buffer) = 512 > start_steps
    out2 = rollout(e, pool, 2, **kw)
    assert out2["updates"] == 2 * P
    total = steps + 2
    for p in range(P):
        assert e.cursor(p) == (total * E, total * E)
        rows = e.read_rows(p, 0, total * E)
        for j in range(0, E, 5):
            env, tr = envs[p * E + j], rows[j::E]                # env j's transitions in time order
            assert len(env.log) == total
            np.testing.assert_array_equal(tr[:, lay.obs_off[0]:lay.obs_off[0] + O], np.stack([l[0] for l in env.log]))
            np.testing.assert_array_equal(tr[:, lay.next_obs_off[0]:lay.next_obs_off[0] + O], np.stack([l[3] for l in env.log]))
            np.testing.assert_allclose(tr[:, lay.rew_off], [l[2] for l in env.log], rtol=1e-6, atol=1e-7)
            np.testing.assert_array_equal(tr[:, lay.done_off], np.array([float(l[4]) for l in env.log], np.float32))
            stored = tr[:, lay.act_off[0]:lay.act_off[0] + A]
            assert np.all(np.abs(stored) <= 1.0) and stored.std() > 1e-3
            np.testing.assert_allclose(np.stack([l[1] for l in env.log]), np.clip(stored * ma, -ma, ma), rtol=0, atol=1e-6)
        for j in (7, 57):                                        # limit 3: two resets inside 7 steps; the row after a reset starts from the reset observation
            env = envs[p * E + j]
            assert env.resets == 1 + total // 3
            tr = rows[j::E]
            assert not np.array_equal(tr[2, lay.next_obs_off[0]:lay.next_obs_off[0] + O], tr[3, lay.obs_off[0]:lay.obs_off[0] + O])
            np.testing.assert_array_equal(tr[1, lay.next_obs_off[0]:lay.next_obs_off[0] + O], tr[2, lay.obs_off[0]:lay.obs_off[0] + O])
        assert e.opt_step(1, learner=p) == total - 1 and e.opt_step(0, learner=p) == total - 1
    st = e.stats()
    assert np.all(np.isfinite(st)) and np.all(st[:, 0, N.STAT_CRITIC_LOSS] > 0)
    assert not np.allclose(before, e.get_params(1, learner=P - 1))
    pool.close(); e.close()


def test_c4_builtin_wide_pool_rows_follow_its_dynamics(N):
    """The built-in SynBandWide-v0 pool (what tools/config4_rollout.py times) at 256 envs: the rows of the ring obey the env's own
    recurrence — next_obs is the banded map of (obs, clip(stored * 0.4) / 0.4) up to its 0.05-sigma noise — and consecutive rows of
    an env chain (next_obs[t] == obs[t + 1]) except across resets."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout
    E, O, A, steps = 256, 376, 17, 6
    e = Engine(N.ALGO_SAC, O, A, 4096, twin_critic=True, batch_max=256, n_learners=1, seed=5)
    _rand_params(e, N, 0.03, seed=6)
    e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0)
    pool = EnvPool("SynBandWide-v0", E, n_threads=4, seed=2)
    out = rollout(e, pool, steps, envs_per_learner=E, start_steps=511, learn_every=1, batch=256, alpha_lr=1e-4, target_entropy=-float(A))
    assert out["env_steps"] == steps * E and out["updates"] == steps - 1
    lay = e.layout
    rows = e.read_rows(0, 0, steps * E)
    obs = rows[:, lay.obs_off[0]:lay.obs_off[0] + O].reshape(steps, E, O)
    nobs = rows[:, lay.next_obs_off[0]:lay.next_obs_off[0] + O].reshape(steps, E, O)
    act = rows[:, lay.act_off[0]:lay.act_off[0] + A].reshape(steps, E, A)
    np.testing.assert_array_equal(nobs[:-1], obs[1:])            # no episode ends in 6 steps (limit 1000, |s| stays small)
    av = np.clip(act, -1, 1)
    pred = 0.6 * obs + 0.25 * np.roll(obs, 1, axis=2) + 0.1 * np.roll(obs, -1, axis=2) + 0.5 * av[:, :, np.arange(O) % A]
    resid = nobs - pred
    assert abs(resid.std() - 0.05) < 0.005 and abs(resid.mean()) < 0.002, (resid.std(), resid.mean())
    sq = (nobs.astype(np.float64) ** 2).mean(axis=2) + 0.1 * (av.astype(np.float64) ** 2).mean(axis=2)
    np.testing.assert_allclose(rows[:, lay.rew_off].reshape(steps, E), -sq, rtol=1e-4, atol=1e-6)
    pool.close(); e.close()
