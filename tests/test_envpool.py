"""The native env pool against the Python envs of freerl_amd/envs.py (same published equations):
host-only, no GPU.  Physical states are injected so both sides start from the same point."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def native():
    from freerl_amd import _native
    _native.build()
    return _native


@pytest.mark.parametrize("name", ["Pendulum-v1", "CartPole-v1", "SynLinear-v0", "SynLinearDiscrete-v0"])
def test_pool_dynamics_match_python_envs(native, name):
    from freerl_amd import envs as E
    from freerl_amd.envpool import EnvPool
    n = 5
    pool = EnvPool(name, n, n_threads=2, seed=3)
    py = [E.make(name, prefer_gymnasium=False) for _ in range(n)]
    g = np.random.default_rng(0)
    pool.reset()
    for i, e in enumerate(py):
        e.reset(seed=100 + i)
        pool.set_state(i, e.state)
    noisy = name.startswith("SynLinear")
    prev_state = [np.array(e.state, dtype=np.float64) for e in py]
    for t in range(30):
        if pool.n_actions:
            act = g.integers(0, pool.n_actions, (n, 1)).astype(np.float32)
        else:
            act = g.uniform(-pool.max_action, pool.max_action, (n, pool.act_dim)).astype(np.float32)
        nobs, rew, term, trunc, onext = pool.step(act)
        for i, e in enumerate(py):
            a = int(act[i, 0]) if pool.n_actions else act[i]
            o2, r2, t2, tr2, _ = e.step(a)
            if noisy:       # process noise (0.05 sigma) comes from different generators: check the mean
                a_vec = np.zeros(2)
                if pool.n_actions:
                    a_vec[int(a) // 2] = 1.0 if int(a) % 2 == 0 else -1.0
                else:
                    a_vec = np.clip(np.asarray(a, np.float64), -1, 1)
                mean = e.A @ prev_state[i] + e.B @ a_vec
                assert np.max(np.abs(nobs[i] - mean)) < 0.35, (nobs[i], mean)
                e.reset(seed=200 + t)
                pool.set_state(i, e.state)      # re-synchronise
                prev_state[i] = e.state.copy()
                continue
            np.testing.assert_allclose(nobs[i], o2, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(rew[i], r2, rtol=1e-5, atol=1e-6)
            assert bool(term[i]) == t2
            if term[i] or trunc[i]:
                assert not np.allclose(onext[i], nobs[i])       # auto-reset observation
                e.reset(seed=7)
                pool.set_state(i, e.state)
            else:
                np.testing.assert_array_equal(onext[i], nobs[i])
    pool.close()


def test_pool_time_limit_and_autoreset(native):
    from freerl_amd.envpool import EnvPool
    pool = EnvPool("PendulumShort-v1", 3, n_threads=1, seed=1)
    pool.reset()
    for t in range(40):
        nobs, rew, term, trunc, onext = pool.step(np.zeros((3, 1), np.float32))
        assert not term.any()
        assert trunc.all() == (t == 39)
    assert pool.max_steps == 40 and pool.max_action == 2.0 and pool.obs_dim == 3
    pool.close()


def test_pool_argument_validation(native):
    import ctypes as C
    L = native.lib()
    h = C.c_void_p()
    assert L.frl_envpool_create(99, 4, 1, 0, None, 0, C.byref(h)) == 1
    assert L.frl_envpool_create(0, 0, 1, 0, None, 0, C.byref(h)) == 1
    assert L.frl_envpool_destroy(None) == 0
