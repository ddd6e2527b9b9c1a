"""Engines of the register-chained kernel family keep their nets in fragment-image order in HBM (NetDesc::frag,
device/chain_net.hpp).  The layout is internal: everything that crosses the C ABI — frl_params_get / set in the reference's
state_dict order, frl_act in every mode, a later frl_obsnorm_enable (which moves the engine to the row-chunk family) — must
be indistinguishable from an engine of the row-chunk family holding the same parameters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    assert _native.device_count() > 0, "no HIP device: the engine has no CPU fallback"
    return _native


def _pair(N, monkeypatch, algo, O, A, twin, P=3):
    from freerl_amd.engine import Engine
    g = np.random.default_rng(77)
    eng = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FRL_CRITIC_V2", flag)
        e = Engine(algo, O, A, 512, n_learners=P, twin_critic=twin, batch_max=128, seed=5)
        eng.append(e)
    ec, er = eng
    assert ec.learn_path(64)[0] and not er.learn_path(64)[0]
    for net in range(2):
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V):
            for p in range(P):
                flat = g.standard_normal(ec.num_params(net)).astype(np.float32) * 0.3
                ec.set_params(net, flat, kind, learner=p)
                er.set_params(net, flat, kind, learner=p)
                np.testing.assert_array_equal(ec.get_params(net, kind, learner=p), flat)      # the layout is invisible
    return ec, er, g


# (7, 3): the narrow standard shape of kernels_critic2 / _actor2 (act_frag_kernel reads the images directly); (40, 6) and (376, 17):
# the K-sliced chained family (kernels_criticw / _actorw), whose select_action goes through a Wk copy of the net (frag_to_wk_kernel)
@pytest.mark.parametrize("O,A", [(7, 3), (40, 6), (376, 17)])
@pytest.mark.parametrize("algo_name,twin", [("TD3", True), ("DDPG", False), ("SAC", True)])
def test_act_is_layout_independent(N, monkeypatch, algo_name, twin, O, A):
    algo = getattr(N, "ALGO_" + algo_name)
    P, rows = 3, 150                                # 150 rows: three 64-row workgroups, the last one ragged
    ec, er, g = _pair(N, monkeypatch, algo, O, A, twin, P)
    obs = g.standard_normal((P, rows, O)).astype(np.float32)
    oa = g.standard_normal((P, rows, O + A)).astype(np.float32)
    for tgt in (False, True):
        a1 = ec.act(0, N.ACT_TANHHEAD, obs, use_target=tgt, out_dim=A)
        a2 = er.act(0, N.ACT_TANHHEAD, obs, use_target=tgt, out_dim=A)
        np.testing.assert_allclose(a1, a2, rtol=2e-5, atol=2e-6)
        for head in range(2 if twin else 1):
            q1 = ec.act(1, N.ACT_RAW, oa, head=head, use_target=tgt, out_dim=1)
            q2 = er.act(1, N.ACT_RAW, oa, head=head, use_target=tgt, out_dim=1)
            np.testing.assert_allclose(q1, q2, rtol=2e-5, atol=2e-5)
    if algo == N.ALGO_SAC:
        eps = g.standard_normal((P, rows, A)).astype(np.float32)
        s1 = ec.act(0, N.ACT_SAC_SAMPLE, obs, eps=eps, out_dim=A)
        s2 = er.act(0, N.ACT_SAC_SAMPLE, obs, eps=eps, out_dim=A)
        np.testing.assert_allclose(s1, s2, rtol=2e-5, atol=2e-6)
    st1, env1 = ec.act_explore(N.ACT_TANHHEAD if algo != N.ALGO_SAC else N.ACT_SAC_SAMPLE, obs, kind=N.EXPLORE_GAUSS, sigma=0.1,
                               max_action=2.0, out_dim=A)
    st2, env2 = er.act_explore(N.ACT_TANHHEAD if algo != N.ALGO_SAC else N.ACT_SAC_SAMPLE, obs, kind=N.EXPLORE_GAUSS, sigma=0.1,
                               max_action=2.0, out_dim=A)
    np.testing.assert_allclose(st1, st2, rtol=2e-5, atol=2e-6)          # same Philox key and counter: same draws
    np.testing.assert_allclose(env1, env2, rtol=2e-5, atol=4e-6)
    ec.close(); er.close()


@pytest.mark.parametrize("O,A", [(6, 2), (30, 5)])
def test_obsnorm_enable_moves_a_chained_engine_to_the_row_chunk_family(N, monkeypatch, O, A):
    """Batch_ObsNorm is the row-chunk family's: enabling it re-lays every parameter array out as Wk; parameters read back
    unchanged and learn() then matches an engine that was row-chunk from the start.  (6, 2): the narrow chained family,
    (30, 5): the K-sliced one."""
    P, B = 2, 64
    ec, er, g = _pair(N, monkeypatch, N.ALGO_TD3, O, A, True, P)
    before = [[ec.get_params(net, kind, learner=p) for p in range(P)] for net in range(2)
              for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)]
    ec.obsnorm_enable(True); er.obsnorm_enable(True)
    assert not ec.learn_path(B)[0]
    after = [[ec.get_params(net, kind, learner=p) for p in range(P)] for net in range(2)
             for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)]
    for x, y in zip(before, after):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(u, v)
    # Adam's second moment must be non-negative for a meaningful step: overwrite v with squares on both engines
    for net in range(2):
        for p in range(P):
            v = np.abs(ec.get_params(net, N.PARAM_ADAM_V, learner=p)) * 1e-3
            ec.set_params(net, v, N.PARAM_ADAM_V, learner=p); er.set_params(net, v, N.PARAM_ADAM_V, learner=p)
    n_rows = 300
    recs = g.standard_normal((n_rows, ec.width)).astype(np.float32)
    recs[:, ec.layout.done_off] = g.random(n_rows) < 0.1
    for p in range(P):
        ec.add_batch(recs, learners=[p] * n_rows); er.add_batch(recs, learners=[p] * n_rows)
    idx = np.stack([g.choice(n_rows, B, replace=False) for _ in range(P)]).astype(np.int64)[:, None, :]
    nz = g.standard_normal((P, 1, 2, B, A)).astype(np.float32)
    kw = dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=True, use_policy_noise=True, policy_noise=0.2,
              noise_clip=0.5, max_action=1.0, idx=idx, noise=nz, want_stats=True)
    s1, s2 = ec.learn(B, **kw), er.learn(B, **kw)
    np.testing.assert_allclose(s1[:, 0, :2], s2[:, 0, :2], rtol=1e-6)
    for net in range(2):
        np.testing.assert_allclose(ec.get_params(net, learner=1), er.get_params(net, learner=1), rtol=1e-6, atol=1e-7)
    ec.close(); er.close()
