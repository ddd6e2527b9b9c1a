"""kernels_solo.hip draws a call's batch rows a launch AHEAD (a spare workgroup of the previous critic launch, SoloArgs::pre_read /
pre_write) and uses them only when their tag — Philox counter, ring size, batch — matches the call's own arguments.  The rows, and
with them every array of the engine, must be bit-identical to the in-launch draw (FRL_SOLO_PREDRAW=0), also across the events that
invalidate a pre-drawn set: add() between two calls (ring size), a changed batch, select_action's device draw in between (counter)."""
import numpy as np
import pytest

from tests.golden import cases, synth
from tests.hip_helpers import flat_params, records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    assert _native.device_count() > 0
    return _native


def _run(N, monkeypatch, predraw, algo, P, O=8, A=2):
    from freerl_amd.engine import Engine
    monkeypatch.setenv("FRL_SOLO_PREDRAW", "1" if predraw else "0")
    for v in ("FRL_CRITIC_V2", "FRL_SOLO", "FRL_SOLOW"):
        monkeypatch.delenv(v, raising=False)
    B, cap = 256, 4096
    twin = algo != N.ALGO_DDPG
    e = Engine(algo, O, A, cap, n_learners=P, twin_critic=twin, batch_max=B, seed=11)
    # (narrow shape: kernels_solo.hip, a spare workgroup draws; wide first layer: kernels_solow.hip, the learner's first helper workgroup)
    assert e.learn_path(B)[0] and e.learn_path(B)[2] == 16 and (e.learn_path(B)[1] == 117376) == (O + A <= 16 and A <= 4), "not the sixteen-workgroup kernels"
    g = np.random.default_rng(5)
    for p in range(P):
        for net in range(2):
            flat = (g.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
        if algo == N.ALGO_SAC:
            e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0, learner=p)
        tab = synth.transitions(700 + p, 1500, O, A)
        rec = records([tab])
        e.add_batch(rec, learners=np.full(len(rec), p, np.int32))
    extra = records([synth.transitions(990, 64, O, A)])
    kw = dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3)
    if algo == N.ALGO_TD3:
        kw.update(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)
    if algo == N.ALGO_SAC:
        kw.update(alpha_lr=1e-4, target_entropy=-float(A))
    out = []
    for k in range(14):
        batch = 200 if k in (9, 10) else B                             # a changed batch: the pre-drawn 256 rows do not fit
        if k in (4, 5):                                                # the ring grows between two calls: the tag's size is stale
            for p in range(P):
                e.add_batch(extra, learners=np.full(len(extra), p, np.int32))
        e.learn(batch, do_actor=(algo != N.ALGO_TD3 or k % 2 == 1), **kw)
        out.append(np.array(e.last_indices(batch)))
    arrays = [e.get_params(net, kind, learner=p) for p in range(P) for net in range(2)
              for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)]
    e.close()
    return out, arrays


@pytest.mark.parametrize("algo_name,P,O,A", [("td3", 1, 8, 2), ("sac", 3, 8, 2), ("ddpg", 15, 8, 2), ("sac", 1, 376, 17), ("td3", 5, 17, 6)])
def test_predrawn_rows_equal_the_in_launch_draw(N, monkeypatch, algo_name, P, O, A):
    algo = {"td3": N.ALGO_TD3, "sac": N.ALGO_SAC, "ddpg": N.ALGO_DDPG}[algo_name]
    idx1, arr1 = _run(N, monkeypatch, True, algo, P, O, A)
    idx0, arr0 = _run(N, monkeypatch, False, algo, P, O, A)
    for k, (a, b) in enumerate(zip(idx1, idx0)):
        np.testing.assert_array_equal(a, b, err_msg="rows of call %d" % k)
        assert len(np.unique(a.reshape(P, -1)[0])) == a.reshape(P, -1).shape[1], "a batch holds a row twice"
    for i, (a, b) in enumerate(zip(arr1, arr0)):
        np.testing.assert_array_equal(a, b, err_msg="array %d" % i)


def _run_maddpg(N, monkeypatch, predraw, B, P):
    """MADDPG on kernels_solow.hip: a spare workgroup per (learner, agent) unit of a critic launch draws the unit's rows for the next call;
    the host skips draw_kernel when the next call asks for exactly that counter / ring size / batch."""
    from freerl_amd.engine import Engine
    monkeypatch.setenv("FRL_SOLO_PREDRAW", "1" if predraw else "0")
    for v in ("FRL_CRITIC_V2", "FRL_SOLOW"):
        monkeypatch.delenv(v, raising=False)
    e = Engine(N.ALGO_MADDPG, [6, 5, 7], [2, 3, 2], 4096, n_learners=P, batch_max=B, seed=17)
    assert e.learn_path(B)[0] and e.learn_path(B)[2] == 16, "not kernels_solow.hip"
    e.fill_synthetic(3000, seed=4)
    out = []
    for k in range(12):
        batch = B // 2 if k in (7, 8) else B                         # a changed batch: the pre-drawn rows are for another one
        if k in (3, 4):                                                # the rings grow between two calls: drawn for a stale size
            e.fill_synthetic(3000 + 200 * (k - 2), seed=4)
        e.learn(batch, gamma=0.95, tau=0.01, actor_lr=1e-3, critic_lr=1e-3)
        out.append(np.array(e.last_indices(batch)))
    arrays = [e.get_params(net, kind, learner=p) for p in range(P) for net in range(6)
              for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)]
    e.close()
    return out, arrays


@pytest.mark.parametrize("B,P", [(1024, 1), (128, 4)])
def test_maddpg_predrawn_rows_equal_draw_kernel_s(N, monkeypatch, B, P):
    idx1, arr1 = _run_maddpg(N, monkeypatch, True, B, P)
    idx0, arr0 = _run_maddpg(N, monkeypatch, False, B, P)
    for k, (a, b) in enumerate(zip(idx1, idx0)):
        np.testing.assert_array_equal(a, b, err_msg="rows of call %d" % k)
    for i, (a, b) in enumerate(zip(arr1, arr0)):
        np.testing.assert_array_equal(a, b, err_msg="array %d" % i)
