"""The narrow standard shape (obs 8, act 2, batch 256, hidden 128 — bench.py's workload) AT POPULATION SIZE, the family chosen by
frl_create on its own (no FRL_CRITIC_V2 / FRL_SOLO):

  * P = 512: the register-chained kernels bench.py's headline number comes from (kernels_critic2 / _actor2, eight-wave workgroups) —
    two rounds of workgroups on 256 CUs; learners 0, 255 (last of the first round), 256 (first of the second) and 511 are watched;
  * P = 24 / 30: kernels_solo.hip with eight workgroups per learner, two 16-row tiles each (populations of 17 .. 32 learners, round 6);
  * P = 40 and P = 128: the row-chunk kernels (ac_critic_kernel / ac_actor_kernel + adam_fused_kernel), which serve populations of
    17 .. 128 learners and whose only oracle test at population size moved to the solo kernels when those took P <= 16.

Every learner has its OWN parameters and its OWN transition table; the watched learners are compared with oracles run on exactly
their inputs: losses to 1e-4 per call, online nets, targets and Adam's first moment element-wise (tolerances and their derivation:
tests/test_gpu_wide_population.py).  TD3_file/TD3.py:189-233, SAC_file/SAC.py:222-260."""
import numpy as np
import pytest

from tests.golden import cases, synth
from tests.hip_helpers import flat_params, records
from tests.test_gpu_wide_population import AC, SAC_A, TWIN, _assert_adam_m, _assert_net, _fill

pytestmark = pytest.mark.gpu
O, A, B, N_TAB, CAP = 8, 2, 256, 600, 1024


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    assert _native.device_count() > 0
    return _native


def _watched(P):
    return (0, 255, 256, 511) if P == 512 else (0, P // 3, (2 * P) // 3 + 1, P - 1)


def _expect_family(e, P):
    chained, lds, rows = e.learn_path(B)
    if 16 < P <= 32:
        assert chained and rows == 32, "P = %d is meant to run kernels_solo.hip with eight workgroups per learner: %r" % (P, (chained, lds, rows))
    elif P > 128:
        assert chained and rows == B, "P = %d did not select the one-workgroup-per-learner chained kernels: %r" % (P, (chained, lds, rows))
    else:
        assert not chained, "P = %d is meant to run the row-chunk kernels: %r" % (P, (chained, lds, rows))


@pytest.mark.parametrize("P", [24, 40, 128, 512])
def test_td3_population_vs_oracles(N, monkeypatch, P):
    from freerl_amd.engine import Engine
    from oracle import algos
    for v in ("FRL_CRITIC_V2", "FRL_SOLO", "FRL_CHAIN_WAVES"):
        monkeypatch.delenv(v, raising=False)
    watch = _watched(P)
    e = Engine(N.ALGO_TD3, O, A, CAP, n_learners=P, twin_critic=True, batch_max=B)
    _expect_family(e, P)
    g = np.random.default_rng(1201)
    na, nc = e.num_params(0), e.num_params(1)
    tabs, actors, critics = {}, {}, {}
    for p in range(P):
        tabs[p] = synth.transitions(20000 + p, N_TAB, O, A)
        if p in watch:
            actors[p] = synth.mlp_params(21000 + p, cases.actor_layers(O, A))
            critics[p] = synth.mlp_params(22000 + p, cases.critic_layers(O + A, twin=True))
            fa, fc = flat_params(actors[p], AC), flat_params(critics[p], TWIN)
        else:
            fa, fc = (g.standard_normal(na) * 0.05).astype(np.float32), (g.standard_normal(nc) * 0.05).astype(np.float32)
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            e.set_params(0, fa, kind, learner=p)
            e.set_params(1, fc, kind, learner=p)
        recs = records([tabs[p]])
        e.add_batch(recs, learners=np.full(len(recs), p, np.int32))
    orcs = {}
    for p in watch:
        orcs[p] = algos.TD3(actors[p], critics[p], O, A, 1e-3, 1e-3, CAP)
        _fill(orcs[p], tabs[p])
    for k in range(2):
        idx = np.stack([synth.indices(23000 + 600 * k + p, N_TAB, B) for p in range(P)])[:, None]
        nz = np.zeros((P, 1, 2, B, A), np.float32)
        for p in range(P):
            nz[p, 0, 0] = synth.normal(24000 + 600 * k + p, (B, A))
        st = e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k % 2 == 1), use_policy_noise=True, policy_noise=0.2,
                     noise_clip=0.5, max_action=1.0, idx=idx, noise=nz, want_stats=True)
        for p in watch:
            cl, al = orcs[p].learn_with(idx[p, 0], nz[p, 0, 0], 0.99, 0.005, 0.2, 0.5, 1.0, 2, 1.0)
            np.testing.assert_allclose(st[p, 0, N.STAT_CRITIC_LOSS], cl, rtol=1e-4, err_msg="critic loss, learner %d call %d" % (p, k))
            if k % 2 == 1:
                np.testing.assert_allclose(st[p, 0, N.STAT_ACTOR_LOSS], al, rtol=1e-4, atol=1e-6, err_msg="actor loss, learner %d" % p)
    assert np.all(np.isfinite(st)), "a learner nobody watches produced a non-finite loss"
    for p in watch:
        o, lab = orcs[p], "td3 P=%d learner %d" % (P, p)
        _assert_net(e.get_params(1, N.PARAM_ONLINE, learner=p), o.critic, TWIN, None, 5e-4, 5e-6, lab + " critic")
        _assert_net(e.get_params(1, N.PARAM_TARGET, learner=p), o.critic_t, TWIN, None, 5e-4, 5e-6, lab + " critic_target")
        _assert_net(e.get_params(0, N.PARAM_ONLINE, learner=p), o.actor, AC, None, 5e-4, 5e-6, lab + " actor")
        _assert_net(e.get_params(0, N.PARAM_TARGET, learner=p), o.actor_t, AC, None, 5e-4, 5e-6, lab + " actor_target")
        _assert_adam_m(e.get_params(1, N.PARAM_ADAM_M, learner=p), o.critic_opt.m, TWIN, None, lab + " critic")
        _assert_adam_m(e.get_params(0, N.PARAM_ADAM_M, learner=p), o.actor_opt.m, AC, None, lab + " actor")
    e.close()


@pytest.mark.parametrize("P", [30, 128, 512])
def test_sac_population_vs_oracles(N, monkeypatch, P):
    """SAC at the same shape: the single-pass twin critic with the tanh-Gaussian target, the actor stage with both heads' dQ/da, the
    log_std and alpha steps — row-chunk kernels at 128 learners, the eight-wave chained ones at 512."""
    from freerl_amd.engine import Engine
    from oracle import algos
    for v in ("FRL_CRITIC_V2", "FRL_SOLO", "FRL_CHAIN_WAVES"):
        monkeypatch.delenv(v, raising=False)
    watch = _watched(P)
    e = Engine(N.ALGO_SAC, O, A, CAP, n_learners=P, twin_critic=True, batch_max=B)
    _expect_family(e, P)
    g = np.random.default_rng(1301)
    na, nc = e.num_params(0), e.num_params(1)
    tabs, actors, critics = {}, {}, {}
    for p in range(P):
        tabs[p] = synth.transitions(30000 + p, N_TAB, O, A)
        if p in watch:
            a = synth.mlp_params(31000 + p, cases.actor_layers(O, A, head="mean_layer"))
            actors[p] = dict([("log_std", np.random.default_rng(32000 + p).uniform(-0.5, 0.2, (1, A)).astype(np.float32))] + list(a.items()))
            critics[p] = synth.mlp_params(33000 + p, cases.critic_layers(O + A, twin=True))
            fa, fc = flat_params(actors[p], SAC_A, "log_std"), flat_params(critics[p], TWIN)
        else:
            fa, fc = (g.standard_normal(na) * 0.05).astype(np.float32), (g.standard_normal(nc) * 0.05).astype(np.float32)
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            e.set_params(0, fa, kind, learner=p)
            e.set_params(1, fc, kind, learner=p)
        e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0, learner=p)
        recs = records([tabs[p]])
        e.add_batch(recs, learners=np.full(len(recs), p, np.int32))
    orcs = {}
    for p in watch:
        orcs[p] = algos.SAC(actors[p], critics[p], O, A, 1e-3, 1e-3, CAP)
        _fill(orcs[p], tabs[p])
    for k in range(2):
        idx = np.stack([synth.indices(34000 + 600 * k + p, N_TAB, B) for p in range(P)])[:, None]
        nz = np.zeros((P, 1, 2, B, A), np.float32)
        for p in range(P):
            nz[p, 0] = np.random.default_rng(35000 + 600 * k + p).standard_normal((2, B, A)).astype(np.float32)
        st = e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, alpha_lr=1e-4, target_entropy=-float(A), idx=idx, noise=nz,
                     want_stats=True)
        for p in watch:
            cl, al, ll = orcs[p].learn_with(idx[p, 0], nz[p, 0, 0], nz[p, 0, 1], 0.99, 0.005)
            np.testing.assert_allclose(st[p, 0, N.STAT_CRITIC_LOSS], cl, rtol=1e-4, err_msg="critic loss, learner %d call %d" % (p, k))
            np.testing.assert_allclose(st[p, 0, N.STAT_ACTOR_LOSS], al, rtol=1e-4, atol=1e-5, err_msg="actor loss, learner %d call %d" % (p, k))
            np.testing.assert_allclose(st[p, 0, N.STAT_ALPHA_LOSS], ll, rtol=1e-4, err_msg="alpha loss, learner %d call %d" % (p, k))
    assert np.all(np.isfinite(st))
    for p in watch:
        o, lab = orcs[p], "sac P=%d learner %d" % (P, p)
        _assert_net(e.get_params(1, N.PARAM_ONLINE, learner=p), o.critic, TWIN, None, 5e-4, 5e-6, lab + " critic")
        _assert_net(e.get_params(1, N.PARAM_TARGET, learner=p), o.critic_t, TWIN, None, 5e-4, 5e-6, lab + " critic_target")
        _assert_net(e.get_params(0, N.PARAM_ONLINE, learner=p), o.actor, SAC_A, "log_std", 5e-4, 5e-6, lab + " actor")
        _assert_adam_m(e.get_params(1, N.PARAM_ADAM_M, learner=p), o.critic_opt.m, TWIN, None, lab + " critic")
        _assert_adam_m(e.get_params(0, N.PARAM_ADAM_M, learner=p), o.actor_opt.m, SAC_A, "log_std", lab + " actor")
    e.close()
