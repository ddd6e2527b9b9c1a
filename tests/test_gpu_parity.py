"""GPU parity: the HIP engine (through the C ABI) against the oracle run live on the same seeded
inputs, and against the golden outputs of the reference (tests/golden/*.npz).

Tolerances (fp32, different reduction orders: MFMA k-permuted fma chains vs BLAS):
  * per-call losses: 1e-4 relative — the bar BASELINE.json's north_star states ("TD-loss curve
    matching reference seed=0 to 1e-4 rel-tol"); observed ~1e-6;
  * parameters after n Adam steps: rtol 5e-4 / atol 5e-6 (Adam's m/sqrt(v) turns 1-ulp gradient
    differences on near-zero gradients into visible parameter differences of order lr*1e-3);
  * Buffer.sample: bit-exact (pure data movement).
"""
import json
import os

import numpy as np
import pytest

from tests.golden import cases, synth
from tests.hip_helpers import flat_params, records, rel_err, unflat_params

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LOSS_RTOL = 1e-4
P_RTOL, P_ATOL = 5e-4, 5e-6
REPORT = {}


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    assert _native.device_count() > 0, "no HIP device: the engine has no CPU fallback"
    return _native


@pytest.fixture(scope="module", autouse=True)
def _dump_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_report.json")
    merged = {}
    if os.path.exists(path):          # a partial run (-k ...) updates its entries and keeps the others
        try:
            merged = json.load(open(path))
        except Exception:
            merged = {}
    merged.update(REPORT)
    with open(path, "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)


def note(key, val):
    REPORT[key] = float(val)


def assert_params_close(got, want, label, atol=P_ATOL):
    worst = 0.0
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=P_RTOL, atol=atol, err_msg="%s %s" % (label, k))
        worst = max(worst, rel_err(got[k], want[k]))
    note("param_relerr/" + label, worst)


# ------------------------------------------------------------------------------------ buffer
def test_buffer_add_wrap_sample_bit_exact(N):
    import torch
    from freerl_amd.engine import Engine
    c = cases.CASES["buffer"]
    inp = cases.buffer_inputs(c)
    fx = gold("buffer")
    e = Engine(N.ALGO_REPLAY_ONLY, c["obs_dim"], c["act_dim"], c["capacity"], batch_max=c["batch"])
    recs = records([inp["table"]])
    for i in range(c["n_add"]):                      # one by one: exercises the pinned staging + wrap
        e.add(0, recs[i])
    assert e.cursor(0) == (int(fx["index"]), int(fx["size"]))
    lay = e.layout
    O, A = c["obs_dim"], c["act_dim"]
    fields = [(lay.obs_off[0], O), (lay.act_off[0], A), (lay.rew_off, 1), (lay.next_obs_off[0], O), (lay.done_off, 1)]
    outs = [torch.empty((c["batch"], w), dtype=torch.float32, device="cuda") for _, w in fields]
    e.sample_into(0, inp["idx"], fields, [t.data_ptr() for t in outs])
    for t, key in zip(outs, ["obs", "act", "rew", "next_obs", "done"]):
        np.testing.assert_array_equal(t.cpu().numpy(), fx[key])
    # whole-record read-back equals what the ring must hold after the wrap
    rows = e.read_rows(0, 0, c["capacity"])
    expect = np.zeros_like(rows)
    for i in range(c["n_add"]):
        expect[i % c["capacity"]] = recs[i]
    np.testing.assert_array_equal(rows, expect)
    e.close()


def test_buffer_empty_ragged_and_errors(N):
    import torch
    from freerl_amd.engine import Engine
    e = Engine(N.ALGO_REPLAY_ONLY, 3, 1, 8, batch_max=8)
    assert e.cursor(0) == (0, 0)
    out = torch.empty((0, 3), dtype=torch.float32, device="cuda")
    e.sample_into(0, np.zeros(0, np.int64), [(0, 3)], [out.data_ptr()])        # empty sample is a no-op
    recs = np.arange(3 * e.width, dtype=np.float32).reshape(3, e.width)
    e.add_batch(recs)
    got = torch.empty((4, 3), dtype=torch.float32, device="cuda")
    e.sample_into(0, np.array([2, 0, -8, 1]), [(0, 3)], [got.data_ptr()])      # -8 wraps like NumPy indexing
    np.testing.assert_array_equal(got.cpu().numpy(), recs[[2, 0, 0, 1], :3])
    with pytest.raises(N.FrlError):
        e.sample_into(0, np.array([8]), [(0, 3)], [got.data_ptr()])            # IndexError in the reference
    with pytest.raises(N.FrlError):
        e.learn(4, gamma=0.99, tau=0.01)                                       # replay-only engine
    e.close()


# --------------------------------------------------------------------------------------- DQN
@pytest.fixture(params=["rowchunk", "fused", "fused_split"])
def dqn_path(request, monkeypatch):
    """DQN.learn has two implementations behind frl_learn: draw_kernel -> dqn_grad_kernel (row chunks, slabs) ->
    adam_fused_kernel (any head), and dqn_fused_kernel (kernels_dqn2.hip: the plain / Double Q-net, everything in one launch) —
    with one workgroup per learner or the batch's 64-row chunks on several (the last to arrive reduces and steps)."""
    monkeypatch.setenv("FRL_DQN_FUSED", "0" if request.param == "rowchunk" else "1")
    monkeypatch.setenv("FRL_DQN_SPLIT", "1" if request.param == "fused" else "4")
    return request.param


def test_dqn_learn_matches_oracle_and_reference(N, dqn_path):
    from freerl_amd.engine import Engine
    from oracle import algos
    c = cases.CASES["dqn"]
    inp = cases.dqn_inputs(c)
    fx = gold("dqn")
    names = ["l1", "l2"]
    e = Engine(N.ALGO_DQN, c["obs_dim"], c["n_actions"], c["capacity"], discrete=True, batch_max=c["batch"])
    flat = flat_params(inp["params"]["Qnet"], names)
    e.set_params(0, flat, N.PARAM_ONLINE)
    e.set_params(0, flat, N.PARAM_TARGET)
    e.add_batch(records([inp["table"]]))
    orc = algos.DQN(inp["params"]["Qnet"], c["obs_dim"], c["n_actions"], c["lr"], c["capacity"])
    tab = inp["table"]
    for i in range(c["n_table"]):
        orc.add(tab["obs"][i], tab["act"][i][0], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    q0 = e.act(0, N.ACT_RAW, tab["obs"][:32], out_dim=c["n_actions"])[0]
    np.testing.assert_allclose(q0, fx["q0"], rtol=1e-5, atol=1e-6)
    greedy = e.act(0, N.ACT_ARGMAX, tab["obs"][:32])[0, :, 0].astype(np.int64)
    np.testing.assert_array_equal(greedy, fx["select_action"])
    losses = []
    for k in range(c["n_learn"]):
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], critic_lr=c["lr"], clip_norm=0.0,
                     idx=inp["idx"][k], want_stats=True)
        losses.append(st[0, 0, N.STAT_CRITIC_LOSS])
        orc.learn_with(inp["idx"][k], c["gamma"], c["tau"])
    np.testing.assert_allclose(losses, np.array(orc.losses), rtol=LOSS_RTOL)
    np.testing.assert_allclose(losses, fx["loss"], rtol=LOSS_RTOL)
    note("loss_relerr/dqn", rel_err(losses, fx["loss"], 1e-6))
    got = unflat_params(e.get_params(0, N.PARAM_ONLINE), orc.q, names)
    assert_params_close(got, orc.q, "dqn/Qnet")
    got_t = unflat_params(e.get_params(0, N.PARAM_TARGET), orc.q_t, names)
    assert_params_close(got_t, orc.q_t, "dqn/Qnet_target")
    synth.check_digest("Qnet", got, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    got_m = unflat_params(e.get_params(0, N.PARAM_ADAM_M), orc.q, names)
    for k in got_m:
        np.testing.assert_allclose(got_m[k], orc.opt.m[k], rtol=1e-3, atol=1e-7)
    assert e.opt_step(0) == int(fx["step"])
    e.close()


@pytest.mark.parametrize("double,clip,wd,batch", [(True, 0.0, 0.0, 256), (False, 0.05, 0.0, 256), (True, 0.5, 1e-2, 200), (False, 0.0, 0.0, 37)])
def test_dqn_double_clip_weight_decay_ragged_batches(N, dqn_path, double, clip, wd, batch):
    """The Double target (DQN_with_tricks.py:263-265: the online net picks, the target net values), clip_grad_norm_, L2 weight
    decay and batches that end inside a 64-row chunk / inside a 16-row tile, three learners with different tables: both
    implementations against the oracle."""
    from freerl_amd.engine import Engine
    from oracle import algos, nn
    c = cases.CASES["dqn"]
    P = 3
    e = Engine(N.ALGO_DQN, c["obs_dim"], c["n_actions"], c["capacity"], discrete=True, batch_max=256, n_learners=P)
    orcs = []
    for p in range(P):
        inp = cases.dqn_inputs(dict(c, table_seed=c["table_seed"] + p, param_seed=c["param_seed"] + p))
        flat = flat_params(inp["params"]["Qnet"], ["l1", "l2"])
        e.set_params(0, flat, N.PARAM_ONLINE, learner=p); e.set_params(0, flat, N.PARAM_TARGET, learner=p)
        recs = records([inp["table"]])
        e.add_batch(recs, learners=np.full(len(recs), p, np.int32))
        orc = algos.DQN(inp["params"]["Qnet"], c["obs_dim"], c["n_actions"], c["lr"], c["capacity"])
        orc.opt.wd = wd
        orc.clip = clip
        tab = inp["table"]
        for i in range(c["n_table"]):
            orc.add(tab["obs"][i], tab["act"][i][0], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
        orcs.append(orc)
    g = np.random.default_rng(5)
    for k in range(4):
        idx = np.stack([g.choice(c["n_table"], batch, replace=False) for _ in range(P)]).astype(np.int32)
        st = e.learn(batch, gamma=c["gamma"], tau=c["tau"], critic_lr=c["lr"], clip_norm=clip, critic_weight_decay=wd,
                     double_dqn=double, idx=idx[:, None, :], want_stats=True)
        for p in range(P):
            orcs[p].learn_with(idx[p], c["gamma"], c["tau"], double=double)
            np.testing.assert_allclose(st[p, 0, N.STAT_CRITIC_LOSS], orcs[p].losses[-1], rtol=LOSS_RTOL)
    for p in range(P):
        got = unflat_params(e.get_params(0, N.PARAM_ONLINE, learner=p), orcs[p].q, ["l1", "l2"])
        assert_params_close(got, orcs[p].q, "dqn2/Qnet/%d" % p)
        got_t = unflat_params(e.get_params(0, N.PARAM_TARGET, learner=p), orcs[p].q_t, ["l1", "l2"])
        assert_params_close(got_t, orcs[p].q_t, "dqn2/Qnet_target/%d" % p)
        assert e.opt_step(0, learner=p) == 4
    e.close()


@pytest.mark.parametrize("obs_dim,n_act,dueling", [(16, 16, False), (3, 2, False), (5, 15, True), (16, 7, True)])
def test_dqn_head_shapes_at_the_tile_edges(N, dqn_path, obs_dim, n_act, dueling):
    """Inputs and heads that fill (or barely use) the 16-wide tiles of both DQN implementations — 16 observation columns, 16
    actions, a Dueling head of 1 + 15 rows — Double target, batch 96 (a full chunk and a half): against the oracle."""
    from freerl_amd.engine import Engine
    from oracle import algos
    c = dict(cases.CASES["dqn"], obs_dim=obs_dim, n_actions=n_act, batch=96, n_learn=3)
    inp = (cases.dqn_dueling_inputs if dueling else cases.dqn_inputs)(c)
    prm = inp["params"]["Qnet"]
    if dueling:                      # engine head rows: [V ; A]
        eng = {"l1.weight": prm["l1.weight"], "l1.bias": prm["l1.bias"],
               "l2.weight": np.vstack([prm["V.weight"], prm["A.weight"]]), "l2.bias": np.concatenate([prm["V.bias"], prm["A.bias"]])}
    else:
        eng = prm
    e = Engine(N.ALGO_DQN, obs_dim, n_act, c["capacity"], discrete=True, batch_max=96, dueling=dueling)
    flat = flat_params(eng, ["l1", "l2"])
    e.set_params(0, flat, N.PARAM_ONLINE); e.set_params(0, flat, N.PARAM_TARGET)
    e.add_batch(records([inp["table"]]))
    orc = algos.DQN(prm, obs_dim, n_act, c["lr"], c["capacity"], dueling=dueling)
    tab = inp["table"]
    for i in range(c["n_table"]):
        orc.add(tab["obs"][i], tab["act"][i][0], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    greedy = e.act(0, N.ACT_ARGMAX, tab["obs"][:64])[0, :, 0].astype(np.int64)
    np.testing.assert_array_equal(greedy, [orc.select_action(tab["obs"][i]) for i in range(64)])
    for k in range(c["n_learn"]):
        st = e.learn(96, gamma=c["gamma"], tau=c["tau"], critic_lr=c["lr"], clip_norm=0.0, double_dqn=True, idx=inp["idx"][k], want_stats=True)
        orc.learn_with(inp["idx"][k], c["gamma"], c["tau"], double=True)
        np.testing.assert_allclose(st[0, 0, N.STAT_CRITIC_LOSS], orc.losses[-1], rtol=LOSS_RTOL)
    for kind, want in ((N.PARAM_ONLINE, orc.q), (N.PARAM_TARGET, orc.q_t)):
        got = unflat_params(e.get_params(0, kind), eng, ["l1", "l2"])
        if dueling:
            got = {"l1.weight": got["l1.weight"], "l1.bias": got["l1.bias"], "V.weight": got["l2.weight"][:1], "V.bias": got["l2.bias"][:1],
                   "A.weight": got["l2.weight"][1:], "A.bias": got["l2.bias"][1:]}
        assert_params_close(got, want, "dqn_edges/%d/%d/%d" % (obs_dim, n_act, dueling))
    e.close()


# ------------------------------------------------------------------------- DDPG / TD3 / SAC
AC_NAMES = ["l1", "l2", "l3"]
TWIN_NAMES = ["l1", "l2", "l3", "l4", "l5", "l6"]


def _setup_ac(N, algo, c, inp, twin, actor_names, actor_extra=None, n_learners=1):
    from freerl_amd.engine import Engine
    e = Engine(algo, c["obs_dim"], c["act_dim"], c["capacity"], twin_critic=twin, batch_max=c["batch"],
               n_learners=n_learners)
    fa = flat_params(inp["params"]["actor"], actor_names, actor_extra)
    fc = flat_params(inp["params"]["critic"], TWIN_NAMES if twin else AC_NAMES)
    for p in range(n_learners):
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            e.set_params(0, fa, kind, learner=p)
            e.set_params(1, fc, kind, learner=p)
    recs = records([inp["table"]])
    for p in range(n_learners):
        e.add_batch(recs, learners=np.full(len(recs), p, np.int32))
    return e


def _fill_oracle(orc, tab):
    for i in range(len(tab["rew"])):
        orc.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))


def _check_ac_params(N, e, orc, twin, actor_names, label, actor_extra=None, learner=0):
    cn = TWIN_NAMES if twin else AC_NAMES
    online = None
    for kind, oa, oc, tag in ((N.PARAM_ONLINE, orc.actor, orc.critic, ""), (N.PARAM_TARGET, orc.actor_t, orc.critic_t, "_target")):
        ga = unflat_params(e.get_params(0, kind, learner=learner), oa, actor_names, actor_extra)
        gc = unflat_params(e.get_params(1, kind, learner=learner), oc, cn)
        assert_params_close(ga, oa, "%s/actor%s" % (label, tag))
        assert_params_close(gc, oc, "%s/critic%s" % (label, tag))
        if online is None:
            online = (ga, gc)
    return online


@pytest.fixture(params=["rowchunk", "chained", "solo"])
def ac_path(request, monkeypatch):
    """The critic stage of DDPG / TD3 / SAC has two implementations behind frl_learn: the row-chunk kernels + reduce / Adam
    launches (any shape; what populations up to 128 learners get) and the one-workgroup-per-learner register-chained kernel
    with Adam fused (kernels_critic2.hip; the bench's path).  FRL_CRITIC_V2 forces either, so both meet the same oracle."""
    if request.param == "solo":       # nothing forced: a single learner of the narrow shape runs kernels_solo.hip (sixteen workgroups)
        monkeypatch.delenv("FRL_CRITIC_V2", raising=False)
        monkeypatch.delenv("FRL_SOLO", raising=False)
        monkeypatch.delenv("FRL_DQN_FUSED", raising=False)
        return request.param
    monkeypatch.setenv("FRL_CRITIC_V2", "1" if request.param == "chained" else "0")
    monkeypatch.setenv("FRL_DQN_FUSED", "1" if request.param == "chained" else "0")      # (tests that also touch DQN: both of its paths)
    return request.param


def test_ddpg_learn(N, ac_path):
    from oracle import algos
    c = cases.CASES["ddpg"]
    inp = cases.ac_inputs(c, twin=False)
    fx = gold("ddpg")
    e = _setup_ac(N, N.ALGO_DDPG, c, inp, False, AC_NAMES)
    orc = algos.DDPG(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"],
                     c["critic_lr"], c["capacity"])
    _fill_oracle(orc, inp["table"])
    sa = e.act(0, N.ACT_TANHHEAD, inp["table"]["obs"][:32], out_dim=c["act_dim"])[0]
    np.testing.assert_allclose(sa, fx["select_action"], rtol=1e-5, atol=1e-6)
    cl, al = [], []
    for k in range(c["n_learn"]):
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     idx=inp["idx"][k], want_stats=True)
        cl.append(st[0, 0, N.STAT_CRITIC_LOSS]); al.append(st[0, 0, N.STAT_ACTOR_LOSS])
        orc.learn_with(inp["idx"][k], None, c["gamma"], c["tau"])
    np.testing.assert_allclose(cl, fx["loss_critic"], rtol=LOSS_RTOL)
    np.testing.assert_allclose(al, fx["loss_actor"], rtol=LOSS_RTOL, atol=1e-6)
    np.testing.assert_allclose(cl, np.array(orc.critic_losses), rtol=LOSS_RTOL)
    note("loss_relerr/ddpg_critic", rel_err(cl, fx["loss_critic"], 1e-6))
    _check_ac_params(N, e, orc, False, AC_NAMES, "ddpg")
    assert e.opt_step(0) == int(fx["actor_step"]) and e.opt_step(1) == int(fx["critic_step"])
    e.close()


def test_ddpg_critic_weight_decay(N, ac_path):
    """DDPG.py's supplement['weight_decay'] on its own (critic Adam weight_decay 1e-3, L2-in-gradient form, DDPG.py:131-134) on
    both critic-stage implementations (with Batch_ObsNorm, as in the ddpg_full golden, only the row-chunk kernels run)."""
    from oracle import algos
    c = cases.CASES["ddpg"]
    inp = cases.ac_inputs(c, twin=False)
    e = _setup_ac(N, N.ALGO_DDPG, c, inp, False, AC_NAMES)
    orc = algos.DDPG(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"], c["critic_lr"],
                     c["capacity"], critic_weight_decay=1e-3)
    _fill_oracle(orc, inp["table"])
    cl = []
    for k in range(c["n_learn"]):
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     critic_weight_decay=1e-3, idx=inp["idx"][k], want_stats=True)
        cl.append(st[0, 0, N.STAT_CRITIC_LOSS])
        orc.learn_with(inp["idx"][k], None, c["gamma"], c["tau"])
    np.testing.assert_allclose(cl, np.array(orc.critic_losses), rtol=LOSS_RTOL)
    _check_ac_params(N, e, orc, False, AC_NAMES, "ddpg_wd/" + ac_path)
    e.close()


@pytest.mark.parametrize("name", ["td3", "td3_pendulum"])
def test_td3_learn(N, name, ac_path):
    from oracle import algos
    c = cases.CASES[name]
    inp = cases.ac_inputs(c, twin=True)
    fx = gold(name)
    e = _setup_ac(N, N.ALGO_TD3, c, inp, True, AC_NAMES)
    orc = algos.TD3(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"],
                    c["critic_lr"], c["capacity"])
    _fill_oracle(orc, inp["table"])
    cl, al = [], []
    for k in range(c["n_learn"]):
        total_it = k + 1
        do_actor = total_it % c["policy_freq"] == 0
        nz = np.zeros((1, 1, 2, c["batch"], c["act_dim"]), np.float32)
        nz[0, 0, 0] = inp["noise"][k][0]
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     do_actor=do_actor, use_policy_noise=True, policy_noise=c["policy_noise"],
                     noise_clip=c["noise_clip"], max_action=c["max_action"], policy_noise_scale=c["policy_noise_scale"],
                     idx=inp["idx"][k], noise=nz, want_stats=True)
        cl.append(st[0, 0, N.STAT_CRITIC_LOSS])
        if do_actor:
            al.append(st[0, 0, N.STAT_ACTOR_LOSS])
        orc.learn_with(inp["idx"][k], inp["noise"][k][0], c["gamma"], c["tau"], c["policy_noise"], c["noise_clip"],
                       c["max_action"], c["policy_freq"], c["policy_noise_scale"])
    np.testing.assert_allclose(cl, fx["loss_critic"], rtol=LOSS_RTOL)
    np.testing.assert_allclose(al, fx["loss_actor"], rtol=LOSS_RTOL, atol=1e-6)
    note("loss_relerr/%s_critic" % name, rel_err(cl, fx["loss_critic"], 1e-6))
    note("loss_relerr/%s_actor" % name, rel_err(al, fx["loss_actor"], 1e-6))
    ga, gc = _check_ac_params(N, e, orc, True, AC_NAMES, name)
    synth.check_digest("actor", ga, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    synth.check_digest("critic", gc, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    assert e.opt_step(0) == int(fx["actor_step"]) and e.opt_step(1) == int(fx["critic_step"])
    e.close()


def test_sac_learn(N, ac_path):
    from oracle import algos
    c = cases.CASES["sac"]
    inp = cases.ac_inputs(c, twin=True, gaussian=True)
    fx = gold("sac")
    an = ["l1", "l2", "mean_layer"]
    e = _setup_ac(N, N.ALGO_SAC, c, inp, True, an, "log_std")
    e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0)        # Alpha(alpha=0.01) (SAC.py:188)
    orc = algos.SAC(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"],
                    c["critic_lr"], c["capacity"])
    _fill_oracle(orc, inp["table"])
    ev = e.act(0, N.ACT_TANHHEAD, inp["table"]["obs"][:32], out_dim=c["act_dim"])[0]
    np.testing.assert_allclose(ev, fx["evaluate_action"], rtol=1e-5, atol=1e-6)
    cl, al, ll, alphas = [], [], [], []
    for k in range(c["n_learn"]):
        nz = np.stack([inp["noise"][k][0], inp["noise"][k][1]])[None, None]
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     alpha_lr=1e-4, target_entropy=-float(c["act_dim"]), idx=inp["idx"][k], noise=nz, want_stats=True)
        cl.append(st[0, 0, N.STAT_CRITIC_LOSS]); al.append(st[0, 0, N.STAT_ACTOR_LOSS])
        ll.append(st[0, 0, N.STAT_ALPHA_LOSS]); alphas.append(st[0, 0, N.STAT_ALPHA])
        orc.learn_with(inp["idx"][k], inp["noise"][k][0], inp["noise"][k][1], c["gamma"], c["tau"])
    np.testing.assert_allclose(cl, fx["loss_critic"], rtol=LOSS_RTOL)
    np.testing.assert_allclose(al, fx["loss_actor"], rtol=LOSS_RTOL, atol=2e-6)
    np.testing.assert_allclose(ll, fx["loss_alpha"], rtol=LOSS_RTOL)
    np.testing.assert_allclose(alphas, fx["alpha"], rtol=1e-5)
    note("loss_relerr/sac_critic", rel_err(cl, fx["loss_critic"], 1e-6))
    note("loss_relerr/sac_actor", rel_err(al, fx["loss_actor"], 1e-6))
    ga, gc = _check_ac_params(N, e, orc, True, an, "sac", "log_std")
    synth.check_digest("actor", ga, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    synth.check_digest("critic", gc, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    # stochastic select_action on the UPDATED actor with the reference's eps (generated after learn)
    eps = np.stack([synth.normal(c["noise_seed"] + 900 + i, (1, c["act_dim"]))[0] for i in range(8)])
    sa = e.act(0, N.ACT_SAC_SAMPLE, inp["table"]["obs"][:8], eps=eps, out_dim=c["act_dim"])[0]
    np.testing.assert_allclose(sa, fx["select_action"], rtol=5e-4, atol=5e-5)
    vals, t = e.alpha_state()
    np.testing.assert_allclose(vals[0], fx["log_alpha"], rtol=1e-5)
    assert t == c["n_learn"]
    e.close()


@pytest.mark.parametrize("delta", [0.5, 10.0])
def test_huber_td_loss_option(N, delta, ac_path):
    """frl_learn_args.loss_kind = FRL_LOSS_HUBER (north_star's "Huber/MSE TD-loss"; the reference's huber_loss, MAPPO.py:
    273-276, pinned on the oracle side by tests/golden/huber.npz): DQN, TD3 and SAC against the oracle with the same
    injected indices / noise.  delta 0.5 puts most TD errors on the linear branch, 10 (the reference's default) on the
    quadratic one — where the loss is exactly half the MSE."""
    from oracle import algos, nn
    # DQN
    c = cases.CASES["dqn"]
    inp = cases.dqn_inputs(c)
    from freerl_amd.engine import Engine
    e = Engine(N.ALGO_DQN, c["obs_dim"], c["n_actions"], c["capacity"], discrete=True, batch_max=c["batch"])
    flat = flat_params(inp["params"]["Qnet"], ["l1", "l2"])
    e.set_params(0, flat, N.PARAM_ONLINE); e.set_params(0, flat, N.PARAM_TARGET)
    e.add_batch(records([inp["table"]]))
    orc = algos.DQN(inp["params"]["Qnet"], c["obs_dim"], c["n_actions"], c["lr"], c["capacity"])
    orc.td_loss = nn.huber(delta)
    tab = inp["table"]
    for i in range(c["n_table"]):
        orc.add(tab["obs"][i], tab["act"][i][0], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    losses = []
    for k in range(c["n_learn"]):
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], critic_lr=c["lr"], clip_norm=0.0, idx=inp["idx"][k],
                     want_stats=True, huber_delta=delta)
        losses.append(st[0, 0, N.STAT_CRITIC_LOSS])
        orc.learn_with(inp["idx"][k], c["gamma"], c["tau"])
    np.testing.assert_allclose(losses, np.array(orc.losses), rtol=LOSS_RTOL)
    mse = gold("dqn")["loss"]
    if delta >= 10:
        np.testing.assert_allclose(losses[0], 0.5 * mse[0], rtol=1e-5)      # all errors quadratic: half the MSE on the first call
    else:
        assert losses[0] < 0.5 * mse[0]
    assert_params_close(unflat_params(e.get_params(0), orc.q, ["l1", "l2"]), orc.q, "huber/dqn")
    e.close()
    # TD3 (twin critic: both heads) and SAC
    for algo_id, name, gaussian in ((N.ALGO_TD3, "td3", False), (N.ALGO_SAC, "sac", True)):
        c = cases.CASES[name]
        inp = cases.ac_inputs(c, twin=True, gaussian=gaussian)
        an = ["l1", "l2", "mean_layer"] if gaussian else AC_NAMES
        e = _setup_ac(N, algo_id, c, inp, True, an, "log_std" if gaussian else None)
        if gaussian:
            e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0)
            orc = algos.SAC(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"], c["critic_lr"], c["capacity"])
        else:
            orc = algos.TD3(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"], c["critic_lr"], c["capacity"])
        orc.td_loss = nn.huber(delta)
        _fill_oracle(orc, inp["table"])
        got, want = [], []
        for k in range(c["n_learn"]):
            if gaussian:
                nz = np.stack([inp["noise"][k][0], inp["noise"][k][1]])[None, None]
                st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"], alpha_lr=1e-4,
                             target_entropy=-float(c["act_dim"]), idx=inp["idx"][k], noise=nz, want_stats=True, huber_delta=delta)
                want.append(orc.learn_with(inp["idx"][k], inp["noise"][k][0], inp["noise"][k][1], c["gamma"], c["tau"])[0])
            else:
                nz = np.zeros((1, 1, 2, c["batch"], c["act_dim"]), np.float32)
                nz[0, 0, 0] = inp["noise"][k][0]
                st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                             do_actor=(k + 1) % c["policy_freq"] == 0, use_policy_noise=True, policy_noise=c["policy_noise"],
                             noise_clip=c["noise_clip"], max_action=c["max_action"], policy_noise_scale=c["policy_noise_scale"],
                             idx=inp["idx"][k], noise=nz, want_stats=True, huber_delta=delta)
                want.append(orc.learn_with(inp["idx"][k], inp["noise"][k][0], c["gamma"], c["tau"], c["policy_noise"], c["noise_clip"],
                                           c["max_action"], c["policy_freq"], c["policy_noise_scale"])[0])
            got.append(st[0, 0, N.STAT_CRITIC_LOSS])
        np.testing.assert_allclose(got, want, rtol=LOSS_RTOL)
        _check_ac_params(N, e, orc, True, an, "huber/" + name, "log_std" if gaussian else None)
        e.close()
    # refused where it has no meaning
    e = Engine(N.ALGO_DQN, 4, 3, 64, discrete=True, batch_max=8, c51=(51, -10.0, 10.0))
    e.fill_synthetic(32, seed=1)
    with pytest.raises(N.FrlError):
        e.learn(8, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0, huber_delta=1.0)
    e.close()


@pytest.fixture(params=["rowchunk", "chained"])
def ma_path(request, monkeypatch):
    """MADDPG / MATD3 on both families: the row-chunk kernels and the K-sliced chained ones (kernels_criticw / _actorw: one
    workgroup per (learner, agent); what populations of more than 128 units run)."""
    monkeypatch.setenv("FRL_CRITIC_V2", "1" if request.param == "chained" else "0")
    return request.param == "chained"


def test_maddpg_learn(N, ma_path):
    from freerl_amd.engine import Engine
    from oracle import algos
    c = cases.CASES["maddpg"]
    inp = cases.maddpg_inputs(c)
    fx = gold("maddpg")
    ids = inp["ids"]
    od = [c["dims"][a][0] for a in ids]
    ad = [c["dims"][a][1] for a in ids]
    e = Engine(N.ALGO_MADDPG, od, ad, c["capacity"], batch_max=c["batch"])
    assert e.learn_path(c["batch"])[0] == bool(ma_path)
    for j, a in enumerate(ids):
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            e.set_params(2 * j, flat_params(inp["params"][a]["actor"], AC_NAMES), kind)
            e.set_params(2 * j + 1, flat_params(inp["params"][a]["critic"], AC_NAMES), kind)
    e.add_batch(records([inp["tables"][a] for a in ids]))
    orc = algos.MADDPG(inp["params"], c["dims"], c["actor_lr"], c["critic_lr"], c["capacity"])
    for i in range(c["n_table"]):
        orc.add({a: inp["tables"][a]["obs"][i] for a in ids}, {a: inp["tables"][a]["act"][i] for a in ids},
                {a: float(inp["tables"][a]["rew"][i]) for a in ids}, {a: inp["tables"][a]["next_obs"][i] for a in ids},
                {a: bool(inp["tables"][a]["done"][i]) for a in ids})
    for j, a in enumerate(ids):
        act = e.act(2 * j, N.ACT_TANHHEAD, inp["tables"][a]["obs"][:1], out_dim=ad[j])[0, 0]
        np.testing.assert_allclose(act, fx["select_action/" + a], rtol=1e-5, atol=1e-6)
    cl = {a: [] for a in ids}
    al = {a: [] for a in ids}
    for k in range(c["n_learn"]):
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     idx=np.stack(inp["idx"][k])[None], want_stats=True)
        for j, a in enumerate(ids):
            cl[a].append(st[0, j, N.STAT_CRITIC_LOSS]); al[a].append(st[0, j, N.STAT_ACTOR_LOSS])
        orc.learn_with(inp["idx"][k], c["gamma"], c["tau"])
    for j, a in enumerate(ids):
        np.testing.assert_allclose(cl[a], fx["loss_critic/" + a], rtol=LOSS_RTOL)
        np.testing.assert_allclose(al[a], fx["loss_actor/" + a], rtol=LOSS_RTOL, atol=1e-6)
        note("loss_relerr/maddpg_critic_" + a, rel_err(cl[a], fx["loss_critic/" + a], 1e-6))
        for kind, oa, oc, tag in ((N.PARAM_ONLINE, orc.actor, orc.critic, ""), (N.PARAM_TARGET, orc.actor_t, orc.critic_t, "_t")):
            assert_params_close(unflat_params(e.get_params(2 * j, kind), oa[a], AC_NAMES), oa[a], "maddpg/%s/actor%s" % (a, tag))
            assert_params_close(unflat_params(e.get_params(2 * j + 1, kind), oc[a], AC_NAMES), oc[a], "maddpg/%s/critic%s" % (a, tag))
    e.close()


def test_matd3_learn(N, ma_path):
    """MATD3_simple.learn = FRL_ALGO_MADDPG + twin_critic + policy noise on every agent's target action + do_actor."""
    from freerl_amd.engine import Engine
    from oracle import algos
    c = cases.CASES["matd3"]
    inp = cases.maddpg_inputs(c, twin=True)
    fx = gold("matd3")
    ids = inp["ids"]
    n = len(ids)
    od = [c["dims"][a][0] for a in ids]
    ad = [c["dims"][a][1] for a in ids]
    am = max(ad)
    e = Engine(N.ALGO_MADDPG, od, ad, c["capacity"], batch_max=c["batch"], twin_critic=True)
    assert e.learn_path(c["batch"])[0] == bool(ma_path)
    for j, a in enumerate(ids):
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            e.set_params(2 * j, flat_params(inp["params"][a]["actor"], AC_NAMES), kind)
            e.set_params(2 * j + 1, flat_params(inp["params"][a]["critic"], TWIN_NAMES), kind)
    e.add_batch(records([inp["tables"][a] for a in ids]))
    orc = algos.MATD3(inp["params"], c["dims"], c["actor_lr"], c["critic_lr"], c["capacity"])
    for i in range(c["n_table"]):
        orc.add({a: inp["tables"][a]["obs"][i] for a in ids}, {a: inp["tables"][a]["act"][i] for a in ids},
                {a: float(inp["tables"][a]["rew"][i]) for a in ids}, {a: inp["tables"][a]["next_obs"][i] for a in ids},
                {a: bool(inp["tables"][a]["done"][i]) for a in ids})
    cl = {a: [] for a in ids}
    al = {a: [] for a in ids}
    for k in range(c["n_learn"]):
        nz = np.zeros((1, n, max(2, n), c["batch"], am), np.float32)
        for i in range(n):
            for j in range(n):
                nz[0, i, j, :, :ad[j]] = inp["noise"][k][i][j]
        do_actor = ((k + 1) % c["policy_freq"] == 0)
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     use_policy_noise=True, policy_noise=c["policy_noise"], noise_clip=c["noise_clip"],
                     max_action=c["max_action"], policy_noise_scale=c["policy_noise_scale"], do_actor=do_actor,
                     idx=np.stack(inp["idx"][k])[None], noise=nz, want_stats=True)
        for j, a in enumerate(ids):
            cl[a].append(st[0, j, N.STAT_CRITIC_LOSS])
            if do_actor:
                al[a].append(st[0, j, N.STAT_ACTOR_LOSS])
        orc.learn_with(inp["idx"][k], inp["noise"][k], c["gamma"], c["tau"], c["policy_noise_scale"], c["policy_noise"],
                       c["noise_clip"], c["max_action"], c["policy_freq"])
    for j, a in enumerate(ids):
        np.testing.assert_allclose(cl[a], fx["loss_critic/" + a], rtol=LOSS_RTOL)
        np.testing.assert_allclose(al[a], fx["loss_actor/" + a], rtol=LOSS_RTOL, atol=1e-6)
        note("loss_relerr/matd3_critic_" + a, rel_err(cl[a], fx["loss_critic/" + a], 1e-6))
        for kind, oa, oc, tag in ((N.PARAM_ONLINE, orc.actor, orc.critic, ""), (N.PARAM_TARGET, orc.actor_t, orc.critic_t, "_t")):
            assert_params_close(unflat_params(e.get_params(2 * j, kind), oa[a], AC_NAMES), oa[a], "matd3/%s/actor%s" % (a, tag))
            # 8 Adam steps (4 calls x 2 heads' gradients) on elements whose gradient is ~0: Adam's update is ~lr whatever
            # the gradient's size, so a rounding-level gradient difference moves such an element by up to lr/100
            assert_params_close(unflat_params(e.get_params(2 * j + 1, kind), oc[a], TWIN_NAMES), oc[a], "matd3/%s/critic%s" % (a, tag),
                                atol=2e-5)
    e.close()


# --------------------------------------------------------------------------------------- PPO
@pytest.mark.parametrize("name", ["ppo", "ppo_tricks"])
def test_ppo_learn(N, name):
    from freerl_amd.engine import Engine
    from oracle import ppo as oppo
    c = cases.CASES[name]
    inp = cases.ppo_inputs(c)
    fx = gold(name)
    O, A, T = c["obs_dim"], c["act_dim"], c["horizon"]
    an = ["l1", "l2", "mean_layer"]
    e = Engine(N.ALGO_PPO, O, A, T, batch_max=c["minibatch"], extra_cols=A + 1,
               hidden_act=N.ACT_TANH if c["trick"]["tanh"] else N.ACT_RELU)
    e.set_params(0, flat_params(inp["params"]["actor"], an, "log_std"))
    e.set_params(1, flat_params(inp["params"]["critic"], AC_NAMES))
    tab = inp["table"]
    extra = np.concatenate([tab["logp"], tab["adv_done"].astype(np.float32).reshape(-1, 1)], axis=1)
    e.add_batch(records([tab], extra=extra))
    orc = oppo.PPO(inp["params"]["actor"], inp["params"]["critic"], O, A, c["actor_lr"], c["critic_lr"], T, c["trick"])
    for i in range(T):
        orc.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]),
                tab["logp"][i], bool(tab["adv_done"][i]))
    ev = e.act(0, N.ACT_TANHHEAD, tab["obs"][:16], out_dim=A)[0]
    np.testing.assert_allclose(ev, fx["evaluate_action"], rtol=1e-5, atol=1e-6)
    out = e.ppo_learn(T, c["minibatch"], c["k_epochs"], gamma=c["gamma"], lmbda=c["lmbda"], clip=c["clip"],
                      ent_coef=c["ent"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                      adam_eps=1e-5 if c["trick"]["adam_eps"] else 1e-8, adv_norm=c["trick"]["adv_norm"],
                      perms=np.stack(inp["perms"])[None], want_trace=True, want_adv=True)
    orc.learn_with(inp["perms"], c["minibatch"], c["gamma"], c["lmbda"], c["clip"], c["k_epochs"], c["ent"])
    # GAE: wave-scan (affine composition) vs the reference's sequential fp32 recurrence
    np.testing.assert_allclose(out["adv"][0], fx["adv_raw"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["v_target"][0], fx["v_target"], rtol=1e-4, atol=1e-5)
    note("gae_abs_err/" + name, float(np.max(np.abs(out["adv"][0] - fx["adv_raw"]))))
    np.testing.assert_allclose(out["trace"][0, :, 0], fx["loss_actor"], rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(out["trace"][0, :, 1], fx["loss_critic"], rtol=2e-4)
    note("loss_relerr/%s_critic" % name, rel_err(out["trace"][0, :, 1], fx["loss_critic"], 1e-6))
    ga = unflat_params(e.get_params(0), orc.actor, an, "log_std")
    gc = unflat_params(e.get_params(1), orc.critic, AC_NAMES)
    for k in orc.actor:
        np.testing.assert_allclose(ga[k], orc.actor[k], rtol=2e-3, atol=2e-5, err_msg=k)
    for k in orc.critic:
        np.testing.assert_allclose(gc[k], orc.critic[k], rtol=2e-3, atol=2e-5, err_msg=k)
    synth.check_digest("actor", ga, fx, 2e-3, 2e-5, "hip-vs-reference")
    assert e.opt_step(0) == int(fx["actor_step"]) and e.opt_step(1) == int(fx["critic_step"])
    assert e.cursor(0) == (0, 0)
    e.close()


def _beta_flat(p):
    """Actor_Beta state_dict -> the engine's 3-layer net whose head is [alpha_layer ; beta_layer]."""
    return np.concatenate([p["l1.weight"].ravel(), p["l1.bias"], p["l2.weight"].ravel(), p["l2.bias"],
                           p["alpha_layer.weight"].ravel(), p["beta_layer.weight"].ravel(), p["alpha_layer.bias"],
                           p["beta_layer.bias"]]).astype(np.float32)


def _beta_unflat(flat, like):
    out, o = {}, 0
    A = like["alpha_layer.bias"].size
    for k in ("l1.weight", "l1.bias", "l2.weight", "l2.bias"):
        out[k] = flat[o:o + like[k].size].reshape(like[k].shape); o += like[k].size
    for k in ("alpha_layer.weight", "beta_layer.weight", "alpha_layer.bias", "beta_layer.bias"):
        out[k] = flat[o:o + like[k].size].reshape(like[k].shape); o += like[k].size
    assert o == flat.size and A > 0
    return out


def test_ppo_beta_actor(N):
    """Actor_Beta (PPO_with_tricks.py:120-151,325-332): frl_config.actor_dist = 1."""
    from freerl_amd.engine import Engine
    from oracle import ppo as oppo
    c = cases.CASES["ppo_beta"]
    inp = cases.ppo_beta_inputs(c)
    fx = gold("ppo_beta")
    O, A, T = c["obs_dim"], c["act_dim"], c["horizon"]
    e = Engine(N.ALGO_PPO, O, A, T, batch_max=c["minibatch"], extra_cols=A + 1, actor_dist=1)
    e.set_params(0, _beta_flat(inp["params"]["actor"]))
    e.set_params(1, flat_params(inp["params"]["critic"], AC_NAMES))
    tab = inp["table"]
    extra = np.concatenate([tab["logp"], tab["adv_done"].astype(np.float32).reshape(-1, 1)], axis=1)
    e.add_batch(records([tab], extra=extra))
    orc = oppo.PPO(inp["params"]["actor"], inp["params"]["critic"], O, A, c["actor_lr"], c["critic_lr"], T, c["trick"], beta=True)
    for i in range(T):
        orc.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]),
                tab["logp"][i], bool(tab["adv_done"][i]))
    z = e.act(0, N.ACT_RAW, tab["obs"][:16], out_dim=2 * A)[0]                # head pre-activations [alpha | beta]
    sp = lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20)))) + 1
    np.testing.assert_allclose(sp(z[:, :A]), fx["alpha"], rtol=1e-5)
    np.testing.assert_allclose(sp(z[:, A:]), fx["beta"], rtol=1e-5)
    out = e.ppo_learn(T, c["minibatch"], c["k_epochs"], gamma=c["gamma"], lmbda=c["lmbda"], clip=c["clip"], ent_coef=c["ent"],
                      actor_lr=c["actor_lr"], critic_lr=c["critic_lr"], adv_norm=c["trick"]["adv_norm"],
                      perms=np.stack(inp["perms"])[None], want_trace=True, want_adv=True)
    orc.learn_with(inp["perms"], c["minibatch"], c["gamma"], c["lmbda"], c["clip"], c["k_epochs"], c["ent"])
    np.testing.assert_allclose(out["adv"][0], fx["adv_raw"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["trace"][0, :, 0], fx["loss_actor"], rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(out["trace"][0, :, 1], fx["loss_critic"], rtol=2e-4)
    ga = _beta_unflat(e.get_params(0), orc.actor)
    for k in orc.actor:
        np.testing.assert_allclose(ga[k], orc.actor[k], rtol=2e-3, atol=2e-5, err_msg=k)
    synth.check_digest("actor", ga, fx, 2e-3, 2e-5, "hip-vs-reference")
    e.close()


def test_ppo_py_cautious_adamw(N):
    """PPO_file/PPO.py: frl_ppo_learn with optimizer = 1 (c_adamw.py's cautious AdamW, lr = actor_lr for both nets)."""
    from freerl_amd.engine import Engine
    from oracle import ppo as oppo
    c = cases.CASES["ppo_py"]
    inp = cases.ppo_inputs(c)
    fx = gold("ppo_py")
    O, A, T = c["obs_dim"], c["act_dim"], c["horizon"]
    an = ["l1", "l2", "mean_layer"]
    e = Engine(N.ALGO_PPO, O, A, T, batch_max=c["minibatch"], extra_cols=A + 1)
    e.set_params(0, flat_params(inp["params"]["actor"], an, "log_std"))
    e.set_params(1, flat_params(inp["params"]["critic"], AC_NAMES))
    tab = inp["table"]
    extra = np.concatenate([tab["logp"], tab["adv_done"].astype(np.float32).reshape(-1, 1)], axis=1)
    e.add_batch(records([tab], extra=extra))
    orc = oppo.PPO(inp["params"]["actor"], inp["params"]["critic"], O, A, c["actor_lr"], c["critic_lr"], T, c["trick"],
                   optimizer="c_adamw")
    for i in range(T):
        orc.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]),
                tab["logp"][i], bool(tab["adv_done"][i]))
    out = e.ppo_learn(T, c["minibatch"], c["k_epochs"], gamma=c["gamma"], lmbda=c["lmbda"], clip=c["clip"], ent_coef=c["ent"],
                      actor_lr=c["actor_lr"], critic_lr=c["critic_lr"], adam_eps=1e-6, optimizer=1,
                      perms=np.stack(inp["perms"])[None], want_trace=True)
    orc.learn_with(inp["perms"], c["minibatch"], c["gamma"], c["lmbda"], c["clip"], c["k_epochs"], c["ent"])
    np.testing.assert_allclose(out["trace"][0, :, 0], fx["loss_actor"], rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(out["trace"][0, :, 1], fx["loss_critic"], rtol=5e-4)
    note("loss_relerr/ppo_py_critic", rel_err(out["trace"][0, :, 1], fx["loss_critic"], 1e-6))
    ga = unflat_params(e.get_params(0), orc.actor, an, "log_std")
    gc = unflat_params(e.get_params(1), orc.critic, AC_NAMES)
    # the cautious mask flips single elements on rounding-level differences of exp_avg*grad (each flip moves that element
    # by ~lr): per-element agreement for all but a handful, aggregate agreement through the digests
    for got, want in ((ga, orc.actor), (gc, orc.critic)):
        for k in want:
            bad = np.abs(got[k] - want[k]) > (2e-5 + 2e-3 * np.abs(want[k]))
            assert bad.mean() < 2e-3, (k, int(bad.sum()), bad.size)
    synth.check_digest("actor", ga, fx, 5e-3, 5e-4, "hip-vs-reference")
    synth.check_digest("critic", gc, fx, 5e-3, 5e-4, "hip-vs-reference")
    assert e.opt_step(0) == e.opt_step(1) == int(fx["opt_step"])
    e.close()


# --------------------------------------------------------------- population / device RNG / scale
def test_population_learners_are_independent(N):
    """P = 3 learners with identical inputs but different sample indices each equal their own oracle."""
    from oracle import algos
    c = dict(cases.CASES["td3"])
    inp = cases.ac_inputs(c, twin=True)
    P = 3
    e = _setup_ac(N, N.ALGO_TD3, c, inp, True, AC_NAMES, n_learners=P)
    orcs = []
    for p in range(P):
        o = algos.TD3(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"],
                      c["critic_lr"], c["capacity"])
        _fill_oracle(o, inp["table"])
        orcs.append(o)
    for k in range(2):
        idx = np.stack([synth.indices(7000 + 10 * k + p, c["n_table"], c["batch"]) for p in range(P)])[:, None]
        nz = np.zeros((P, 1, 2, c["batch"], c["act_dim"]), np.float32)
        for p in range(P):
            nz[p, 0, 0] = synth.normal(8000 + 10 * k + p, (c["batch"], c["act_dim"]))
        st = e.learn(c["batch"], gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k % 2 == 1),
                     use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=2.0, idx=idx, noise=nz,
                     want_stats=True)
        for p in range(P):
            cl, _ = orcs[p].learn_with(idx[p, 0], nz[p, 0, 0], 0.99, 0.005, 0.2, 0.5, 2.0, 2, 1.0)
            np.testing.assert_allclose(st[p, 0, N.STAT_CRITIC_LOSS], cl, rtol=LOSS_RTOL)
    for p in range(P):
        _check_ac_params(N, e, orcs[p], True, AC_NAMES, "pop%d" % p, learner=p)
    e.close()


def test_bench_sized_population_takes_the_chained_kernels_and_matches(N, monkeypatch):
    """Above 128 learners frl_learn selects the one-workgroup-per-learner chained kernels on its own (what bench.py runs):
    144 learners, no override; four of them (first, last, two in between — different workgroups / CUs) against their own
    oracles with their own indices and noise, actor step included."""
    from oracle import algos
    monkeypatch.delenv("FRL_CRITIC_V2", raising=False)
    c = dict(cases.CASES["td3"])
    inp = cases.ac_inputs(c, twin=True)
    P, watch = 144, (0, 1, 77, 143)
    e = _setup_ac(N, N.ALGO_TD3, c, inp, True, AC_NAMES, n_learners=P)
    fa, fc = flat_params(inp["params"]["actor"], AC_NAMES), flat_params(inp["params"]["critic"], TWIN_NAMES)
    recs = records([inp["table"]])
    for p in range(1, P):                      # _setup_ac fills learner 0: same parameters and table for the others
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            e.set_params(0, fa, kind, learner=p); e.set_params(1, fc, kind, learner=p)
        e.add_batch(recs, learners=np.full(len(recs), p, np.int32))
    orcs = {}
    for p in watch:
        o = algos.TD3(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"], c["critic_lr"], c["capacity"])
        _fill_oracle(o, inp["table"])
        orcs[p] = o
    B, A = c["batch"], c["act_dim"]
    for k in range(2):
        idx = np.stack([synth.indices(7100 + 10 * k + (p % 7), c["n_table"], B) for p in range(P)])[:, None]
        nz = np.zeros((P, 1, 2, B, A), np.float32)
        for p in range(P):
            nz[p, 0, 0] = synth.normal(8100 + 10 * k + (p % 5), (B, A))
        st = e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k % 2 == 1), use_policy_noise=True,
                     policy_noise=0.2, noise_clip=0.5, max_action=2.0, idx=idx, noise=nz, want_stats=True)
        for p in watch:
            cl, al = orcs[p].learn_with(idx[p, 0], nz[p, 0, 0], 0.99, 0.005, 0.2, 0.5, 2.0, 2, 1.0)
            np.testing.assert_allclose(st[p, 0, N.STAT_CRITIC_LOSS], cl, rtol=LOSS_RTOL)
            if k % 2 == 1:
                np.testing.assert_allclose(st[p, 0, N.STAT_ACTOR_LOSS], al, rtol=LOSS_RTOL, atol=1e-6)
    e.profile(True)
    for p in watch:
        _check_ac_params(N, e, orcs[p], True, AC_NAMES, "pop144_%d" % p, learner=p)
    # the chained path leaves no separate Adam launch for the critic: that is how the selection is observable
    e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=True, use_policy_noise=True, policy_noise=0.2,
            noise_clip=0.5, max_action=2.0)
    kern = e.profile_read()
    assert "grad_critic" in kern and "adam_critic" not in kern and "adam_actor" not in kern, kern
    e.close()


def test_chained_population_is_deterministic_and_independent_of_its_size(N, monkeypatch):
    """The bench's kernel family (one workgroup per learner, device-drawn indices and noise) on its own terms: the same seed gives
    bit-identical parameters twice, and learner p's parameters do not depend on how many other learners share the launch (144
    learners = one partial round of workgroups, 300 = two): its Philox key is (seed, p), its ring and nets are its own."""
    from freerl_amd.engine import Engine
    monkeypatch.delenv("FRL_CRITIC_V2", raising=False)
    watch = (0, 5, 77, 143)

    def run(P):
        e = Engine(N.ALGO_TD3, 8, 2, 4096, n_learners=P, twin_critic=True, batch_max=256, seed=77)
        assert e.learn_path(256)[0]
        for p in range(P):
            rng = np.random.default_rng(1000 + p)
            for net in (0, 1):
                flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
                e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
                e.set_params(net, flat, N.PARAM_TARGET, learner=p)
        e.fill_synthetic(4096, seed=3)
        for k in range(6):
            e.learn(256, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k % 2 == 1), use_policy_noise=True,
                    policy_noise=0.2, noise_clip=0.5, max_action=1.0)
        out = [np.concatenate([e.get_params(net, kind, learner=p) for net in (0, 1)
                               for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)]) for p in watch]
        e.close()
        return out
    a, b, c = run(144), run(144), run(300)
    for x, y, z in zip(a, b, c):
        assert np.all(np.isfinite(x))
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(x, z)
    assert not np.array_equal(a[0], a[1])


def test_full_size_device_rng_properties(N):
    """BASELINE config-2 scale (replay 1e6 rows, batch 256), device-drawn indices and noise.
    Size-independent properties: (1) bitwise determinism from the seed; (2) tau = 1 makes the
    target nets equal the online nets; (3) the ring content survives the update untouched."""
    from freerl_amd.engine import Engine
    cap = 1_000_000

    def run():
        e = Engine(N.ALGO_TD3, 8, 2, cap, twin_critic=True, batch_max=256, seed=1234)
        rng = np.random.default_rng(5)
        for net in (0, 1):
            flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE)
            e.set_params(net, flat, N.PARAM_TARGET)
        e.fill_synthetic(cap, seed=99)
        before = e.read_rows(0, 123456, 64)
        stats = []
        for k in range(4):
            st = e.learn(256, gamma=0.99, tau=1.0, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k % 2 == 1),
                         use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0, want_stats=True)
            stats.append(st.copy())
        after = e.read_rows(0, 123456, 64)
        out = dict(stats=np.stack(stats), a=e.get_params(0), c=e.get_params(1), at=e.get_params(0, N.PARAM_TARGET),
                   ct=e.get_params(1, N.PARAM_TARGET))
        np.testing.assert_array_equal(before, after)
        e.close()
        return out
    r1, r2 = run(), run()
    assert np.all(np.isfinite(r1["stats"])) and np.all(r1["stats"][:, 0, 0, N.STAT_CRITIC_LOSS] > 0)
    for k in ("stats", "a", "c"):
        np.testing.assert_array_equal(r1[k], r2[k])
    np.testing.assert_array_equal(r1["a"], r1["at"])
    np.testing.assert_array_equal(r1["c"], r1["ct"])


def test_full_size_chained_population_properties(N, monkeypatch):
    """The bench's configuration in full: replay 1e6 rows per learner, batch 256, 160 learners on the chained kernels (fragment-image
    parameters), device-drawn indices and noise.  Size-independent properties: tau = 1 makes every target net equal its online
    net after a policy step (through two different parameter layouts in HBM: frl_params_get translates both); a zero learning
    rate moves the Adam moments but not the parameters; the rings survive untouched; losses are finite and positive."""
    from freerl_amd.engine import Engine
    monkeypatch.delenv("FRL_CRITIC_V2", raising=False)
    cap, P, watch = 1_000_000, 160, (0, 81, 159)
    e = Engine(N.ALGO_TD3, 8, 2, cap, n_learners=P, twin_critic=True, batch_max=256, seed=4321)
    assert e.learn_path(256)[0]
    rng = np.random.default_rng(6)
    start = {}
    for p in range(P):
        for net in (0, 1):
            flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
            e.set_params(net, flat, N.PARAM_TARGET, learner=p)
            if p in watch:
                start[p, net] = flat
    e.fill_synthetic(cap, seed=98)
    before = [e.read_rows(p, 654321, 32) for p in watch]
    kw = dict(gamma=0.99, use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0, want_stats=True)
    st = e.learn(256, tau=0.005, actor_lr=0.0, critic_lr=0.0, do_actor=True, **kw)          # lr = 0: nothing but m, v and the targets move
    assert np.all(np.isfinite(st)) and np.all(st[:, 0, N.STAT_CRITIC_LOSS] > 0)
    for p in watch:
        for net in (0, 1):
            np.testing.assert_array_equal(e.get_params(net, learner=p), start[p, net])
            assert np.abs(e.get_params(net, N.PARAM_ADAM_M, learner=p)).max() > 0
    for k in range(3):
        st = e.learn(256, tau=1.0, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k == 2), **kw)
        assert np.all(np.isfinite(st))
    for p in watch:
        for net in (0, 1):
            th = e.get_params(net, learner=p)
            assert not np.array_equal(th, start[p, net])
            np.testing.assert_array_equal(th, e.get_params(net, N.PARAM_TARGET, learner=p))
    for p, b in zip(watch, before):
        np.testing.assert_array_equal(b, e.read_rows(p, 654321, 32))
    e.close()


def test_device_index_draw_is_a_uniform_subset_without_replacement(N):
    """The device-side stand-in for `np.random.choice(len(buffer), B, replace=False)` (DQN.py:97) at its hardest point,
    len(buffer) == 2 * B (every second draw collides and is redrawn): distinct rows, all in range, every row equally
    likely, a fresh subset per call, the same subsets from the same seed, different ones per learner."""
    from freerl_amd.engine import Engine
    B, size, P, calls = 256, 512, 3, 60

    def run(seed):
        e = Engine(N.ALGO_DQN, 4, 3, size, discrete=True, batch_max=B, n_learners=P, seed=seed)
        e.fill_synthetic(size, seed=1)
        got = []
        for _ in range(calls):
            e.learn(B, gamma=0.99, tau=0.01, critic_lr=0.0, clip_norm=0.0)
            got.append(e.last_indices(B)[:, 0])
        e.close()
        return np.stack(got)                     # [calls][P][B]
    a, b = run(7), run(7)
    np.testing.assert_array_equal(a, b)
    assert a.min() >= 0 and a.max() < size
    for k in range(calls):
        for p in range(P):
            assert len(np.unique(a[k, p])) == B
    assert not np.array_equal(a[0, 0], a[1, 0]) and not np.array_equal(a[0, 0], a[0, 1])
    counts = np.bincount(a[:, 0].reshape(-1), minlength=size)          # each row ~ Binomial(calls, 1/2)
    assert abs(counts.mean() - calls / 2) < 1e-9 and counts.min() > 8 and counts.max() < 52
    assert 2.8 < counts.std() < 5.0                                        # sqrt(60 / 4) = 3.87
    assert not np.array_equal(run(8)[0], a[0])


def test_rollout_collector_fills_the_ring_consistently(N):
    """frl_rollout on a population: collect-only first (transitions land in the right rings, next_obs
    of step t is obs of step t+1 unless the episode ended), then with learning (updates counted,
    losses finite, parameters move)."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout
    P, E = 3, 2
    e = Engine(N.ALGO_TD3, 3, 1, 400, twin_critic=True, batch_max=32, n_learners=P, seed=5)
    rng = np.random.default_rng(1)
    for p in range(P):
        for net in (0, 1):
            flat = (rng.standard_normal(e.num_params(net)) * 0.1).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
            e.set_params(net, flat, N.PARAM_TARGET, learner=p)
    pool = EnvPool("PendulumShort-v1", P * E, n_threads=2, seed=9)
    out = rollout(e, pool, 50, envs_per_learner=E, learn_every=0, explore_sigma=0.1, batch=32)
    assert out["env_steps"] == 50 * P * E and out["updates"] == 0 and out["episodes"] == P * E      # 40-step episodes
    lay = e.layout
    for p in range(P):
        assert e.cursor(p) == (100, 100)
        rows = e.read_rows(p, 0, 100)
        for env in range(E):
            tr = rows[env::E]                                  # this env's transitions in time order
            obs, nobs = tr[:, lay.obs_off[0]:lay.obs_off[0] + 3], tr[:, lay.next_obs_off[0]:lay.next_obs_off[0] + 3]
            same = np.all(np.abs(nobs[:-1] - obs[1:]) < 1e-6, axis=1)
            assert same.sum() == len(same) - 1                 # exactly one episode boundary (step 40) in 50 steps
            assert np.all(np.abs(np.linalg.norm(obs[:, :2], axis=1) - 1) < 1e-5)      # (cos, sin)
            assert np.all(tr[:, lay.done_off] == 0)            # Pendulum never terminates (truncation is not `done`)
            assert np.all(np.abs(tr[:, lay.act_off[0]]) <= 1.0)
    before = e.get_params(1, learner=1).copy()
    out = rollout(e, pool, 20, envs_per_learner=E, start_steps=64, learn_every=1, policy_freq=2, batch=32)
    assert out["updates"] == 20 * P
    st = e.stats()
    assert np.all(np.isfinite(st)) and np.all(st[:, 0, N.STAT_CRITIC_LOSS] > 0)
    assert not np.allclose(before, e.get_params(1, learner=1))
    assert e.opt_step(1, learner=2) == 20 and e.opt_step(0, learner=2) == 10
    pool.close(); e.close()


def test_ppo_rollout_collector_lays_out_env_segments(N):
    """frl_ppo_rollout: every env's steps form one contiguous time-ordered segment of the learner's ring, the stored
    log-probs are those of the stored actions under the collecting policy, the segment ends carry adv_done, and a
    cycle performs K_epochs x n_minibatch steps per learner."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, ppo_rollout
    P, E, Tseg, O, A = 2, 4, 16, 3, 1
    T = E * Tseg
    e = Engine(N.ALGO_PPO, O, A, T, batch_max=32, n_learners=P, extra_cols=A + 1, seed=3)
    rng = np.random.default_rng(2)
    for p in range(P):
        fa = (rng.standard_normal(e.num_params(0)) * 0.1).astype(np.float32)
        fa[-A:] = -0.3                                           # log_std
        e.set_params(0, fa, learner=p)
        e.set_params(1, (rng.standard_normal(e.num_params(1)) * 0.1).astype(np.float32), learner=p)
    pool = EnvPool("PendulumShort-v1", P * E, n_threads=2, seed=4)          # 40-step episodes, max_action 2
    out = ppo_rollout(e, pool, 1, envs_per_learner=E, steps_per_env=Tseg, minibatch=32, k_epochs=2, actor_lr=0.0, critic_lr=0.0)
    assert out["env_steps"] == P * T and out["updates"] == P * 2 * (T // 32)
    lay = e.layout
    for p in range(P):
        assert e.cursor(p) == (0, 0)                             # learn() cleared the buffer (PPO_with_tricks.py:354)
        rows = e.read_rows(p, 0, T)
        obs = rows[:, lay.obs_off[0]:lay.obs_off[0] + O]
        nobs = rows[:, lay.next_obs_off[0]:lay.next_obs_off[0] + O]
        act, logp, adv_done = rows[:, lay.act_off[0]], rows[:, lay.extra_off], rows[:, lay.extra_off + A]
        for env in range(E):
            seg = slice(env * Tseg, (env + 1) * Tseg)
            cont = np.all(np.abs(nobs[seg][:-1] - obs[seg][1:]) < 1e-6, axis=1)
            assert cont.all()                                    # 16 steps of a 40-step episode: no boundary inside
            assert adv_done[seg][-1] == 1 and np.all(adv_done[seg][:-1] == 0) and np.all(rows[seg, lay.done_off] == 0)
        # lr = 0: the parameters are still the collecting policy's -> recompute the Gaussian log-prob of the stored actions
        full = np.zeros((P, T, O), np.float32)
        full[p] = obs
        mean = e.act(0, N.ACT_TANHHEAD, full, out_dim=A)[p][:, 0]
        ls = -0.3
        want = -((act - mean) ** 2) / (2 * np.exp(2 * ls)) - ls - 0.9189385332
        np.testing.assert_allclose(logp, want, rtol=2e-4, atol=2e-5)
    before = e.get_params(0, learner=1).copy()
    out = ppo_rollout(e, pool, 3, envs_per_learner=E, steps_per_env=Tseg, minibatch=32, k_epochs=2, actor_lr=1e-3, critic_lr=1e-3)
    assert out["updates"] == 3 * P * 2 * 2 and out["env_steps"] == 3 * P * T and out["episodes"] >= P * E
    assert not np.allclose(before, e.get_params(0, learner=1))
    assert e.opt_step(0, learner=0) == 2 * 2 + 3 * 2 * 2 and np.all(np.isfinite(e.get_params(1, learner=0)))
    pool.close(); e.close()


def test_ppo_device_permutations_visit_every_row_once_per_epoch(N):
    """frl_ppo_learn without caller-supplied permutations (np.random.permutation per epoch, PPO_with_tricks.py:320) draws
    them on the device, horizon 200 (not a power of two): (1) one full-batch minibatch is order-independent, so device-drawn
    and host-supplied orders give the same parameters; (2) with 8-row minibatches the loss traces depend on the order: the
    same seed reproduces them, another learner / call / seed draws another order; (3) every row is visited exactly once
    per epoch: at lr = 0 the mean of the 25 minibatch losses equals the full-batch loss."""
    from freerl_amd.engine import Engine
    O, A, T = 5, 2, 200
    def make(seed):
        e = Engine(N.ALGO_PPO, O, A, T, batch_max=T, n_learners=2, extra_cols=A + 1, seed=seed)
        g = np.random.default_rng(11)
        a0 = (g.standard_normal(e.num_params(0)) * 0.1).astype(np.float32)
        c0 = (g.standard_normal(e.num_params(1)) * 0.1).astype(np.float32)
        for p in range(2):
            e.set_params(0, a0, learner=p); e.set_params(1, c0, learner=p)
        rec = g.standard_normal((T, e.width)).astype(np.float32) * 0.5
        lay = e.layout
        rec[:, lay.done_off] = 0; rec[:, lay.extra_off + A] = (g.random(T) < 0.05)
        rec[:, lay.extra_off:lay.extra_off + A] = -1.0
        return e, rec

    def fill(e, rec):
        for p in range(2):
            e.set_cursor(p, 0, 0)
        e.add_batch(np.concatenate([rec, rec]), learners=np.repeat(np.arange(2), T))
    kw = dict(gamma=0.99, lmbda=0.95, clip=0.2, ent_coef=0.01, actor_lr=1e-3, critic_lr=1e-3)
    # full-batch minibatch: order-independent up to float summation order inside the chunk loop (rows are summed per chunk)
    e, rec = make(3)
    fill(e, rec)
    e.ppo_learn(T, T, 1, **kw)
    dev = e.get_params(1, learner=0).copy()
    e.close()
    e, rec = make(3)
    fill(e, rec)
    e.ppo_learn(T, T, 1, perms=np.tile(np.arange(T), (2, 1, 1)).reshape(2, 1, T), **kw)
    np.testing.assert_allclose(dev, e.get_params(1, learner=0), rtol=2e-4, atol=2e-6)
    e.close()
    # minibatches of 8 rows: the traces depend on the order; same seed -> same, other learner / call / seed -> different
    def traces(seed):
        e, rec = make(seed)
        out = []
        for _ in range(2):
            fill(e, rec)
            out.append(e.ppo_learn(T, 8, 2, want_trace=True, **kw)["trace"].copy())
        e.close()
        return np.stack(out)                       # [call][learner][step][2]
    t1, t2, t3 = traces(3), traces(3), traces(4)
    assert np.all(np.isfinite(t1))
    np.testing.assert_array_equal(t1, t2)
    assert not np.array_equal(t1[0, 0], t1[0, 1]) and not np.array_equal(t1[0, 0, :25], t3[0, 0, :25])
    # every row exactly once per epoch: the critic's epoch-mean loss of an lr = 0 run equals the full-batch loss
    e, rec = make(5)
    fill(e, rec)
    full = e.ppo_learn(T, T, 1, want_trace=True, **dict(kw, actor_lr=0.0, critic_lr=0.0))["trace"][0, 0, 1]
    fill(e, rec)
    tr = e.ppo_learn(T, 8, 1, want_trace=True, **dict(kw, actor_lr=0.0, critic_lr=0.0))["trace"][0, :, 1]
    np.testing.assert_allclose(tr.mean(), full, rtol=1e-4)         # 25 minibatches of 8 = all 200 rows once
    e.close()


def test_ppo_rollout_collector_discrete_policy(N):
    """frl_ppo_rollout with a Categorical policy on CartPole: stored actions are valid action indices, stored log-probs are
    log-softmax values of the collecting policy (lr = 0 keeps it), episodes terminate inside the segments and carry
    done / adv_done accordingly."""
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, ppo_rollout
    P, E, Tseg, O, nA = 2, 4, 32, 4, 2
    T = E * Tseg
    e = Engine(N.ALGO_PPO, O, nA, T, batch_max=32, n_learners=P, discrete=True, extra_cols=2, seed=5)
    rng = np.random.default_rng(3)
    for p in range(P):
        e.set_params(0, (rng.standard_normal(e.num_params(0)) * 0.03).astype(np.float32), learner=p)      # near-uniform policy
        e.set_params(1, (rng.standard_normal(e.num_params(1)) * 0.1).astype(np.float32), learner=p)
    pool = EnvPool("CartPole-v1", P * E, n_threads=2, seed=6)
    out = ppo_rollout(e, pool, 1, envs_per_learner=E, steps_per_env=Tseg, minibatch=32, k_epochs=1, actor_lr=0.0, critic_lr=0.0)
    assert out["env_steps"] == P * T and out["episodes"] >= 1          # a random policy drops the pole within 32 steps somewhere
    lay = e.layout
    for p in range(P):
        rows = e.read_rows(p, 0, T)
        act, logp, adv_done, done = rows[:, lay.act_off[0]], rows[:, lay.extra_off], rows[:, lay.extra_off + 1], rows[:, lay.done_off]
        assert set(np.unique(act)) <= {0.0, 1.0}
        assert np.all(adv_done >= done) and np.all(adv_done.reshape(E, Tseg)[:, -1] == 1)
        assert done.sum() >= 1 or p > 0
        full = np.zeros((P, T, O), np.float32)
        full[p] = rows[:, lay.obs_off[0]:lay.obs_off[0] + O]
        logits = e.act(0, N.ACT_RAW, full, out_dim=nA)[p]
        lsm = logits - np.log(np.exp(logits - logits.max(1, keepdims=True)).sum(1, keepdims=True)) - logits.max(1, keepdims=True)
        np.testing.assert_allclose(logp, lsm[np.arange(T), act.astype(int)], rtol=2e-4, atol=2e-5)
        assert 0.1 < act.mean() < 0.9                                    # both actions get sampled
    pool.close(); e.close()


def test_ppo_discrete_learn(N):
    """Actor_discrete + Categorical (PPO_with_tricks.py:110-121, 249-251, 333-336)."""
    import torch
    from freerl_amd.engine import Engine
    from oracle import ppo as oppo
    c = cases.CASES["ppo_discrete"]
    inp = cases.ppo_discrete_inputs(c)
    fx = gold("ppo_discrete")
    O, nA, T = c["obs_dim"], c["n_actions"], c["horizon"]
    e = Engine(N.ALGO_PPO, O, nA, T, batch_max=c["minibatch"], extra_cols=2, discrete=True)
    e.set_params(0, flat_params(inp["params"]["actor"], AC_NAMES))
    e.set_params(1, flat_params(inp["params"]["critic"], AC_NAMES))
    tab = inp["table"]
    extra = np.concatenate([tab["logp"], tab["adv_done"].astype(np.float32).reshape(-1, 1)], axis=1)
    e.add_batch(records([tab], extra=extra))
    orc = oppo.PPO(inp["params"]["actor"], inp["params"]["critic"], O, nA, c["actor_lr"], c["critic_lr"], T, c["trick"],
                   discrete=True)
    for i in range(T):
        orc.add(tab["obs"][i], tab["act"][i][0], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]),
                tab["logp"][i], bool(tab["adv_done"][i]))
    ev = e.act(0, N.ACT_ARGMAX, tab["obs"][:16])[0, :, 0].astype(np.int64)
    np.testing.assert_array_equal(ev, fx["evaluate_action"])
    qs = []
    for i in range(12):
        torch.manual_seed(900 + i)
        qs.append(torch.empty(1, nA).exponential_(1).numpy()[0])
    a, lp = e.act(0, N.ACT_CAT_SAMPLE, tab["obs"][:12], eps=np.stack(qs), want_logp=True)
    np.testing.assert_array_equal(a[0, :, 0].astype(np.int64), fx["select_action"])
    np.testing.assert_allclose(lp[0, :, 0], fx["select_logp"], rtol=1e-5, atol=1e-6)
    out = e.ppo_learn(T, c["minibatch"], c["k_epochs"], gamma=c["gamma"], lmbda=c["lmbda"], clip=c["clip"],
                      ent_coef=c["ent"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"], adv_norm=True,
                      perms=np.stack(inp["perms"])[None], want_trace=True, want_adv=True)
    orc.learn_with(inp["perms"], c["minibatch"], c["gamma"], c["lmbda"], c["clip"], c["k_epochs"], c["ent"])
    np.testing.assert_allclose(out["adv"][0], fx["adv_raw"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["trace"][0, :, 0], fx["loss_actor"], rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(out["trace"][0, :, 1], fx["loss_critic"], rtol=2e-4)
    ga = unflat_params(e.get_params(0), orc.actor, AC_NAMES)
    for k in orc.actor:
        np.testing.assert_allclose(ga[k], orc.actor[k], rtol=2e-3, atol=2e-5, err_msg=k)
    synth.check_digest("actor", ga, fx, 2e-3, 2e-5, "hip-vs-reference")
    e.close()


def test_ppo_py_discrete_categorical_logits(N):
    """PPO_file/PPO.py's discrete policy: Categorical(logits=l3(...)) (PPO.py:78-90,176,257), frl_config.actor_dist = 2, with the
    cautious AdamW.  The case's logits spread past float eps, so the clamped probs= form gives other numbers (checked on the
    oracle side); here: engine vs reference golden and vs oracle, through the class (`PPO(..., trick=None)`)."""
    import torch
    from freerl_amd.PPO import PPO
    from oracle import ppo as oppo
    c = cases.CASES["ppo_py_discrete"]
    inp = cases.ppo_discrete_inputs(c)
    fx = gold("ppo_py_discrete")
    O, nA, T = c["obs_dim"], c["n_actions"], c["horizon"]
    pol = PPO([O, nA], False, c["actor_lr"], c["critic_lr"], T, "cuda", trick=None, minibatch_max=c["minibatch"])
    assert pol._e.cfg.actor_dist == 2
    pol.agent.actor.load_state_dict({k: torch.as_tensor(v) for k, v in inp["params"]["actor"].items()})
    pol.agent.critic.load_state_dict({k: torch.as_tensor(v) for k, v in inp["params"]["critic"].items()})
    tab = inp["table"]
    np.testing.assert_array_equal(np.array([pol.evaluate_action(tab["obs"][i]) for i in range(16)]), fx["evaluate_action"])
    sel = []
    for i in range(12):
        torch.manual_seed(900 + i)              # the class draws q = empty(1, nA).exponential_(1) like Categorical.sample()
        sel.append(pol.select_action(tab["obs"][i]))
    np.testing.assert_array_equal(np.array([int(a) for a, _ in sel]), fx["select_action"])
    np.testing.assert_allclose(np.array([float(lp) for _, lp in sel]), fx["select_logp"], rtol=1e-5, atol=1e-6)
    orc = oppo.PPO(inp["params"]["actor"], inp["params"]["critic"], O, nA, c["actor_lr"], c["critic_lr"], T, c["trick"],
                   discrete=True, optimizer="c_adamw", cat_logits=True)
    for i in range(T):
        args = (tab["obs"][i], tab["act"][i][0], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]), tab["logp"][i],
                bool(tab["adv_done"][i]))
        pol.add(*args); orc.add(*args)
    pol.track_loss = True
    perms = iter(inp["perms"])
    orig = np.random.permutation
    np.random.permutation = lambda n: next(perms)
    try:
        pol.learn(c["minibatch"], c["gamma"], c["lmbda"], c["clip"], c["k_epochs"], c["ent"])
    finally:
        np.random.permutation = orig
    orc.learn_with(inp["perms"], c["minibatch"], c["gamma"], c["lmbda"], c["clip"], c["k_epochs"], c["ent"])
    tr = pol.last_trace
    np.testing.assert_allclose(tr[0, :, 0], fx["loss_actor"], rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(tr[0, :, 1], fx["loss_critic"], rtol=5e-4)
    np.testing.assert_allclose(tr[0, :, 0], np.array(orc.actor_losses), rtol=5e-4, atol=5e-6)
    ga = {k: v.numpy() for k, v in pol.agent.actor.state_dict().items()}
    synth.check_digest("actor", ga, fx, 5e-3, 5e-4, "hip-vs-reference")


def test_ddpg_full_batch_obs_norm_and_weight_decay(N):
    """DDPG.py supplements on the engine: critic Adam weight_decay 1e-3 + device-side Batch_ObsNorm."""
    from oracle import algos
    c = cases.CASES["ddpg_full"]
    inp = cases.ac_inputs(c, twin=False)
    fx = gold("ddpg_full")
    e = _setup_ac(N, N.ALGO_DDPG, c, inp, False, AC_NAMES)
    e.obsnorm_enable(True)
    orc = algos.DDPG(inp["params"]["actor"], inp["params"]["critic"], c["obs_dim"], c["act_dim"], c["actor_lr"],
                     c["critic_lr"], c["capacity"], critic_weight_decay=1e-3, batch_obs_norm=True)
    _fill_oracle(orc, inp["table"])
    cl, al = [], []
    for k in range(c["n_learn"]):
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     critic_weight_decay=1e-3, idx=inp["idx"][k], want_stats=True)
        cl.append(st[0, 0, N.STAT_CRITIC_LOSS]); al.append(st[0, 0, N.STAT_ACTOR_LOSS])
        orc.learn_with(inp["idx"][k], None, c["gamma"], c["tau"])
    # normalised inputs are O(x/std), std ~ 0.03-0.08 (the reference's first update sets std = batch mean)
    np.testing.assert_allclose(cl, fx["loss_critic"], rtol=2e-4)
    np.testing.assert_allclose(al, fx["loss_actor"], rtol=5e-4, atol=2e-5)
    stats = e.obsnorm_stats()
    assert stats["n"] == c["n_learn"]
    np.testing.assert_allclose(stats["mean"], fx["bn_mean"][0], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(stats["std"], fx["bn_std"][0], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(stats["mean"], orc.bn.running_ms.mean[0], rtol=1e-5, atol=1e-7)
    sa = e.act(0, N.ACT_TANHHEAD, inp["table"]["obs"][:16], out_dim=c["act_dim"])[0]
    np.testing.assert_allclose(sa, fx["select_action"], rtol=5e-3, atol=5e-4)
    ga = unflat_params(e.get_params(0), orc.actor, AC_NAMES)
    gc = unflat_params(e.get_params(1), orc.critic, AC_NAMES)
    for k in orc.critic:
        np.testing.assert_allclose(gc[k], orc.critic[k], rtol=5e-3, atol=5e-5, err_msg=k)
    synth.check_digest("actor", ga, fx, 5e-3, 5e-5, "hip-vs-reference")
    e.close()


def test_sac_batch_obs_norm(N):
    from oracle import algos
    c = cases.CASES["sac_bn"]
    inp = cases.ac_inputs(c, twin=True, gaussian=True)
    fx = gold("sac_bn")
    an = ["l1", "l2", "mean_layer"]
    e = _setup_ac(N, N.ALGO_SAC, c, inp, True, an, "log_std")
    e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0)
    e.obsnorm_enable(True)
    cl, al = [], []
    for k in range(c["n_learn"]):
        nz = np.stack([inp["noise"][k][0], inp["noise"][k][1]])[None, None]
        st = e.learn(c["batch"], gamma=c["gamma"], tau=c["tau"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                     alpha_lr=1e-4, target_entropy=-float(c["act_dim"]), idx=inp["idx"][k], noise=nz, want_stats=True)
        cl.append(st[0, 0, N.STAT_CRITIC_LOSS]); al.append(st[0, 0, N.STAT_ACTOR_LOSS])
    np.testing.assert_allclose(cl, fx["loss_critic"], rtol=2e-4)
    np.testing.assert_allclose(al, fx["loss_actor"], rtol=5e-4, atol=2e-5)
    np.testing.assert_allclose(e.obsnorm_stats()["std"], fx["bn_std"][0], rtol=1e-4, atol=1e-7)
    # SAC.evaluate_action does not normalise (SAC.py:200-204), select_action does (:194-195)
    ev = e.act(0, N.ACT_TANHHEAD, inp["table"]["obs"][:8], out_dim=c["act_dim"], normalize=False)[0]
    np.testing.assert_allclose(ev, fx["evaluate_action"], rtol=5e-3, atol=5e-4)
    eps = np.stack([synth.normal(c["noise_seed"] + 900 + i, (1, c["act_dim"]))[0] for i in range(8)])
    sa = e.act(0, N.ACT_SAC_SAMPLE, inp["table"]["obs"][:8], eps=eps, out_dim=c["act_dim"])[0]
    np.testing.assert_allclose(sa, fx["select_action"], rtol=5e-3, atol=5e-4)
    e.close()


# ------------------------------------------------------------------ the population-sized row chunk
@pytest.mark.parametrize("rows", ["64", "16"])
def test_other_row_chunks_give_the_same_answers(N, rows, monkeypatch):
    """frl_create picks 32-row chunks for small populations and 64-row chunks (4x2 / 2x4 register blocks, two workgroups
    per CU) for the bench-sized ones; FRL_RC forces a size so the single-learner golden cases cover those kernels too."""
    monkeypatch.setenv("FRL_RC", rows)
    from freerl_amd.engine import Engine
    e = Engine(N.ALGO_TD3, 8, 2, 512, twin_critic=True, batch_max=256)
    assert e.lds_bytes()[1] == int(rows)
    e.close()
    monkeypatch.setenv("FRL_CRITIC_V2", "0")                 # these are the row-chunk kernels' knobs
    monkeypatch.setenv("FRL_DQN_FUSED", "0")
    test_dqn_learn_matches_oracle_and_reference(N, "rowchunk")
    test_td3_learn(N, "td3", "rowchunk")
    test_td3_learn(N, "td3_pendulum", "rowchunk")
    test_sac_learn(N, "rowchunk")
    test_maddpg_learn(N, False)
    test_matd3_learn(N, False)


@pytest.mark.parametrize("rows,cps", [("16", "2"), ("16", "4"), ("32", "2")])
def test_workgroups_walking_several_row_chunks_give_the_same_answers(N, rows, cps, monkeypatch):
    """Bench-sized populations give one gradient workgroup several consecutive row chunks (first chunk stores its slab, the
    others add to it; frl_create's schedule).  FRL_CPS forces that for the single-learner golden cases, Rainbow included."""
    monkeypatch.setenv("FRL_RC", rows)
    monkeypatch.setenv("FRL_CPS", cps)
    monkeypatch.setenv("FRL_CRITIC_V2", "0")
    monkeypatch.setenv("FRL_DQN_FUSED", "0")
    test_dqn_learn_matches_oracle_and_reference(N, "rowchunk")
    test_td3_learn(N, "td3", "rowchunk")
    test_sac_learn(N, "rowchunk")
    test_maddpg_learn(N, False)
    test_matd3_learn(N, False)
    test_dqn_rainbow_all_six_tricks(N)


# ------------------------------------------------------------------ PER / N-step / Double
def test_per_buffer_sumtree_on_device(N):
    """frl_per_*: new rows at the max priority, stratified descents with injected uniforms (bit-exact indices), IS
    weights, float32 priorities from TD errors, ring wrap — against the golden of the reference's PER_Buffer/SumTree."""
    from freerl_amd.engine import Engine
    c = cases.CASES["per_buffer"]
    inp = cases.per_buffer_inputs(c)
    fx = gold("per_buffer")
    e = Engine(N.ALGO_REPLAY_ONLY, c["obs_dim"], 1, c["capacity"], batch_max=c["batch"])
    e.per_enable(0.5, 0.4, 0.001, 0.01)
    tab = inp["table"]
    recs = records([tab])
    half = c["n_add"] // 2
    e.add_batch(recs[:half])
    np.testing.assert_allclose(e.per_state()["sum"], float(fx["sum_after_first_adds"]), rtol=1e-14)
    added = half
    for k in range(c["n_rounds"]):
        idx, w = e.per_sample(c["batch"], uniforms=inp["uniforms"][k])
        np.testing.assert_array_equal(idx[0], fx["idx/%d" % k])
        np.testing.assert_allclose(w[0], fx["is_weight/%d" % k], rtol=2e-6)
        e.per_update(c["batch"], idx=idx, td_error=inp["td"][k].reshape(1, -1))
        st = e.per_state()
        np.testing.assert_allclose(st["sum"], float(fx["sum/%d" % k]), rtol=1e-6)     # float32 pow: powf vs NumPy, 1 ulp
        np.testing.assert_allclose(st["max"], float(fx["max/%d" % k]), rtol=1e-6)
        stop = min(added + 60, c["n_add"])
        if stop > added:
            e.add_batch(recs[added:stop])
        added = stop
        np.testing.assert_allclose(e.per_state()["sum"], float(fx["sum_after_adds/%d" % k]), rtol=1e-6)
    assert abs(e.per_state()["beta"] - float(fx["beta"])) < 1e-12 and e.cursor(0)[1] == int(fx["size"])
    e.close()


def test_dqn_with_tricks_double_per_nstep(N, dqn_path):
    """DQN_with_tricks.learn (Double + PER + N_Step) through the class: n-step fold on add, PER sample -> fused update with
    the reference's weight broadcasting -> priority update; vs the golden from the imported reference."""
    from freerl_amd.DQN_with_tricks import DQN
    import torch
    c = cases.CASES["dqn_tricks"]
    inp = cases.dqn_tricks_inputs(c)
    fx = gold("dqn_tricks")
    trick = dict(Double=True, Dueling=False, PER=True, Noisy=False, N_Step=True, Categorical=False)
    pol = DQN([c["obs_dim"], c["n_actions"]], False, c["lr"], c["capacity"], "cuda", trick=trick, gamma=c["gamma"],
              batch_size=c["batch"], batch_max=c["batch"])
    sd = {k: torch.from_numpy(v.copy()) for k, v in inp["params"]["Qnet"].items()}
    pol.agent.Qnet.load_state_dict(sd)
    pol.agent.Qnet_target.load_state_dict(sd)
    tab = inp["table"]
    for i in range(c["n_table"]):
        pol.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    assert len(pol.buffer) == int(fx["size"])
    np.testing.assert_array_equal(pol.buffer.buffer.rewards[:len(pol.buffer)], fx["stored_rewards"])
    np.testing.assert_array_equal(pol.buffer.buffer.dones[:len(pol.buffer)], fx["stored_dones"])
    pol.track_loss = True
    losses = []
    us = iter([u for b in inp["uniforms"] for u in b])
    orig = np.random.random_sample
    np.random.random_sample = lambda *a: next(us)
    try:
        for k in range(c["n_learn"]):
            pol.learn(c["batch"], c["gamma"], c["tau"])
            losses.append(pol.last_loss)
            np.testing.assert_allclose(pol.buffer.sumtree.sum(), float(fx["tree_sum/%d" % k]), rtol=2e-4)
    finally:
        np.random.random_sample = orig
    np.testing.assert_allclose(losses, fx["loss"], rtol=5e-4)
    got = {k: v.numpy() for k, v in pol.agent.Qnet.state_dict().items()}
    synth.check_digest("Qnet", got, fx, 2e-3, 2e-5, "hip-vs-reference")


def test_dqn_dueling_double(N, dqn_path):
    """Dueling + Double through the class (DQN_with_tricks.py:60-79,263-265): head [V ; A] in the engine, Q = V + A - mean(A)
    in the act and update kernels, the reference's l1/V/A state_dict layout."""
    from freerl_amd.DQN_with_tricks import DQN
    import torch
    c = cases.CASES["dqn_dueling"]
    inp = cases.dqn_dueling_inputs(c)
    fx = gold("dqn_dueling")
    trick = dict(Double=True, Dueling=True, PER=False, Noisy=False, N_Step=False, Categorical=False)
    pol = DQN([c["obs_dim"], c["n_actions"]], False, c["lr"], c["capacity"], "cuda", trick=trick, gamma=c["gamma"],
              batch_size=c["batch"], batch_max=c["batch"])
    assert list(pol.agent.Qnet.state_dict().keys()) == ["l1.weight", "l1.bias", "V.weight", "V.bias", "A.weight", "A.bias"]
    sd = {k: torch.from_numpy(v.copy()) for k, v in inp["params"]["Qnet"].items()}
    pol.agent.Qnet.load_state_dict(sd)
    pol.agent.Qnet_target.load_state_dict(sd)
    tab = inp["table"]
    for i in range(c["n_table"]):
        pol.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    np.testing.assert_array_equal([pol.select_action(tab["obs"][i]) for i in range(32)], fx["select_action"])
    np.testing.assert_allclose(pol.agent.Qnet(torch.as_tensor(tab["obs"][:8])).numpy(), fx["q_values"], rtol=1e-5, atol=1e-6)
    pol.track_loss = True
    losses = []
    it = iter(inp["idx"])
    orig = np.random.choice
    np.random.choice = lambda *a, **k: next(it)
    try:
        for _ in range(c["n_learn"]):
            pol.learn(c["batch"], c["gamma"], c["tau"])
            losses.append(pol.last_loss)
    finally:
        np.random.choice = orig
    np.testing.assert_allclose(losses, fx["loss"], rtol=LOSS_RTOL)
    got = {k: v.numpy() for k, v in pol.agent.Qnet.state_dict().items()}
    synth.check_digest("Qnet", got, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    got_t = {k: v.numpy() for k, v in pol.agent.Qnet_target.state_dict().items()}
    synth.check_digest("Qnet_target", got_t, fx, P_RTOL, P_ATOL, "hip-vs-reference")


def test_dqn_noisy_dueling_double(N):
    """NoisyLinear heads (Noisy_net.py:17-76) under Dueling + Double through the class: the reference's state_dict keys,
    per-forward noise drawn from torch's generator in the reference's order, mu / sigma gradients."""
    from freerl_amd.DQN_with_tricks import DQN
    import torch
    c = cases.CASES["dqn_noisy"]
    inp = cases.dqn_noisy_inputs(c)
    fx = gold("dqn_noisy")
    trick = dict(Double=True, Dueling=True, PER=False, Noisy=True, N_Step=False, Categorical=False)
    pol = DQN([c["obs_dim"], c["n_actions"]], False, c["lr"], c["capacity"], "cuda", trick=trick, gamma=c["gamma"],
              batch_size=c["batch"], batch_max=c["batch"])
    assert list(pol.agent.Qnet.state_dict().keys()) == list(fx["state_dict_keys"])
    assert torch.initial_seed() == 100                                   # NoisyLinear.__init__ reseeds (Noisy_net.py:33)
    sd = pol.agent.Qnet.state_dict()
    for k, v in inp["params"]["Qnet"].items():
        sd[k] = torch.from_numpy(v.copy())
    pol.agent.Qnet.load_state_dict(sd)
    pol.agent.Qnet_target.load_state_dict(sd)
    tab = inp["table"]
    for i in range(c["n_table"]):
        pol.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    order = lambda one: [torch.from_numpy(t.copy()) for h in ("V", "A") for t in one[h]]
    seq = iter(order(inp["probe"]) + [t for per_call in inp["raw"] for one in per_call for t in order(one)])
    orig = torch.randn
    torch.randn = lambda *a, **k: next(seq)
    pol.track_loss = True
    losses = []
    it = iter(inp["idx"])
    orig_choice = np.random.choice
    np.random.choice = lambda *a, **k: next(it)
    try:
        # a noisy forward with the probe noise: Q = V + A - mean(A) from the raw head of effective set 0
        pol._e.noisy_resample(pol.agent.Qnet.draw())
        raw = pol._e.act(0, N.ACT_RAW, tab["obs"][:8], out_dim=1 + c["n_actions"], use_target=2)[0]
        q = raw[:, :1] + raw[:, 1:] - raw[:, 1:].mean(axis=1, keepdims=True)
        np.testing.assert_allclose(q, fx["q_probe"], rtol=1e-5, atol=1e-6)
        for _ in range(c["n_learn"]):
            pol.learn(c["batch"], c["gamma"], c["tau"])
            losses.append(pol.last_loss)
    finally:
        torch.randn = orig
        np.random.choice = orig_choice
    np.testing.assert_allclose(losses, fx["loss"], rtol=LOSS_RTOL)
    got = {k: v.numpy() for k, v in pol.agent.Qnet.state_dict().items() if "epsilon" not in k}
    synth.check_digest("Qnet", got, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    got_t = {k: v.numpy() for k, v in pol.agent.Qnet_target.state_dict().items() if "epsilon" not in k}
    synth.check_digest("Qnet_target", got_t, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    # the epsilon buffers are the last noise drawn (the online net's: Qnet(obs) of the last learn)
    last = cases.noisy_eps(inp["raw"][-1][2])
    np.testing.assert_allclose(pol.agent.Qnet.state_dict()["A.bias_epsilon"].numpy(), last["A"][1], rtol=1e-6)


def test_dqn_categorical(N):
    """Categorical DQN alone through the class (DQN_with_tricks.py:82-158,248-260)."""
    from freerl_amd.DQN_with_tricks import DQN
    import torch
    c = cases.CASES["dqn_c51"]
    inp = cases.dqn_c51_inputs(c)
    fx = gold("dqn_c51")
    trick = dict(Double=False, Dueling=False, PER=False, Noisy=False, N_Step=False, Categorical=True)
    pol = DQN([c["obs_dim"], c["n_actions"]], False, c["lr"], c["capacity"], "cuda", trick=trick, gamma=c["gamma"],
              batch_size=c["batch"], batch_max=c["batch"])
    sd = {k: torch.from_numpy(v.copy()) for k, v in inp["params"]["Qnet"].items()}
    pol.agent.Qnet.load_state_dict(sd)
    pol.agent.Qnet_target.load_state_dict(sd)
    tab = inp["table"]
    for i in range(c["n_table"]):
        pol.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    np.testing.assert_array_equal([pol.select_action(tab["obs"][i]) for i in range(32)], fx["select_action"])
    pol.track_loss = True
    losses = []
    it = iter(inp["idx"])
    orig = np.random.choice
    np.random.choice = lambda *a, **k: next(it)
    try:
        for _ in range(c["n_learn"]):
            pol.learn(c["batch"], c["gamma"], c["tau"])
            losses.append(pol.last_loss)
    finally:
        np.random.choice = orig
    np.testing.assert_allclose(losses, fx["loss"], rtol=LOSS_RTOL)
    got = {k: v.numpy() for k, v in pol.agent.Qnet.state_dict().items()}
    synth.check_digest("Qnet", got, fx, P_RTOL, P_ATOL, "hip-vs-reference")
    synth.check_digest("Qnet_target", {k: v.numpy() for k, v in pol.agent.Qnet_target.state_dict().items()}, fx, P_RTOL, P_ATOL)


def test_dqn_rainbow_all_six_tricks(N):
    """The reference's default configuration (DQN_with_tricks.py:416): Double + Dueling + PER + Noisy + N_Step + Categorical in one
    fused update; n-step fold on add, stratified PER draw, three noisy forwards, projected distribution, priorities from the
    cross-entropy errors."""
    from freerl_amd.DQN_with_tricks import DQN
    import torch
    c = cases.CASES["dqn_rainbow"]
    inp = cases.dqn_rainbow_inputs(c)
    fx = gold("dqn_rainbow")
    trick = dict(Double=True, Dueling=True, PER=True, Noisy=True, N_Step=True, Categorical=True)
    pol = DQN([c["obs_dim"], c["n_actions"]], False, c["lr"], c["capacity"], "cuda", trick=trick, gamma=c["gamma"],
              batch_size=c["batch"], batch_max=c["batch"])
    assert list(pol.agent.Qnet.state_dict().keys()) == list(fx["state_dict_keys"])
    sd = pol.agent.Qnet.state_dict()
    for k, v in inp["params"]["Qnet"].items():
        sd[k] = torch.from_numpy(v.copy())
    pol.agent.Qnet.load_state_dict(sd)
    pol.agent.Qnet_target.load_state_dict(sd)
    tab = inp["table"]
    for i in range(c["n_table"]):
        pol.add(tab["obs"][i], tab["act"][i], float(tab["rew"][i]), tab["next_obs"][i], bool(tab["done"][i]))
    order = lambda one: [torch.from_numpy(t.copy()) for h in ("V", "A") for t in one[h]]
    seq = iter(order(inp["probe"]) + [t for per_call in inp["raw"] for one in per_call for t in order(one)])
    us = iter([u for b in inp["uniforms"] for u in b])
    orig_randn, orig_rs = torch.randn, np.random.random_sample
    torch.randn = lambda *a, **k: next(seq)
    np.random.random_sample = lambda *a: next(us)
    pol.track_loss = True
    losses = []
    try:
        assert int(pol.select_action(tab["obs"][3])) == int(fx["select_action_probe"])
        for k in range(c["n_learn"]):
            pol.learn(c["batch"], c["gamma"], c["tau"])
            losses.append(pol.last_loss)
            np.testing.assert_allclose(pol.buffer.sumtree.sum(), float(fx["tree_sum/%d" % k]), rtol=2e-4)
    finally:
        torch.randn, np.random.random_sample = orig_randn, orig_rs
    np.testing.assert_allclose(losses, fx["loss"], rtol=5e-4)
    got = {k: v.numpy() for k, v in pol.agent.Qnet.state_dict().items() if "epsilon" not in k}
    synth.check_digest("Qnet", got, fx, 2e-3, 2e-5, "hip-vs-reference")
    got_t = {k: v.numpy() for k, v in pol.agent.Qnet_target.state_dict().items() if "epsilon" not in k}
    synth.check_digest("Qnet_target", got_t, fx, 2e-3, 2e-5, "hip-vs-reference")
    with pytest.raises(RuntimeError):
        DQN([4, 2], False, 1e-3, 64, "cuda", trick=dict(trick, Dueling=False), gamma=0.99)
