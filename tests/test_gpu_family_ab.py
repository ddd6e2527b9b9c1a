"""Chained kernel families against the row-chunk kernels on the SAME inputs (tests/family_ab.py; the advisor's round-4 finding: the
long-curve envelopes are loose past their tight window, so a bug that only shows after several updates — wrong dW1 tile masking on a
ragged chunk, say — could pass them).  20 learn() calls with injected indices and noise, two learners with their own parameters, then
every net's theta / target / Adam m / Adam v compared array by array; ragged batches, config 4 / config 5, hidden 256 included.

Also here: the two properties the K-sliced sweeps' unmasked row reads rest on (chain_wide.hpp: xfrag) — the padding of every weight block
stays EXACTLY zero through training, and a non-finite field in a row nobody samples changes nothing."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPORT = {}

# measured on MI355X (profiles/r05/family_ab_report.json): after 20 calls the families agree to 1e-7 .. 1e-5 of an array's largest element
# (TD3 / DDPG / MADDPG / MATD3, hidden 128 and 256) unless a ReLU unit within rounding of zero opened in one family only — then that
# sample's dQ/da differs, every element of the actor's m by ~1/B of a per-row term (SAC at 376 / 17: 7e-4 of max |m|) and theta by up to lr
# per call (2e-3 of the largest weight).  Tolerances, relative to the array's largest |x|:
TOL_MOMENT, TOL_THETA = 2e-3, 2e-2


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    assert _native.device_count() > 0
    return _native


@pytest.fixture(scope="module", autouse=True)
def _dump():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "family_ab_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("case", ["sac_c4", "td3_wide", "td3_b1000", "maddpg_c5", "matd3_het", "td3_h256", "sac_h256", "matd3_h256",
                                  "td3_narrow_b100", "sac_380_20_b17"])
def test_chained_vs_rowchunk_20_calls(N, case):
    from tests import family_ab as AB
    # (B = 17: a 17-sample gradient is mostly noise-sized elements, whose Adam steps differ between any two fp32 implementations —
    # measured after 20 calls: m 2.4e-2, theta 2.4e-2 of the arrays' largest; 5 calls keep the comparison on the kernels)
    calls = 5 if AB.CASES[case]["B"] < 64 else (20 if AB.CASES[case]["B"] <= 256 else 10)
    a, b = AB.run(case, 0, calls, 2), AB.run(case, 1, calls, 2)
    assert not a["family"] and b["family"], (a["family"], b["family"])
    d = AB.diff(a, b)
    st = d.pop("stats")
    REPORT[case] = dict(calls=calls, arrays={k: v[0] for k, v in d.items()}, arrays_q99={k: v[3] for k, v in d.items()},
                        loss_rel_first5=float(st[:5, :, :, :2].max()), loss_rel_all=float(st[:, :, :, :2].max()))
    # losses: rounding level while the trajectories coincide (5 calls), the drift envelope of DESIGN.md 2.1 afterwards
    assert st[:5, :, :, :2].max() <= 1e-4, (case, st[:5, :, :, :2].max())
    assert st[:, :, :, :2].max() <= 5e-3, (case, st[:, :, :, :2].max())
    for key, (w, at, mx, q99) in d.items():
        tol = TOL_THETA if key.startswith(("theta", "target", "act")) else TOL_MOMENT       # (act = the policy after `calls` updates)
        # 99 % of an array's elements within tol, the rest (a ReLU unit open in one family only moves one row's share of < 1 % of a
        # layer's gradient: at B = 17 that is 1/17 of it) within 5e-2
        assert q99 <= tol, "%s %s: 99th percentile of |diff| / max |x| = %.3e (max %.3e at flat index %d, |x| max %.3g)" % (case, key, q99, w, at, mx)
        assert w <= 5e-2, "%s %s: max |diff| / max |x| = %.3e at flat index %d (|x| max %.3g)" % (case, key, w, at, mx)


@pytest.mark.parametrize("case,P", [("td3_syn", 16), ("sac_narrow_b200", 16), ("ddpg_narrow_b37", 13), ("td3_narrow_b100", 5),
                                    ("td3_syn", 24), ("sac_narrow_b200", 32), ("ddpg_narrow_b37", 29), ("td3_narrow_b100", 17)])
def test_sixteen_workgroups_per_learner_vs_rowchunk_at_population_size(N, case, P):
    """kernels_solo.hip with every learner of a FULL population of its family (sixteen learners = all 256 CUs; per-learner slabs, flag
    words and mailboxes) against the row-chunk kernels on the same injected indices and noise, own parameters per learner, 20 calls,
    every array of every learner — the other solo tests run one learner, or compare two solo runs with each other.  Populations of
    17 .. 32 learners (round 6): eight workgroups per learner walking two 16-row tiles each, a slab per tile — ragged batches (200,
    100, 37 rows) leave some workgroups with one tile or none."""
    from tests import family_ab as AB
    calls = 20 if AB.CASES[case]["B"] >= 64 else 5
    a, b = AB.run(case, 0, calls, P), AB.run(case, None, calls, P)
    rows = 16 if P <= 16 else 32
    assert not a["family"] and b["path"] == (True, 117376, rows), (a["path"], b["path"])
    d = AB.diff(a, b)
    st = d.pop("stats")
    REPORT["solo/%s/P%d" % (case, P)] = dict(calls=calls, P=P, arrays={k: v[0] for k, v in d.items()}, arrays_q99={k: v[3] for k, v in d.items()},
                                  loss_rel_first5=float(st[:5, :, :, :2].max()), loss_rel_all=float(st[:, :, :, :2].max()))
    assert st[:5, :, :, :2].max() <= 1e-4, (case, st[:5, :, :, :2].max())
    assert st[:, :, :, :2].max() <= 5e-3, (case, st[:, :, :, :2].max())
    for key, (w, at, mx, q99) in d.items():
        tol = TOL_THETA if key.startswith(("theta", "target", "act")) else TOL_MOMENT
        assert q99 <= tol, "%s %s: 99th percentile of |diff| / max |x| = %.3e (max %.3e at flat index %d, |x| max %.3g)" % (case, key, q99, w, at, mx)
        assert w <= 5e-2, "%s %s: max |diff| / max |x| = %.3e at flat index %d (|x| max %.3g)" % (case, key, w, at, mx)


SOLOW_LDS = 4 * (64 * 256 + 2 * 8 * 256 + 128 + 128 + 32 + 32 + 4 * 8 * 256 + 26 * 256 + 3 * 256 + 256 + 4 * 2 * 256 + 16 * 32 + 16 * 48 + 128)


@pytest.mark.parametrize("case,P", [("sac_c4", 1), ("td3_wide", 3), ("ddpg_wide", 2), ("sac_100_7", 1), ("td3_201_12", 2), ("sac_380_20_b17", 1),
                                    ("sac_380_20_b256", 16), ("sac_c4", 5), ("td3_8_6", 2), ("ddpg_30_20", 1),
                                    ("maddpg_c5", 1), ("maddpg_het", 2), ("matd3_het", 5), ("matd3_c5", 1)])
def test_sixteen_workgroups_wide_first_layer_vs_rowchunk(N, case, P):
    """kernels_solow.hip (round 6: a handful of learners with a first layer of up to 416 columns and heads of up to 32 outputs on
    sixteen workgroups each, W1 streamed from the block) against the row-chunk kernels on the same injected indices and noise, own
    parameters per learner, every array of every learner: config 4's dims at one and five learners, 25 k-tiles with 20 actions at a
    FULL population of sixteen, one / seven / thirteen k-tiles, single and twin critics, ragged batches (200 rows; 17 rows = two
    tiles, fourteen workgroups without rows); MADDPG / MATD3 with a unit = (learner, agent): config 5's three agents at batch 1024 (64
    row tiles per unit), heterogeneous agents (own rows behind the joint rows in LDS, action columns off every boundary) at two and five learners."""
    from tests import family_ab as AB
    calls = 5 if AB.CASES[case]["B"] < 64 else (20 if AB.CASES[case]["B"] <= 256 else 10)
    a, b = AB.run(case, 0, calls, P), AB.run(case, None, calls, P)
    assert not a["family"] and b["path"] == (True, SOLOW_LDS, 16), (a["path"], b["path"])
    d = AB.diff(a, b)
    st = d.pop("stats")
    REPORT["solow/%s/P%d" % (case, P)] = dict(calls=calls, P=P, arrays={k: v[0] for k, v in d.items()}, arrays_q99={k: v[3] for k, v in d.items()},
                                   loss_rel_first5=float(st[:5, :, :, :2].max()), loss_rel_all=float(st[:, :, :, :2].max()))
    # critic losses by their own size; actor losses (-Q mean: they cross zero — maddpg_c5's agent 1 passes -0.0005 in its tenth call,
    # where 4e-6 of absolute difference reads as 0.9 relative) by the critic-loss scale, as tests/test_gpu_longrun.py reports them
    sa, sb = a["stats"], b["stats"]
    scale = float(np.abs(sa[..., 0]).mean())
    act_err = np.abs(sa[..., 1] - sb[..., 1]) / max(scale, 1e-6)
    REPORT["solow/%s/P%d" % (case, P)]["actor_abs_err_over_critic_scale"] = [float(act_err[:5].max()), float(act_err.max())]
    assert st[:5, :, :, 0].max() <= 1e-4 and act_err[:5].max() <= 1e-4, (case, st[:5, :, :, 0].max(), act_err[:5].max())
    assert st[:, :, :, 0].max() <= 5e-3 and act_err.max() <= 5e-3, (case, st[:, :, :, 0].max(), act_err.max())
    for key, (w, at, mx, q99) in d.items():
        tol = TOL_THETA if key.startswith(("theta", "target", "act")) else TOL_MOMENT
        assert q99 <= tol, "%s %s: 99th percentile of |diff| / max |x| = %.3e (max %.3e at flat index %d, |x| max %.3g)" % (case, key, q99, w, at, mx)
        assert w <= 5e-2, "%s %s: max |diff| / max |x| = %.3e at flat index %d (|x| max %.3g)" % (case, key, w, at, mx)


@pytest.mark.parametrize("case,P", [("sac_c4", 2), ("td3_wide", 1), ("sac_380_20_b17", 3)])
def test_wide_first_layer_device_draws_are_the_rowchunk_chain_s(N, case, P):
    """kernels_solow.hip draws the batch's rows inside its critic launch and regenerates the noise sets where they are used; the
    row-chunk chain has draw_kernel write both to EngineDesc::idx / noise.  Same seed, no injected rows: the rows of every call are
    identical and the updates agree as in the injected-rows test."""
    from tests import family_ab as AB
    calls = 6
    a, b = AB.run(case, 0, calls, P, device_rng=True), AB.run(case, None, calls, P, device_rng=True)
    assert not a["family"] and b["path"] == (True, SOLOW_LDS, 16), (a["path"], b["path"])
    assert np.array_equal(a["rows_drawn"], b["rows_drawn"])
    d = AB.diff(a, b)
    st = d.pop("stats")
    assert st[:5, :, :, :2].max() <= 1e-4, (case, st[:5, :, :, :2].max())
    for key, (w, at, mx, q99) in d.items():
        tol = TOL_THETA if key.startswith(("theta", "target", "act")) else TOL_MOMENT
        assert q99 <= tol and w <= 5e-2, "%s %s: |diff| / max |x|: 99th percentile %.3e, max %.3e at flat index %d" % (case, key, q99, w, at)


@pytest.mark.parametrize("case,P", [("sac_c4", 1), ("td3_wide", 4), ("sac_380_20_b17", 8), ("ddpg_wide", 5)])
def test_policy_step_in_one_launch_matches_two_launches(N, monkeypatch, case, P):
    """kernels_solow.hip with helper workgroups runs a policy step as ONE launch (solow_step_*): a workgroup with a row tile flags its
    critic slab and goes on to the policy's forward while the helpers alone step the critic; FRL_SOLOW_FUSE=0 keeps the two launches.
    Same arithmetic per element — only the gradient norm's partial sums are grouped by other workgroups — so 20 calls agree to
    rounding; ragged batches (200 rows, 17 rows: workgroups without rows walk through both halves at once), TD3's alternation of
    critic-only launches and fused steps, 1 .. 8 learners (48 / 32 / 16 helpers per learner)."""
    from tests import family_ab as AB
    calls = 5 if AB.CASES[case]["B"] < 64 else 20
    monkeypatch.setenv("FRL_SOLOW_FUSE", "0")
    a = AB.run(case, None, calls, P, device_rng=True)
    monkeypatch.delenv("FRL_SOLOW_FUSE")
    b = AB.run(case, None, calls, P, device_rng=True)
    assert a["path"] == b["path"] == (True, SOLOW_LDS, 16)
    assert np.array_equal(a["rows_drawn"], b["rows_drawn"])
    d = AB.diff(a, b)
    st = d.pop("stats")
    scale = float(np.abs(a["stats"][..., 0]).mean())
    act_err = np.abs(a["stats"][..., 1] - b["stats"][..., 1]) / max(scale, 1e-6)
    assert st[:, :, :, 0].max() <= 1e-4 and act_err.max() <= 1e-4, (case, st[:, :, :, 0].max(), act_err.max())
    for key, (w, at, mx, q99) in d.items():
        assert q99 <= 1e-4 and w <= 5e-2, "%s %s: |diff| / max |x|: 99th percentile %.3e, max %.3e at flat index %d" % (case, key, q99, w, at)


@pytest.mark.parametrize("case", ["sac_c4", "maddpg_c5", "td3_h256", "td3_narrow_b100"])
def test_padding_stays_zero_and_unsampled_nonfinite_rows_are_inert(N, monkeypatch, case):
    """(1) frl_params_pad_max == 0 for theta / target / m / v of every net after 12 updates on the chained families; (2) the same run with
    inf / NaN written into the reward, done and next_obs fields of ring rows that no index set touches is bit-identical."""
    from tests import family_ab as AB
    from freerl_amd.engine import Engine
    c = AB.CASES[case]
    monkeypatch.setenv("FRL_CRITIC_V2", "1")

    def run(poison):
        e = Engine(c["algo"], c["obs"], c["act"], 4096, n_learners=2, twin_critic=c["twin"], batch_max=c["B"], hidden=c.get("hidden", 128), seed=3)
        assert e.learn_path(c["B"])[0]
        g = np.random.default_rng(7)
        for net in range(e.n_nets):
            for p in range(2):
                flat = (g.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
                e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
                e.set_params(net, flat, N.PARAM_TARGET, learner=p)
        e.fill_synthetic(3000, seed=5)
        if poison:                                  # rows 2990 .. 2999 of both learners: never sampled below (indices < 2990)
            lay = e.layout
            for p in range(2):
                rows = e.read_rows(p, 2990, 10)
                rows[:, lay.rew_off:lay.rew_off + e.n_agents] = np.inf
                rows[:, lay.next_obs_off[0]:lay.next_obs_off[0] + 3] = np.nan
                rows[:, lay.obs_off[0]] = -np.inf
                e.set_cursor(p, 2990, 2990)
                e.add_batch(rows, learners=np.full(10, p, np.int32))
                e.flush()
                assert e.cursor(p) == (3000, 3000)
        na = e.n_agents
        am = max(c["act"]) if isinstance(c["act"], list) else c["act"]
        for k in range(12):
            idx = np.stack([[g.choice(2990, c["B"], replace=False) for _ in range(na)] for _ in range(2)]).astype(np.int64)
            noise = g.standard_normal((2, na, max(2, na), c["B"], am)).astype(np.float32)
            kw = {}
            if c["algo"] == N.ALGO_TD3 or c.get("matd3"):
                kw = dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0, do_actor=(k % 2 == 1))
            if c["algo"] == N.ALGO_SAC:
                kw = dict(alpha_lr=1e-3, target_entropy=-float(am))
            need = c["algo"] in (N.ALGO_TD3, N.ALGO_SAC) or c.get("matd3")
            st = e.learn(c["B"], gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3, idx=idx if na > 1 else idx[:, 0],
                         noise=noise if need else None, want_stats=True, **kw)
            assert np.all(np.isfinite(st)), (case, k)
        pads = {}
        out = []
        for net in range(e.n_nets):
            for kind, nm in ((N.PARAM_ONLINE, "theta"), (N.PARAM_TARGET, "target"), (N.PARAM_ADAM_M, "m"), (N.PARAM_ADAM_V, "v")):
                for p in range(2):
                    pads["%s%d/%d" % (nm, net, p)] = e.pad_max(net, kind, learner=p)
                    out.append(e.get_params(net, kind, learner=p))
        e.close()
        return pads, np.concatenate(out)
    pads, clean = run(False)
    bad = {k: v for k, v in pads.items() if v != 0.0}
    assert not bad, "padding slots moved away from zero: %r" % bad
    _, poisoned = run(True)
    np.testing.assert_array_equal(clean, poisoned)
