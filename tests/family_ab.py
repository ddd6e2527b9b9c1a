"""Kernel family against kernel family on the same inputs: the row-chunk kernels (FRL_CRITIC_V2=0) and the chained ones (=1: narrow
register-chained, K-sliced hidden 128, x-stationary hidden 256) — stats, theta / target / Adam moments of every net, array by array.
Shared by tests/test_gpu_family_ab.py (the assertions) and tools/wide_ab.py (the developer CLI that prints every array)."""
import os

import numpy as np

from freerl_amd import _native as N
from freerl_amd.engine import Engine

CASES = {
    "sac_c4": dict(algo=N.ALGO_SAC, obs=376, act=17, B=256, twin=True),
    "td3_wide": dict(algo=N.ALGO_TD3, obs=17, act=6, B=200, twin=True),
    "ddpg_wide": dict(algo=N.ALGO_DDPG, obs=40, act=3, B=256, twin=False),
    "td3_b1000": dict(algo=N.ALGO_TD3, obs=30, act=5, B=1000, twin=True),
    "maddpg_c5": dict(algo=N.ALGO_MADDPG, obs=[18] * 3, act=[5] * 3, B=1024, twin=False),
    "maddpg_het": dict(algo=N.ALGO_MADDPG, obs=[6, 5, 7], act=[2, 3, 2], B=64, twin=False),
    "matd3_het": dict(algo=N.ALGO_MADDPG, obs=[6, 5, 7], act=[2, 3, 2], B=64, twin=True, matd3=True),
    "matd3_c5": dict(algo=N.ALGO_MADDPG, obs=[18] * 3, act=[5] * 3, B=1024, twin=True, matd3=True),
    "td3_h256": dict(algo=N.ALGO_TD3, obs=8, act=2, B=256, twin=True, hidden=256),
    "sac_h256": dict(algo=N.ALGO_SAC, obs=40, act=17, B=200, twin=True, hidden=256),
    "ddpg_h256": dict(algo=N.ALGO_DDPG, obs=11, act=3, B=96, twin=False, hidden=256),
    "td3_8_6": dict(algo=N.ALGO_TD3, obs=8, act=6, B=256, twin=True),                 # one first-layer k-tile, six actions: past the narrow kernels' four
    "ddpg_30_20": dict(algo=N.ALGO_DDPG, obs=30, act=20, B=128, twin=False),          # a single critic head with two actor head tiles
    "td3_syn": dict(algo=N.ALGO_TD3, obs=8, act=2, B=256, twin=True),                 # the bench shape
    "td3_narrow_b100": dict(algo=N.ALGO_TD3, obs=8, act=2, B=100, twin=True),        # the narrow register-chained kernels, ragged batches
    "sac_narrow_b200": dict(algo=N.ALGO_SAC, obs=11, act=3, B=200, twin=True),
    "ddpg_narrow_b37": dict(algo=N.ALGO_DDPG, obs=3, act=1, B=37, twin=False),
    "sac_380_20_b17": dict(algo=N.ALGO_SAC, obs=380, act=20, B=17, twin=True),
    "sac_380_20_b256": dict(algo=N.ALGO_SAC, obs=380, act=20, B=256, twin=True),
    "sac_380_20_h256": dict(algo=N.ALGO_SAC, obs=380, act=20, B=256, twin=True, hidden=256),
    "sac_100_7": dict(algo=N.ALGO_SAC, obs=100, act=7, B=200, twin=True),            # seven first-layer k-blocks: the cooperative dW1 pass, 7 k-tiles per wave
    "td3_201_12": dict(algo=N.ALGO_TD3, obs=201, act=12, B=256, twin=True),
    "maddpg_h256": dict(algo=N.ALGO_MADDPG, obs=[6, 5, 7], act=[2, 3, 2], B=64, twin=False, hidden=256),
    "matd3_h256": dict(algo=N.ALGO_MADDPG, obs=[18] * 3, act=[5] * 3, B=300, twin=True, matd3=True, hidden=256),
    "sac_h256_wide": dict(algo=N.ALGO_SAC, obs=120, act=20, B=256, twin=True, hidden=256),
}


def run(name, family, calls, P=2, device_rng=False):
    c = CASES[name]
    old = os.environ.get("FRL_CRITIC_V2")
    if family is None:                         # nothing forced: up to sixteen learners of the narrow shape take kernels_solo.hip
        os.environ.pop("FRL_CRITIC_V2", None)
    else:
        os.environ["FRL_CRITIC_V2"] = str(family)
    try:
        e = Engine(c["algo"], c["obs"], c["act"], 4096, n_learners=P, twin_critic=c["twin"], batch_max=c["B"],
                   hidden=c.get("hidden", 128), seed=3)
    finally:
        if old is None:
            os.environ.pop("FRL_CRITIC_V2", None)
        else:
            os.environ["FRL_CRITIC_V2"] = old
    g = np.random.default_rng(0)
    for net in range(e.n_nets):
        for p in range(P):
            flat = (g.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
            e.set_params(net, flat + np.float32(0.01) * g.standard_normal(flat.size).astype(np.float32), N.PARAM_TARGET, learner=p)
    if c["algo"] == N.ALGO_SAC:
        for p in range(P):
            e.set_alpha_state([np.log(0.2), 0, 0, 0.2], learner=p)
    e.fill_synthetic(3000, seed=5)
    na = e.n_agents
    am = max(c["act"]) if isinstance(c["act"], list) else c["act"]
    stats, rows_drawn = [], []
    for k in range(calls):
        idx = np.stack([[g.choice(3000, c["B"], replace=False) for _ in range(na)] for _ in range(P)]).astype(np.int64)
        noise = g.standard_normal((P, na, max(2, na), c["B"], am)).astype(np.float32)
        kw = {}
        if c["algo"] == N.ALGO_TD3:
            kw = dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0, do_actor=(k % 2 == 1))
        if c["algo"] == N.ALGO_SAC:
            kw = dict(alpha_lr=1e-3, target_entropy=-float(am))
        if c.get("matd3"):
            kw = dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0, do_actor=(k % 2 == 1))
        need_noise = c["algo"] in (N.ALGO_TD3, N.ALGO_SAC) or c.get("matd3")
        if device_rng:              # the engine draws rows and noise itself (Philox): the same bits whatever family runs the update
            st = e.learn(c["B"], gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3, want_stats=True, **kw)
            rows_drawn.append(e.last_indices(c["B"]).copy())
        else:
            st = e.learn(c["B"], gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3, idx=idx if na > 1 else idx[:, 0],
                         noise=noise if need_noise else None, want_stats=True, **kw)
        stats.append(st.copy())
    out = dict(stats=np.stack(stats), family=e.learn_path(c["B"])[0], path=e.learn_path(c["B"]))
    if device_rng:
        out["rows_drawn"] = np.stack(rows_drawn)
    for net in range(e.n_nets):
        for kind, nm in ((N.PARAM_ONLINE, "theta"), (N.PARAM_TARGET, "target"), (N.PARAM_ADAM_M, "m"), (N.PARAM_ADAM_V, "v")):
            out["%s%d" % (nm, net)] = np.stack([e.get_params(net, kind, learner=p) for p in range(P)])
    out["layers"] = [e.net_layers(net) if hasattr(e, "net_layers") else None for net in range(e.n_nets)]
    obs_dim = c["obs"][0] if isinstance(c["obs"], list) else c["obs"]
    ob = g.standard_normal((P, 7, obs_dim)).astype(np.float32)
    out["act"] = e.act(0, N.ACT_TANHHEAD, ob, out_dim=(c["act"][0] if isinstance(c["act"], list) else c["act"]))
    e.close()
    return out



def diff(a, b):
    """{array name: (max |x - y| / max |x|, where, max |x|, 99th percentile of |x - y| / max |x|)} over theta / target / m / v of every net and the actions, plus the per-call relative difference
    of the stats.  Against the array's largest element: an element whose gradient is rounding noise (1e-3 of the largest and below, a sum
    of 256-1024 cancelling terms in two different orders) gets a different Adam step m / sqrt(v) in each family — up to lr per call in
    theta — and a ReLU unit within an ulp of zero may open in one family and not in the other; element-relative errors of such entries
    are large in BOTH families against the oracle and say nothing (the oracle tests hold each family to it)."""
    out = {}
    for key in sorted(a):
        if key in ("stats", "family", "layers", "path", "rows_drawn"):
            continue
        x, y = a[key], b[key]
        rel = np.abs(x - y) / (np.abs(x).max() + 1e-30)
        out[key] = (float(rel.max()), int(rel.argmax()), float(np.abs(x).max()), float(np.quantile(rel, 0.99)))
    sa, sb = a["stats"], b["stats"]
    out["stats"] = np.abs(sa - sb) / np.maximum(np.abs(sa), 1e-6)         # [calls][P][agents][FRL_STAT_COUNT]
    return out
