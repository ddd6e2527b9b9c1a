"""env-steps/s of the rollout-and-update loop (frl_rollout), device-side exploration against round 1's host loop.

    python tools/rollout_bench.py [P ...]        (default 1 512)
TD3 on SynLinear-v0 (obs 8, act 2) and DQN on SynLinearDiscrete-v0 (obs 8, 4 actions), batch 256, one learn() per vector
step, E envs per learner.  "per" = DQN_with_tricks' Double + PER loop (frl_per_sample -> learn -> frl_per_update around every
step, new rows entering the sum-tree at the maximum priority), "rainbow" = its default trick set (Double + Dueling + PER + Noisy +
Categorical); both collect through the staged add path (the priorities enter there)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N
from freerl_amd.engine import Engine
from freerl_amd.envpool import EnvPool, rollout

B, CAP = 256, 100_000


def run(algo, P, E, host, steps=300):
    dqn = algo in ("dqn", "per", "rainbow")
    kw_e = dict(dueling=True, noisy=True, c51=(51, -100.0, 100.0)) if algo == "rainbow" else {}
    e = Engine(N.ALGO_DQN if dqn else N.ALGO_TD3, 8, 4 if dqn else 2, CAP, discrete=dqn, twin_critic=not dqn, batch_max=B,
               n_learners=P, seed=1, **kw_e)
    g = np.random.default_rng(0)
    for p in range(P):
        for net in range(e.n_nets):
            flat = (g.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
            e.set_params(net, flat, N.PARAM_TARGET, learner=p)
    per = algo in ("per", "rainbow")
    if per:                                                  # PER priorities are assigned by add(): fill through the add path
        e.per_enable(0.5, 0.4, 0.001, 0.01)
        rec = g.standard_normal((4096, e.width)).astype(np.float32)
        rec[:, e.layout.act_off[0]] = g.integers(0, 4, 4096)
        rec[:, e.layout.done_off] = g.random(4096) < 0.05
        for p in range(P):
            for _ in range(2):
                e.add_batch(rec, learners=np.full(4096, p, np.int32))
    else:
        e.fill_synthetic(CAP // 2, seed=5)
    pool = EnvPool("SynLinearDiscrete-v0" if dqn else "SynLinear-v0", P * E, n_threads=min(8, os.cpu_count() or 1), seed=2)
    kw = dict(envs_per_learner=E, start_steps=0, learn_every=1, batch=B, host_explore=host)
    if per:
        kw.update(per=1, double_dqn=True)
    if dqn:
        kw.update(clip_norm=0.0)
    rollout(e, pool, 5, **kw)
    out = rollout(e, pool, steps, **kw)
    print("%-4s P=%4d E=%2d %-6s %.3f ms per vector step -> %9.0f env-steps/s, %8.0f updates/s" %
          (algo, P, E, "host" if host else "device", out["seconds"] / steps * 1e3, out["env_steps"] / out["seconds"],
           out["updates"] / out["seconds"]), flush=True)
    pool.close(); e.close()


if __name__ == "__main__":
    for P in [int(x) for x in sys.argv[1:]] or [1, 512]:
        for algo in ("td3", "dqn"):
            for E in (1, 8):
                for host in (True, False):
                    run(algo, P, E, host)
        for algo in ("per", "rainbow"):
            for E in (1, 8):
                run(algo, P, E, False, steps=100 if algo == "rainbow" else 300)
