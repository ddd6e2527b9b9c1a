"""Stability soak: long rollout-and-update runs at the bench population, checked for finite losses / parameters at the end.
    python tools/soak.py [vector steps]          (default 10000 TD3 + 20000 DQN steps at 512 learners, 30 PPO cycles at 64)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402
from freerl_amd.envpool import EnvPool, rollout  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
P = 512
for algo, dqn in (("td3", False), ("dqn", True)):
    e = Engine(N.ALGO_DQN if dqn else N.ALGO_TD3, 8, 4 if dqn else 2, 100_000, discrete=dqn, twin_critic=not dqn, batch_max=256,
               n_learners=P, seed=1)
    e.fill_synthetic(1000, seed=5)
    pool = EnvPool("SynLinearDiscrete-v0" if dqn else "SynLinear-v0", P, n_threads=8, seed=2)
    kw = dict(envs_per_learner=1, start_steps=0, learn_every=1, batch=256)
    if dqn:
        kw.update(clip_norm=0.0)
    n = steps * (2 if dqn else 1)
    t0 = time.perf_counter()
    out = rollout(e, pool, n, **kw)
    dt = time.perf_counter() - t0
    st = e.stats()
    par = np.concatenate([e.get_params(net, learner=p) for net in range(e.n_nets) for p in (0, P // 2, P - 1)])
    ok = np.all(np.isfinite(st)) and np.all(np.isfinite(par)) and np.abs(par).max() < 1e3
    print("%s: %d vector steps x %d learners in %.1f s (%.0f env-steps/s), %d updates; stats finite %s, |theta| max %.2f, critic loss mean %.4g -> %s"
          % (algo, n, P, dt, out["env_steps"] / dt, out["updates"], bool(np.all(np.isfinite(st))), float(np.abs(par).max()),
             float(st[:, 0, N.STAT_CRITIC_LOSS].mean()), "OK" if ok else "FAILED"), flush=True)
    assert ok
    pool.close(); e.close()
