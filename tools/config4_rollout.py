"""BASELINE config 4's rollout leg on one GPU: SAC at Humanoid-v4's dims (obs 376, act 17) with 256 vectorised env instances per
learner (SAC_file/SAC.py:519-576: select_action -> env.step -> add -> learn per step), P learners = seeds.  The env is the
built-in SynBandWide-v0 (banded linear dynamics at those dims: what is measured is the engine's rollout path — batched device-side
select_action on 376-wide observations, the 3 KB records through the pinned block, the 393-column first layers of the update —
not MuJoCo).  One learn() per vector step (256 env steps), batch 256.
    python tools/config4_rollout.py [P ...]      (default 1 8 32)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N
from freerl_amd.engine import Engine
from freerl_amd.envpool import EnvPool, rollout

O, A, E, B, CAP = 376, 17, 256, 256, 100_000


def run(P, steps=60):
    e = Engine(N.ALGO_SAC, O, A, CAP, n_learners=P, twin_critic=True, batch_max=B, seed=1)
    g = np.random.default_rng(0)
    for p in range(P):
        for net in range(e.n_nets):
            flat = (g.standard_normal(e.num_params(net)) * 0.03).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
        e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0, learner=p)
    e.fill_synthetic(2 * B, seed=5)
    pool = EnvPool("SynBandWide-v0", P * E, n_threads=min(64, os.cpu_count() or 1), seed=2)
    kw = dict(envs_per_learner=E, start_steps=0, learn_every=1, batch=B, alpha_lr=1e-4, target_entropy=-float(A))
    rollout(e, pool, 5, **kw)
    out = rollout(e, pool, steps, **kw)
    print("SAC C4 dims  P=%3d learners x %d envs: %.3f ms per vector step -> %9.0f env-steps/s, %7.0f updates/s" %
          (P, E, out["seconds"] / steps * 1e3, out["env_steps"] / out["seconds"], out["updates"] / out["seconds"]), flush=True)
    pool.close(); e.close()


if __name__ == "__main__":
    for P in [int(x) for x in sys.argv[1:]] or [1, 8, 32]:
        run(P)
