"""TD3 learn() throughput of SMALL populations (SYN dims obs 8 / act 2, batch 256, hidden 128): which kernel family serves P learners best.
    python tools/small_pop_bench.py 1 2 4 8 12 16 24 32 64 128          (FRL_SOLO=0/1, FRL_CRITIC_V2=0/1 force a family)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402


def run(P, steps=400):
    e = Engine(N.ALGO_TD3, 8, 2, 100_000, n_learners=P, twin_critic=True, batch_max=256, hidden=128, seed=1)
    rng = np.random.default_rng(0)
    for net in range(e.n_nets):
        for p in range(P):
            flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
    e.fill_synthetic(50_000, seed=5)
    kw = dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, clip_norm=0.5, use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)
    for k in range(20):
        e.learn(256, do_actor=(k % 2 == 1), **kw)
    e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        e.learn(256, do_actor=(k % 2 == 1), **kw)
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    path = e.learn_path(256)
    fam = "solo" if path[2] == 16 else ("solo x8" if path[0] and path[2] == 32 else ("chained" if path[0] else "row-chunk"))
    print("P=%4d  %-9s %8.1f us per learn() -> %9.0f updates/s" % (P, fam, dt * 1e6, P / dt), flush=True)
    e.close()


if __name__ == "__main__":
    for a in sys.argv[1:] or ["1", "8", "16"]:
        run(int(a))
