"""Latency of ONE learner's learn() (P = 1, SYN dims obs 8 / act 2, batch 256, hidden 128): TD3 / DDPG / SAC on the row-chunk
launch chain and, with FRL_CRITIC_V2=1, on the one-workgroup chained kernels.
    python tools/single_bench.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402

CASES = [("TD3", N.ALGO_TD3, dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)),
         ("DDPG", N.ALGO_DDPG, {}), ("SAC", N.ALGO_SAC, dict(alpha_lr=1e-4, target_entropy=-2.0))]


def run(name, algo, kw, steps, want_stats):
    twin = algo in (N.ALGO_TD3, N.ALGO_SAC)
    e = Engine(algo, 8, 2, 100_000, n_learners=1, twin_critic=twin, batch_max=256, hidden=128, seed=1)
    rng = np.random.default_rng(0)
    for net in range(e.n_nets):
        flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE); e.set_params(net, flat, N.PARAM_TARGET)
    if algo == N.ALGO_SAC:
        e.set_alpha_state([np.log(0.01), 0, 0, 0.01])
    e.fill_synthetic(50_000, seed=5)
    chained = e.learn_path(256)[0]

    def step(k):
        return e.learn(256, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(algo != N.ALGO_TD3 or k % 2 == 1),
                       want_stats=want_stats, **kw)
    for k in range(20):
        step(k)
    e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    print("%-5s %-9s %s  %7.1f us per learn() -> %7.0f updates/s" % (name, "chained" if chained else "row-chunk",
          "stats read back every call" if want_stats else "asynchronous            ", dt * 1e6, 1 / dt), flush=True)
    e.close()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for c in CASES:
        run(*c, steps, False)
        run(*c, steps, True)
