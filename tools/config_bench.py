"""learn() throughput of the actor-critic family at the BASELINE / SURVEY §8(d) shapes other than bench.py's headline:

  TD3  SYN  obs 8  act 2   batch 256  hidden 128      (bench.py's workload, for reference)
  TD3  SYN  obs 8  act 2   batch 256  hidden 256      (north_star's wider contraction)
  TD3  C2   obs 3  act 1   batch 256                  (Pendulum-v1)
  DDPG SYN, SAC SYN
  SAC  C4   obs 376 act 17 batch 256                  (Humanoid-v4 dims)
  MADDPG C5 3 agents x (obs 18, act 5), batch 1024    (simple_spread_v3)

Device-drawn indices and noise, replay filled; weights random (timing only).  Replay capacity is 1e5 rows per learner
here, 2e4 for C4 (its 3 KB rows x 1e6 x hundreds of learners would not fit even in 288 GB); the index draw is O(batch) on the
device, so capacity does not enter the timing.
    python tools/config_bench.py [P ...]      (default P = 1 64 512)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402

CAP = 100_000
CASES = [
    # name, algo, obs, act, batch, hidden, learn kwargs
    ("TD3 SYN h128", N.ALGO_TD3, 8, 2, 256, 128, dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)),
    ("TD3 SYN h256", N.ALGO_TD3, 8, 2, 256, 256, dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)),
    ("TD3 C2", N.ALGO_TD3, 3, 1, 256, 128, dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=2.0)),
    ("DDPG SYN", N.ALGO_DDPG, 8, 2, 256, 128, {}),
    ("SAC SYN", N.ALGO_SAC, 8, 2, 256, 128, dict(alpha_lr=1e-4, target_entropy=-2.0)),
    ("SAC C4", N.ALGO_SAC, 376, 17, 256, 128, dict(alpha_lr=1e-4, target_entropy=-17.0)),
    ("MADDPG C5", N.ALGO_MADDPG, [18] * 3, [5] * 3, 1024, 128, {}),
]


def run(case, P, steps=20):
    name, algo, obs, act, B, H, kw = case
    twin = algo in (N.ALGO_TD3, N.ALGO_SAC)
    cap = 20_000 if name == "SAC C4" else CAP
    e = Engine(algo, obs, act, cap, n_learners=P, twin_critic=twin, batch_max=B, hidden=H, seed=1)
    rng = np.random.default_rng(0)
    for net in range(e.n_nets):
        n = e.get_params(net, learner=0).size
        for p in range(P):
            flat = (rng.standard_normal(n) * 0.05).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p)
            e.set_params(net, flat, N.PARAM_TARGET, learner=p)
    if algo == N.ALGO_SAC:
        for p in range(P):
            e.set_alpha_state([np.log(0.01), 0, 0, 0.01], learner=p)
    e.fill_synthetic(cap, seed=5)
    e.sync()
    chained, lds, rc = e.learn_path(B)

    def step(k):
        e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(algo != N.ALGO_TD3 or k % 2 == 1), **kw)
    for k in range(4):
        step(k)
    e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    w1, w0 = e.learn_work(B, True), e.learn_work(B, False)          # (flops, bytes) of the whole population's learn()
    fl = (0.5 * (w1[0] + w0[0]) if algo == N.ALGO_TD3 else w1[0])
    x1, x0 = e.learn_work_executed(B, True), e.learn_work_executed(B, False)
    fx = 0.5 * (x1 + x0) if algo == N.ALGO_TD3 else x1
    # two flop counts, named: EXECUTED (what autograd and the kernels compute: no first-layer dX of a trained net, action columns
    # only for dQ/da) is the one fractions of the 157.3 TFLOP/s peak are taken from; SURVEY 8(d)'s formula counts a dX per layer
    print("%-13s P=%4d  %-9s rows/workgroup %3d  LDS %3d KB  %8.3f ms per learn() -> %9.0f updates/s  executed %6.1f TFLOP/s = %.3f of peak  (8d formula %6.1f)" %
          (name, P, "chained" if chained else "row-chunk", rc, lds // 1024, dt * 1e3, P / dt, fx / dt / 1e12, fx / dt / 1e12 / 157.3, fl / dt / 1e12), flush=True)
    e.close()


if __name__ == "__main__":
    Ps = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 64, 512]
    only = [a for a in sys.argv[1:] if not a.isdigit()]          # e.g. "C4" "C5": substring filter on the case names
    for case in CASES:
        if only and not any(o in case[0] for o in only):
            continue
        for P in Ps:
            try:
                run(case, P)
            except Exception as ex:      # report and go on to the next shape
                print("%-13s P=%4d  FAILED: %s" % (case[0], P, ex), flush=True)
