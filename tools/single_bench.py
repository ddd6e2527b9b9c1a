"""Latency of ONE learner's learn() (P = 1, SYN dims obs 8 / act 2, batch 256, hidden 128) — the reference's own use case, one learn()
per env step (TD3.py:403-450, DQN.py:294-343): TD3 / DDPG / SAC on the sixteen-workgroups-per-learner kernels (kernels_solo.hip: the
default for one learner; FRL_CRITIC_V2=0 the row-chunk launch chain, =1 the one-workgroup chained kernels), plain DQN on its one-launch
update, and the rollout-and-update LOOP of one learner stepping one env (and eight) for TD3 and DQN.
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
    e_path = e.learn_path(256)

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
    fam = "solo" if e_path == (True, 117376, 16) else ("chained" if chained else "row-chunk")
    print("%-5s %-9s %s  %7.1f us per learn() -> %7.0f updates/s" % (name, fam,
          "stats read back every call" if want_stats else "asynchronous            ", dt * 1e6, 1 / dt), flush=True)
    e.close()


def run_dqn(steps):
    e = Engine(N.ALGO_DQN, 8, 4, 100_000, discrete=True, batch_max=256, n_learners=1, seed=1)
    flat = (np.random.default_rng(0).standard_normal(e.num_params(0)) * 0.05).astype(np.float32)
    e.set_params(0, flat, N.PARAM_ONLINE); e.set_params(0, flat, N.PARAM_TARGET)
    e.fill_synthetic(100_000, seed=5)
    for k in range(20):
        e.learn(256, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
    e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        e.learn(256, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    print("DQN   fused     asynchronous              %7.1f us per learn() -> %7.0f updates/s" % (dt * 1e6, 1 / dt), flush=True)
    e.close()


def run_loop(algo, E, steps):
    """frl_rollout with one learner: select_action + exploration -> env.step (host pool) -> add -> learn, every vector step."""
    from freerl_amd.envpool import EnvPool, rollout
    dqn = algo == "dqn"
    e = Engine(N.ALGO_DQN if dqn else N.ALGO_TD3, 8, 4 if dqn else 2, 100_000, discrete=dqn, twin_critic=not dqn, batch_max=256, n_learners=1, seed=1)
    g = np.random.default_rng(0)
    for net in range(e.n_nets):
        flat = (g.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE); e.set_params(net, flat, N.PARAM_TARGET)
    e.fill_synthetic(100_000, seed=5)
    pool = EnvPool("SynLinearDiscrete-v0" if dqn else "SynLinear-v0", E, n_threads=1, seed=2)
    kw = dict(envs_per_learner=E, start_steps=0, learn_every=1, batch=256)
    if dqn:
        kw.update(epsilon=0.1, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
    rollout(e, pool, 50, **kw)
    r = rollout(e, pool, steps, **kw)
    print("%-5s loop, %d env(s): %7.1f us per vector step -> %8.0f env-steps/s, %7.0f updates/s" %
          (algo.upper(), E, 1e6 * r["seconds"] / steps, r["env_steps"] / r["seconds"], r["updates"] / r["seconds"]), flush=True)
    pool.close(); e.close()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for c in CASES:
        run(*c, steps, False)
        run(*c, steps, True)
    run_dqn(steps)
    for algo in ("td3", "dqn"):
        for E in (1, 8):
            run_loop(algo, E, steps)
