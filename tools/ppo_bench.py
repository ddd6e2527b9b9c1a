"""PPO.learn() timing at BASELINE config C3's shape (HalfCheetah-v4: obs 17, act 6, horizon 2048, minibatch 64,
K_epochs 10 -> 320 actor + 320 critic steps per learn): P learners, one persistent workgroup each.
    python tools/ppo_bench.py [P ...]
and the whole collect + learn cycle (frl_ppo_rollout) with 64 vectorised envs per learner (config C3's "64 vec-envs") on the
synthetic linear-Gaussian env (obs 8, act 2): 64 envs x 32 steps = horizon 2048 per cycle.
    python tools/ppo_bench.py rollout [P ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402

O, A, T, MB, K = 17, 6, 2048, 64, 10


def run(P):
    e = Engine(N.ALGO_PPO, O, A, T, n_learners=P, batch_max=MB, extra_cols=A + 1, seed=1)
    rng = np.random.default_rng(0)
    dims_a = [(128, O), (128, 128), (A, 128)]
    dims_c = [(128, O), (128, 128), (1, 128)]
    def init(dims, extra=0):
        parts = []
        for o, i in dims:
            b = 1 / np.sqrt(i)
            parts += [rng.uniform(-b, b, o * i), rng.uniform(-b, b, o)]
        return np.concatenate(parts + [np.zeros(extra)]).astype(np.float32)
    for p in range(P):
        e.set_params(0, init(dims_a, A), learner=p)
        e.set_params(1, init(dims_c), learner=p)
    w = e.width
    rec = rng.standard_normal((P, T, w)).astype(np.float32) * 0.5
    lay = e.layout
    rec[:, :, lay.done_off] = (rng.random((P, T)) < 0.01)
    rec[:, :, lay.extra_off + A] = (rng.random((P, T)) < 0.01)
    kw = dict(gamma=0.99, lmbda=0.95, clip=0.2, ent_coef=0.01, actor_lr=3e-4, critic_lr=3e-4, adv_norm=True)
    times = []
    for it in range(4):
        for p in range(P):
            e.set_cursor(p, 0, 0)
        e.add_batch(rec.reshape(P * T, w), learners=np.repeat(np.arange(P), T))
        e.sync()
        t0 = time.perf_counter()
        e.ppo_learn(T, MB, K, **kw)
        e.sync()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    fl, by = e.ppo_work(T, K)
    print("P=%4d  PPO.learn (horizon %d, mb %d, K %d): %.1f ms  -> %.0f samples/s consumed, %.0f minibatch steps/s; "
          "%.2f GFLOP per learner = %.1f TFLOP/s (%.3f of the fp32 MFMA peak 157.3)"
          % (P, T, MB, K, best * 1e3, P * T / best, P * K * (T // MB) / best, fl / P / 1e9, fl / best / 1e12, fl / best / 1e12 / 157.3),
          flush=True)
    e.close()


def run_rollout(P, E=64, Tseg=32, iters=3):
    from freerl_amd.envpool import EnvPool, ppo_rollout
    o, a = 8, 2
    e = Engine(N.ALGO_PPO, o, a, E * Tseg, n_learners=P, batch_max=MB, extra_cols=a + 1, seed=1)
    rng = np.random.default_rng(0)
    for p in range(P):
        fa = (rng.standard_normal(e.num_params(0)) * 0.05).astype(np.float32)
        fa[-a:] = 0.0
        e.set_params(0, fa, learner=p)
        e.set_params(1, (rng.standard_normal(e.num_params(1)) * 0.05).astype(np.float32), learner=p)
    pool = EnvPool("SynLinear-v0", P * E, n_threads=min(8, max(1, os.cpu_count() or 1)), seed=3)
    kw = dict(envs_per_learner=E, steps_per_env=Tseg, minibatch=MB, k_epochs=K, adv_norm=True)
    ppo_rollout(e, pool, 1, **kw)
    out = ppo_rollout(e, pool, iters, **kw)
    print("P=%4d  %d envs/learner: %.1f ms per cycle (collect %d x %d + learn) -> %.0f env-steps/s, %.0f minibatch steps/s"
          % (P, E, out["seconds"] / iters * 1e3, E, Tseg, out["env_steps"] / out["seconds"], out["updates"] / out["seconds"]),
          flush=True)
    pool.close(); e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rollout":
        for P in [int(x) for x in sys.argv[2:]] or [1, 16, 64]:
            run_rollout(P)
    else:
        for P in [int(x) for x in sys.argv[1:]] or [1, 64, 256]:
            run(P)
