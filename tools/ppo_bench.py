"""PPO.learn() timing at BASELINE config C3's shape (HalfCheetah-v4: obs 17, act 6, horizon 2048, minibatch 64,
K_epochs 10 -> 320 actor + 320 critic steps per learn): P learners, one persistent workgroup each.
    python tools/ppo_bench.py [P ...]"""
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
    print("P=%4d  PPO.learn (horizon %d, mb %d, K %d): %.1f ms  -> %.0f samples/s consumed, %.0f minibatch steps/s"
          % (P, T, MB, K, best * 1e3, P * T / best, P * K * (T // MB) / best), flush=True)
    e.close()


if __name__ == "__main__":
    for P in [int(x) for x in sys.argv[1:]] or [1, 64, 256]:
        run(P)
