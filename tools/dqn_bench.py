"""DQN-family learn() throughput at BASELINE config C1's shape (LunarLander-v2: obs 8, 4 actions, batch 64 -> here 256 to
match the north-star shape, replay 1e5 filled): plain DQN, Double + PER, and the reference's default Rainbow set
(Double + Dueling + PER + Noisy + Categorical; the N_Step fold is host-side add() work and not part of learn()).
    python tools/dqn_bench.py [P ...]
and the whole DQN.py loop (select_action -> epsilon-greedy -> env.step -> add -> learn, one learn per env step; DQN.py:294-339)
through frl_rollout on the synthetic discrete env (obs 8, 4 actions), one env per learner and 8 envs per learner:
    python tools/dqn_bench.py rollout [P ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402

O, NA, B, CAP = 8, 4, 256, 100_000


def run(P, variant, steps=30):
    kw = dict(discrete=True, batch_max=B, n_learners=P, seed=1)
    if variant == "rainbow":
        kw.update(dueling=True, noisy=True, c51=(51, -100.0, 100.0))
    e = Engine(N.ALGO_DQN, O, NA, CAP, **kw)
    rng = np.random.default_rng(0)
    n = e.num_params(0) if hasattr(e, "num_params") else None
    for p in range(P):
        flat = (rng.standard_normal(e.get_params(0, learner=p).size) * 0.05).astype(np.float32)
        if variant == "rainbow":
            flat = np.abs(flat) * 0 + flat            # sigma slots may be any sign for timing purposes
        e.set_params(0, flat, N.PARAM_ONLINE, learner=p)
        e.set_params(0, flat, N.PARAM_TARGET, learner=p)
    per = variant in ("per", "rainbow")
    if per:
        e.per_enable(0.5, 0.4, 0.001, 0.01)
        rec = rng.standard_normal((4096, e.width)).astype(np.float32)
        lay = e.layout
        rec[:, lay.act_off[0]] = rng.integers(0, NA, 4096)
        rec[:, lay.done_off] = rng.random(4096) < 0.05
        for p in range(P):                                  # PER priorities are assigned by add(): fill through the add path
            for s in range(0, 20480, 4096):
                e.add_batch(rec, learners=np.full(4096, p, np.int32))
    else:
        e.fill_synthetic(CAP, seed=5)
    e.sync()

    def step():
        if per:
            e.per_sample(B, want_outputs=False)
            e.learn(B, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0, double_dqn=True, per=1)
            e.per_update(B)
        else:
            e.learn(B, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
    for _ in range(5):
        step()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    print("P=%4d  %-8s %.3f ms per learn() -> %.0f updates/s" % (P, variant, dt * 1e3, P / dt), flush=True)
    e.close()


def run_rollout(P, E, steps=300):
    from freerl_amd.envpool import EnvPool, rollout
    e = Engine(N.ALGO_DQN, O, NA, CAP, discrete=True, batch_max=B, n_learners=P, seed=1)
    rng = np.random.default_rng(0)
    for p in range(P):
        flat = (rng.standard_normal(e.get_params(0, learner=p).size) * 0.05).astype(np.float32)
        e.set_params(0, flat, N.PARAM_ONLINE, learner=p)
        e.set_params(0, flat, N.PARAM_TARGET, learner=p)
    e.fill_synthetic(CAP // 2, seed=5)                     # a run's steady state (ring half full): every vector step is followed by a learn()
    pool = EnvPool("SynLinearDiscrete-v0", P * E, n_threads=min(8, os.cpu_count() or 1), seed=2)
    kw = dict(envs_per_learner=E, start_steps=0, learn_every=1, epsilon=0.1, batch=B, gamma=0.99, tau=0.01, critic_lr=1e-3,
              clip_norm=0.0)
    rollout(e, pool, 5, **kw)
    out = rollout(e, pool, steps, **kw)
    print("P=%4d  %d env(s) per learner: %.3f ms per vector step -> %.0f env-steps/s, %.0f updates/s" %
          (P, E, out["seconds"] / steps * 1e3, out["env_steps"] / out["seconds"], out["updates"] / out["seconds"]), flush=True)
    pool.close(); e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rollout":
        for P in [int(x) for x in sys.argv[2:]] or [1, 512]:
            for E in (1, 8):
                run_rollout(P, E)
        sys.exit(0)
    for P in [int(x) for x in sys.argv[1:]] or [1, 512]:
        for v in ("dqn", "per", "rainbow"):
            run(P, v)
