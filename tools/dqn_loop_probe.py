import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from freerl_amd import _native as N
from freerl_amd.engine import Engine
from freerl_amd.envpool import EnvPool, rollout
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
e = Engine(N.ALGO_DQN, 8, 4, rows, discrete=True, batch_max=256, n_learners=1, seed=1)
g = np.random.default_rng(0)
flat = (g.standard_normal(e.num_params(0)) * 0.05).astype(np.float32)
e.set_params(0, flat, N.PARAM_ONLINE); e.set_params(0, flat, N.PARAM_TARGET)
e.fill_synthetic(rows, seed=5)
pool = EnvPool("SynLinearDiscrete-v0", 1, n_threads=1, seed=2)
kw = dict(envs_per_learner=1, start_steps=0, learn_every=1, epsilon=0.1, batch=256, gamma=0.99, tau=0.01, critic_lr=1e-3)
rollout(e, pool, 50, **kw)
r = rollout(e, pool, 2000, **kw)
print("env-steps/s %.0f  (%.2f us per step)" % (r["env_steps"] / r["seconds"], 1e6 * r["seconds"] / r["env_steps"]))
# learn() alone, asynchronous
for k in range(20): e.learn(256, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
e.sync(); t0 = time.perf_counter()
for k in range(2000): e.learn(256, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
e.sync(); print("learn() alone: %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
e.profile(True)
for k in range(200): e.learn(256, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
print({k: round(1e3 * v[0] / v[1], 2) for k, v in e.profile_read().items()})
