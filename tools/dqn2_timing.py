"""Developer instrument: shader-clock cycles per section of dqn_fused_kernel (kernels_dqn2.hip), learner 0's first workgroup.
    python tools/dqn2_timing.py [P]       (builds the `ppot` variant: -DFRL_PPO_TIMING, unity; one workgroup per learner)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRL_HIP_VARIANT", "ppot")
os.environ.setdefault("FRL_HIPCC_FLAGS", "-DFRL_PPO_TIMING")
os.environ.setdefault("FRL_DQN_SPLIT", "1")
from freerl_amd import _native as N  # noqa: E402

N.build()
from freerl_amd.engine import Engine  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
e = Engine(N.ALGO_DQN, 8, 4, 100_000, discrete=True, n_learners=P, batch_max=256, seed=1)
rng = np.random.default_rng(0)
for p in range(P):
    flat = (rng.standard_normal(e.num_params(0)) * 0.05).astype(np.float32)
    e.set_params(0, flat, N.PARAM_ONLINE, learner=p); e.set_params(0, flat, N.PARAM_TARGET, learner=p)
e.fill_synthetic(100_000, seed=5)
if len(sys.argv) > 2 and sys.argv[2] == "rollout":          # the folded step of frl_rollout (add() + learn() + select_action)
    from freerl_amd.envpool import EnvPool, rollout
    pool = EnvPool("SynLinearDiscrete-v0", P, n_threads=1, seed=2)
    rollout(e, pool, 50, envs_per_learner=1, start_steps=0, learn_every=1, epsilon=0.1, batch=256, gamma=0.99, tau=0.01, critic_lr=1e-3)
    pool.close()
else:
    for k in range(6):
        e.learn(256, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
fn = N.lib().frl_debug_ppo_clocks
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 16)()
assert fn(buf) == 0
clk = np.array(buf[:8], dtype=np.float64)
names = ["index draw", "weight + block loads issued, add() stores, images stored", "row prefetch issue", "target + online forward, TD delta (4 chunks)",
         "exchanges + dW2 + dH1 + dW1 (4 chunks)", "bias / loss reductions", "norm + Adam + soft update", "select_action + hand-over"]
tot = clk[:8].sum()
print("P=%d: %.0f cycles per learner" % (P, tot))
for i, n in enumerate(names[:8]):
    print("   %-50s %8.0f  %5.1f%%" % (n, clk[i], 100 * clk[i] / tot))
e.close()
