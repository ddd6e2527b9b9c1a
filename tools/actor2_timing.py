"""Developer instrument: shader-clock cycles per section of ac_actor_v2_kernel (kernels_actor2.hip), learner 0.
    python tools/actor2_timing.py [P]       (builds the `ppot` variant: -DFRL_PPO_TIMING, unity)
The critic kernel of the same learn() call stamps the same clock array first; the actor stage runs last and its dump is what is read."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRL_HIP_VARIANT", "ppot")
os.environ.setdefault("FRL_HIPCC_FLAGS", "-DFRL_PPO_TIMING")
os.environ.setdefault("FRL_CRITIC_V2", "1")
from freerl_amd import _native as N  # noqa: E402

N.build()
from freerl_amd.engine import Engine  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
algo = {"td3": N.ALGO_TD3, "sac": N.ALGO_SAC, "ddpg": N.ALGO_DDPG}[sys.argv[2] if len(sys.argv) > 2 else "td3"]
e = Engine(algo, 8, 2, 100_000, n_learners=P, twin_critic=algo != N.ALGO_DDPG, batch_max=256, seed=1)
rng = np.random.default_rng(0)
for net in range(2):
    for p in range(P):
        flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
e.fill_synthetic(100_000, seed=5)
kw = dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3)
if algo == N.ALGO_TD3: kw.update(do_actor=True, use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)
if algo == N.ALGO_SAC: kw.update(alpha_lr=1e-4, target_entropy=-2.0)
for k in range(6):
    e.learn(256, **kw)
fn = N.lib().frl_debug_ppo_clocks
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 16)()
assert fn(buf) == 0
clk = np.array(buf[:8], dtype=np.float64)
names = ["weight staging (actor, critic head(s), actor again)", "pass A: actor forward (two tiles per wave)",
         "pass B: critic forward + dX chain (two tiles per wave)", "pass C: actor forward again (4 chunks)",
         "pass C: delta + exchanges + dW + dH (4 chunks)", "row prefetch issue + norm / reductions", "clip + Adam + soft update"]
tot = clk[:7].sum()
print("P=%d: %.0f cycles per learner (actor stage of %s)" % (P, tot, sys.argv[2] if len(sys.argv) > 2 else "td3"))
for i, n in enumerate(names):
    print("   %-58s %8.0f  %5.1f%%" % (n, clk[i], 100 * clk[i] / tot))
e.close()
