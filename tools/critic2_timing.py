"""Developer instrument: shader-clock cycles per section of ac_critic_v2_*_kernel (kernels_critic2.hip), learner 0.
    python tools/critic2_timing.py [P]       (builds the `ppot` variant: -DFRL_PPO_TIMING, unity)"""
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
e = Engine(N.ALGO_TD3, 8, 2, 100_000, n_learners=P, twin_critic=True, batch_max=256, seed=1)
rng = np.random.default_rng(0)
for net in range(2):
    for p in range(P):
        flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
e.fill_synthetic(100_000, seed=5)
for k in range(7):      # (the last call is critic-only: the actor stage stamps the same clock array)
    e.learn(256, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k % 2 == 1), use_policy_noise=True, policy_noise=0.2,
            noise_clip=0.5, max_action=1.0)
fn = N.lib().frl_debug_ppo_clocks
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 16)()
assert fn(buf) == 0
clk = np.array(buf[:8], dtype=np.float64)
persist = os.environ.get("FRL_CRITIC_PERSIST", "0") != "0"
names = ["weight staging (5 nets)", "target actor pass", "target critic passes", "grad zero / bias reductions",
         "row prefetch issue (critic pass)", "forward (critic pass, 8 chunks)", "delta + exchanges + dW + dH (critic pass, 8 chunks)",
         "norm + clip + Adam + soft update in the open (persistent: the LAST learner's only) + row index loads"]
tot = clk[:8].sum()
per = max(1, -(-P // 256)) if persist else 1
print("P=%d: %.0f cycles per workgroup = %d learner(s) (critic stage, %s)" % (P, tot, per, "persistent kernels_critic3" if persist else "kernels_critic2"))
for i, n in enumerate(names):
    print("   %-50s %8.0f  %5.1f%%" % (n, clk[i], 100 * clk[i] / tot))
fine = np.array(buf[8:16], dtype=np.float64)
fnames = ["target actor: row loads + next net's fetch issued", "target actor: forward_vh<2> (2 chunks of 128 rows)", "target actor: action rule",
          "target critics: stage_commit (2 heads)", "target critics: row loads, a' from LDS, fetch issued", "target critics: forward_vh<2> (4 chunk passes)",
          "target critics: min / TD target"]
print("   inside the target passes:")
for i, n in enumerate(fnames):
    print("      %-60s %8.0f" % (n, fine[i]))
e.close()
