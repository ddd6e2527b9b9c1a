"""Developer instrument: shader-clock cycles per section of kernels_criticw.hip / kernels_actorw.hip (the K-sliced chained
family), workgroup 0.     python tools/wide_timing.py [sac_c4 | maddpg_c5] [P]     (builds the `widet` variant: -DFRL_WIDE_TIMING, unity)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRL_HIP_VARIANT", "widet")
os.environ.setdefault("FRL_HIPCC_FLAGS", "-DFRL_WIDE_TIMING")
os.environ.setdefault("FRL_CRITIC_V2", "1")
from freerl_amd import _native as N  # noqa: E402

N.build()
from freerl_amd.engine import Engine  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "sac_c4"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if case == "sac_c4":
    e = Engine(N.ALGO_SAC, 376, 17, 20_000, n_learners=P, twin_critic=True, batch_max=256, seed=1)
    B, kw = 256, dict(alpha_lr=1e-4, target_entropy=-17.0)
elif case == "td3_h256":
    e = Engine(N.ALGO_TD3, 8, 2, 20_000, n_learners=P, twin_critic=True, batch_max=256, hidden=256, seed=1)
    B, kw = 256, dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)
else:
    e = Engine(N.ALGO_MADDPG, [18] * 3, [5] * 3, 20_000, n_learners=P, batch_max=1024, seed=1)
    B, kw = 1024, {}
rng = np.random.default_rng(0)
for net in range(e.n_nets):
    for p in range(P):
        flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
if case == "sac_c4":
    for p in range(P):
        e.set_alpha_state([np.log(0.01), 0, 0, 0.01], learner=p)
e.fill_synthetic(20_000, seed=5)
for k in range(4):
    e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, **kw)
fn = N.lib().frl_debug_wide_clocks
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 32)()
assert fn(buf) == 0
clk = np.array(buf[:32], dtype=np.float64).reshape(2, 16)
names = [["weight staging (layers 2, 3 of every net)", "target actor(s): first-layer sweeps", "target actor(s): layers 2-3 + action rule",
          "target critic(s): first-layer sweeps", "target critic(s): layers 2-3 + TD target", "critic: first-layer sweeps",
          "critic: layers 2-3 forward", "critic: TD delta + backward (exchanges, dW2/3, dH, deltas -> scratch)", "dW1 pass (+ layer 2-3 gradient stores)",
          "dW1 stores, norm reduction", "clip + Adam (+ soft update) stream"],
         ["weight staging", "actor: first-layer sweep (+ h1 -> scratch)", "actor: layers 2-3 + action rule (+ h2 -> scratch)",
          "critic: first-layer sweeps on [s | a]", "action k-blocks of W1 -> LDS", "critic: layers 2-3 + dX chain -> dQ/da",
          "actor pass C: activations back + head", "actor pass C: deltas + backward", "dW1 pass (+ layer 2-3 gradient stores)",
          "dW1 stores, log_std / norm reductions", "clip + Adam + soft update stream"]]
if case == "td3_h256":
    names[0] = ["row copies, head staging", "target passes: first-layer sweeps (+ h1 -> scratch)", "target passes: second-layer sweeps (+ head partials)",
                "target passes: action rule / TD target", "critic: first-layer sweeps (+ h1 -> scratch)", "critic: second-layer sweeps (+ h2 -> scratch, head partials)",
                "critic: TD delta, layer-2 deltas -> scratch", "critic: transposed sweeps (d1 -> scratch)", "dW2 pass", "dW3, dW1, bias passes", "norm reduction",
                "clip + Adam (+ soft update) stream"]
    names[1] = ["sweep_x forward: first fetch -> barrier (all launches, all sweeps)", "   commit + barrier, 8 slices", "   fetch issue + 512 MFMAs, 8 slices", "   epilogue, 8 slices", "sweep_x transposed: first fetch -> barrier", "   commit + barrier", "   fetch issue + 512 MFMAs", "   epilogue", "l1_x: loads issued -> first barrier", "LDS staging -> second barrier", "wait for the row operands", "256 MFMAs + ReLU", "l1_x calls (all launches)"]
for row, title in ((0, "kernels_criticw"), (1, "kernels_actorw")):
    tot = clk[row].sum()
    print("%s %s P=%d: %.0f cycles per workgroup" % (case, title, P, tot))
    for i, n in enumerate(names[row]):
        print("   %-75s %9.0f  %5.1f%%" % (n, clk[row][i], 100 * clk[row][i] / max(tot, 1)))
e.close()
