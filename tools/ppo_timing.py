"""Developer instrument: shader-clock cycles per section of a PPO minibatch step (ppo_update_kernel), learner 0.

    python tools/ppo_timing.py [P]        (builds the `ppot` variant: -DFRL_PPO_TIMING, unity)
Sections: 0 gather, 1 forward, 2 per-row surrogate / delta, 3 backward (dX + dW, gradients stored), 4 clip + Adam."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRL_HIP_VARIANT", "ppot")
os.environ.setdefault("FRL_HIPCC_FLAGS", "-DFRL_PPO_TIMING")
from freerl_amd import _native as N  # noqa: E402

N.build()
from freerl_amd.engine import Engine  # noqa: E402

O, A, T, MB, K = 17, 6, 2048, 64, 10
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = Engine(N.ALGO_PPO, O, A, T, n_learners=P, batch_max=MB, extra_cols=A + 1, seed=1)
rng = np.random.default_rng(0)
for p in range(P):
    fa = (rng.standard_normal(e.num_params(0)) * 0.05).astype(np.float32)
    fa[-A:] = 0
    e.set_params(0, fa, learner=p)
    e.set_params(1, (rng.standard_normal(e.num_params(1)) * 0.05).astype(np.float32), learner=p)
rec = rng.standard_normal((P, T, e.width)).astype(np.float32) * 0.5
lay = e.layout
rec[:, :, lay.done_off] = 0
rec[:, :, lay.extra_off + A] = (rng.random((P, T)) < 0.01)
kw = dict(gamma=0.99, lmbda=0.95, clip=0.2, ent_coef=0.01, actor_lr=3e-4, critic_lr=3e-4, adv_norm=True)
for it in range(2):
    for p in range(P):
        e.set_cursor(p, 0, 0)
    e.add_batch(rec.reshape(P * T, e.width), learners=np.repeat(np.arange(P), T))
    e.ppo_learn(T, MB, K, **kw)
fn = N.lib().frl_debug_ppo_clocks
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 16)()
assert fn(buf) == 0
clk = np.array(buf[:], dtype=np.float64).reshape(2, 8)
steps = K * (T // MB)
names = ["gather", "forward", "row math", "backward", "clip + Adam"]
if not os.environ.get("FRL_PPO_STREAMING"):        # the on-chip kernel (kernels_ppo2.hip) has its own sections
    names = ["gather + forward", "row math", "exchange 1 + dW3 + dH2", "exchange 2 + dW2 + dH1", "exchange 3 + dW1",
             "bias / norm / loss reductions", "clip + Adam"]
for w, who in enumerate(("actor", "critic")):
    tot = clk[w, :len(names)].sum()
    print("P=%d %s: %.0f cycles per minibatch step" % (P, who, tot / steps))
    for i, n in enumerate(names):
        print("   %-12s %8.0f  %5.1f%%" % (n, clk[w, i] / steps, 100 * clk[w, i] / tot))
e.close()
