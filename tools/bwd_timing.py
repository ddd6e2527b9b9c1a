"""Developer instrument: shader-clock cycles per section of the eight-wave backward (ChainNetT<8>::backward8), wave 0 of learner 0,
summed over the critic stage's chunks and heads.
    FRL_UNIT_FLAGS=-DFRL_BWD_TIMING bash tools/build_unit_timing.sh kernels_critic2; FRL_HIP_VARIANT=ppot python tools/bwd_timing.py [P]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRL_HIP_VARIANT", "ppot")
os.environ.setdefault("FRL_CRITIC_V2", "1")
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CALLS = 6
e = Engine(N.ALGO_TD3, 8, 2, 100_000, n_learners=P, twin_critic=True, batch_max=256, seed=1)
rng = np.random.default_rng(0)
for net in range(2):
    for p in range(P):
        flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
e.fill_synthetic(100_000, seed=5)
for k in range(CALLS):
    e.learn(256, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=False, use_policy_noise=True, policy_noise=0.2,
            noise_clip=0.5, max_action=1.0)
fn = N.lib().frl_debug_ppo_clocks
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 16)()
assert fn(buf) == 0
b = np.array(buf[8:16], dtype=np.float64) / CALLS
names = ["waiting at the 8 barriers", "exchange writes (three exchanges)", "dH2 (head delta, VALU)", "head contraction (32 MFMAs)",
         "layer-2 contraction (2 x 128 MFMAs)", "dH1 chain (256 MFMAs)", "first-layer contraction (32 MFMAs)"]
print("P=%d: backward of one learner (4 chunk-heads), %.0f cycles on wave 0; whole stage %.0f" % (P, b[:7].sum(), np.array(buf[:8]).sum()))
floor = [0, 0, 0, 4 * 32 * 32, 4 * 256 * 32, 4 * 256 * 32, 4 * 32 * 32]
for n, v, f in zip(names, b, floor):
    print("   %-44s %8.0f   (own MFMA issue %6d)" % (n, v, f))
e.close()
