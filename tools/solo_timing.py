"""Developer instrument: where one learner's update goes on the sixteen-workgroup kernels (kernels_solo.hip): wall-clock stamps
(100 MHz) of thread 0 of every workgroup, in a library whose kernels_solo.hip was compiled with -DFRL_SOLO_TIMING
(tools/build_solo_timing.sh -> tools/_bin/libfreerl_hip_solot.so), plus the per-kernel HIP-event times of the product library's chain.
    FRL_HIP_VARIANT=solot python tools/solo_timing.py [td3|ddpg|sac]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "td3"
algo = dict(td3=N.ALGO_TD3, ddpg=N.ALGO_DDPG, sac=N.ALGO_SAC)[which]
kw = dict(td3=dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0), ddpg={}, sac=dict(alpha_lr=1e-4, target_entropy=-2.0))[which]
e = Engine(algo, 8, 2, 100_000, n_learners=1, twin_critic=algo != N.ALGO_DDPG, batch_max=256, seed=1)
assert e.learn_path(256) == (True, 117376, 16), e.learn_path(256)
rng = np.random.default_rng(0)
for net in range(2):
    flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
    e.set_params(net, flat, N.PARAM_ONLINE); e.set_params(net, flat, N.PARAM_TARGET)
if algo == N.ALGO_SAC:
    e.set_alpha_state([np.log(0.01), 0, 0, 0.01])
e.fill_synthetic(50_000, seed=5)


def stamps(do_actor):
    for k in range(6):
        e.learn(256, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=do_actor, **kw)
    buf = np.zeros((16, 32), np.float32)
    N.check(N.lib().frl_solo_debug_read(e._h, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size))
    if buf[:, 16:18].any():
        print("   (inside the first section, workgroup 0: the index draw starts at %.2f us, indices drawn at %.2f us, row fields + noise issued at %.2f us)"
              % (buf[0, 18] * 0.01, buf[0, 16] * 0.01, buf[0, 17] * 0.01))
    return buf[:, 8:16] * 0.01          # us


names_c = ["first image staged (+ row fields)", "target actor forward", "target critic heads", "critic heads fwd + bwd -> slab", "grid barrier 1",
           "slab sum + partial norm", "norm mailboxes", "clip + Adam + soft update"]
names_a = ["first image staged (+ row fields)", "A: actor forward", "B: critic fwd + dX chain", "C: actor backward -> slab", "grid barrier 1",
           "slab sum + partial norm", "norm mailboxes", "clip + Adam + soft update"]
for title, names, do_actor in (("critic stage (last launch of a critic-only call)", names_c, False), ("actor stage", names_a, True)):
    if which != "td3" and not do_actor:
        continue
    t = stamps(do_actor)
    if not t.any():
        print("no stamps: this library's kernels_solo.hip was not compiled with -DFRL_SOLO_TIMING")
        break
    d = np.diff(np.concatenate([np.zeros((16, 1), np.float32), t], axis=1), axis=1)
    print("%s %s: us per section, workgroup 0 | mean over 16 | max; end of kernel at %.1f us (slowest workgroup)" % (which, title, t[:, 7].max()))
    for i, n in enumerate(names):
        print("   %-40s %6.2f | %6.2f | %6.2f" % (n, d[0, i], d[:, i].mean(), d[:, i].max()))
e.profile(True)
for k in range(200):
    e.learn(256, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(algo != N.ALGO_TD3 or k % 2 == 1), **kw)
pr = e.profile_read()
print("HIP-event time per launch (us):", {k: round(1e3 * v[0] / v[1], 2) for k, v in pr.items()})
e.close()
