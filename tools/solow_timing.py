"""Developer instrument: where one learner's update goes on kernels_solow.hip (sixteen workgroups, wide first layer): wall-clock
stamps (100 MHz) of thread 0 of every workgroup, in a library whose kernels_solow.hip was compiled with -DFRL_SOLO_TIMING
(bash tools/build_unit_variant.sh kernels_solow solowt -DFRL_SOLO_TIMING), plus the per-kernel HIP-event times of the chain.
    FRL_HIP_VARIANT=solowt python tools/solow_timing.py [sac|td3] [obs act]        (default: sac 376 17 = config 4)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N  # noqa: E402
from freerl_amd.engine import Engine  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sac"
obs, act = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (376, 17)
algo = dict(td3=N.ALGO_TD3, ddpg=N.ALGO_DDPG, sac=N.ALGO_SAC, maddpg=N.ALGO_MADDPG)[which]
kw = dict(td3=dict(use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0), ddpg={}, sac=dict(alpha_lr=1e-4, target_entropy=-float(act)), maddpg={})[which]
BATCH = 1024 if which == "maddpg" else 256          # (maddpg: config 5 — three agents of 18 / 5, batch 1024; the stamps are agent 0's first sixteen workgroups)
if which == "maddpg":
    e = Engine(algo, [18] * 3, [5] * 3, 20_000, n_learners=1, twin_critic=False, batch_max=BATCH, seed=1)
else:
    e = Engine(algo, obs, act, 20_000, n_learners=1, twin_critic=algo != N.ALGO_DDPG, batch_max=256, seed=1)
assert e.learn_path(BATCH)[2] == 16 and e.learn_path(BATCH)[0], e.learn_path(BATCH)
rng = np.random.default_rng(0)
for net in range(e.n_nets):
    flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
    e.set_params(net, flat, N.PARAM_ONLINE); e.set_params(net, flat, N.PARAM_TARGET)
if algo == N.ALGO_SAC:
    e.set_alpha_state([np.log(0.01), 0, 0, 0.01])
e.fill_synthetic(20_000, seed=5)


def stamps(do_actor):
    for k in range(6):
        e.learn(BATCH, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=do_actor, **kw)
    buf = np.zeros((16, 32), np.float32)
    N.check(N.lib().frl_solo_debug_read(e._h, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size))
    if do_actor and buf[0, 16]:
        print("   (pass A of workgroup 0: its first-layer sweep ends at %.2f us, layers 2 at %.2f, head tiles at %.2f, the policy's row rule at %.2f)"
              % tuple(buf[0, 16:20] * 0.01))
        print("   (... the sweep is entered at %.2f us, its first batch of four k-tiles swept at %.2f)" % tuple(buf[0, 20:22] * 0.01))
    return buf[:, 8:16] * 0.01          # us


names_c = ["first image staged (+ row fields)", "target actor forward", "target critic heads", "critic heads fwd + bwd -> slab", "slab hand-over",
           "slab sum -> grad + partial norm", "norm mailboxes", "clip + Adam + soft update"]
names_a = ["first image staged (+ row fields)", "A: actor forward", "B: critic fwd + dX chain", "C: actor backward -> slab", "slab hand-over",
           "slab sum -> grad + partial norm", "norm mailboxes", "clip + Adam + soft update"]
for title, names, do_actor in (("critic stage (last launch of a critic-only call)", names_c, False), ("actor stage", names_a, True)):
    if which != "td3" and not do_actor:
        continue
    t = stamps(do_actor)
    if not t.any():
        print("no stamps: this library's kernels_solow.hip was not compiled with -DFRL_SOLO_TIMING")
        break
    d = np.diff(np.concatenate([np.zeros((16, 1), np.float32), t], axis=1), axis=1)
    print("%s %s: us per section, workgroup 0 | mean over 16 | max; end of kernel at %.1f us (slowest workgroup)" % (which, title, t[:, 7].max()))
    for i, n in enumerate(names):
        print("   %-40s %6.2f | %6.2f | %6.2f" % (n, d[0, i], d[:, i].mean(), d[:, i].max()))
e.profile(True)
for k in range(200):
    e.learn(BATCH, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(algo != N.ALGO_TD3 or k % 2 == 1), **kw)
pr = e.profile_read()
print("HIP-event time per launch (us):", {k: round(1e3 * v[0] / v[1], 2) for k, v in pr.items()})
e.close()
