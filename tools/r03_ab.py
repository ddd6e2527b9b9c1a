"""Developer check: the persistent critic kernel (FRL_CRITIC_PERSIST=1) against round 2's launch shape (=0) on the same inputs,
bitwise, for DDPG / TD3 / SAC at P learners; then both against the oracle-free invariant that learner p's result does not depend
on which workgroup slot handled it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FRL_CRITIC_V2"] = "1"
from freerl_amd import _native as N
from freerl_amd.engine import Engine

def run(algo, twin, persist, P, calls, grid=None):
    os.environ["FRL_CRITIC_PERSIST"] = str(persist)
    if grid: os.environ["FRL_CRITIC_GRID"] = str(grid)
    else: os.environ.pop("FRL_CRITIC_GRID", None)
    O, A, B = 8, 2, 256
    e = Engine(algo, O, A, 2048, n_learners=P, twin_critic=twin, batch_max=B, seed=3)
    g = np.random.default_rng(5)
    for net in range(2):
        for p in range(P):
            flat = (g.standard_normal(e.num_params(net)) * 0.1).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
    if algo == N.ALGO_SAC:
        for p in range(P): e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0, learner=p)
    recs = g.standard_normal((1024, e.width)).astype(np.float32)
    recs[:, e.layout.done_off] = g.random(1024) < 0.05
    for p in range(P):
        e.add_batch(recs, learners=[p] * 1024)
    losses = []
    for k in range(calls):
        idx = np.stack([g.choice(1024, B, replace=False) for _ in range(P)]).astype(np.int64)[:, None, :]
        nz = g.standard_normal((P, 1, 2, B, A)).astype(np.float32)
        kw = dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, idx=idx, noise=nz, want_stats=True)
        if algo == N.ALGO_TD3: kw.update(do_actor=(k % 2 == 1), use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)
        if algo == N.ALGO_SAC: kw.update(alpha_lr=1e-4, target_entropy=-2.0)
        st = e.learn(B, **kw)
        losses.append(st[:, 0, :2].copy())
    params = [np.stack([e.get_params(net, kind, learner=p) for p in range(P)]) for net in range(2) for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)]
    e.close()
    return np.array(losses), params

for name, algo, twin in (("ddpg", N.ALGO_DDPG, False), ("td3", N.ALGO_TD3, True), ("sac", N.ALGO_SAC, True)):
    for P in (1, 5):
        l0, p0 = run(algo, twin, 0, P, 12)
        l1, p1 = run(algo, twin, 1, P, 12)
        l2, p2 = run(algo, twin, 1, P, 12, grid=2)          # P = 5 on 2 workgroups: 3 + 2 learners, updates pipelined
        d01 = max(float(np.abs(a - b).max()) for a, b in zip(p0, p1))
        d12 = max(float(np.abs(a - b).max()) for a, b in zip(p1, p2))
        print("%-5s P=%d  loss max|v2 - v3| %.3e   params max|v2 - v3| %.3e   params max|v3 - v3(grid 2)| %.3e   first losses v2 %s v3 %s" % (
            name, P, float(np.abs(l0 - l1).max()), d01, d12, l0[:3, 0, 0], l1[:3, 0, 0]), flush=True)
