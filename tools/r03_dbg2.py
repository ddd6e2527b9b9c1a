"""Developer check: the background (pipelined) update of the persistent critic kernel, learner by learner: P = 2 learners on ONE
workgroup (learner 0's update runs inside learner 1's target passes) against P = 2 on two workgroups (both updates in the open)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FRL_CRITIC_V2"] = "1"; os.environ["FRL_CRITIC_PERSIST"] = "1"
from freerl_amd import _native as N
from freerl_amd.engine import Engine

def run(algo, twin, grid, calls=1, P=2, B=256):
    if grid: os.environ["FRL_CRITIC_GRID"] = str(grid)
    else: os.environ.pop("FRL_CRITIC_GRID", None)
    O, A = 8, 2
    e = Engine(algo, O, A, 2048, n_learners=P, twin_critic=twin, batch_max=B, seed=3)
    g = np.random.default_rng(5)
    for net in range(2):
        for p in range(P):
            flat = (g.standard_normal(e.num_params(net)) * 0.1).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
    recs = g.standard_normal((1024, e.width)).astype(np.float32)
    recs[:, e.layout.done_off] = g.random(1024) < 0.05
    for p in range(P): e.add_batch(recs, learners=[p] * 1024)
    for k in range(calls):
        idx = np.stack([g.choice(1024, B, replace=False) for _ in range(P)]).astype(np.int64)[:, None, :]
        nz = g.standard_normal((P, 1, 2, B, A)).astype(np.float32)
        kw = dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, idx=idx, noise=nz, want_stats=True)
        if algo == N.ALGO_TD3: kw.update(do_actor=True, use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)
        if algo == N.ALGO_SAC: kw.update(alpha_lr=1e-4, target_entropy=-2.0)
        st = e.learn(B, **kw)
    params = {(p, net, kind): e.get_params(net, kind, learner=p) for p in range(P) for net in range(2) for kind in range(4)}
    e.close()
    return st[:, 0, :2].copy(), params

def where(net, twin, i):
    dims = [(128, 8), (128, 128), (2, 128)] if net == 0 else ([(128, 10), (128, 128), (1, 128)] * (2 if twin else 1))
    o = 0
    for li, (n, k) in enumerate(dims):
        if i < o + n * k: return "L%d.w[%d][%d]" % (li, (i - o) // k, (i - o) % k)
        o += n * k
        if i < o + n: return "L%d.b[%d]" % (li, i - o)
        o += n
    return "extra[%d]" % (i - o)

def units_of(net, twin, bad):
    """(head, layer, wave, unit J) histogram of differing weight elements of the critic"""
    out = {}
    heads = 2 if twin else 1
    dims = [(128, 10), (128, 128), (1, 128)] * heads
    o = 0
    spans = []
    for li, (n, k) in enumerate(dims):
        spans.append((o, n, k, li)); o += n * k + n
    for i in bad:
        for (o, n, k, li) in spans:
            if o <= i < o + n * k:
                r, c = (i - o) // k, (i - o) % k
                hd, l3 = li // 3, li % 3
                if l3 == 1: w, J = (r >> 4) >> 1, ((r >> 4) & 1) * 8 + (c >> 4)
                elif l3 == 0: w, J = (r >> 4) >> 1, 16 + ((r >> 4) & 1)
                else: w, J = (c >> 4) >> 1, 18 + ((c >> 4) & 1)
                out[(hd, w, J)] = out.get((hd, w, J), 0) + 1
    return out

for name, algo, twin in (("ddpg", N.ALGO_DDPG, False), ("ddpg twin", N.ALGO_DDPG, True), ("td3 single", N.ALGO_TD3, False), ("td3", N.ALGO_TD3, True)):
    for B in (256,):
        ref_l, ref = run(algo, twin, 2, B=B)
        l, p = run(algo, twin, 1, B=B)
        print("%s B=%d losses open %s pipelined %s" % (name, B, ref_l.ravel(), l.ravel()))
        for key in sorted(p):
            d = np.abs(p[key] - ref[key])
            if d.max() > 0:
                bad = np.nonzero(d > 0)[0]
                print("    learner %d net %d kind %d: %d elements differ, max %.3e; first %s" % (key[0], key[1], key[2], bad.size, d.max(),
                      [where(key[1], twin, int(i)) for i in bad[:6]]))
                if key[1] == 1 and key[2] == 2:
                    big = np.nonzero(d > 1e-4)[0]
                    print("       critic m, |diff| > 1e-4 by (head, wave, unit): %s" % sorted(units_of(1, twin, big).items()))
