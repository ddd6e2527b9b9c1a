"""Developer check: one learn() call on the chained family against the row-chunk family, parameter by parameter."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N
from freerl_amd.engine import Engine

def run(algo, twin, chained, persist, calls=1, seed=5):
    os.environ["FRL_CRITIC_V2"] = "1" if chained else "0"
    os.environ["FRL_CRITIC_PERSIST"] = str(persist)
    O, A, B, P = 8, 2, 256, 1
    e = Engine(algo, O, A, 2048, n_learners=P, twin_critic=twin, batch_max=B, seed=3)
    g = np.random.default_rng(seed)
    for net in range(2):
        flat = (g.standard_normal(e.num_params(net)) * 0.1).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE); e.set_params(net, flat, N.PARAM_TARGET)
    recs = g.standard_normal((1024, e.width)).astype(np.float32)
    recs[:, e.layout.done_off] = g.random(1024) < 0.05
    e.add_batch(recs)
    out = []
    for k in range(calls):
        idx = g.choice(1024, B, replace=False).astype(np.int64)[None, None, :]
        nz = g.standard_normal((P, 1, 2, B, A)).astype(np.float32)
        kw = dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, idx=idx, noise=nz, want_stats=True)
        if algo == N.ALGO_TD3: kw.update(do_actor=True, use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)
        if algo == N.ALGO_SAC: kw.update(alpha_lr=1e-4, target_entropy=-2.0)
        st = e.learn(B, **kw)
    params = {(net, kind): e.get_params(net, kind) for net in range(2) for kind in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V)}
    e.close()
    return st[0, 0, :2].copy(), params

def layer_of(net, twin, i):
    dims = [(128, 8), (128, 128), (2, 128)] if net == 0 else ([(128, 10), (128, 128), (1, 128)] * (2 if twin else 1))
    o = 0
    for li, (n, k) in enumerate(dims):
        if i < o + n * k: return "L%d.w[%d][%d]" % (li, (i - o) // k, (i - o) % k)
        o += n * k
        if i < o + n: return "L%d.b[%d]" % (li, i - o)
        o += n
    return "extra[%d]" % (i - o)

for name, algo, twin in (("ddpg", N.ALGO_DDPG, False), ("td3", N.ALGO_TD3, True)):
    for calls in (1, 3):
        ref_l, ref = run(algo, twin, False, 0, calls)
        for tag, persist in (("v2", 0), ("v3", 1), ("v3 again", 1)):
            l, p = run(algo, twin, True, persist, calls)
            print("%s calls=%d %-8s loss %s (row-chunk %s)" % (name, calls, tag, l, ref_l))
            for key in sorted(p):
                d = np.abs(p[key] - ref[key])
                i = int(d.argmax())
                nbad = int((d > 1e-5 + 1e-3 * np.abs(ref[key])).sum())
                print("    net %d kind %d  max |diff| %.3e at %s (ref %.5f got %.5f)  elements off: %d of %d" % (
                    key[0], key[1], d.max(), layer_of(key[0], twin, i), ref[key][i], p[key][i], nbad, d.size))
