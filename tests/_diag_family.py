"""Diagnostic (not collected by pytest): when tools/wide_ab.py / wide_stress.py flag a difference between the kernel families, which
of them agrees with the oracle — tools/wide_ab.py's SAC 380 / 20 data replayed learner by learner.
    python tests/_diag_family.py [batch] [seed] [hidden] [calls]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freerl_amd import _native as N
from freerl_amd.engine import Engine
from oracle import algos
from tests.hip_helpers import flat_params, unflat_params
from tests.golden import cases, synth

O, A, B, P = 380, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 17, 2
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
CALLS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
an = ["l1", "l2", "mean_layer"]
TWIN = ["l1", "l2", "l3", "l4", "l5", "l6"]
tmpl_a = synth.mlp_params(1, cases.actor_layers(O, A, head="mean_layer", hidden=H))
tmpl_a = dict([("log_std", np.zeros((1, A), np.float32))] + list(tmpl_a.items()))
tmpl_c = synth.mlp_params(2, cases.critic_layers(O + A, twin=True, hidden=H))
res = {}
for fam in (0, 1):
    os.environ["FRL_CRITIC_V2"] = str(fam)
    e = Engine(N.ALGO_SAC, O, A, 4096, n_learners=P, twin_critic=True, batch_max=B, hidden=H, seed=3)
    g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    par = {}
    for net in range(e.n_nets):
        for p in range(P):
            flat = (g.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
            tg = flat + np.float32(0.01) * g.standard_normal(flat.size).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, tg, N.PARAM_TARGET, learner=p)
            par[(net, p)] = (flat, tg)
    for p in range(P):
        e.set_alpha_state([np.log(0.2), 0, 0, 0.2], learner=p)
    e.fill_synthetic(3000, seed=5)
    rows = {p: e.read_rows(p, 0, 3000) for p in range(P)}
    idxs, noises = [], []
    for k in range(CALLS):
        idx = np.stack([[g.choice(3000, B, replace=False)] for _ in range(P)]).astype(np.int64)
        noise = g.standard_normal((P, 1, 2, B, A)).astype(np.float32)
        idxs.append(idx); noises.append(noise)
        st = e.learn(B, gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3, alpha_lr=1e-3, target_entropy=-float(A), idx=idx[:, 0], noise=noise, want_stats=True)
    res[fam] = dict(m=[e.get_params(0, N.PARAM_ADAM_M, learner=p) for p in range(P)], st=st.copy(), fam=e.learn_path(B)[0])
    if fam == 0:
        keep = (par, rows, idxs, noises)
    e.close()
par, rows, idxs, noises = keep
for p in range(P):
    actor = unflat_params(par[(0, p)][0], tmpl_a, an, "log_std"); actor_t = unflat_params(par[(0, p)][1], tmpl_a, an, "log_std")
    critic = unflat_params(par[(1, p)][0], tmpl_c, TWIN); critic_t = unflat_params(par[(1, p)][1], tmpl_c, TWIN)
    # flat order is [layers..., log_std]; the dict wants log_std too
    orc = algos.SAC(actor, critic, O, A, 1e-3, 1e-3, 4096, alpha0=0.2, alpha_lr=1e-3)
    orc.actor_t, orc.critic_t = actor_t, critic_t
    r = rows[p]
    for i in range(3000):
        orc.add(r[i, :O], r[i, O:O + A], float(r[i, O + A]), r[i, O + A + 2:O + A + 2 + O], bool(r[i, O + A + 1]))
    for k in range(CALLS):
        cl, al, ll = orc.learn_with(idxs[k][p, 0], noises[k][p, 0, 0], noises[k][p, 0, 1], 0.99, 0.01)
    om = flat_params(orc.actor_opt.m, an, "log_std")
    for fam in (0, 1):
        d = np.abs(res[fam]["m"][p] - om)
        k = int(d.argmax())
        big = np.nonzero(d > 1e-6)[0]
        if big.size:
            print("   %d elements off by > 1e-6: indices %s ... (W1 | b1 | W2 | b2 128 | W3 | b3 | log_std; W2 starts at 49280); index - 49280 = (out, in) %s" %
                  (big.size, big[:12], [divmod(int(i) - 49280, 128) for i in big[:12]]))
        print("learner %d family %d (%s): actor loss %.7g (oracle %.7g)  max |m - oracle m| %.3e at %d (%.6g vs %.6g), max |m| %.3e" %
              (p, fam, "chained" if res[fam]["fam"] else "row-chunk", res[fam]["st"][p, 0, N.STAT_ACTOR_LOSS], al, d.max(), k, res[fam]["m"][p][k], om[k], np.abs(om).max()))
