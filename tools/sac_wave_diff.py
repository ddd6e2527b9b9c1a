"""Developer instrument: SAC (long_sac fixture) on the eight-wave and the four-wave chained kernels, every array after every call:
where and when the two diverge (round 6: a dead-unit ReLU flip at call 10).   python tools/sac_wave_diff.py"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["FRL_CRITIC_V2"] = "1"
from freerl_amd import _native as N
from freerl_amd.engine import Engine
from tests.golden import long_cases as LC
from tests.hip_helpers import flat_params, records
c = LC.LONG["long_sac"]; inp = LC.ac_inputs(c)
O, A, B = c["obs_dim"], c["act_dim"], c["batch"]
tab, actor, critic = inp["table"], inp["params"]["actor"], inp["params"]["critic"]
an = ["l1", "l2", "mean_layer"]; cn = ["l1", "l2", "l3", "l4", "l5", "l6"]
def run(waves, ncalls):
    os.environ["FRL_CHAIN_WAVES"] = str(waves)
    e = Engine(N.ALGO_SAC, O, A, 4096, twin_critic=True, batch_max=B)
    for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
        e.set_params(0, flat_params(actor, an, "log_std"), kind); e.set_params(1, flat_params(critic, cn), kind)
    e.add_batch(records([tab])); e.set_alpha_state([np.log(0.01), 0, 0, 0.01], 0)
    out = []
    for k in range(ncalls):
        nz = np.zeros((1, 1, 2, B, A), np.float32); nz[0, 0, 0] = inp["noise"][k][0]; nz[0, 0, 1] = inp["noise"][k][1]
        st = e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, alpha_lr=1e-4, target_entropy=-float(A), idx=inp["idx"][k], noise=nz, want_stats=True)
        out.append(dict(stats=st[0, 0].copy(), a=e.get_params(0), at=e.get_params(0, N.PARAM_TARGET), am=e.get_params(0, N.PARAM_ADAM_M),
                        c=e.get_params(1), ct=e.get_params(1, N.PARAM_TARGET), cm=e.get_params(1, N.PARAM_ADAM_M), alpha=np.array(e.alpha_state(0)[0])))
    e.close()
    return out
r8, r4 = run(8, 14), run(4, 14)
for k in range(14):
    row = []
    for key in ("stats", "a", "at", "am", "c", "ct", "cm", "alpha"):
        d = np.abs(r8[k][key] - r4[k][key]); s = np.abs(r4[k][key]).max() + 1e-30
        row.append("%s %.1e@%d" % (key, d.max() / s, int(d.argmax())))
    print(k, " ".join(row))
print("actor params", r4[0]["a"].size, "log_std at", r4[0]["a"].size - A)
