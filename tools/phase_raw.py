"""Developer instrument: raw barrier-to-barrier cycle counts (wave 0's view) of the row-chunk gradient kernels at shapes other
than the bench's, one line per sampled workgroup — read them against the kernel's FRL_PHASE sequence.
    python tools/phase_raw.py {c4|c5|syn} {0|1} [hidden]     (c4: SAC at Humanoid dims, c5: MADDPG simple_spread; 0 critic / 1 actor kernel)
Builds the `phase` variant (-DFRL_PHASE_TIMING, unity).  This is how the serial gather (94 k cycles per 376-column gather at c4)
was found."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FRL_HIP_VARIANT"] = "phase"; os.environ["FRL_HIPCC_FLAGS"] = "-DFRL_PHASE_TIMING"
os.environ.setdefault("FRL_CRITIC_V2", "0")
from freerl_amd import _native as N
from freerl_amd.engine import Engine
L = N.lib()
fn = L.frl_debug_phase_clocks
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_int), C.c_int]
kind = sys.argv[1]
if kind == "c4":
    P, O, A, B, algo, kid = 128, 376, 17, 256, N.ALGO_SAC, int(sys.argv[2])
elif kind == "c5":
    P, O, A, B, algo, kid = 64, [18, 18, 18], [5, 5, 5], 1024, N.ALGO_MADDPG, int(sys.argv[2])
else:
    P, O, A, B, algo, kid = 256, 8, 2, 256, N.ALGO_TD3, int(sys.argv[2])
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
e = Engine(algo, O, A, 20000, n_learners=P, twin_critic=(kind != 'c5'), batch_max=B, hidden=H, seed=1)
rng = np.random.default_rng(0)
for net in range(e.n_nets):
    n = e.get_params(net, learner=0).size
    for p in range(P):
        flat = (rng.standard_normal(n) * 0.05).astype(np.float32)
        e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
if algo == N.ALGO_SAC:
    for p in range(P): e.set_alpha_state([np.log(0.01), 0, 0, 0.01], learner=p)
e.fill_synthetic(20000, seed=5)
print("lds, rc", e.lds_bytes())
buf = (C.c_int * (8 * 5 * 64))()
assert fn(buf, 61) == 0
assert fn(buf, -1 - kid) == 0
for it in range(4):
    e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=True, alpha_lr=1e-4, target_entropy=-1.0)
assert fn(buf, 0) == 0
raw = np.array(buf[:], dtype=np.int64).reshape(8, 5, 64)
for b in range(2):
    d = np.diff(raw[b, 4, :])
    print("WG", b, "diffs:", d.tolist(), "sum", d[d > 0][:40].sum())
