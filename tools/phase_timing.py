"""Developer instrument: where does a gradient-kernel workgroup spend its cycles?

Builds the `phase` variant of the library (-DFRL_PHASE_TIMING), runs the bench workload's TD3 learn step
and prints, for 8 sampled workgroups of `ac_critic_kernel`, the shader-clock cycles between consecutive
barrier-delimited phases (thread 0's view; 4 workgroups share a CU, so a phase's time includes the other
three's interleaved work — read the numbers as shares, not latencies).

    FRL_HIP_VARIANT=phase FRL_HIPCC_FLAGS=-DFRL_PHASE_TIMING python tools/phase_timing.py [P] > gpurun_out/phase.txt
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRL_HIP_VARIANT", "phase")
os.environ.setdefault("FRL_HIPCC_FLAGS", "-DFRL_PHASE_TIMING")
from freerl_amd import _native as N  # noqa: E402

FWD = lambda n: [n + " l1", n + " l2 + head dots"]
BWD = ["head bwd (dW3,dX3)", "dW2", "dX2", "dW1"]
# barrier-delimited phases of one row chunk (TD3, single agent: the in-place paths of the kernels)
LABELS_TD3 = (["gather s'"] + FWD("pi'") + ["pi' finalize + a' into the Q input", "Q1' and Q2' l1", "Q1' and Q2' l2 + head dots",
              "y = r + g min q", "gather [s|a]"] + FWD("Q1") + ["Q1 finalize + delta"] + BWD +
              ["(second head: input still in place)"] + FWD("Q2") + ["Q2 finalize + delta"] + BWD)

LABELS_TD3_ACTOR = (["gather s"] + FWD("pi") + ["pi finalize + action into the Q input + spill h1,h2"] + FWD("Q1") +
                    ["Q1 finalize + q + dq", "head bwd (dX3)", "dX2", "dX1 (action columns)", "da += dx",
                     "reload h1,h2 + gather s + head delta", "head bwd (dW3,dX3)", "dW2", "dX2", "dW1"])


def c51(P):
    """Wave 0's stamp-to-stamp cycles through c51_grad_kernel (the reference's default Rainbow set), first row chunk."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    N.build()
    from freerl_amd.engine import Engine
    L = N.lib()
    fn = L.frl_debug_phase_clocks
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_int), C.c_int]
    e = Engine(N.ALGO_DQN, 8, 4, 100_000, discrete=True, batch_max=256, n_learners=P, seed=1, dueling=True, noisy=True,
               c51=(51, -100.0, 100.0))
    rng = np.random.default_rng(0)
    for p in range(P):
        flat = (rng.standard_normal(e.get_params(0, learner=p).size) * 0.05).astype(np.float32)
        e.set_params(0, flat, N.PARAM_ONLINE, learner=p)
        e.set_params(0, flat, N.PARAM_TARGET, learner=p)
    e.fill_synthetic(100_000, seed=5)
    rc = e.lds_bytes()[1]
    nblk = P * (256 // rc)
    buf = (C.c_int * (8 * 5 * 64))()
    assert fn(buf, max(1, nblk // 8 - 3)) == 0
    assert fn(buf, -3) == 0
    for it in range(4):
        e.learn(256, gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0, double_dqn=True)
    assert fn(buf, 0) == 0
    raw = np.array(buf[:], dtype=np.int64).reshape(8, 5, 64)
    d = np.diff(raw[:, 4, :].astype(np.float64), axis=1)
    d = d[:, :(d[0] > 0).sum()]
    print("P=%d, %d rows per workgroup; wave 0 barrier-to-barrier cycles, mean over sampled workgroups (total %.0f):" % (P, rc, d.mean(axis=0).sum()))
    print(np.round(d.mean(axis=0)).astype(int).tolist())


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "c51":
        return c51(int(sys.argv[2]) if len(sys.argv) > 2 else 512)
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    actor = len(sys.argv) > 2 and sys.argv[2] == "actor"
    N.build()
    from freerl_amd.engine import Engine
    L = N.lib()
    fn = L.frl_debug_phase_clocks
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_int), C.c_int]
    import bench
    B = bench.BATCH
    e = bench.make_engine(N, Engine, P, 0, 1)    # the bench workload: replay 1e6 filled, random-init weights
    rc = e.lds_bytes()[1]
    nblk = P * ((B + rc - 1) // rc)
    KMAX = 64
    buf = (C.c_int * (8 * 5 * KMAX))()
    stride = max(1, nblk // 8 - 3) if P >= 8 else 8       # P < 8: blocks 0, 8, 16 ... are unit 0's slices (update_common.hpp)
    assert fn(buf, stride) == 0
    assert fn(buf, -2 if actor else -1) == 0            # which kernel dumps its stamps
    for it in range(6):
        e.learn(B, **bench.td3_kwargs(1 if actor else 0))
    assert fn(buf, 0) == 0
    labels = LABELS_TD3_ACTOR if actor else LABELS_TD3
    raw = np.array(buf[:], dtype=np.int64).reshape(8, 5, KMAX)
    if os.environ.get("FRL_RAW_MARKS"):        # wave 0's stamp row as-is (FRL_MARK() experiments)
        row = raw[:, 4, :].astype(np.float64)
        d = np.diff(row, axis=1)
        print("wave-0 stamp-to-stamp cycles, mean over sampled workgroups:")
        print(np.round(d.mean(axis=0)).astype(int).tolist())
        return
    nb = len(labels)
    arrive = raw[:, :4, :nb].astype(np.float64)                              # [block][wave][barrier]
    rel0 = raw[:, 4, :nb + 1].astype(np.float64)                             # init stamp, then wave 0's release of each barrier
    t0 = rel0[:, :1]
    unwrap = lambda a, ref: a + 2.0 ** 32 * (a < ref - 2.0 ** 31)            # 32-bit stamps
    arrive = unwrap(arrive, t0[:, None, :])
    rel0 = unwrap(rel0, t0)
    release = np.repeat(rel0[:, None, 1:], 4, axis=1)
    prev = np.repeat(rel0[:, None, :-1], 4, axis=1)
    work = arrive - prev
    wait = release - arrive
    t = None
    total = (rel0[:, -1] - rel0[:, 0]).mean()
    print("P=%d, %d rows per workgroup, sampled every %d blocks; mean cycles per workgroup %.0f" % (P, rc, stride, total))
    print("%-24s %9s %9s %9s %9s %7s" % ("phase (ends at barrier)", "work mean", "work max", "work min", "wait mean", "share"))
    for k, lab in enumerate(labels):
        w = work[:, :, k]
        print("%-24s %9.0f %9.0f %9.0f %9.0f %6.1f%%" % (lab, w.mean(), w.max(axis=1).mean(), w.min(axis=1).mean(),
              wait[:, :, k].mean(), 100 * (w.mean() + wait[:, :, k].mean()) / total))
    print("sum work %.0f  sum wait %.0f" % (work.mean(axis=(0, 1)).sum(), wait.mean(axis=(0, 1)).sum()))
    e.close()


if __name__ == "__main__":
    main()
