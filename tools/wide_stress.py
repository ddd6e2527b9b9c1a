"""Developer check: the K-sliced chained families against the row-chunk kernels over a spread of shapes — input widths on both
sides of every k-block and alignment boundary, one- and two-tile heads, batches from 1 row to ragged super-chunks, hidden 128
and 256, single- and multi-agent.      python tools/wide_stress.py [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wide_ab  # noqa: E402
from freerl_amd import _native as N  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
shapes = []
for obs, act in ((13, 3), (14, 2), (15, 1), (16, 16), (17, 15), (29, 3), (31, 17), (45, 4), (63, 1), (64, 32), (100, 7), (201, 12), (380, 20), (383, 17)):
    for hidden in (128, 256):
        B = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, 128, 129, 255, 256, 257, 300, 511, 512]))
        algo = [N.ALGO_TD3, N.ALGO_SAC, N.ALGO_DDPG][int(rng.integers(3))]
        shapes.append(dict(algo=algo, obs=obs, act=act, B=B, twin=algo != N.ALGO_DDPG, hidden=hidden))
for obs, act in (([5, 9, 3], [1, 2, 3]), ([17, 17], [6, 6]), ([18, 18, 18, 18], [5, 5, 5, 5]), ([40, 7], [9, 2])):
    for hidden in (128, 256):
        B = int(rng.choice([1, 33, 64, 200, 256, 320]))
        tw = bool(rng.integers(2))
        shapes.append(dict(algo=N.ALGO_MADDPG, obs=obs, act=act, B=B, twin=tw, matd3=tw, hidden=hidden))
bad = 0
for i, c in enumerate(shapes):
    name = "stress_%d" % i
    wide_ab.CASES[name] = c
    try:
        a, b = wide_ab.run(name, 0, 2), wide_ab.run(name, 1, 2)
    except Exception as ex:
        print("%-10s %s FAILED: %r" % (name, c, ex)); bad += 1
        continue
    worst = 0.0
    for key in a:
        if key in ("family", "layers"):
            continue
        x, y = np.asarray(a[key], np.float64), np.asarray(b[key], np.float64)
        if not (np.isfinite(x).all() and np.isfinite(y).all()):
            worst = float("inf")
            continue
        tol = 2e-2 if key.startswith(("theta", "target")) else 2e-3
        worst = max(worst, float(np.abs(x - y).max() / (np.abs(x).max() + 1e-30)) * (2e-3 / tol))
    ok = worst < 2e-3
    bad += not ok
    print("%-10s obs %-18s act %-14s B %4d hidden %3d algo %d twin %d  families %s/%s  worst %.1e %s" %
          (name, c["obs"], c["act"], c["B"], c["hidden"], c["algo"], c["twin"], a["family"], b["family"], worst, "OK" if ok else "MISMATCH"), flush=True)
print("%d shapes, %d bad" % (len(shapes), bad))
sys.exit(1 if bad else 0)
