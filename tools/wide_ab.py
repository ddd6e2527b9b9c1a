"""Developer check: the chained kernel families (FRL_CRITIC_V2=1) against the row-chunk kernels (=0) on the same inputs, array by array —
the printing front end of tests/family_ab.py (tests/test_gpu_family_ab.py asserts the same comparison under pytest).
    python tools/wide_ab.py [sac_c4 | td3_wide | ddpg_wide | maddpg_c5 | ... | all] [calls] [learners]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.family_ab import CASES, diff, run  # noqa: E402


def compare(name, calls, P=2):
    a, b = run(name, 0, calls, P), run(name, 1, calls, P)
    print("== %s: families %s / %s, %d calls" % (name, a["family"], b["family"], calls))
    d = diff(a, b)
    st = d.pop("stats")
    for k in range(calls):
        for ag in range(st.shape[2]):
            print("  call %d agent %d stats rel diff (critic, actor, alpha_loss, alpha, cgnorm, agnorm, ent): %s" %
                  (k, ag, " ".join("%.1e" % x for x in st[k, :, ag].max(axis=0)[:7])))
    worst = 0.0
    for key, (w, at, mx, _q99) in d.items():
        # theta / target: one Adam step is +-lr whatever the gradient's size, so a unit that is dead in one family and barely alive in
        # the other moves its weights by a full lr (1e-3 here, ~5e-3 of the largest weight): 2e-2 for those arrays
        tol = 2e-2 if key.startswith(("theta", "target")) else 2e-3
        worst = max(worst, w * (2e-3 / tol))
        print("  %-9s max |diff| / max |x| %.2e  (|x| max %.3g)%s" % (key, w, mx, "" if w < tol else "   <-- at flat index %d" % at))
    print("  WORST (scaled to a 2e-3 tolerance) %.2e %s" % (worst, "OK" if worst < 2e-3 else "MISMATCH"))
    return worst < 2e-3


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    ok = True
    for name in (CASES if which == "all" else [which]):
        try:
            ok &= compare(name, calls, P)
        except Exception as ex:
            print("== %s FAILED: %r" % (name, ex))
            ok = False
    sys.exit(0 if ok else 1)
