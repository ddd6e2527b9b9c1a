"""Print per-kernel register / spill / scratch figures from hipcc's -save-temps assembly.
    FRL_HIP_VARIANT=dev FRL_HIPCC_FLAGS=-save-temps=obj python tools/kernel_regs.py [--check]"""
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRL_HIP_VARIANT", "dev")
os.environ["FRL_HIPCC_FLAGS"] = os.environ.get("FRL_HIPCC_FLAGS", "") + " -save-temps=obj"
from freerl_amd import _native as N  # noqa: E402

if not (os.environ.get("FRL_REGS_REUSE") and os.path.exists(N.LIB_PATH)):
    N.build(force=True)
out_dir = os.path.dirname(N.LIB_PATH)                      # variants are built into tools/_bin/, next to their temporaries
asm = glob.glob(os.path.join(out_dir, "*gfx950.s"))[0]
s = open(asm).read()
md = s[s.index("amdhsa.kernels:"):]
# upper bounds on spilled VGPRs (`--check`: exit 1 when a kernel passes its bound — a regression gate for the hot kernels, which must
# stay spill-free, and a ratchet for the K-sliced / x-stationary / PPO ones, whose spills sit outside their MFMA loops: DESIGN.md 8)
# (round 6: the lane constants re-derived per phase took the K-sliced kernels from 97-397 spilled VGPRs to 0-51, PPO's from 135-140 to 0)
BOUNDS = [("ac_critic_v2_", 8), ("ac_actor_v2_", 0), ("solo_critic_twin_w8_", 8), ("solo_", 0), ("solow_", 0), ("dqn_fused_", 0), ("c51_grad_", 0), ("ac_critic_kernel", 0),
          ("ac_critic_wide_", 60), ("ac_actor_wide_", 16), ("ac_critic_x_", 700), ("ac_actor_x_", 460), ("ppo_update_v2_", 0),
          ("", 20)]
bad = []
for blk in md.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    short = re.sub(r"^_ZN3frl\d+", "", g("name"))
    bound = next(b for pre, b in BOUNDS if short.startswith(pre))
    if int(g("vgpr_spill_count")) > bound:
        bad.append("%s: %s spilled VGPRs > %d" % (short[:40], g("vgpr_spill_count"), bound))
    print("%-34s agpr %3s vgpr %3s spill %3s scratch %4s" % (re.sub(r"^_ZN3frl\d+", "", g("name"))[:34], blk.split()[0],
          g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
for f in glob.glob(os.path.join(out_dir, "frl_api-*")) + glob.glob(os.path.join(out_dir, "frl_api.hip-*")) + [N.LIB_PATH]:      # leave nothing for gpurun to ship
    if os.environ.get("FRL_KEEP_ASM") and f.endswith(".s"):
        continue
    os.remove(f)
if "--check" in sys.argv:
    if bad:
        print("spill bounds exceeded:\n  " + "\n  ".join(bad))
        sys.exit(1)
    print("spill bounds hold")
