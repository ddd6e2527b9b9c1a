"""Print per-kernel register / spill / scratch figures from hipcc's -save-temps assembly.
    FRL_HIP_VARIANT=dev FRL_HIPCC_FLAGS=-save-temps=obj python tools/kernel_regs.py"""
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
for blk in md.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    print("%-34s agpr %3s vgpr %3s spill %3s scratch %4s" % (re.sub(r"^_ZN3frl\d+", "", g("name"))[:34], blk.split()[0],
          g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
for f in glob.glob(os.path.join(out_dir, "frl_api-*")) + glob.glob(os.path.join(out_dir, "frl_api.hip-*")) + [N.LIB_PATH]:      # leave nothing for gpurun to ship
    if os.environ.get("FRL_KEEP_ASM") and f.endswith(".s"):
        continue
    os.remove(f)
