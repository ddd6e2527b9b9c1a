"""Register / spill / scratch figures of every kernel in one hipcc -save-temps assembly file:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c freerl_amd/csrc/kernels_critic2.hip -o /tmp/x.o -save-temps=obj
    python tools/asm_regs.py /tmp/kernels_critic2-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys

s = open(sys.argv[1]).read()
md = s[s.index("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    print("%-40s agpr %3s vgpr %3s sgpr %3s spill %3s scratch %4s lds %s" % (re.sub(r"^_ZN3frl\d+", "", g("name"))[:40], blk.split()[0],
          g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
