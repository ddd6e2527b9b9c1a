"""Where a kernel's spilled VGPRs are touched: scratch loads / stores per basic block of hipcc's -save-temps assembly, next to the
block's MFMA count (the loops that matter are the blocks with hundreds of MFMAs).
    FRL_KEEP_ASM=1 python tools/kernel_regs.py; python tools/spill_blocks.py tools/_bin/frl_api-hip-amdgcn-amd-amdhsa-gfx950.s ac_critic_x_h2a1 [min_mfma]"""
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cur, blocks, name = None, [], None
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        cur = dict(k=name, b=m.group(1), n=0, mfma=0, st=0, ld=0)
        blocks.append(cur)
        continue
    t = line.strip()
    if cur is None or not t or t[0] in ";.":
        continue
    cur["n"] += 1
    cur["mfma"] += "v_mfma" in t
    cur["st"] += t.startswith("scratch_store")
    cur["ld"] += t.startswith("scratch_load")
mine = [b for b in blocks if b["k"] and kern in b["k"]]
hot = [b for b in mine if b["mfma"] >= min_mfma]
print("%s: %d basic blocks, %d scratch stores + %d scratch loads in all; blocks with >= %d MFMAs:" %
      (kern, len(mine), sum(b["st"] for b in mine), sum(b["ld"] for b in mine), min_mfma))
for b in hot:
    print("   %-12s %5d instructions  %4d MFMAs  %3d scratch stores  %3d scratch loads" % (b["b"], b["n"], b["mfma"], b["st"], b["ld"]))
print("   -> in the MFMA loops: %d stores, %d loads (%.1f per 100 MFMAs); elsewhere (set-up, epilogues, reductions): %d stores, %d loads" %
      (sum(b["st"] for b in hot), sum(b["ld"] for b in hot), 100.0 * sum(b["st"] + b["ld"] for b in hot) / max(1, sum(b["mfma"] for b in hot)),
       sum(b["st"] for b in mine) - sum(b["st"] for b in hot), sum(b["ld"] for b in mine) - sum(b["ld"] for b in hot)))
