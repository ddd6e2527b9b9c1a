"""Every scratch (spill) access of one kernel with its place in the instruction stream and its source line:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -c freerl_amd/csrc/kernels_critic2.hip -o /tmp/x.o -save-temps=obj
    python tools/asm_spills.py /tmp/kernels_critic2-hip-amdgcn-amd-amdhsa-gfx950.s ac_critic_v2_twin_nv
(barrier / MFMA ordinals locate the pass; file numbers are the .s file's `.file` directives)"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % sys.argv[2], l))
cur, n, bars, mf = None, 0, 0, 0
for l in lines[start:]:
    if "s_endpgm" in l:
        break
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = "%s:%s" % (files.get(m.group(1), m.group(1)), m.group(2))
        continue
    if not l.startswith("\t") or l.strip().startswith((".", ";")):
        continue
    n += 1
    bars += "s_barrier" in l
    mf += "v_mfma" in l
    if "scratch_" in l:
        print("%6d  barrier %3d  mfma %5d  %-52s %s" % (n, bars, mf, l.strip().split(";")[0][:52], cur))
