"""Where a kernel's scratch accesses sit: per kernel of a -save-temps assembly, the instruction stream cut into blocks of N
instructions, each with its MFMA / LDS / VMEM / scratch counts (a spill inside an MFMA block costs; one between passes does not).
    python tools/asm_spillmap.py <file.s> <kernel substring> [block=200]"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % pat, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
ins = [l.strip() for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
print("%d instructions" % len(ins))
for b in range(0, len(ins), blk):
    seg = ins[b:b + blk]
    c = lambda p: sum(1 for x in seg if x.startswith(p))
    print("%6d  mfma %3d  ds_r %3d  ds_w %3d  vmem %3d  scratch_ld %3d  scratch_st %3d  barrier %d" % (
        b, c("v_mfma"), c("ds_read") + c("ds_load"), c("ds_write") + c("ds_store"), c("global_") + c("buffer_"),
        c("scratch_load"), c("scratch_store"), c("s_barrier")))
