"""Summarise rocprofv3 counter-collection CSVs into per-kernel means (one JSON per pass) and derive
profiles/traffic.json for bench.py's roofline.traffic.

    python tools/pmc_summary.py <rocprof output dir> <out.json> [--traffic profiles/traffic.json --fetch A.json --write B.json]

Kernels are keyed "name grid=<threads>" so the P = 512 bench dispatches stay apart from the single-learner ones."""
import collections
import csv
import glob
import json
import os
import re
import sys


def summarise(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            out["%s grid=%s" % (name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: dict(dispatches=len(v), mean=sum(v) / len(v)) for c, v in cs.items()} for k, cs in sorted(out.items())}


def kernel_source_digest():
    """sha256 over the sources the headline gradient kernel is compiled from: profiles/traffic.json records it, bench.py
    compares it, so a traffic figure measured on an older kernel is reported as stale instead of silently reused."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "freerl_amd", "csrc")
    h = hashlib.sha256()
    for f in ("kernels_critic2.hip", "kernels_critic.hip", "kernels.h", "frl_desc.h", "device/chain.hpp", "device/chain_net.hpp", "device/net.hpp", "device/tile.hpp",
              "device/update_common.hpp"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    a = sys.argv[1:]
    if a and a[0] == "--traffic":
        out, fetch, write, kernel = a[1], json.load(open(a[2])), json.load(open(a[3])), a[4]
        pick = lambda d, c: max(((v[c]["mean"], k) for k, v in d.items() if k.startswith("frl::" + kernel) and c in v))
        f, key = pick(fetch, "FETCH_SIZE")
        w, _ = pick(write, "WRITE_SIZE")
        json.dump({"kernel": kernel, "dispatch": key, "kernel_source_digest": kernel_source_digest(), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                   "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled "
                                 "(MI355X_MICROARCH.md, HBM); WRITE_SIZE uncalibrated, taken as is",
                   "hbm_bytes_per_launch": (2 * f + w) * 1024.0}, open(out, "w"), indent=1)
        return
    json.dump(summarise(a[0]), open(a[1], "w"), indent=1)


if __name__ == "__main__":
    main()
