#!/usr/bin/env python3
"""Summarises a profiles/collect.sh output directory: per-kernel time from the kernel-trace
stats and per-launch PMC averages for the sweep kernel, with the gfx950 FETCH_SIZE
correction (it reports half the bytes of wide coalesced reads; WRITE_SIZE is exact)."""
import csv, glob, json, os, sys
from collections import defaultdict

src = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "fdtd2d_sweep"
out = {"kernel": kernel, "source": src}
def newest(pattern):
    """gpurun merges every call's files into the same directory: keep the latest run only."""
    paths = sorted(glob.glob(pattern), key=os.path.getmtime)
    return paths[-1:]


for path in newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    out["kernel_stats"] = [dict(r) for r in csv.DictReader(open(path))]
counters = defaultdict(list)
for group in glob.glob(os.path.join(src, "pmc_*")):
  for path in newest(os.path.join(group, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
out["pmc_per_launch_mean"] = {k: sum(v) / len(v) for k, v in counters.items()}
out["pmc_launches"] = {k: len(v) for k, v in counters.items()}
pm = out["pmc_per_launch_mean"]
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    fetch = pm["FETCH_SIZE"] * 1024 * 2      # KB units; x2: gfx950 counts 128-B requests as 64 B
    write = pm["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = {"read_corrected": fetch, "write": write, "total": fetch + write,
                                   "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024"}
if "TCC_HIT_sum" in pm:
    out["l2_hit_rate"] = pm["TCC_HIT_sum"] / (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
