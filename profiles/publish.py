#!/usr/bin/env python3
"""Copies what profiles/collect.sh left under gpurun_out/<tag>/ into profiles/<tag>/ (kernel
stats, the per-launch PMC CSVs, the bench line printed under the profiler, a summary) and enters
the sweep's HBM traffic per launch into profiles/traffic.json under the key bench.py looks up.
Usage: publish.py <tag> <kernel substring> [<steps per launch>]"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, kernel = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    paths = sorted(glob.glob(pattern), key=os.path.getmtime)
    return paths[-1] if paths else None


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
for group in glob.glob(os.path.join(src, "pmc_*")):
    if os.path.isdir(group):
        path = newest(os.path.join(group, "*", "*_counter_collection.csv"))
        if path:
            shutil.copy(path, os.path.join(dst, os.path.basename(group) + ".csv"))
if os.path.exists(os.path.join(src, "bench_trace.json")):
    shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))
summary = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "profiles", "summarize.py"),
                                              os.path.join("gpurun_out", tag), kernel], cwd=ROOT))
summary["source"] = f"profiles/{tag}"
name = kernel.rstrip("<")
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_{name}_summary.json"), "w"), indent=1)
bench = json.load(open(os.path.join(dst, "bench_under_rocprof.json")))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
key = f"{bench['config']['workload']} [{name} x{steps}]"
tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
h = summary["hbm_bytes_per_launch"]
avg = [float(r["AverageNs"]) for r in summary.get("kernel_stats", []) if name in r["Name"] and (kernel.endswith("<") and name + "<" in r["Name"] or not kernel.endswith("<"))]
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16
traffic[key] = {"kernel_source_sha16": kernel_source_sha16(name), "hbm_bytes_per_launch": h["total"], "read_corrected": h["read_corrected"], "write": h["write"],
                "kernel": name, "steps_per_launch": steps, "kernel_avg_ns_rocprofv3": avg[0] if avg else None,
                "l2_hit_rate": summary.get("l2_hit_rate"),
                "source": f"profiles/{tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 on gfx950)"}
json.dump(traffic, open(tpath, "w"), indent=1)
print(key, traffic[key])
