#!/usr/bin/env python3
"""Copies what profiles/collect_mid.sh left under gpurun_out/r03_mid/ into profiles/r03_mid/ (kernel
stats, the two PMC CSVs, the wave timeline of one pass) and writes profiles/r03_mid/summary.json: per
grid and band height the waves of a launch against the 2048 wave slots, the pass time, the share of
a wave's cycles spent issuing / waiting, the read latency, and how the launch ends (when the number
of waves in flight falls below the slot count and how long the rest takes)."""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r03_mid"), os.path.join(ROOT, "profiles", "r03_mid")
KERNEL = "fdtd2d_sweep_tb"
SLOTS = 2048


def newest(pattern):
    paths = sorted(glob.glob(pattern), key=os.path.getmtime)
    return paths[-1] if paths else None


def pmc_means(path):
    vals = {}
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


summary = {}
for d in sorted(glob.glob(os.path.join(src, "*_rows*"))):
    name = os.path.basename(d)
    out = os.path.join(dst, name)
    os.makedirs(out, exist_ok=True)
    row = {}
    stats = newest(os.path.join(d, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(out, "kernel_stats.csv"))
        for r in csv.DictReader(open(stats)):
            if KERNEL in r["Name"]:
                row["rocprofv3_avg_us"] = float(r["AverageNs"]) / 1e3
                row["launches"] = int(r["Calls"])
    for grp in ("sq", "tcp"):
        p = newest(os.path.join(d, "pmc_" + grp, "*", "*_counter_collection.csv"))
        if p:
            shutil.copy(p, os.path.join(out, f"pmc_{grp}.csv"))
            row.update(pmc_means(p))
    tl = os.path.join(d, "timeline.json")
    if os.path.exists(tl):
        shutil.copy(tl, os.path.join(out, "timeline.json"))
        t = json.load(open(tl))
        row["rows_per_wave"] = t["rows_per_wave"]
        row["waves_per_launch"] = t["waves"]
        row["waves_over_slots"] = t["waves"] / SLOTS
        row["launch_us_hip_events"] = t["launch_ms_hip_events"] * 1e3
        row["wave_life_us"] = {k: v["life_us_median"] for k, v in t["kinds"].items()}
        f = t["in_flight"]
        full = [x for x, w in zip(f["t_us"], f["waves"]) if w >= 0.95 * min(SLOTS, t["waves"])]
        row["full_until_us"] = max(full) if full else 0.0
        row["last_round_us"] = t["span_us"] - row["full_until_us"]
    b = os.path.join(d, "bench_trace.json")
    if os.path.exists(b):
        row["bench_under_rocprof_Gcell_updates_per_s"] = json.load(open(b))["value"] / 1e3
    if "SQ_WAVE_CYCLES" in row:
        wc = row["SQ_WAVE_CYCLES"]
        row["share_of_wave_cycles"] = {"valu_issue": row.get("SQ_ACTIVE_INST_VALU", 0) / wc, "waiting_any": row.get("SQ_WAIT_ANY", 0) / wc,
                                       "waiting_for_issue": row.get("SQ_WAIT_INST_ANY", 0) / wc}
    if row.get("TCP_TCC_READ_REQ_sum"):
        row["read_latency_cycles"] = row["TCP_TCC_READ_REQ_LATENCY_sum"] / row["TCP_TCC_READ_REQ_sum"]
    summary[name] = row
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
for k, v in summary.items():
    print(k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items() if x in (
        "rows_per_wave", "waves_per_launch", "waves_over_slots", "rocprofv3_avg_us", "launch_us_hip_events", "full_until_us", "last_round_us",
        "read_latency_cycles", "share_of_wave_cycles", "wave_life_us", "SQ_WAVES")})
