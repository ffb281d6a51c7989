"""Reads a rocprofv3 kernel trace (CSV) of tools/overlap_trace.py and reports, for every halo
pack/unpack kernel, whether a sweep kernel of the main stream was running at that moment."""
import csv
import glob
import json
import re
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "")))
      for r in rows if "fdtd" in r["Kernel_Name"]]
sweeps = [e for e in ev if "sweep" in e[0]]
big = max(e[2] - e[1] for e in sweeps)
interior = [e for e in sweeps if e[2] - e[1] > 0.5 * big]           # the launches over the interior rows
halo = [e for e in ev if "halo_deep" in e[0]]
inside = sum(1 for h in halo if any(s[1] <= h[1] and h[2] <= s[2] for s in interior))
out = {"trace": path, "halo_kernels": len(halo), "halo_kernels_inside_an_interior_sweep": inside,
       "interior_sweeps": len(interior), "interior_sweep_avg_ms": sum(e[2] - e[1] for e in interior) / len(interior) / 1e6,
       "edge_sweeps": len(sweeps) - len(interior),
       "edge_sweep_avg_ms": (sum(e[2] - e[1] for e in sweeps if e not in interior) / max(1, len(sweeps) - len(interior))) / 1e6,
       "halo_kernel_avg_us": sum(h[2] - h[1] for h in halo) / max(1, len(halo)) / 1e3}
t0 = min(e[1] for e in ev)
out["timeline_excerpt_us"] = [[re.search(r"fdtd\w+", e[0].replace("fdtd::", "")).group(0), round((e[1] - t0) / 1e3, 1), round((e[2] - t0) / 1e3, 1), e[3]]
                              for e in sorted(ev, key=lambda e: e[1])[len(ev) // 2: len(ev) // 2 + 16]]
print(json.dumps(out, indent=1))
