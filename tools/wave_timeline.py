#!/usr/bin/env python3
"""Wave timeline of one pass of the multi-step 2-D sweep: how long each kind of wave lives, how many
are in flight over the launch, when the launch would end if every wave were as fast as the plain ones.
Usage: wave_timeline.py NX NY [K] [ROWS] [--tissue U] -> one JSON line (and a table on stderr)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fdtd_amd
from fdtd_amd import binding as B

args = [a for a in sys.argv[1:] if not a.startswith("--")]
nx, ny = int(args[0]), int(args[1])
K = int(args[2]) if len(args) > 2 else 0
if len(args) > 3 and int(args[3]) > 0:
    os.environ["FDTD_TB_ROWS"] = args[3]
tissue = int(sys.argv[sys.argv.index("--tissue") + 1]) if "--tissue" in sys.argv else 0
if tissue:
    nx, ny = 265 * tissue, 490 * tissue
src = [(40 * tissue, 245 * tissue, 0)] if tissue else [(nx // 2, ny // 2, 0)]
s = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, 1, src), ecoef=B.ECOEF_CONSISTENT, kernel=B.KERNEL_MULTISTEP, steps_per_pass=K)
if tissue:
    data = os.path.join(ROOT, "data")
    s.load_materials(os.path.join(data, "mat_specs_riken"), os.path.join(data, "pgmdata"), first_slice=800, upsample=tissue)
K = s.steps_per_pass
s.step(4 * K)
waves, iters = s.sweep_work
s.profile_begin(1)
s.profile_waves_begin(waves + 64)
s.step(K)
ms, _ = s.profile_read_launches(1)
t = s.profile_waves_read(waves + 64)
t = t[t[:, 1] > 0]
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
t0 = start.min()
start, end = (start - t0) * 1e-2, (end - t0) * 1e-2          # microseconds
life = end - start
flags = (t[:, 3] & np.uint64(3)).astype(int)
band = (t[:, 2] >> np.uint64(32)).astype(int)
strip = (t[:, 2] & np.uint64(0xffffffff)).astype(int)
xcc = ((t[:, 3] >> np.uint64(8)) & np.uint64(15)).astype(int)
kinds = {3: "plain (no guards)", 2: "x-face strip (y guards folded)", 0: "fully guarded", 1: "fully guarded"}
out = {"grid": [nx, ny], "steps_per_pass": K, "rows_per_wave": int(os.environ.get("FDTD_TB_ROWS", 0)) or None,
       "waves": int(len(t)), "launch_ms_hip_events": float(ms[0]), "span_us": float(end.max()), "kinds": {}}
for f in (3, 2, 0):
    m = (flags == f) | ((f == 0) & (flags == 1))
    if not m.any():
        continue
    out["kinds"][kinds[f]] = {"waves": int(m.sum()), "life_us_median": float(np.median(life[m])), "life_us_max": float(life[m].max()),
                              "life_us_min": float(life[m].min()), "last_end_us": float(end[m].max()), "first_start_us": float(start[m].min())}
# waves in flight over time
grid_t = np.linspace(0, end.max(), 41)
inflight = [int(((start <= x) & (end > x)).sum()) for x in grid_t]
out["in_flight"] = {"t_us": [float(x) for x in grid_t], "waves": inflight}
out["start_of_last_wave_us"] = float(start.max())
out["waves_per_xcc"] = [int((xcc == i).sum()) for i in range(8)]
# the launch if it ended with the last plain wave
plain = flags == 3
if plain.any():
    out["end_of_last_plain_wave_us"] = float(end[plain].max())
print(json.dumps(out))
print(f"{nx}x{ny} K={K}: {len(t)} waves, launch {ms[0] * 1e3:.1f} us (span of waves {end.max():.1f} us)", file=sys.stderr)
for k, v in out["kinds"].items():
    print(f"  {k:34s} {v['waves']:6d} waves  life median {v['life_us_median']:8.1f} us  max {v['life_us_max']:8.1f}  "
          f"first start {v['first_start_us']:7.1f}  last end {v['last_end_us']:8.1f}", file=sys.stderr)
print("  in flight: " + " ".join(str(x) for x in inflight), file=sys.stderr)
s.close()
