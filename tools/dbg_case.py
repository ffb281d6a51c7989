#!/usr/bin/env python3
"""Debug helper: one small 2-D case, multi-step kernel against the oracle; prints where fields differ."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fdtd_amd
from fdtd_amd import binding as B
from oracle import oracle
import util
nx, ny, K, steps = [int(v) for v in sys.argv[1:5]]
src = [(int(a), int(b), 0) for a, b in (p.split(",") for p in sys.argv[5:])] or [(nx // 2, ny // 2, 0)]
s, o = util.pair(oracle, (nx, ny, 1), src, B.ECOEF_CONSISTENT, kernel=(B.KERNEL_MULTISTEP, K))
s.step(steps); o.step(steps)
for fid in (B.HX, B.HY, B.DZ, B.EZ):
    got, want = s.field(fid), np.asarray(o.field(fid))
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    idx = np.argwhere(bad.reshape(ny, nx))
    print(B.FIELD_NAMES[fid], "mismatches", len(idx), "nan", int(np.isnan(got).sum()), "first", idx[:6].tolist(), "last", idx[-3:].tolist())
    if len(idx):
        ys = np.unique(idx[:, 0]); xs = np.unique(idx[:, 1])
        print("   rows", ys[:20].tolist(), "cols", xs[:20].tolist())
