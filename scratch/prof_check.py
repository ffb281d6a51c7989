import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import fdtd_amd
from fdtd_amd import binding as B
for n in (512, 4096, 16384):
    p = fdtd_amd.make_params(n, n, 1, [(n // 2, n // 2, 0)])
    s = fdtd_amd.Solver(p, ecoef=B.ECOEF_CONSISTENT)
    s.set_probes([[n // 2, n // 2, 0]], 64)
    s.step(12)
    pr = s.read_probes(0, 12)
    ez = s.field(B.EZ)
    snap = np.empty_like(ez); s.snapshot_async(B.EZ, snap); s.snapshot_wait()
    print(n, "probe ez", pr[-1, 0, 3], "field max", np.abs(ez).max(), "at src", ez[0, n // 2, n // 2], "snapshot max", np.abs(snap).max(), flush=True)
    s.close()
