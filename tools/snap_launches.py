"""Per-launch sweep times around snapshots (16384^2, multi-step sweep): which launches a snapshot slows down."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import fdtd_amd
from fdtd_amd import binding as B

every = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 16384
s = fdtd_amd.Solver(fdtd_amd.make_params(n, n, 1, [(n // 2, n // 2, 0)]), ecoef=B.ECOEF_CONSISTENT, kernel=B.KERNEL_MULTISTEP)
s.step(16)
for _ in range(2):
    s.snapshot_async(B.EZ)
s.snapshot_wait()
s.profile_begin(400)
s.sync()
t0 = time.perf_counter()
marks = []
for i in range(400 // every):
    s.step(every)
    ta = time.perf_counter()
    s.snapshot_async(B.EZ)
    marks.append((time.perf_counter() - ta) * 1e3)
s.sync()
t1 = time.perf_counter()
s.snapshot_wait()
t2 = time.perf_counter()
ms, k = s.profile_read_launches(400)
print(f"every {every}: steps {1e3*(t1-t0):.1f} ms, tail {1e3*(t2-t1):.1f} ms; host time in snapshot_async: {[round(m, 2) for m in marks]}")
print(" ".join(f"{a:.2f}/{b}" for a, b in zip(ms, k)))
s.close()
