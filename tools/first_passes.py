"""Duration of the first passes of the multi-step sweep after a context is created (16384^2 vacuum),
pass by pass: how long start-up effects last.  Arguments: the steps of the calls to make, e.g.
"5 8 8 4" (a 5-step warm-up pass, two full passes, a remainder pass)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import fdtd_amd
from fdtd_amd import binding as B

n = 16384
calls = [int(v) for v in sys.argv[1:]] or [160]
s = fdtd_amd.Solver(fdtd_amd.make_params(n, n, 1, [(n // 2, n // 2, 0)]), ecoef=B.ECOEF_CONSISTENT, kernel=(B.KERNEL_MULTISTEP, 8))
s.sync()
s.profile_begin(sum(calls) + 8)
for c in calls:
    s.step(c)
s.sync()
ms, steps = s.profile_read_launches(sum(calls))
print("calls", calls, "-> per launch (steps: ms):", " ".join("%d: %.3f" % (k, m) for m, k in zip(ms, steps)), flush=True)
s.close()
