import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
mode = sys.argv[1]
if mode != "notorch":
    import torch
    if mode == "torchcuda":
        torch.cuda.set_device(0); torch.cuda.synchronize()
import fdtd_amd
from fdtd_amd import binding as B
os.system(f"grep -E 'libamdhip64|libhsa-runtime' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
n = 4096
p = fdtd_amd.make_params(n, n, 1, [(n // 2, n // 2, 0)])
s = fdtd_amd.Solver(p, ecoef=B.ECOEF_CONSISTENT)
s.set_probes([[n // 2, n // 2, 0]], 64)
s.step(2)
s.profile_begin(10)
s.step(10)
s.sync()
ms = s.profile_read(10)
pr = s.read_probes(0, 12)
ez = s.field(B.EZ)
print(mode, "sweep ms", ms[:3], "probe ez", pr[-1, 0, 3], "field max", np.abs(ez).max(), "at src", ez[0, n // 2, n // 2], flush=True)
s.close()
