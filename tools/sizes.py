"""Step time of the one-step and the multi-step 2-D sweeps over grid sizes (vacuum)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fdtd_amd
from fdtd_amd import binding as B

def run(n, kernel, steps):
    s = fdtd_amd.Solver(fdtd_amd.make_params(n, n, 1, [(n // 2, n // 2, 0)]), ecoef=B.ECOEF_CONSISTENT, kernel=kernel)
    s.step(24); s.sync()
    t0 = time.perf_counter(); s.step(steps); s.sync(); el = time.perf_counter() - t0
    s.close()
    return el / steps * 1e6

for n in (256, 512, 1024, 2048, 4096, 8192):
    steps = 2400 if n <= 1024 else 480
    row = [run(n, k, steps) for k in (B.KERNEL_FUSED, (B.KERNEL_MULTISTEP, 2), (B.KERNEL_MULTISTEP, 4))]
    print(f"{n:5d}^2  one-step {row[0]:9.2f} us/step   x2 {row[1]:9.2f}   x4 {row[2]:9.2f}", flush=True)
