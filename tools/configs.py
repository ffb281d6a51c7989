import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.getcwd())
import numpy as np, fdtd_amd
from fdtd_amd import binding as B
DATA = os.path.join(os.getcwd(), 'data')
def run(nx, ny, kernel, pgm=0, steps=120, src=None):
    p = fdtd_amd.make_params(nx, ny, 1, [src or (nx // 8, ny // 2, 0)])
    s = fdtd_amd.Solver(p, ecoef=B.ECOEF_CONSISTENT, kernel=kernel)
    if pgm:
        s.load_materials(os.path.join(DATA, "mat_specs_riken"), os.path.join(DATA, "pgmdata"), first_slice=800, upsample=pgm)
    s.step(12); s.sync()
    t0 = time.perf_counter(); s.step(steps); s.sync(); el = time.perf_counter() - t0
    print(f"{nx}x{ny} pgm={pgm} kernel={kernel} {s.sweep_kernel_name}: {el/steps*1e3:.4f} ms/step {nx*ny*steps/el/1e9:.1f} Gcell/s", flush=True)
    s.close()
for k in (B.KERNEL_FUSED, (B.KERNEL_MULTISTEP, 4), (B.KERNEL_MULTISTEP, 5)):
    run(8480, 15680, k, pgm=32)
    run(265, 490, k, pgm=1, steps=1200, src=(40, 245, 0))
