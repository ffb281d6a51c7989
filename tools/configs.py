"""Step time of the 2-D kernels on the tissue-map workload (v1_00800.pgm upsampled x32, Debye + Mur)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fdtd_amd
from fdtd_amd import binding as B
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'data')

def run(nx, ny, kernel, pgm=0, steps=168, src=None):
    p = fdtd_amd.make_params(nx, ny, 1, [src or (nx // 8, ny // 2, 0)])
    s = fdtd_amd.Solver(p, ecoef=B.ECOEF_CONSISTENT, kernel=kernel)
    if pgm:
        s.load_materials(os.path.join(DATA, "mat_specs_riken"), os.path.join(DATA, "pgmdata"), first_slice=800, upsample=pgm)
    s.step(2 * 840 // 10); s.sync()
    t0 = time.perf_counter(); s.step(steps); s.sync(); el = time.perf_counter() - t0
    print(f"{nx}x{ny} pgm={pgm} kernel={kernel} {s.sweep_kernel_name}: {el/steps*1e3:.4f} ms/step {nx*ny*steps/el/1e9:.1f} Gcell/s", flush=True)
    s.close()

kernels = [B.KERNEL_FUSED] + [(B.KERNEL_MULTISTEP, k) for k in (int(v) for v in (sys.argv[1:] or ["4", "5", "6", "7"]))]
for k in kernels:
    run(8480, 15680, k, pgm=32, steps=840)
print("same grid, vacuum:")
for k in kernels[1:]:
    run(8480, 15680, k, pgm=0, steps=840)
