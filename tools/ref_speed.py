"""The reference's own kernels (oracle/_ref, see oracle/ref_driver.hip) timed on this GPU beside the
product's 3-D sweep, same grids, vacuum."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fdtd_amd
from fdtd_amd import binding as B
from oracle import oracle as orc, reference
for dims in ((100, 100, 100), (512, 512, 512), (265, 490, 1687)):
    po = orc.make_params(*dims, [(dims[0] // 2, dims[1] // 2, dims[2] // 2)])
    ms_ref = reference.time_per_step(po, 3, 12 if dims[0] > 100 else 300)
    s = fdtd_amd.Solver(fdtd_amd.make_params(*dims, [(dims[0] // 2, dims[1] // 2, dims[2] // 2)]), ecoef=B.ECOEF_CONSISTENT)
    n = 40 if dims[0] > 100 else 1000
    s.step(5); s.sync()
    t0 = time.perf_counter(); s.step(n); s.sync(); ms = (time.perf_counter() - t0) / n * 1e3
    s.close()
    cells = dims[0] * dims[1] * dims[2]
    print(f"{dims}: reference kernels {ms_ref:.4f} ms/step ({cells / ms_ref / 1e6:.1f} Gcell-updates/s), "
          f"this library {ms:.4f} ms/step ({cells / ms / 1e6:.1f} Gcell-updates/s), x{ms_ref / ms:.1f}", flush=True)
