"""Does a 3-D grid that fits into the 256 MB memory-side cache sweep faster per cell?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import fdtd_amd
from fdtd_amd import binding as B

for nx, ny in ((512, 512), (1024, 1024)):
    for nz in (3, 4, 6, 8, 12, 16, 24, 32, 64, 128):
        s = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, nz, [(nx // 2, ny // 2, nz // 2)]), ecoef=B.ECOEF_CONSISTENT)
        s.step(20); s.sync()
        s.profile_begin(60); s.step(60); ms = s.profile_read(60)
        cells = nx * ny * nz
        t = float(np.median(ms))
        print(f"{nx}x{ny}x{nz:4d}: {cells * 73 / 1e6:8.1f} MB resident, sweep {t * 1e3:8.1f} us, "
              f"{cells / t / 1e6:6.1f} Gcell/s, {cells * 97 / t / 1e9:6.2f} TB/s algorithmic", flush=True)
        s.close()
