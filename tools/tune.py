"""Parameter sweep of the fused 2-D sweep (prints sweep ms / GB/s)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import fdtd_amd
from fdtd_amd import binding as B
DATA = os.path.join(os.path.dirname(__file__), '..', 'data')

def run(nx, ny, chunk, steps=40, warm=5, skew=0, variant=0, pgm=0, band=0, nopad=0):
    os.environ["FDTD_PITCH_SKEW"] = str(skew)
    os.environ["FDTD_SWEEP_VARIANT"] = str(variant)
    os.environ["FDTD_SWEEP_BAND"] = str(band)
    os.environ["FDTD_SWEEP_NOPAD"] = str(nopad)
    p = fdtd_amd.make_params(nx, ny, 1, [(nx // 8, ny // 2, 0)])
    s = fdtd_amd.Solver(p, ecoef=B.ECOEF_CONSISTENT, chunk_rows=chunk)
    if pgm:
        s.load_materials(os.path.join(DATA, "mat_specs_riken"), os.path.join(DATA, "pgmdata"), first_slice=800, upsample=pgm)
    s.step(warm); s.sync()
    s.profile_begin(steps)
    t0 = time.perf_counter(); s.step(steps); s.sync(); el = time.perf_counter() - t0
    ms = s.profile_read(steps)
    ez = s.field(B.EZ) if nx * ny < 2**27 else None
    ok = "" if ez is None else f" max|ez| {np.abs(ez).max():.3e} finite {np.isfinite(ez).all()}"
    s.close()
    return float(np.median(ms)), float(np.mean(ms)), el / steps * 1e3, ok

if __name__ == "__main__":
    variants = [int(v) for v in sys.argv[1].split(",")]
    chunks = [int(v) for v in sys.argv[2].split(",")]
    cfgs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["vac"]
    for cfg in cfgs:
        if cfg == "vac":
            nx = ny = 16384; pgm = 0
        elif cfg == "pgm":
            nx, ny, pgm = 265 * 32, 490 * 32, 32
        elif cfg == "vac2":
            nx, ny, pgm = 265 * 32, 490 * 32, 0
        elif cfg == "wide":
            nx, ny, pgm = 65536, 4096, 0
        cells = nx * ny
        bands = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
        for variant in variants:
          for band in bands:
            for chunk in chunks:
                med, mean, wall, ok = run(nx, ny, chunk, variant=variant, pgm=pgm, band=max(band, 0), nopad=1 if band < 0 else 0)
                print(f"{cfg} {nx}x{ny} variant {variant} band {band} chunk {chunk:3d}: sweep median {med:.3f} ms mean {mean:.3f} wall/step {wall:.3f} ms -> {cells/med/1e6:.1f} Gcell/s, {cells * 41 / med / 1e6:.0f} GB/s algorithmic ({cells*41/med/1e6/80:.1f}% of 8 TB/s){ok}", flush=True)
