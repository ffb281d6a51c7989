"""Step time of the 3-D sweep on the full body model (265 x 490 x 1687, mat_specs_riken, the codes of all
1687 slices from tests/golden/body_codes_265x490x1687.u8.xz) and on 512^3 vacuum."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import fdtd_amd
from fdtd_amd import binding as B
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'data')

def run(dims, body, steps=60):
    nx, ny, nz = dims
    s = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, nz, [(nx // 8, ny // 2, nz // 2)]), ecoef=B.ECOEF_CONSISTENT)
    if body:
        table, _ = B.read_material_specs(os.path.join(DATA, "mat_specs_riken"))
        for code, row in enumerate(table):
            s.set_material(code, *[float(v) for v in row])
        import lzma
        path = os.path.join(DATA, "..", "tests", "golden", "body_codes_265x490x1687.u8.xz")
        s.set_codes(np.frombuffer(lzma.decompress(open(path, "rb").read()), dtype=np.uint8).reshape(nz, ny, nx))
        s.setup_mur_from_materials()
    s.step(10); s.sync()
    s.profile_begin(steps)
    t0 = time.perf_counter(); s.step(steps); s.sync(); el = time.perf_counter() - t0
    ms = float(np.mean(s.profile_read(steps)))
    cells = nx * ny * nz
    print(f"{dims} body={body}: {el / steps * 1e3:.3f} ms/step, sweep {ms:.3f} ms, {cells * steps / el / 1e9:.1f} Gcell/s, "
          f"{cells * 97 / ms / 1e9:.2f} TB/s algorithmic", flush=True)
    s.close()

run((265, 490, 1687), True)
run((265, 490, 1687), False)
run((512, 512, 512), False)
