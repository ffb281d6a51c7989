"""Writes tests/golden/reference_host_vectors.json: outputs of the reference's own host functions
(oracle/_ref/libfdtd_ref_host.so: initParamsWithPath, setupSources, loadMaterials, setupMurBoundary
of fdtd.cu compiled as C, see oracle/ref_host_driver.c) on the inputs tests/test_reference_host.py
uses.  Runs only where the reference tree exists (`make -C oracle ref_host` first); the JSON file is
the part that travels.  Floats are recorded as their 32-bit patterns, long arrays as SHA-256 of
their bytes.

    python tests/golden/make_reference_host_vectors.py
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import reference_host as RH  # noqa: E402
import test_reference_host as T  # noqa: E402


def u32(a):
    return [int(v) for v in np.ascontiguousarray(a, dtype=np.float32).ravel().view(np.uint32)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def main():
    out = {"_comment": "recorded from the reference's host functions by tests/golden/make_reference_host_vectors.py",
           "params": [], "materials": [], "mur_defaults": []}
    with tempfile.TemporaryDirectory() as tmp:
        from pathlib import Path
        for changes in T.VARIANTS:
            ref = RH.RefParams(T.write_params(Path(tmp), **changes))
            jz = ref.setup_sources()
            nz = int(np.flatnonzero(jz)[-1]) + 1
            out["params"].append({
                "changes": changes,
                "ints": {n: getattr(ref, n) for n in RH.INT_NAMES},
                "floats_u32": {n: u32(getattr(ref, n))[0] for n in RH.FLOAT_NAMES},
                "sources": [int(v) for v in ref.sources],
                "jz_nonzero_prefix_u32": u32(jz[:nz]),
                "jz_sha256": sha(jz)})
            ref.close()
        # loadMaterials on the shipped slice, then on random slices over every code + setupMurBoundary
        for dims in [(265, 490, 3), (100, 120, 4)]:
            d = Path(tmp) / f"pgm_{dims[0]}"
            d.mkdir()
            for iz in range(dims[2]):
                os.symlink(os.path.join(T.DATA, "pgmdata", "v1_00800.pgm"), d / f"v1_{iz + 1:05d}.pgm")
            ref = RH.RefParams(T.SHIPPED)
            ref.set(*dims, 0.0, 1.0, 1.0, 0.0)
            mats = ref.load_materials(T.SPECS, str(d))
            out["materials"].append({"dims": dims, "source": "data/pgmdata/v1_00800.pgm as every slice",
                                     "sha256": [sha(m) for m in mats]})
            ref.close()
        nx, ny, nz = 23, 19, 7
        from fdtd_amd import binding
        _, present = binding.read_material_specs(T.SPECS)
        codes = np.random.default_rng(42).choice(np.flatnonzero(present), size=(nz, ny, nx))
        d = Path(tmp) / "pgm_random"
        d.mkdir()
        for iz in range(nz):
            T.write_pgm(d / f"v1_{iz + 1:05d}.pgm", nx, ny, codes[iz])
        ref = RH.RefParams(T.SHIPPED)
        ref.set(nx, ny, nz, 0.0, 1.0, 1.0, 0.0)
        mats = ref.load_materials(T.SPECS, str(d))
        planes = ref.setup_mur(*mats)
        out["materials"].append({"dims": (nx, ny, nz), "source": "default_rng(42).choice(codes of mat_specs_riken)",
                                 "sha256": [sha(m) for m in mats], "rp_sha256": [sha(p) for p in planes],
                                 "rp_first8_u32": [u32(p[:8]) for p in planes]})
        ref.close()
        for mat in [(0.0, 1.0, 1.0, 0.0), (0.9, 52.7, 4.0, 8.4e-12), (5.0, 80.0, 4.0, 9.0e-12), (1e-3, 3.0, 1.5, 2e-10)]:
            ref = RH.RefParams(T.SHIPPED)
            ref.set(9, 8, 7, *mat)
            planes = ref.setup_mur(*[np.full((7, 8, 9), v, dtype=np.float32) for v in mat])
            assert all(np.all(p == p[0]) for p in planes)
            out["mur_defaults"].append({"material": mat, "rp_u32": [u32(p[0])[0] for p in planes]})
            ref.close()
    with open(os.path.join(HERE, "reference_host_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
