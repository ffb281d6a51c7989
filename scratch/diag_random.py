import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import fdtd_amd
from fdtd_amd import binding as B
from oracle import oracle
import util
nx, ny = 265, 49
for variant in ["full", "no_rp", "no_codes", "no_rand"]:
    rng = np.random.default_rng(99 + nx)
    s, o = util.pair(oracle, (nx, ny, 1), [(nx // 2, ny // 2, 0), (nx - 1, ny - 1, 0)], B.ECOEF_CONSISTENT, kernel=B.KERNEL_SPLIT,
                     sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(1, ny, nx), dtype=np.uint8)
    if variant != "no_codes":
        util.set_materials(s, o, codes)
    for face in (0, 1, 3, 4):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        if variant != "no_rp":
            s.set_rp(face, rp); o.set_rp(face, rp)
    if variant != "no_rand":
        util.randomize(s, o, rng, oracle)
    for step in range(1, 4):
        s.step(1); o.step(1)
        for fid in util.fields_of(s):
            g, w = s.field(fid)[0], np.asarray(o.field(fid))[0]
            bad = np.argwhere(g.view(np.uint32) != w.view(np.uint32))
            if len(bad):
                ulp = np.abs(g.view(np.int32).astype(np.int64) - w.view(np.int32).astype(np.int64))
                print(variant, "step", step, B.FIELD_NAMES[fid], "nbad", len(bad), "max ulp", ulp.max(),
                      "first", [(int(y), int(x), float(g[y, x]), float(w[y, x])) for y, x in bad[:6]],
                      "x range", bad[:, 1].min(), bad[:, 1].max(), "y range", bad[:, 0].min(), bad[:, 0].max())
    print(variant, "done")
