"""Where do 16 probes cost time in the multi-step sweep?  16384^2, 400 steps: no probes, probes in the first
band, on the diagonal, in the last band."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fdtd_amd
from fdtd_amd import binding as B
n = 16384
def run(name, probes):
    s = fdtd_amd.Solver(fdtd_amd.make_params(n, n, 1, [(n // 2, n // 2, 0)]), ecoef=B.ECOEF_CONSISTENT, kernel=(B.KERNEL_MULTISTEP, 8))
    if probes:
        s.set_probes(probes, capacity=512)
    s.step(16); s.sync()
    t0 = time.perf_counter(); s.step(400); s.sync(); el = time.perf_counter() - t0
    print(f"{name:28s} {n * n * 400 / el / 1e9:7.1f} Gcell-updates/s", flush=True)
    s.close()
xs = [int(n * (i + 0.5) / 16) for i in range(16)]
run("no probes", None)
run("16 probes, first band", [(x, 40, 0) for x in xs])
run("16 probes, diagonal", [(x, x, 0) for x in xs])
run("16 probes, last band", [(x, n - 20, 0) for x in xs])
run("16 probes, one wave", [(4000 + i, 8000, 0) for i in range(16)])
