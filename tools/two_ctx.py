"""Do the ramp and the tail of a pass overlap with another stream's work?  N independent contexts of
16384 x (16384 / N) cells each step concurrently from N threads; the aggregate is compared with one
context of 16384 x 16384."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fdtd_amd
from fdtd_amd import binding as B

def make(nx, ny):
    s = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, 1, [(nx // 2, ny // 2, 0)]), ecoef=B.ECOEF_CONSISTENT,
                        kernel=(B.KERNEL_MULTISTEP, 8))
    s.step(16); s.sync()
    return s

steps = 400
for n in (1, 2, 3, 4):
    ny = 16384 // n // 96 * 96 if n > 1 else 16384
    ctx = [make(16384, ny) for _ in range(n)]
    def run(s):
        s.step(steps); s.sync()
    th = [threading.Thread(target=run, args=(s,)) for s in ctx]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print(f"{n} context(s) of 16384 x {ny}: {16384 * ny * n * steps / el / 1e9:.1f} Gcell-updates/s aggregate", flush=True)
    for s in ctx: s.close()
