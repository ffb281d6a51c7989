"""Two slabs of a 16384-wide grid on ONE GPU, each driven by its own thread, halos exchanged by the
library over the in-process transport: the same schedule as the RCCL run (rows next to the
neighbour first on the second stream, pack -> send/recv -> unpack there, interior on the main
stream).  Run under `rocprofv3 --kernel-trace`; tools/overlap_report.py then checks in the trace
that the pack/unpack kernels ran inside the interval of the interior sweep.
Usage: overlap_trace.py [rows per slab] [passes] [steps per pass, 1 = one-step sweep]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fdtd_amd
from fdtd_amd import binding as B

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nx, world = 16384, 2
p = fdtd_amd.make_params(nx, rows * world, 1, [(nx // 2, rows, 0)])
group = B.LocalGroup(world)
kernel = (B.KERNEL_MULTISTEP, spp) if spp > 1 else B.KERNEL_FUSED
slabs = [fdtd_amd.Solver(p, ecoef=B.ECOEF_CONSISTENT, kernel=kernel, slab=(r * rows, rows)) for r in range(world)]
for r, s in enumerate(slabs):
    s.comm_init_local(group.handle, r, world)


def run(r):
    slabs[r].step(spp)                 # first pass: allocates the halo buffers, primes the halos
    slabs[r].sync()
    t0 = time.perf_counter()
    slabs[r].step(spp * passes)
    slabs[r].sync()
    print(f"rank {r}: {(time.perf_counter() - t0) / passes * 1e3:.3f} ms per pass of {spp} steps "
          f"(two slabs of {nx}x{rows} share the GPU)", flush=True)


threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join()
for s in slabs:
    s.close()
group.close()
