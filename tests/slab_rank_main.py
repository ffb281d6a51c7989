"""One rank of a two-process slab run on one GPU (started by test_parity_gpu.py):
HIP engine per slab, halo rows staged through the host and exchanged over gloo."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch.distributed as dist

import fdtd_amd
from fdtd_amd import binding as B

rank, world, port, nx, ny, steps = (int(v) for v in sys.argv[1:7])
out_dir = sys.argv[7]
spp = int(sys.argv[8]) if len(sys.argv) > 8 else 0     # > 1: the multi-step engine with that many steps per pass
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("gloo", rank=rank, world_size=world)
p = fdtd_amd.make_params(nx, ny, 1, [(nx // 2, ny // 2, 0), (10, ny // 2 - 1, 0)])
part = fdtd_amd.partition_rows(ny, world)[rank]
solver = fdtd_amd.Solver(p, ecoef=B.ECOEF_CONSISTENT, slab=part,
                         kernel=(B.KERNEL_MULTISTEP, spp) if spp > 1 else B.KERNEL_AUTO)
runner = fdtd_amd.SlabRunner(solver, rank, world, transport="host")
runner.step(steps)
for fid in (B.HX, B.HY, B.DZ, B.EZ):
    np.save(os.path.join(out_dir, f"f{fid}_r{rank}.npy"), solver.field(fid)[0])
dist.barrier()
dist.destroy_process_group()
