"""RCCL through the library's own communicator in a process that has torch (and its bundled RCCL)
loaded and a gloo process group up -- the situation of bench.py --gpus N.  One rank: the packed
deep-halo block goes to this very rank through ncclSend/ncclRecv."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd
from fdtd_amd import binding as B

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
torch.zeros(4, device="cuda").sum().item()
m = fdtd_amd.Solver(fdtd_amd.make_params(4096, 512, 1, [(2048, 3, 0)]), ecoef=B.ECOEF_CONSISTENT, slab=(0, 512),
                    kernel=(B.KERNEL_MULTISTEP, 8))
m.step(16)
m.comm_init(B.comm_unique_id(), 0, 1)
m.comm_selftest()
m.step(16)
m.sync()
print("rccl ok with torch", torch.__version__, "loaded; ez max", float(np.abs(m.field(B.EZ)).max()))
dist.destroy_process_group()
