"""Slab decomposition across ranks: one process per GPU, one slab per process.

The grid is cut along its slowest axis (y in 2-D, z in 3-D).  A slab keeps one
halo row (plane) of E on either side; after every step each rank sends the
first and the last row of its newest E level to the neighbours and receives
theirs into its halo.  (The multi-step 2-D engine keeps halos as deep as its
steps per pass, of every array, and exchanges once per pass.)  H needs no exchange: the one H row below the slab that
the D update reads is recomputed locally from the E halo every step.  The Mur
faces across the slab axis belong to the first and last rank, sources and
probes to the rank that holds their row -- the kernels decide that from global
coordinates, so a slab engine only has to be told where it sits.

Two transports:
  * "rccl": the library exchanges halos itself with ncclSend/ncclRecv on its
    stream (fdtd_comm_init); `step(n)` is one library call.
  * "host": halo rows staged through host memory and moved with
    torch.distributed point-to-point calls of whatever backend the process
    group has (gloo on CPUs); one exchange per step, driven from here.
The engine is any object with step(n), halo_get(side) and halo_put(side, rows);
fdtd_amd.Solver is the GPU engine.
"""
import numpy as np


def partition_rows(n_rows, world):
    """[(begin, count)] per rank, as even as possible, every slab >= 2 rows."""
    if world < 1 or n_rows < 2 * world:
        raise ValueError(f"cannot cut {n_rows} rows into {world} slabs of at least 2 rows")
    base, extra = divmod(n_rows, world)
    out, begin = [], 0
    for r in range(world):
        count = base + (1 if r < extra else 0)
        out.append((begin, count))
        begin += count
    return out


class SlabRunner:
    def __init__(self, engine, rank, world, transport="host", group=None):
        if transport not in ("host", "rccl"):
            raise ValueError("transport must be 'host' or 'rccl'")
        self.engine, self.rank, self.world = engine, rank, world
        self.transport, self.group = transport, group
        # An engine that advances several steps per pass over memory keeps halos that deep (every
        # live array, material codes included) and recomputes the rows in them: it exchanges once
        # per pass, and once before the first pass so that the halos hold the neighbours' state.
        self.primed = False

    @property
    def stride(self):
        """Steps between two exchanges, asked of the engine every time (it may change when
        materials are loaded)."""
        return int(getattr(self.engine, "steps_per_pass", 1))

    def invalidate(self):
        """Call after anything that changes the engine's state behind the stepper's back (fields,
        material codes, a checkpoint): the next step() exchanges first."""
        self.primed = False

    def exchange(self):
        """Host-staged halo exchange with ranks rank-1 and rank+1."""
        import torch
        import torch.distributed as dist
        ops, recvs = [], []
        for side, peer in ((0, self.rank - 1), (1, self.rank + 1)):
            if peer < 0 or peer >= self.world:
                continue
            out = torch.from_numpy(np.ascontiguousarray(self.engine.halo_get(side)))
            inc = torch.empty_like(out)
            ops.append(dist.P2POp(dist.isend, out, peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, inc, peer, group=self.group))
            recvs.append((side, inc))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for side, inc in recvs:
            self.engine.halo_put(side, inc.numpy())

    def step(self, nsteps=1):
        if self.world == 1 or self.transport == "rccl":
            self.engine.step(nsteps)
            return
        if not self.primed:
            self.exchange()
            self.primed = True
        done = 0
        while done < nsteps:
            n = min(self.stride, nsteps - done)
            self.engine.step(n)
            self.exchange()
            done += n
