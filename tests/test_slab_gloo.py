"""The N > 1 path on CPUs: world_size-2 and -3 process groups (gloo) run the slab
protocol of fdtd_amd.slab -- partition_rows, SlabRunner, one ez row per
neighbour per step, the hx halo row recomputed locally -- with the CPU oracle as
the per-slab engine, and must reproduce the single-slab oracle bit for bit.  The
deep-halo form of the protocol (several steps per exchange, what the multi-step
GPU engine does) is restated with the oracle as well.
The GPU engine (fdtd_amd.Solver) implements the same three calls; its slab
tests are in test_parity_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleSlabEngine:
    """step / halo_get / halo_put over an oracle slab (2-D TMz)."""

    def __init__(self, orc, params, ecoef, slab, codes=None, table=None):
        self.orc = orc
        self.o = orc.Oracle(params, ecoef=ecoef, slab=slab)
        self.nyl = slab[1]
        if codes is not None:
            self.o.set_materials(codes[slab[0]:slab[0] + slab[1]][None], table)

    def step(self, n):
        self.o.step(n)

    def halo_get(self, side):
        ez = self.o.field_storage(self.orc.EZ)[0]          # rows: halo, owned..., halo
        return np.array(ez[1] if side == 0 else ez[self.nyl])

    def halo_put(self, side, rows):
        ez = self.o.field_storage(self.orc.EZ)[0]
        ez[0 if side == 0 else self.nyl + 1] = rows


class OracleDeepSlabEngine:
    """The deep-halo protocol of the multi-step GPU engine, restated with the oracle: the slab also
    holds `depth` rows of each neighbour, advances `depth` steps without talking to anybody (the
    borrowed rows go stale from the outside in, one row per step, and never reach an owned row), and
    then takes the neighbours' boundary rows of every array -- H, both levels of ez and dz."""
    ARRAYS = None

    def __init__(self, orc, params, ecoef, slab, depth, codes=None, table=None):
        b, n = slab
        self.steps_per_pass = depth
        self.depth, self.n = depth, n
        self.lo = min(depth, b)                              # neighbour rows held below / above
        self.hi = min(depth, params.ny - (b + n))
        self.o = orc.Oracle(params, ecoef=ecoef, slab=(b - self.lo, n + self.lo + self.hi))
        if codes is not None:
            self.o.set_materials(codes[b - self.lo:b + n + self.hi][None], table)
        self.ARRAYS = [(orc.HX, 0), (orc.HY, 0), (orc.EZ, 0), (orc.EZ, 1), (orc.DZ, 0), (orc.DZ, 1)]

    def step(self, n):
        self.o.step(n)

    def field(self, fid):
        return np.array(self.o.field(fid)[0][self.lo:self.lo + self.n])

    def halo_get(self, side):
        first = self.lo if side == 0 else self.lo + self.n - self.depth       # my first / last `depth` rows
        return np.stack([self.o.field(f, l)[0][first:first + self.depth] for f, l in self.ARRAYS])

    def halo_put(self, side, rows):
        have = self.lo if side == 0 else self.hi
        assert have == self.depth
        first = 0 if side == 0 else self.lo + self.n
        for (f, l), block in zip(self.ARRAYS, rows):
            self.o.field(f, l)[0][first:first + self.depth] = block


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nx, ny, steps, out_dir, depth=0):
    import sys
    sys.path.insert(0, ROOT)
    import fdtd_amd.slab as slab
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        table = np.array([[0, 1, 1, 0], [0.6, 56.4, 33.1, 3.5e-11], [0.1, 14.2, 7.4, 3.4e-11]], dtype=np.float32)
        codes = rng.integers(0, 3, size=(ny, nx), dtype=np.uint8)
        srcs = [(nx // 2, ny // 2, 0), (3, ny // world, 0), (nx - 2, ny // world - 1, 0)]
        p = orc.make_params(nx, ny, 1, srcs, sigma=0.05, eps_s=3.0, eps_i=2.0, tau_d=1e-11)
        part = slab.partition_rows(ny, world)[rank]
        if depth:
            eng = OracleDeepSlabEngine(orc, p, orc.ECOEF_CONSISTENT, part, depth, codes, table)
        else:
            eng = OracleSlabEngine(orc, p, orc.ECOEF_CONSISTENT, part, codes, table)
        runner = slab.SlabRunner(eng, rank, world, transport="host")
        runner.step(steps)
        for fid in (orc.HX, orc.HY, orc.DZ, orc.EZ):
            got = eng.field(fid) if depth else np.array(eng.o.field(fid)[0])
            np.save(os.path.join(out_dir, f"f{fid}_r{rank}.npy"), got)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,depth", [(2, 0), (3, 0), (2, 4), (3, 3)])
def test_slabs_over_gloo_equal_single_slab(oracle, tmp_path, world, depth):
    """depth 0: one ez row per neighbour per step.  depth K: the multi-step protocol, K rows of
    every array per neighbour once per K steps (90 steps = passes plus single steps at the end)."""
    nx, ny, steps = 61, 47, 90
    mp.spawn(_worker, args=(world, _free_port(), nx, ny, steps, str(tmp_path), depth), nprocs=world, join=True)

    rng = np.random.default_rng(3)
    table = np.array([[0, 1, 1, 0], [0.6, 56.4, 33.1, 3.5e-11], [0.1, 14.2, 7.4, 3.4e-11]], dtype=np.float32)
    codes = rng.integers(0, 3, size=(ny, nx), dtype=np.uint8)
    srcs = [(nx // 2, ny // 2, 0), (3, ny // world, 0), (nx - 2, ny // world - 1, 0)]
    p = oracle.make_params(nx, ny, 1, srcs, sigma=0.05, eps_s=3.0, eps_i=2.0, tau_d=1e-11)
    whole = oracle.Oracle(p, ecoef=oracle.ECOEF_CONSISTENT)
    whole.set_materials(codes[None], table)
    whole.step(steps)
    for fid in (oracle.HX, oracle.HY, oracle.DZ, oracle.EZ):
        got = np.concatenate([np.load(tmp_path / f"f{fid}_r{r}.npy") for r in range(world)], axis=0)
        want = np.asarray(whole.field(fid)[0])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), oracle.FIELD_NAMES[fid]
    assert np.abs(whole.field(oracle.EZ)).max() > 0
