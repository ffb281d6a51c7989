"""Helpers shared by the GPU parity tests: build the HIP solver and the oracle
from the same inputs and compare their fields."""
import numpy as np

import fdtd_amd
from fdtd_amd import binding as B

# max |hip - oracle| <= TOL * (largest |oracle| value the field reached), per field
TOL = 1e-5

RIKEN_LIKE = np.array([
    # sigma, eps_s, eps_inf, tau_d
    [0.0, 1.0, 1.0, 0.0],
    [0.59515458, 56.44398499, 33.05709076, 3.51972652e-11],
    [0.10401546, 14.16905880, 7.36329556, 3.41064989e-11],
    [0.03710634, 5.53071189, 3.99812841, 2.36329289e-11],
    [0.74710494, 56.93144607, 28.00134277, 1.86721420e-11],
    [2.14390564, 70.39963913, 33.14797211, 1.81622616e-11],
    [0.5, 4.0, 4.0, 0.0],
], dtype=np.float32)


def pair(oracle, dims, sources, ecoef, kernel=B.KERNEL_AUTO, chunk_rows=0, **kw):
    """(Solver, Oracle) on identical parameters."""
    nx, ny, nz = dims
    pa = fdtd_amd.make_params(nx, ny, nz, sources, **kw)
    po = oracle.make_params(nx, ny, nz, sources, **kw)
    s = fdtd_amd.Solver(pa, ecoef=ecoef, kernel=kernel, chunk_rows=chunk_rows)
    o = oracle.Oracle(po, ecoef=ecoef)
    return s, o


def fields_of(s):
    return [B.HX, B.HY, B.DZ, B.EZ] if s.is2d else list(range(9))


def set_materials(s, o, codes, table=RIKEN_LIKE):
    for code, row in enumerate(table):
        s.set_material(code, *[float(v) for v in row])
    s.set_codes(codes)
    o.set_materials(codes, table)


def written_masks(oracle_mod, o):
    """Cells of D and E that a step writes, found by stepping a scratch oracle
    whose target buffers hold a sentinel."""
    scratch = oracle_mod.Oracle(o.p, ecoef=oracle_mod.ECOEF_CONSISTENT)
    fids = [f for f in (fields_of_oracle(o)) if f > B.HZ]
    for fid in fids:
        scratch.field(fid, 2)[...] = 12345.0      # level 2 = the buffer the first step writes
    scratch.step(1)
    return {fid: np.asarray(scratch.field(fid)) != 12345.0 for fid in fids}


def fields_of_oracle(o):
    return [B.HX, B.HY, B.DZ, B.EZ] if o.p.nz == 1 else list(range(9))


def randomize(s, o, rng, oracle_mod, scale=1.0):
    """Same random state in both: H, and both time levels of D and E.  The
    reference zeroes all three of its rotating buffers (fdtd.cu:1090-1118), so a
    cell no kernel ever writes is zero at every time level; the random state
    keeps that property (such cells stay zero)."""
    masks = written_masks(oracle_mod, o)
    for fid in fields_of(s):
        levels = (0,) if fid <= B.HZ else (0, 1)
        for lvl in levels:
            amp = scale * (1e-12 if B.DX <= fid <= B.DZ else (1e-3 if fid <= B.HZ else 1.0))
            data = (rng.standard_normal(s.shape) * amp).astype(np.float32)
            if fid in masks:
                data = np.where(masks[fid], data, np.float32(0.0))     # +0.0, never -0.0
            s.set_field(fid, data, lvl)
            o.set_field(fid, data, lvl)


class Peak:
    """Running max |oracle| per field, the denominator of the tolerance."""

    def __init__(self):
        self.peak = {}

    def update(self, o, fids):
        for fid in fids:
            v = float(np.max(np.abs(o.field(fid)))) if np.all(np.isfinite(o.field(fid))) else np.inf
            self.peak[fid] = max(self.peak.get(fid, 0.0), v)


def compare(s, o, peak, where=""):
    """Assert the tolerance contract; return the number of cells that are not
    bit-identical (0 expected: both sides evaluate the same IEEE operations)."""
    notbit = 0
    for fid in fields_of(s):
        got, want = s.field(fid), np.asarray(o.field(fid))
        assert got.shape == want.shape
        pk = peak.peak.get(fid, 0.0)
        err = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))
        assert err <= TOL * pk or err == 0.0, \
            f"{where} field {B.FIELD_NAMES[fid]}: max|diff| {err:.3e} > {TOL} * peak {pk:.3e}"
        notbit += int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
    return notbit


def run_and_compare(s, o, steps, check_every, where=""):
    peak = Peak()
    fids = fields_of(s)
    done, notbit = 0, 0
    while done < steps:
        n = min(check_every, steps - done)
        for _ in range(n):          # the peak is tracked every step
            o.step(1, threads=4)
            peak.update(o, fids)
        s.step(n)
        done += n
        notbit += compare(s, o, peak, f"{where} step {done}")
    return notbit


def assert_bits_equal(got, want, what=""):
    """Bit identity with a readable report: how many cells differ, and in which rows / planes."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = got.view(np.uint32) != want.view(np.uint32)
    if bad.any():
        idx = np.argwhere(bad)
        rows = sorted(set(int(i[-2]) for i in idx))
        planes = sorted(set(int(i[0]) for i in idx))
        first = tuple(int(v) for v in idx[0])
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} cells differ; planes {planes[:8]}, rows "
                             f"{rows[:12]}{'...' if len(rows) > 12 else ''} (last {rows[-1]}); first at {first}: "
                             f"got {got[first]!r}, want {want[first]!r}")
