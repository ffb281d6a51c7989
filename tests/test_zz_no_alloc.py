"""fdtd_halo_get_noalloc(): the packed boundary block of a multi-step slab staged in the array
that is part of the library's code object, against fdtd_halo_get() (buffers allocated on first
use) -- bit for bit, on two calls in a row, the second on a smaller slab (so the array still
holds the first call's data), both sides of the slab; and the size limit."""
import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import binding as B
import util

pytestmark = pytest.mark.gpu


def make(nx, ny, k, seed):
    rng = np.random.default_rng(seed)
    s = fdtd_amd.Solver(fdtd_amd.make_params(nx, 3 * ny, 1, [(nx // 2, ny + ny // 2, 0)], sigma=0.2, eps_s=9.0, eps_i=3.0,
                                             tau_d=2e-11),
                        ecoef=B.ECOEF_CONSISTENT, kernel=(B.KERNEL_MULTISTEP, k), slab=(ny, ny))
    for code, row in enumerate(util.RIKEN_LIKE):
        s.set_material(code, *[float(v) for v in row])
    s.set_codes(rng.integers(0, len(util.RIKEN_LIKE), size=(ny, nx), dtype=np.uint8))
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        s.set_field(fid, rng.standard_normal(s.shape).astype(np.float32))
    s.step(k)
    return s


def test_halo_get_noalloc_equals_halo_get_on_two_calls_in_a_row():
    big = make(1001, 64, 8, 1)          # 6 * 8 * 1001 floats + 8008 code bytes
    small = make(333, 40, 5, 2)         # 5 * 333 = 1665 code bytes: the last float is written in part only
    for s in (big, small):
        for side in (0, 1):
            got = s.halo_get_noalloc(side)
            want = s.halo_get(side)
            assert got.size == s.halo_count and np.abs(want).max() > 0
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (s.p.nx, side)
    # one-row form (one-step kernels): no staging at all, same rows
    f = fdtd_amd.Solver(fdtd_amd.make_params(700, 90, 1, [(350, 45, 0)]), ecoef=B.ECOEF_CONSISTENT, kernel=B.KERNEL_FUSED,
                        slab=(30, 30))
    f.step(40)
    assert np.array_equal(f.halo_get_noalloc(1), f.halo_get(1))


def test_halo_get_noalloc_rejects_blocks_beyond_its_staging_memory():
    s = fdtd_amd.Solver(fdtd_amd.make_params(24000, 64, 1, [(12000, 32, 0)]), ecoef=B.ECOEF_CONSISTENT,
                        kernel=(B.KERNEL_MULTISTEP, 8), slab=(16, 32))
    assert s.halo_count > (1 << 20)
    with pytest.raises(fdtd_amd.FdtdError, match="4 MiB"):
        s.halo_get_noalloc(0)
    assert s.halo_get(0).size == s.halo_count       # the allocating form still serves it


def test_halo_get_noalloc_3d_plane_above_the_staging_limit_is_not_staged():
    # 3 * 640 * 600 floats > 1 Mi floats: the one-plane form copies straight out of the field arrays
    s = fdtd_amd.Solver(fdtd_amd.make_params(640, 600, 12, [(320, 300, 6)]), ecoef=B.ECOEF_CONSISTENT, kernel=B.KERNEL_FUSED,
                        slab=(4, 4))
    s.step(6)
    assert s.halo_count > (1 << 20)
    for side in (0, 1):
        got, want = s.halo_get_noalloc(side), s.halo_get(side)
        assert np.abs(want).max() > 0 or side == 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
