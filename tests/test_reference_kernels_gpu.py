"""The reference's OWN kernels against the CPU oracle and the HIP path.

oracle/_ref/libfdtd_ref.so is /root/reference/fdtd_calculations.cu itself, passed through the
image's hipify-perl (cuda* runtime names -> hip*, nothing else) and compiled for gfx950 with
-ffp-contract=off by `make -C oracle ref` in the container that holds the reference tree
(oracle/ref_driver.hip launches the five kernels with main's geometry, order and buffer
rotation).  These tests pin the oracle to it bit for bit -- vacuum, Debye tissue maps, random rp
planes (all twelve Mur cases), sources, odd extents, every phase of the three-buffer rotation --
and hold the HIP kernels against the same outputs directly.  The reference has no 2-D mode (with
nz = 1 none of its guards passes), but ONE step of the 2-D TMz reduction is what the reference's
kernels do to the middle plane of a three-plane grid whose state does not depend on z and has
ex = ey = hz = dx = dy = 0: the z differences the reduction drops are then exact zeros.  The last
test pins the oracle's 2-D step to that, from random states."""
import os

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import binding as B

import util

pytestmark = pytest.mark.gpu
REF = B.ECOEF_REFERENCE


@pytest.fixture(scope="module")
def ref():
    from oracle import reference
    if not reference.available():
        pytest.skip("oracle/_ref/libfdtd_ref.so is not built (needs the reference tree: make -C oracle ref)")
    return reference


def assert_same(orc, o, got, what):
    for fid in range(9):
        for level, arr in enumerate(got[fid]):
            want = np.asarray(o.field(fid, level))
            util.assert_bits_equal(arr, want, f"{what}: {orc.FIELD_NAMES[fid]} level {level}")


def random_case(orc, dims, srcs, seed, warm=0):
    nx, ny, nz = dims
    rng = np.random.default_rng(seed)
    o = orc.Oracle(orc.make_params(nx, ny, nz, srcs, sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11), ecoef=orc.ECOEF_REFERENCE)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(nz, ny, nx), dtype=np.uint8)
    o.set_materials(codes, util.RIKEN_LIKE)
    for face in range(6):
        o.set_rp(face, (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32))
    # every buffer of the rotation holds data: nothing relies on untouched cells being zero
    for fid in range(9):
        amp = 1e-12 if B.DX <= fid <= B.DZ else (1e-3 if fid <= B.HZ else 1.0)
        for level in ((0,) if fid <= B.HZ else (0, 1, 2)):
            o.set_field(fid, (rng.standard_normal((nz, ny, nx)) * amp).astype(np.float32), level)
    if warm:
        o.step(warm)
    return o, codes


def test_reference_kernels_vacuum_from_rest_equal_oracle(oracle, ref):
    """The shape of the shipped case (vacuum defaults, default rp planes, one source, all fields
    zero at the start) on a small grid: 60 steps, every field and every buffer."""
    dims, src = (31, 26, 22), [(15, 13, 11)]
    o = oracle.Oracle(oracle.make_params(*dims, src), ecoef=oracle.ECOEF_REFERENCE)
    got = ref.run_like(o, 60)
    o.step(60, threads=4)
    assert_same(oracle, o, got, "vacuum")
    assert np.abs(got[B.EZ][0]).max() > 0 and np.all(np.isfinite(got[B.EZ][0]))


def test_reference_kernels_shipped_case_100cube(oracle, ref, golden):
    """data/input_params as shipped (100^3, vacuum, source index (49,49,49)), 111 steps: the
    reference's kernels reproduce the recorded probe rows (overflow to -inf at step 111 included)
    and agree with the oracle on every cell of every buffer on the way."""
    from conftest import DATA
    p = oracle.load_params(os.path.join(DATA, "input_params"))
    o = oracle.Oracle(p, ecoef=oracle.ECOEF_REFERENCE)
    sx, sy, sz = p.sources[0], p.sources[1], p.sources[2]
    done = 0
    for step, dz, ez, ezx, hy in golden["probe_series_100cube_vacuum_reference"]["rows"]:
        got = ref.run_like(o, step - done)
        o.step(step - done, threads=8)
        done = step
        with np.errstate(invalid="ignore"):
            assert_same(oracle, o, got, f"shipped case, step {step}")
        vals = (got[B.DZ][0][sz, sy, sx], got[B.EZ][0][sz, sy, sx], got[B.EZ][0][sz, sy, sx + 1], got[B.HY][0][sz, sy, sx])
        for g, w in zip(vals, (dz, ez, ezx, hy)):
            w = float(w)
            assert g == w if np.isinf(w) else f"{float(g):.8e}" == f"{w:.8e}", (step, vals)


@pytest.mark.parametrize("dims,srcs,warm,steps", [
    ((37, 29, 23), [(18, 14, 11), (1, 1, 0), (36, 28, 22)], 0, 7),
    ((37, 29, 23), [(18, 14, 11)], 4, 5),          # starts in the middle of the rotation
    ((37, 29, 23), [(18, 14, 11)], 2, 9),
    ((131, 5, 7), [(129, 2, 3), (127, 3, 3)], 1, 6),   # more than one 128-thread block in x
    ((3, 3, 3), [(1, 1, 1)], 0, 6),
    ((4, 9, 5), [(2, 4, 2)], 5, 4),
])
def test_reference_kernels_debye_mur_random_state_equal_oracle(oracle, ref, dims, srcs, warm, steps):
    o, _ = random_case(oracle, dims, srcs, seed=sum(dims) + warm, warm=warm)
    got = ref.run_like(o, steps)
    o.step(steps, threads=4)
    assert_same(oracle, o, got, f"{dims} from step {warm}")
    assert np.abs(got[B.EX][0]).max() > 0


@pytest.mark.parametrize("kernel", [B.KERNEL_FUSED, B.KERNEL_SPLIT])
def test_hip_kernels_equal_reference_kernels(oracle, ref, kernel):
    """No oracle in between: the HIP path and the reference's kernels from the same state (random
    fields, tissue codes, random rp planes), two newest levels of E and D and H, bit for bit."""
    dims, srcs, steps = (45, 31, 19), [(22, 15, 9), (2, 3, 4)], 8
    rng = np.random.default_rng(77)
    s, o = util.pair(oracle, dims, srcs, REF, kernel=kernel, sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=dims[::-1], dtype=np.uint8)
    util.set_materials(s, o, codes)
    for face in range(6):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    util.randomize(s, o, rng, oracle)
    got = ref.run_like(o, steps)          # the oracle only lends its state; it is not stepped
    s.step(steps)
    for fid in range(9):
        for level in ((0,) if fid <= B.HZ else (0, 1)):
            util.assert_bits_equal(s.field(fid, level), got[fid][level], f"{B.FIELD_NAMES[fid]} level {level}")
    assert np.abs(s.field(B.EZ)).max() > 0


@pytest.mark.parametrize("dims,warm,seed", [((37, 29), 0, 1), ((37, 29), 1, 2), ((37, 29), 2, 3), ((131, 6), 4, 4),
                                            ((3, 3), 0, 5), ((5, 40), 5, 6)])
def test_2d_tmz_step_is_the_reference_step_of_a_z_uniform_state(oracle, ref, dims, warm, seed):
    """hx, hy, dz, ez of a random 2-D state (tissue codes, random rp planes, sources, all three
    buffers filled) go into every plane of an nx x ny x 3 grid, the other components are zero; one
    step of the reference's kernels on that grid, and plane 1 must be the oracle's 2-D step: every
    buffer of hx, hy, dz, ez, bit for bit.  Every phase of the buffer rotation is covered."""
    nx, ny = dims
    rng = np.random.default_rng(seed)
    srcs2 = [(nx // 2, ny // 2, 0), (1, 1, 0)]
    kw = dict(sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    o2 = oracle.Oracle(oracle.make_params(nx, ny, 1, srcs2, **kw), ecoef=oracle.ECOEF_REFERENCE)
    o3 = oracle.Oracle(oracle.make_params(nx, ny, 3, [(x, y, 1) for x, y, _ in srcs2], **kw), ecoef=oracle.ECOEF_REFERENCE)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(ny, nx), dtype=np.uint8)
    o2.set_materials(codes[None], util.RIKEN_LIKE)
    o3.set_materials(np.repeat(codes[None], 3, axis=0), util.RIKEN_LIKE)
    # x faces: [z][y][2]; y faces: [z][2][x] -- the 2-D planes repeated for every z
    for face in (0, 1, 3, 4):
        rp = (1.0 + 60.0 * rng.random(o2.rp(face).size)).astype(np.float32)
        o2.set_rp(face, rp)
        o3.set_rp(face, np.tile(rp, 3))
    if warm:                       # only to move both step counters (source sample, buffer rotation)
        o2.step(warm)
        o3.step(warm)
    for fid in range(9):
        for level in ((0,) if fid <= B.HZ else (0, 1, 2)):
            if fid in (B.HX, B.HY, B.DZ, B.EZ):
                amp = 1e-12 if fid == B.DZ else (1e-3 if fid <= B.HZ else 1.0)
                plane = (rng.standard_normal((ny, nx)) * amp).astype(np.float32)
                o2.set_field(fid, plane[None], level)
                o3.set_field(fid, np.repeat(plane[None], 3, axis=0), level)
            else:
                o3.set_field(fid, np.zeros((3, ny, nx), dtype=np.float32), level)
    got = ref.run_like(o3, 1)
    o2.step(1)
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        for level, arr in enumerate(got[fid]):
            util.assert_bits_equal(arr[1], np.asarray(o2.field(fid, level))[0], f"{oracle.FIELD_NAMES[fid]} level {level}")
    assert np.abs(np.asarray(o2.field(B.EZ))).max() > 0
