"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on
the same inputs.  Contract per field: max|hip - oracle| <= 1e-5 * running peak
of the oracle field (tests/util.py); both sides evaluate the same IEEE
operations in the same order, so the tests also require bit-identity."""
import os

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import binding as B
from conftest import DATA
import util

pytestmark = pytest.mark.gpu

REF, CON = B.ECOEF_REFERENCE, B.ECOEF_CONSISTENT


# ------------------------------------------------------------------ 3-D

def test_3d_shipped_case_against_recorded_probe_series(golden):
    """data/input_params (100^3 vacuum, REFERENCE coefficients): dz, ez, hy at the
    source after steps 1..110 against the recorded reference outputs."""
    p = fdtd_amd.params_load(os.path.join(DATA, "input_params"))
    s = fdtd_amd.Solver(p, ecoef=REF)
    s.set_probes([[49, 49, 49], [50, 49, 49]], capacity=128)
    s.step(111)
    pr = s.read_probes(0, 111)          # [step][probe][hx hy hz dx dy dz ex ey ez]
    for step, dz, ez, ezx, hy in golden["probe_series_100cube_vacuum_reference"]["rows"]:
        got = (pr[step - 1, 0, B.DZ], pr[step - 1, 0, B.EZ], pr[step - 1, 1, B.EZ], pr[step - 1, 0, B.HY])
        for g, w in zip(got, (dz, ez, ezx, hy)):
            w = float(w)
            if np.isinf(w):
                assert g == w
            else:
                assert f"{float(g):.8e}" == f"{w:.8e}", (step, got)


@pytest.mark.parametrize("kernel", [B.KERNEL_FUSED, B.KERNEL_SPLIT])
def test_3d_vacuum_fields_match_oracle(oracle, kernel):
    s, o = util.pair(oracle, (100, 100, 100), [(49, 49, 49)], REF, kernel=kernel)
    assert util.run_and_compare(s, o, 90, 30, "100^3 vacuum") == 0


@pytest.mark.parametrize("kernel", [B.KERNEL_FUSED, "fused, Mur launch", B.KERNEL_SPLIT])
@pytest.mark.parametrize("dims,ecoef", [((23, 19, 17), REF), ((23, 19, 17), CON), ((64, 5, 9), CON),
                                         ((3, 3, 3), REF), ((130, 7, 4), REF), ((265, 12, 6), CON),
                                         ((300, 9, 5), CON), ((4, 4, 4), REF), ((5, 4, 6), CON),
                                         ((128, 32, 5), CON)])    # planes that fill their XCD segments
def test_3d_random_state_debye_and_mur(oracle, monkeypatch, dims, ecoef, kernel):
    """Random fields, random tissue map, non-uniform rp planes, two sources: every
    term of every kernel (all 12 Mur cases, the ey high-x quirk) is exercised.  Grids of this size
    finish the Mur faces inside the fused sweep; "Mur launch" runs the sweep large grids take."""
    if kernel == "fused, Mur launch":
        monkeypatch.setenv("FDTD_3D_FOLD_MUR", "0")
        kernel = B.KERNEL_FUSED
    nx, ny, nz = dims
    rng = np.random.default_rng(1234 + nx)
    srcs = [(nx // 2, ny // 2, nz // 2), (nx - 1, ny - 1, nz - 1), (1, 1, 0)]
    s, o = util.pair(oracle, dims, srcs, ecoef, kernel=kernel, sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(nz, ny, nx), dtype=np.uint8)
    util.set_materials(s, o, codes)
    for face in range(6):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    util.randomize(s, o, rng, oracle)
    assert util.run_and_compare(s, o, 12, 3, f"{dims}") == 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("FDTD_TEST_3D_SEEDS", "16"))))   # more seeds for a soak run
def test_3d_random_cases(oracle, monkeypatch, seed):
    """Seeded random 3-D cases: extents (rows that are no multiple of four cells, planes that end inside
    a wave or an XCD segment), tissue maps, rp planes, sources and probes anywhere, the way the steps are
    cut into calls; with the Mur faces inside the sweep or as a launch of their own."""
    rng = np.random.default_rng(9000 + seed)
    nx, ny, nz = int(rng.integers(4, 70)), int(rng.integers(4, 40)), int(rng.integers(4, 30))
    if rng.random() < 0.3:
        monkeypatch.setenv("FDTD_3D_FOLD_MUR", "0")
    if rng.random() < 0.3:
        monkeypatch.setenv("FDTD_3D_FLAT", "0")
    srcs = [(int(rng.integers(1, nx)), int(rng.integers(1, ny)), int(rng.integers(0, nz)))
            for _ in range(int(rng.integers(1, 5)))]
    ecoef = REF if rng.random() < 0.3 else CON
    s, o = util.pair(oracle, (nx, ny, nz), srcs, ecoef, kernel=B.KERNEL_FUSED, sigma=0.2, eps_s=9.0, eps_i=3.0,
                     tau_d=2e-11)
    if rng.random() < 0.7:
        codes = rng.integers(0, len(util.RIKEN_LIKE), size=(nz, ny, nx), dtype=np.uint8)
        util.set_materials(s, o, codes)
    for face in range(6):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    util.randomize(s, o, rng, oracle)
    probes = [[int(rng.integers(0, nx)), int(rng.integers(0, ny)), int(rng.integers(0, nz))]
              for _ in range(int(rng.integers(1, 9)))]
    s.set_probes(probes, capacity=64)
    steps = [int(v) for v in rng.integers(1, 7, size=3)]
    series = []
    for n in steps:
        s.step(n)
        for _ in range(n):
            o.step(1)
            series.append([[o.field(f)[z, y, x] for f in range(9)] for x, y, z in probes])
    for fid in util.fields_of(s):
        util.assert_bits_equal(s.field(fid), np.asarray(o.field(fid)), f"seed {seed} {(nx, ny, nz)} {B.FIELD_NAMES[fid]}")
    got = s.read_probes(0, sum(steps))
    want = np.array(series, dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, nx, ny, nz)


def test_3d_mur_from_loaded_materials(oracle):
    dims = (12, 11, 10)
    rng = np.random.default_rng(5)
    s, o = util.pair(oracle, dims, [(6, 5, 5)], CON)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=dims[::-1], dtype=np.uint8)
    util.set_materials(s, o, codes)
    s.setup_mur_from_materials()
    o.setup_mur()
    for face in range(6):
        assert np.array_equal(s.rp(face), o.rp(face)), face
    assert util.run_and_compare(s, o, 30, 10, "rp from materials") == 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kernel", [B.KERNEL_FUSED, B.KERNEL_SPLIT])
def test_3d_body_slices_at_full_plane_size(oracle, kernel):
    """The plane size the reference data has (265 x 490, pitch 272: rows that are no multiple of a
    wave, XCD segments that cut rows), 24 planes of the tissue slice v1_00800 with a different
    tissue swapped in on every third plane, mat_specs_riken, rp planes from the loaded materials,
    sources next to faces: every field against the oracle bit for bit."""
    nx, ny, nz, steps = 265, 490, 24, 10
    srcs = [(40, 245, 12), (1, 1, 0), (264, 489, 23), (130, 200, 1)]
    s, o = util.pair(oracle, (nx, ny, nz), srcs, CON, kernel=kernel)
    specs = os.path.join(DATA, "mat_specs_riken")
    table, present = B.read_material_specs(specs)
    base = B.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), nx, ny)
    codes = np.broadcast_to(base, (nz, ny, nx)).copy()
    codes[::3][codes[::3] == 48] = 45
    for code in np.flatnonzero(present):
        s.set_material(int(code), *[float(v) for v in table[code]])
    s.set_codes(codes)
    o.set_materials(codes, table)
    s.setup_mur_from_materials()
    o.setup_mur()
    assert util.run_and_compare(s, o, steps, 5, "3-D body slices") == 0
    assert np.abs(s.field(B.EX)).max() > 0

@pytest.mark.timeout(900)
def test_3d_whole_body_model_265x490x1687_every_cell(oracle, capsys):
    """The workload the reference's data was made for, at its full size: all 1687 slices of the voxel
    body model (tests/golden/body_codes_265x490x1687.u8.xz -- the codes of the reference's PGM files,
    53 tissues), mat_specs_riken, rp planes from the loaded materials, a random state in every
    cell inside the outermost layer at both time levels (the cells no kernel ever writes lie on the
    faces and stay zero, as in the reference), sources in air and in tissue.  Four steps (every phase
    of the reference's three-buffer rotation and two source samples): all nine fields of all
    219 M cells against the oracle bit for bit.  About a minute, most of it host-side set-up: it
    says where it is (a silent minute looks like a hang to a watchdog)."""
    import gc
    import lzma
    import time
    import psutil
    t_start = time.time()

    def say(what):
        with capsys.disabled():
            print(f"[whole body model, {time.time() - t_start:5.1f} s] {what}", flush=True)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "body_codes_265x490x1687.u8.xz")
    if psutil.virtual_memory().available < 50e9:
        pytest.skip("needs 50 GB of host memory (the oracle's state is 22 GB)")
    nx, ny, nz, steps = 265, 490, 1687, 4
    n = nx * ny * nz
    codes = np.frombuffer(lzma.decompress(open(path, "rb").read()), dtype=np.uint8).reshape(nz, ny, nx)
    base = B.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), nx, ny)
    assert np.array_equal(codes[799], base)                     # slice 800 is the shipped PGM file
    srcs = [(40, 245, 800), (132, 245, 900), (1, 1, 0), (264, 489, 1686)]
    s, o = util.pair(oracle, (nx, ny, nz), srcs, CON)
    assert s.sweep_kernel_name == "fdtd3d_sweep"
    table, present = B.read_material_specs(os.path.join(DATA, "mat_specs_riken"))
    assert set(np.unique(codes).tolist()) <= set(np.flatnonzero(present).tolist())
    for code in np.flatnonzero(present):
        s.set_material(int(code), *[float(v) for v in table[code]])
    s.set_codes(codes)
    o.set_materials(codes, table)
    say("materials set")
    s.setup_mur_from_materials()
    o.setup_mur()
    for face in range(6):
        assert np.array_equal(s.rp(face), o.rp(face)), face
    rng = np.random.default_rng(2024)
    stride = 1000003
    noise = rng.standard_normal(n + 16 * stride, dtype=np.float32)
    i = 0
    for fid in range(9):
        for lvl in ((0,) if fid <= B.HZ else (0, 1)):
            i += 1
            amp = np.float32(1e-12 if B.DX <= fid <= B.DZ else (1e-3 if fid <= B.HZ else 1.0))
            data = (noise[i * stride:i * stride + n] * amp).reshape(nz, ny, nx)
            if fid > B.HZ:
                data[0] = data[-1] = 0.0
                data[:, 0] = data[:, -1] = 0.0
                data[:, :, 0] = data[:, :, -1] = 0.0
            s.set_field(fid, data, lvl)
            o.set_field(fid, data, lvl)
        say(f"state of {B.FIELD_NAMES[fid]} set")
    del noise, data
    gc.collect()
    o.step(steps, threads=16)
    say("oracle stepped")
    s.step(steps)
    s.sync()
    say("GPU stepped")
    for fid in range(9):
        got, want = s.field(fid), np.asarray(o.field(fid))
        assert np.isfinite(want).all() and np.count_nonzero(want) > 0.8 * want.size
        util.assert_bits_equal(got, want, f"whole body model, {B.FIELD_NAMES[fid]}")
        del got
        say(f"{B.FIELD_NAMES[fid]} equal")
    s.close()
    o.close()


# ------------------------------------------------------------------ 2-D TMz

MS2, MS3, MS4, MS5, MS6, MS7, MS8 = [(B.KERNEL_MULTISTEP, k) for k in (2, 3, 4, 5, 6, 7, 8)]
TL1, TL3, TL8 = [(B.KERNEL_TILE, k) for k in (1, 3, 8)]      # the tiled multi-step sweep of small grids
KERNELS = [B.KERNEL_FUSED, B.KERNEL_SPLIT, MS2, MS3, MS4, MS5, MS6, MS7, MS8, TL1, TL3, TL8]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dims,src,chunk", [
    ((265, 490), [(130, 245)], 0),
    ((1030, 37), [(255, 5), (256, 6), (257, 7), (1023, 8), (1024, 9), (1029, 36)], 5),
    ((300, 70), [(1, 1), (299, 69), (150, 63), (150, 64)], 64),
    ((3, 3), [(1, 1)], 1),
    ((7, 5), [(3, 2)], 2),
    ((2049, 11), [(2047, 5), (2048, 6)], 3),
])
def test_2d_vacuum_point_sources(oracle, kernel, dims, src, chunk):
    """Sources on strip edges (x = 255/256, 1023/1024), chunk edges and grid edges."""
    nx, ny = dims
    s, o = util.pair(oracle, (nx, ny, 1), [(x, y, 0) for x, y in src], CON, kernel=kernel, chunk_rows=chunk)
    assert util.run_and_compare(s, o, 120, 40, f"2-D {dims}") == 0


@pytest.mark.parametrize("kernel", KERNELS)
def test_2d_reference_coefficients_until_overflow_window(oracle, kernel):
    s, o = util.pair(oracle, (200, 180, 1), [(100, 90, 0)], REF, kernel=kernel)
    assert util.run_and_compare(s, o, 90, 30, "2-D REFERENCE") == 0


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dims,chunk", [((265, 49), 7), ((1100, 21), 4), ((70, 130), 64)])
def test_2d_random_state_debye_and_mur(oracle, kernel, dims, chunk):
    nx, ny = dims
    rng = np.random.default_rng(99 + nx)
    s, o = util.pair(oracle, (nx, ny, 1), [(nx // 2, ny // 2, 0), (nx - 1, ny - 1, 0)], CON, kernel=kernel,
                     chunk_rows=chunk, sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(1, ny, nx), dtype=np.uint8)
    util.set_materials(s, o, codes)
    for face in (0, 1, 3, 4):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    util.randomize(s, o, rng, oracle)
    assert util.run_and_compare(s, o, 15, 5, f"2-D random {dims}") == 0


@pytest.mark.parametrize("kernel", KERNELS)
def test_2d_pgm_tissue_slice_1000_steps(oracle, kernel):
    """Config 1b: data/input_params physics on the 265x490 body slice v1_00800,
    mat_specs_riken tissues, source in air next to the body, 1000 steps."""
    nx, ny = 265, 490
    pa = fdtd_amd.make_params(nx, ny, 1, [(40, 245, 0)])
    po = oracle.make_params(nx, ny, 1, [(40, 245, 0)])
    s = fdtd_amd.Solver(pa, ecoef=CON, kernel=kernel)
    o = oracle.Oracle(po, ecoef=CON)
    specs, pgm = os.path.join(DATA, "mat_specs_riken"), os.path.join(DATA, "pgmdata")
    s.load_materials(specs, pgm, first_slice=800)
    o.load_materials(specs, pgm, first_slice=800)
    assert util.run_and_compare(s, o, 1000, 250, "PGM slice") == 0
    ez = s.field(B.EZ)
    assert np.isfinite(ez).all() and np.abs(ez).max() > 0


def test_2d_upsampled_pgm_matches_oracle_on_repeated_codes(oracle):
    nx, ny, u = 265 * 2, 490 * 2, 2
    pa = fdtd_amd.make_params(nx, ny, 1, [(60, 490, 0)])
    po = oracle.make_params(nx, ny, 1, [(60, 490, 0)])
    s = fdtd_amd.Solver(pa, ecoef=CON)
    o = oracle.Oracle(po, ecoef=CON)
    specs, pgm = os.path.join(DATA, "mat_specs_riken"), os.path.join(DATA, "pgmdata")
    s.load_materials(specs, pgm, first_slice=800, upsample=u)
    table, _ = B.read_material_specs(specs)
    codes = B.read_pgm_codes(os.path.join(pgm, "v1_00800.pgm"), 265, 490)
    o.set_materials(np.repeat(np.repeat(codes, u, axis=0), u, axis=1)[None], table)
    assert util.run_and_compare(s, o, 300, 100, "upsampled PGM") == 0


def test_3d_upsampled_pgm_slices_expanded_on_the_device(oracle, tmp_path):
    """loadMaterials with upsample 3 in 3-D: every slice file is read as a (nx/3) x (ny/3) image and
    repeated 3 x 3 times by a kernel; the oracle gets the repeated codes from numpy."""
    bx, by, nz, u = 9, 7, 6, 3
    nx, ny = bx * u, by * u
    rng = np.random.default_rng(31)
    base = rng.choice([0, 1, 45, 48, 94], size=(nz, by, bx)).astype(np.uint8)
    for iz in range(nz):
        (tmp_path / f"v1_{iz + 3:05d}.pgm").write_text(
            f"P2\n# test\n{bx} {by}\n94\n" + "\n".join(str(c) for c in base[iz].ravel()) + "\n")
    src = [(nx // 2, ny // 2, nz // 2)]
    s = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, nz, src), ecoef=CON, kernel=B.KERNEL_FUSED)
    o = oracle.Oracle(oracle.make_params(nx, ny, nz, src), ecoef=CON)
    specs = os.path.join(DATA, "mat_specs_riken")
    s.load_materials(specs, str(tmp_path), first_slice=3, upsample=u)
    table, _ = B.read_material_specs(specs)
    o.set_materials(np.repeat(np.repeat(base, u, axis=1), u, axis=2), table)
    s.setup_mur_from_materials()
    o.setup_mur()
    for face in range(6):
        assert np.array_equal(s.rp(face), o.rp(face)), face
    assert util.run_and_compare(s, o, 40, 20, "3-D upsampled PGM") == 0


@pytest.mark.parametrize("kind", [B.KERNEL_MULTISTEP, B.KERNEL_TILE])
@pytest.mark.parametrize("seed", range(int(os.environ.get("FDTD_TEST_SEEDS", "24"))))   # more seeds for a soak run
def test_2d_multi_step_random_cases(oracle, seed, kind):
    """Seeded random cases of the multi-step sweep: grid size (strip and band edges included),
    steps per pass, sources (also on the last column and row), tissue maps, rp planes, state and
    the way the steps are cut into calls -- every field bit for bit after every call."""
    rng = np.random.default_rng(1000 + seed)
    nx = int(rng.choice([3, 4, 5, 239, 240, 241, 244, 480, 481, int(rng.integers(6, 700))]))
    ny = int(rng.choice([3, 4, 5, 16, 17, 63, 64, 65, int(rng.integers(6, 200))]))
    k = int(rng.integers(2, 9))
    nsrc = int(rng.integers(1, 5))
    srcs = [(int(rng.integers(1, nx)), int(rng.integers(1, ny)), 0) for _ in range(nsrc)]
    s, o = util.pair(oracle, (nx, ny, 1), srcs, CON, kernel=(kind, k), sigma=0.2, eps_s=9.0, eps_i=3.0,
                     tau_d=2e-11)
    if rng.random() < 0.7:
        codes = rng.integers(0, len(util.RIKEN_LIKE), size=(1, ny, nx), dtype=np.uint8)
        if rng.random() < 0.5:
            codes[:, :, : nx // 2] = codes[0, 0, 0]          # uniform waves next to mixed ones
        util.set_materials(s, o, codes)
    for face in (0, 1, 3, 4):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    if rng.random() < 0.5:
        util.randomize(s, o, rng, oracle)
    peak = util.Peak()
    for n in rng.integers(1, 3 * k + 2, size=5):
        for _ in range(int(n)):
            o.step(1)
            peak.update(o, util.fields_of(s))
        s.step(int(n))
        assert util.compare(s, o, peak, f"seed {seed} nx {nx} ny {ny} K {k}") == 0


@pytest.mark.parametrize("k", [2, 5, 8])
@pytest.mark.parametrize("rows", [2, 4, 9, 16])
def test_2d_multi_step_short_bands_on_the_y_faces(oracle, monkeypatch, k, rows):
    """Launches that touch a y face start with a band of 2K+1 rows and end with one of K+2 when the
    grid is tall enough for it (2 * rows + 3K + 3 rows): grid heights on either side of that
    threshold, band heights that do and do not divide the rest, Mur with random rp on all four
    faces, tissue codes, sources and probes inside the two short bands, next to them and in the
    x-face strips -- every field bit for bit against the oracle, probe series included.  (Bands of two
    rows: a wave then runs three iterations with all its stages after the 2(K-1) in which its pipeline fills.)"""
    monkeypatch.setenv("FDTD_TB_ROWS", str(rows))
    limit = 2 * rows + 3 * k + 3
    for ny in (limit - 1, limit, limit + 7, 4 * limit + 3):
        nx = 487                                   # three strips: both x faces and a plain one
        rng = np.random.default_rng(7000 + 31 * ny + k)
        srcs = [(1, 1, 0), (nx - 1, ny - 1, 0), (nx // 2, 2 * k, 0), (nx // 2 + 1, 2 * k + 1, 0), (250, ny - k - 2, 0),
                (3, ny - k - 3, 0)]
        s, o = util.pair(oracle, (nx, ny, 1), srcs, CON, kernel=(B.KERNEL_MULTISTEP, k), sigma=0.2, eps_s=9.0, eps_i=3.0,
                         tau_d=2e-11)
        util.set_materials(s, o, rng.integers(0, len(util.RIKEN_LIKE), size=(1, ny, nx), dtype=np.uint8))
        for face in (0, 1, 3, 4):
            rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
            s.set_rp(face, rp)
            o.set_rp(face, rp)
        util.randomize(s, o, rng, oracle)
        probes = [(0, 0, 0), (nx - 1, ny - 1, 0), (nx // 2, 2 * k, 0), (241, 2 * k + 1, 0), (5, ny - k - 2, 0), (nx - 2, ny - k - 3, 0)]
        steps = 2 * k + 3                          # two full passes and a remainder pass
        s.set_probes(probes, capacity=steps)
        want = []
        peak = util.Peak()
        for _ in range(steps):
            o.step(1)
            peak.update(o, util.fields_of(s))
            want.append([[float(np.asarray(o.field(f)).reshape(ny, nx)[y, x]) for f in (B.HX, B.HY, B.DZ, B.EZ)] for x, y, _ in probes])
        s.step(steps)
        assert util.compare(s, o, peak, f"short face bands K {k} rows {rows} ny {ny}") == 0
        got = s.read_probes(0, steps)
        assert np.array_equal(got.view(np.uint32), np.asarray(want, dtype=np.float32).view(np.uint32)), (k, rows, ny)
        s.close()


def test_wave_timeline_of_a_multi_step_pass():
    """fdtd_profile_waves_*: one record per wave of the next pass -- every wave of the launch is
    there once, its clock readings are ordered and lie inside the launch, the strips and bands
    cover the grid, and the run is unchanged by it."""
    nx, ny, k = 2000, 700, 8
    a = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, 1, [(1000, 350, 0)]), ecoef=CON, kernel=(B.KERNEL_MULTISTEP, k))
    b = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, 1, [(1000, 350, 0)]), ecoef=CON, kernel=(B.KERNEL_MULTISTEP, k))
    a.step(3 * k)
    b.step(3 * k)
    waves, _ = a.sweep_work
    a.profile_begin(1)
    a.profile_waves_begin(waves + 8)
    a.step(k)
    b.step(k)
    ms, _ = a.profile_read_launches(1)
    t = a.profile_waves_read(waves + 8)
    t = t[t[:, 1] > 0]
    assert len(t) == waves
    start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
    assert (end > start).all()
    assert (end.max() - start.min()) * 1e-5 <= ms[0] * 1.05 + 0.02          # 100 MHz ticks -> ms
    strips = (t[:, 2] & np.uint64(0xffffffff)).astype(int)
    bands = (t[:, 2] >> np.uint64(32)).astype(int)
    assert strips.min() == 0 and strips.max() == (nx // 4 + 59) // 60 - 1
    assert len(set(zip(strips.tolist(), bands.tolist()))) == waves          # every (strip, band) once
    flags = (t[:, 3] & np.uint64(3)).astype(int)
    assert (flags == 3).any() and (flags != 3).any()                        # plain waves and guarded ones
    a.step(k)                                                               # the next pass records nothing
    b.step(k)
    for f in (B.HX, B.HY, B.DZ, B.EZ):
        assert np.array_equal(a.field(f).view(np.uint32), b.field(f).view(np.uint32))


# ------------------------------------------------------------------ slabs

@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("world,blank", [(2, False), (3, False), (3, True)])
def test_2d_slabs_equal_whole_grid(oracle, kernel, world, blank):
    """world slabs on one GPU, halos staged through the host after every step (every pass
    of the multi-step kernels, whose halos hold every array steps_per_pass rows deep):
    bit-identical to the single-slab run (and so to the oracle).  blank: the first slab never
    gets material codes (it holds the default material only) but borders one that does."""
    nx, ny = 300, 101
    rng = np.random.default_rng(7)
    srcs = [(150, 50, 0), (20, 33, 0), (280, 34, 0)]
    pa = fdtd_amd.make_params(nx, ny, 1, srcs, sigma=0.1, eps_s=5.0, eps_i=2.0, tau_d=1e-11)
    whole = fdtd_amd.Solver(pa, ecoef=CON, kernel=kernel, chunk_rows=16)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(ny, nx), dtype=np.uint8)
    parts = fdtd_amd.partition_rows(ny, world)
    if blank:
        codes[:parts[0][1]] = B.DEFAULT_CODE
    slabs = [fdtd_amd.Solver(pa, ecoef=CON, kernel=kernel, chunk_rows=16, slab=part) for part in parts]
    for i, (sv, (b, n)) in enumerate(zip([whole] + slabs, [(0, ny)] + parts)):
        for code, row in enumerate(util.RIKEN_LIKE):
            sv.set_material(code, *[float(v) for v in row])
        if not (blank and i == 1):
            sv.set_codes(codes[b:b + n])
    assert len({sv.steps_per_pass for sv in slabs}) == 1
    if world == 3:
        # rp planes from the loaded materials: a multi-step slab takes the x-face entries of the
        # neighbours' rows it recomputes from the codes that arrive with the halos
        for sv in [whole] + slabs:
            sv.setup_mur_from_materials()

    def exchange():
        for r in range(world - 1):          # neighbours r and r + 1
            up, down = slabs[r].halo_get(1), slabs[r + 1].halo_get(0)
            slabs[r + 1].halo_put(0, up)
            slabs[r].halo_put(1, down)

    probes = [(150, 50, 0), (10, 0, 0), (299, 100, 0), (151, parts[1][0], 0), (152, parts[1][0] - 1, 0)]
    for sv in [whole] + slabs:
        sv.set_probes(probes, capacity=160)
    stride, done, steps = slabs[0].steps_per_pass, 0, 150
    exchange()                              # the deep halos also carry the neighbours' material codes
    while done < steps:
        n = min(stride, steps - done, 1 + done % 7)     # full passes, shorter passes and single steps
        whole.step(n)
        for sv in slabs:
            sv.step(n)
        exchange()
        done += n
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        full = whole.field(fid)
        got = np.concatenate([sv.field(fid) for sv in slabs], axis=1)
        assert np.array_equal(full.view(np.uint32), got.view(np.uint32)), B.FIELD_NAMES[fid]
    assert np.abs(whole.field(B.EZ)).max() > 0
    # a probe is recorded by the slab that holds it and reads as zero on the others
    series = whole.read_probes(0, steps)
    parts_series = [sv.read_probes(0, steps) for sv in slabs]
    assert np.array_equal(sum(parts_series).view(np.uint32), (series + np.float32(0)).view(np.uint32))
    assert np.abs(series[:, 3:]).max() > 0
    for k, (x, y, _) in enumerate(probes):
        for (b, n), got in zip(parts, parts_series):
            if not b <= y < b + n:
                assert not got[:, k].any()


@pytest.mark.parametrize("kernel", [B.KERNEL_FUSED, B.KERNEL_SPLIT])
def test_3d_slabs_equal_whole_grid(oracle, kernel):
    nx, ny, nz = 40, 21, 30
    rng = np.random.default_rng(8)
    srcs = [(20, 10, 15), (5, 5, 9), (5, 5, 10)]
    pa = fdtd_amd.make_params(nx, ny, nz, srcs, sigma=0.1, eps_s=5.0, eps_i=2.0, tau_d=1e-11)
    whole = fdtd_amd.Solver(pa, ecoef=CON, kernel=kernel)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(nz, ny, nx), dtype=np.uint8)
    parts = fdtd_amd.partition_rows(nz, 3)
    slabs = [fdtd_amd.Solver(pa, ecoef=CON, kernel=kernel, slab=part) for part in parts]
    for sv, (b, n) in zip([whole] + slabs, [(0, nz)] + parts):
        for code, row in enumerate(util.RIKEN_LIKE):
            sv.set_material(code, *[float(v) for v in row])
        sv.set_codes(codes[b:b + n])
    for _ in range(60):
        whole.step(1)
        for sv in slabs:
            sv.step(1)
        for r in range(2):
            up, down = slabs[r].halo_get(1), slabs[r + 1].halo_get(0)
            slabs[r + 1].halo_put(0, up)
            slabs[r].halo_put(1, down)
    for fid in range(9):
        full = whole.field(fid)
        got = np.concatenate([sv.field(fid) for sv in slabs], axis=0)
        assert np.array_equal(full.view(np.uint32), got.view(np.uint32)), B.FIELD_NAMES[fid]
    assert np.abs(whole.field(B.EX)).max() > 0


def test_rccl_binding_with_a_one_rank_communicator():
    """The RCCL entry points (dlopen, ncclCommInitRank, grouped ncclSend/ncclRecv on the
    library's stream) on one GPU: a one-rank communicator sends a row to itself."""
    s = fdtd_amd.Solver(fdtd_amd.make_params(700, 40, 1, [(350, 20, 0)]), ecoef=CON, slab=(0, 40),
                        kernel=B.KERNEL_FUSED)
    s.step(30)
    s.comm_init(B.comm_unique_id(), 0, 1)
    s.comm_selftest()
    assert np.array_equal(s.halo_get(0), s.field(B.EZ)[0, 0])
    s.step(5)                      # stepping with a communicator and no neighbours is a no-op exchange
    # multi-step engine: the packed block of boundary rows (4 rows of 6 arrays + codes) through RCCL
    m = fdtd_amd.Solver(fdtd_amd.make_params(700, 40, 1, [(350, 3, 0)]), ecoef=CON, slab=(0, 40), kernel=MS4)
    m.step(30)
    m.comm_init(B.comm_unique_id(), 0, 1)
    m.comm_selftest()
    blk = m.halo_get(0)
    assert blk.size == 6 * 4 * 700 + 700 and np.array_equal(blk[2 * 4 * 700:2 * 4 * 700 + 700], m.field(B.EZ)[0, 0])
    assert np.abs(blk).max() > 0
    m.step(6)
    s3 = fdtd_amd.Solver(fdtd_amd.make_params(40, 30, 20, [(20, 15, 10)]), ecoef=CON)
    s3.step(20)
    s3.comm_init(B.comm_unique_id(), 0, 1)
    s3.comm_selftest()


@pytest.mark.parametrize("kind", [B.KERNEL_MULTISTEP, B.KERNEL_TILE])
@pytest.mark.parametrize("seed", range(int(os.environ.get("FDTD_TEST_WIDE_SEEDS", "6"))))
def test_2d_multi_step_random_cases_many_strips_and_bands(oracle, seed, kind):
    """Wider and taller random grids: several strips and bands per launch, so that the order in which
    the row-marching sweep starts its waves (x-face strips, last band, waves with sources or
    probes first) and the tile map of the tiled sweep matter; sources and probes anywhere, tissue
    codes, rp planes; fields and probe rows bit for bit."""
    rng = np.random.default_rng(5000 + seed)
    nx, ny = int(rng.integers(481, 2600)), int(rng.integers(100, 900))
    k = int(rng.integers(2, 9))
    srcs = [(int(rng.integers(1, nx)), int(rng.integers(1, ny)), 0) for _ in range(int(rng.integers(1, 6)))]
    s, o = util.pair(oracle, (nx, ny, 1), srcs, CON, kernel=(kind, k), sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    if rng.random() < 0.6:
        codes = rng.integers(0, len(util.RIKEN_LIKE), size=(1, ny, nx), dtype=np.uint8)
        util.set_materials(s, o, codes)
    for face in (0, 1, 3, 4):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    util.randomize(s, o, rng, oracle)
    probes = [[int(rng.integers(0, nx)), int(rng.integers(0, ny)), 0] for _ in range(int(rng.integers(1, 17)))]
    s.set_probes(probes, capacity=64)
    steps = [int(v) for v in rng.integers(1, 2 * k + 2, size=3)]
    series = []
    peak = util.Peak()
    for n in steps:
        for _ in range(n):
            o.step(1, threads=8)
            peak.update(o, util.fields_of(s))
            series.append([[o.field(f)[0, y, x] for f in (B.HX, B.HY, B.DZ, B.EZ)] for x, y, _ in probes])
        s.step(n)
        assert util.compare(s, o, peak, f"seed {seed} {nx}x{ny} K {k}") == 0
    got = s.read_probes(0, len(series))
    assert np.array_equal(got.view(np.uint32), np.array(series, dtype=np.float32).view(np.uint32))


@pytest.mark.timeout(300)
def test_2d_multi_step_long_run_with_the_library_s_own_choices(oracle):
    """1000 steps of a 3100 x 2500 grid (7.75 M cells: the row-marching sweep at the steps per pass and
    the band height the library chooses by itself), tissue codes in blocks, Mur with random rp, three
    sources, probes; calls of irregular length, so full passes, remainder passes of every length and
    single steps alternate for a long time.  Fields every 125 steps and the probe series of the whole
    run, bit for bit."""
    nx, ny, steps = 3100, 2500, 1000
    rng = np.random.default_rng(424242)
    srcs = [(nx // 2, ny // 2, 0), (7, ny - 9, 0), (nx - 3, 40, 0)]
    s, o = util.pair(oracle, (nx, ny, 1), srcs, CON, sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    assert s.sweep_kernel_name == "fdtd2d_sweep_tb" and s.steps_per_pass == 7
    blocks = rng.integers(0, len(util.RIKEN_LIKE), size=(1, (ny + 31) // 32, (nx + 31) // 32), dtype=np.uint8)
    codes = np.ascontiguousarray(np.repeat(np.repeat(blocks, 32, axis=1), 32, axis=2)[:, :ny, :nx])
    util.set_materials(s, o, codes)
    for face in (0, 1, 3, 4):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    probes = [(nx // 2, ny // 2, 0), (0, 0, 0), (nx - 1, ny - 1, 0), (1000, 3, 0), (nx // 2 + 200, ny // 2 - 100, 0)]
    s.set_probes(probes, capacity=steps)
    series = np.zeros((steps, len(probes), 4), dtype=np.float32)
    peak = util.Peak()
    done = 0
    while done < steps:
        n = int(min(steps - done, rng.integers(1, 60)))
        for i in range(n):
            o.step(1, threads=16)
            peak.update(o, util.fields_of(s))
            for q, (x, y, _) in enumerate(probes):
                series[done + i, q] = [o.field(f)[0, y, x] for f in (B.HX, B.HY, B.DZ, B.EZ)]
        s.step(n)
        if (done + n) // 125 != done // 125 or done + n == steps:
            assert util.compare(s, o, peak, f"long run, step {done + n}") == 0
        done += n
    assert np.abs(series[:, 0, 3]).max() > 0
    assert np.array_equal(s.read_probes(0, steps).view(np.uint32), series.view(np.uint32))
    s.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("spp", [0, 5])
def test_2d_two_processes_one_gpu_host_transport(oracle, tmp_path, spp):
    """Two ranks (two processes, both on GPU 0) run the slab protocol with the HIP engine
    and gloo-staged halos; the joined result equals the oracle bit for bit."""
    import subprocess
    import sys
    nx, ny, steps = 1500, 200, 122
    script = os.path.join(os.path.dirname(__file__), "slab_rank_main.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(29517 + spp), str(nx), str(ny), str(steps),
                               str(tmp_path), str(spp)])
             for r in range(2)]
    assert [p.wait(timeout=240) for p in procs] == [0, 0]
    o = oracle.Oracle(oracle.make_params(nx, ny, 1, [(nx // 2, ny // 2, 0), (10, ny // 2 - 1, 0)]), ecoef=CON)
    o.step(steps, threads=4)
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        got = np.concatenate([np.load(tmp_path / f"f{fid}_r{r}.npy") for r in range(2)], axis=0)
        assert np.array_equal(got.view(np.uint32), np.asarray(o.field(fid)[0]).view(np.uint32)), B.FIELD_NAMES[fid]


def test_2d_tiled_sweep_tall_tiles_materials_mur_probes(oracle):
    """1000 x 700: more than 256 tiles of 48 x 48 cells, so the tiled sweep takes its 64-row tiles
    (smaller grids run the 32-row ones); tissue codes, random rp planes, sources and probes on tile
    edges (x = 47/48, y = 47/48) and on the faces; fields and probe rows bit for bit."""
    nx, ny = 1000, 700
    rng = np.random.default_rng(5)
    srcs = [(500, 350, 0), (47, 47, 0), (48, 48, 0), (999, 699, 0), (1, 1, 0)]
    s, o = util.pair(oracle, (nx, ny, 1), srcs, CON, kernel=(B.KERNEL_TILE, 8), sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    assert s.sweep_kernel_name == "fdtd2d_tile"
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=(1, ny, nx), dtype=np.uint8)
    util.set_materials(s, o, codes)
    for face in (0, 1, 3, 4):
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    util.randomize(s, o, rng, oracle)
    probes = [[500, 350, 0], [0, 300, 0], [999, 301, 0], [400, 0, 0], [401, 699, 0], [47, 47, 0], [48, 48, 0], [1, 1, 0]]
    s.set_probes(probes, capacity=64)
    series = []
    peak = util.Peak()
    for n in (8, 5, 1, 16, 3):
        for _ in range(n):
            o.step(1)
            peak.update(o, util.fields_of(s))
            series.append([[o.field(f)[0, y, x] for f in (B.HX, B.HY, B.DZ, B.EZ)] for x, y, _ in probes])
        s.step(n)
        assert util.compare(s, o, peak, f"tiled sweep after {len(series)} steps") == 0
    got = s.read_probes(0, len(series))
    assert np.array_equal(got.view(np.uint32), np.array(series, dtype=np.float32).view(np.uint32))


@pytest.mark.parametrize("dims,kernel", [((530, 140, 1), MS8), ((530, 140, 1), TL8), ((530, 140, 1), B.KERNEL_FUSED),
                                         ((41, 27, 19), B.KERNEL_FUSED), ((41, 27, 19), B.KERNEL_SPLIT)])
def test_grid_of_one_non_default_material(oracle, dims, kernel):
    """Every cell holds tissue code 4: the single-material kernels run with that material's row;
    then the row changes; then one cell takes another code and the table kernels are back."""
    src = [(dims[0] // 2, dims[1] // 2, dims[2] // 2)]
    s, o = util.pair(oracle, dims, src, CON, kernel=kernel, sigma=0.2, eps_s=9.0, eps_i=3.0, tau_d=2e-11)
    codes = np.full(dims[::-1], 4, dtype=np.uint8)
    util.set_materials(s, o, codes)
    assert util.run_and_compare(s, o, 40, 20, "one material") == 0
    s.set_material(4, 0.3, 30.0, 12.0, 1.5e-11)          # the row behind the single code changes
    table = util.RIKEN_LIKE.copy()
    table[4] = [0.3, 30.0, 12.0, 1.5e-11]
    o.set_materials(codes, table)
    assert util.run_and_compare(s, o, 24, 12, "one material, new row") == 0
    codes[..., dims[1] // 3, dims[0] // 3] = 2
    s.set_codes(codes)
    o.set_materials(codes, table)
    assert util.run_and_compare(s, o, 30, 15, "two materials") == 0


@pytest.mark.parametrize("dims,spp", [((4096, 3600), 0), ((530, 140), 8), ((530, 140), 5)])
def test_2d_material_codes_beyond_the_wide_table(oracle, dims, spp):
    """Material codes above 83: their table rows (48 bytes each) do not fit beside the parked rows of 8
    steps per pass.  Left to itself the library then takes 7 steps per pass; asked for 8 it takes the
    table of 16-byte rows with three waves per workgroup.  Same numbers either way."""
    nx, ny = dims          # (the library chooses 8 steps per pass from 12 M cells up)
    kernel = B.KERNEL_MULTISTEP if spp == 0 else (B.KERNEL_MULTISTEP, spp)
    s, o = util.pair(oracle, (nx, ny, 1), [(nx // 2, ny // 2, 0), (3, 3, 0)], CON, kernel=kernel, sigma=0.2, eps_s=9.0,
                     eps_i=3.0, tau_d=2e-11)
    rng = np.random.default_rng(99)
    table = np.tile(np.array([0.2, 9.0, 3.0, 2e-11], dtype=np.float32), (201, 1))
    table[:len(util.RIKEN_LIKE)] = util.RIKEN_LIKE
    table[120] = [0.9, 40.0, 10.0, 2.5e-11]
    table[200] = [0.05, 6.0, 2.5, 0.9e-11]
    codes = rng.choice(np.array([0, 1, 3, 120, 200], dtype=np.uint8), size=(ny, nx))
    assert s.steps_per_pass == (spp or 8)
    util.set_materials(s, o, codes, table)
    assert s.steps_per_pass == (spp or 7)
    for face in (0, 1, 3, 4):          # x0, y0, xEnd, yEnd
        rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
        s.set_rp(face, rp)
        o.set_rp(face, rp)
    assert util.run_and_compare(s, o, 45 if spp else 22, 15 if spp else 11, "codes up to 200") == 0


@pytest.mark.parametrize("dims,kernel", [((530, 140, 1), MS8), ((530, 140, 1), TL8), ((530, 140, 1), B.KERNEL_FUSED),
                                         ((40, 30, 20), B.KERNEL_FUSED), ((40, 30, 20), B.KERNEL_SPLIT)])
def test_rp_planes_replaced_between_steps(oracle, dims, kernel):
    """fdtd_set_rp in the middle of a run: whatever a kernel keeps derived from the planes (the
    field-independent Mur factors) follows."""
    nx, ny, nz = dims
    rng = np.random.default_rng(77)
    s, o = util.pair(oracle, dims, [(nx // 2, ny // 2, nz // 2), (2, 2, 0 if nz == 1 else 1)], CON, kernel=kernel)
    util.randomize(s, o, rng, oracle)
    faces = (0, 1, 3, 4) if nz == 1 else range(6)
    for round_ in range(3):
        for face in faces:
            rp = (1.0 + 60.0 * rng.random(o.rp(face).size)).astype(np.float32)
            s.set_rp(face, rp)
            o.set_rp(face, rp)
        assert util.run_and_compare(s, o, 19, 19, f"rp set {round_}") == 0


# ------------------------------------------------------------------ results path

@pytest.mark.parametrize("kernel", [B.KERNEL_FUSED, MS2, MS5, TL1, TL3, TL8])
def test_probe_series_of_multi_step_passes(oracle, kernel):
    """Probes on grid edges, strip edges (x = 247/248) and band edges (y = 31/32) over passes and
    single steps mixed: every step's row of the ring equals the oracle, bit for bit."""
    nx, ny = 520, 100
    s, o = util.pair(oracle, (nx, ny, 1), [(260, 50, 0), (3, 3, 0)], CON, kernel=kernel, sigma=0.3, eps_s=4.0,
                     eps_i=2.0, tau_d=1e-11)
    probes = [[260, 50, 0], [0, 40, 0], [519, 41, 0], [100, 0, 0], [101, 99, 0], [247, 31, 0], [248, 32, 0],
              [4, 4, 0], [1, 1, 0]]
    s.set_probes(probes, capacity=256)
    series = []
    for n in (1, 8, 3, 4, 5, 40, 2, 37):
        s.step(n)
        for _ in range(n):
            o.step(1)
            series.append([[o.field(f)[0, y, x] for f in (B.HX, B.HY, B.DZ, B.EZ)] for x, y, _ in probes])
    got = s.read_probes(0, 100)
    want = np.array(series, dtype=np.float32)
    assert np.abs(want[:, 1:5, 3]).max() > 0          # the wave reached the boundary probes
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_probes_and_snapshot_agree_with_read_field(oracle):
    s, o = util.pair(oracle, (400, 300, 1), [(200, 150, 0)], CON)
    probes = [[200, 150, 0], [210, 150, 0], [200, 170, 0], [399, 299, 0]]
    s.set_probes(probes, capacity=64)
    snap = np.empty(s.shape, dtype=np.float32)
    series = []
    for step in range(50):
        s.step(1)
        o.step(1)
        if step == 30:
            s.snapshot_async(B.EZ, snap)        # the next steps overwrite this level
            want_snap = np.array(o.field(B.EZ))
        series.append([[o.field(f)[0, y, x] for f in (B.HX, B.HY, B.DZ, B.EZ)] for x, y, _ in probes])
    s.snapshot_wait()
    assert np.array_equal(snap.view(np.uint32), want_snap.view(np.uint32))
    s.snapshot_async(B.HY)                      # zero-copy form: the pinned staging buffer
    hy_now = s.field(B.HY)
    s.step(3)
    assert np.array_equal(s.snapshot_wait().view(np.uint32), hy_now.view(np.uint32))
    got = s.read_probes(0, 50)
    assert np.array_equal(got.view(np.uint32), np.array(series, dtype=np.float32).view(np.uint32))
    with pytest.raises(fdtd_amd.FdtdError):
        s.read_probes(0, 54)      # 53 steps were taken


def test_create_rejects_bad_inputs():
    with pytest.raises(fdtd_amd.FdtdError, match="at least 3"):
        fdtd_amd.Solver(fdtd_amd.make_params(2, 10, 1, [(1, 1, 0)]))
    with pytest.raises(fdtd_amd.FdtdError, match="source_location"):
        fdtd_amd.Solver(fdtd_amd.make_params(10, 10, 1, [(0, 5, 0)]))
    with pytest.raises(fdtd_amd.FdtdError, match="slab"):
        fdtd_amd.Solver(fdtd_amd.make_params(10, 10, 1, [(5, 5, 0)]), slab=(8, 5))
    s = fdtd_amd.Solver(fdtd_amd.make_params(10, 10, 1, [(5, 5, 0)]))
    with pytest.raises(fdtd_amd.FdtdError, match="count"):
        s.set_codes(np.zeros(7, dtype=np.uint8))
    with pytest.raises(fdtd_amd.FdtdError, match="2-D"):
        s.field(B.EX)


# ------------------------------------------------------------------ full size

@pytest.mark.timeout(600)
def test_2d_full_size_16384_window_and_causality(oracle):
    """16384 x 16384 vacuum + centre source (the benchmark workload), 200 steps.
    Translation invariance: until the wave meets a boundary the solution only
    depends on the distance to the source, so the 512 x 512 window around the
    source must equal, bit for bit, an oracle run on a 512 x 512 grid.  Causality:
    one step moves the field one cell in the 1-norm, so ez is exactly zero at
    |x - sx| + |y - sy| > steps."""
    n, steps, w = 16384, 200, 512
    sx = sy = n // 2
    s = fdtd_amd.Solver(fdtd_amd.make_params(n, n, 1, [(sx, sy, 0)]), ecoef=CON, kernel=B.KERNEL_MULTISTEP)
    assert s.sweep_kernel_name == "fdtd2d_sweep_tb" and s.steps_per_pass == 8
    o = oracle.Oracle(oracle.make_params(w, w, 1, [(w // 2, w // 2, 0)]), ecoef=CON)
    s.step(steps)
    o.step(steps, threads=8)
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        full = s.field(fid)[0]
        win = full[sy - w // 2: sy + w // 2, sx - w // 2: sx + w // 2]
        assert np.array_equal(win.view(np.uint32), np.asarray(o.field(fid)[0]).view(np.uint32)), B.FIELD_NAMES[fid]
        ys, xs = np.nonzero(full)
        assert ys.size > 0
        assert int(np.max(np.abs(ys - sy) + np.abs(xs - sx))) <= steps + 1
        del full


@pytest.mark.timeout(600)
def test_2d_tiled_sweep_4096_whole_field_and_auto_choice(oracle):
    """4096 x 4096 with the tiled multi-step sweep (7 396 tiles, guard-free and guarded ones).  Sources near a corner so that the wave meets two
    Mur faces; 100 steps (12 launches of 8 + one of 4) against the oracle on the whole grid, bit for
    bit, and the probe series of every step."""
    n, steps = 4096, 100
    src = [(60, 70, 0), (2048, 2048, 0)]
    s = fdtd_amd.Solver(fdtd_amd.make_params(n, n, 1, src), ecoef=CON, kernel=TL8)
    assert s.sweep_kernel_name == "fdtd2d_tile" and s.steps_per_pass == 8
    probes = [[60, 70, 0], [0, 70, 0], [61, 0, 0], [2048, 2049, 0], [4095, 4095, 0]]
    s.set_probes(probes, capacity=steps)
    o = oracle.Oracle(oracle.make_params(n, n, 1, src), ecoef=CON)
    s.step(steps)
    series = []
    for _ in range(steps):
        o.step(1, threads=8)
        series.append([[o.field(f)[0, y, x] for f in (B.HX, B.HY, B.DZ, B.EZ)] for x, y, _ in probes])
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        got, want = s.field(fid), np.asarray(o.field(fid))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), B.FIELD_NAMES[fid]
        assert np.abs(got).max() > 0
    want = np.array(series, dtype=np.float32)
    assert np.abs(want[:, 1:3, 3]).max() > 0          # the wave reached the face probes
    assert np.array_equal(s.read_probes(0, steps).view(np.uint32), want.view(np.uint32))
    s.close()
    # FDTD_KERNEL_AUTO: the tiled sweep below 7.5 M cells, the row-marching one from there on
    # (steps per launch: 8, from 3 M cells up 6; the row-marching sweep 7 below 12 M cells, then 8)
    for dims, name, spp in (((1024, 1024), "fdtd2d_tile", 8), ((2048, 2048), "fdtd2d_tile", 6),
                            ((2896, 2896), "fdtd2d_sweep_tb", 7), ((3072, 3072), "fdtd2d_sweep_tb", 7),
                            ((4096, 4096), "fdtd2d_sweep_tb", 8)):
        auto = fdtd_amd.Solver(fdtd_amd.make_params(*dims, 1, [(100, 100, 0)]), ecoef=CON)
        assert auto.sweep_kernel_name == name and auto.steps_per_pass == spp, dims
        auto.close()


@pytest.mark.timeout(900)
def test_2d_largest_grid_65536_window_and_causality(oracle):
    """65536 x 65536 (4.3 G cells, 210 GB resident: byte offsets need 38 bits), source off
    centre in the upper half so every index of the window lies beyond 2^32 bytes.  Same
    translation-invariance and causality checks as at 16384^2, read through windows."""
    import ctypes
    import gc
    gc.collect()                                   # contexts of earlier tests still hold their arrays
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert ctypes.CDLL("libamdhip64.so").hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    if free.value < 225e9:
        pytest.skip("needs 225 GB of free HBM (the multi-step sweep keeps 49 bytes per cell)")
    n, steps, w = 65536, 60, 256
    sx, sy = 40000, 50000
    s = fdtd_amd.Solver(fdtd_amd.make_params(n, n, 1, [(sx, sy, 0)]), ecoef=CON, kernel=B.KERNEL_MULTISTEP)
    s.set_probes([(sx, sy, 0)], capacity=steps)
    o = oracle.Oracle(oracle.make_params(w, w, 1, [(w // 2, w // 2, 0)]), ecoef=CON)
    s.step(steps)
    o.step(steps, threads=8)
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        win = s.window(fid, sy - w // 2, w, sx - w // 2, w)
        assert np.array_equal(win.view(np.uint32), np.asarray(o.field(fid)[0]).view(np.uint32)), B.FIELD_NAMES[fid]
        assert np.any(win != 0)
        # nothing outside the light cone: a frame of windows around it, and the far corners
        for row0, x0 in ((sy - 2 * w, sx - w // 2), (sy + w, sx - w // 2), (sy - w // 2, sx - 2 * w),
                         (sy - w // 2, sx + w), (0, 0), (n - w, n - w), (0, n - w), (n - w, 0)):
            assert not np.any(s.window(fid, row0, w, x0, w)), (B.FIELD_NAMES[fid], row0, x0)
    assert np.all(np.isfinite(s.read_probes(0, steps)))
    s.close()


@pytest.mark.timeout(900)
def test_2d_tissue_map_8480x15680_whole_field(oracle):
    """The dispersive workload at its full size: v1_00800.pgm upsampled x32 (8480 x 15680, 133 M
    cells, 8 tissues), mat_specs_riken, Mur with the default rp, a source in the air beside the
    body.  36 steps (4 passes of 8 + one of 4) against the oracle on the whole grid: every
    cell of every field bit for bit."""
    u, steps = 32, 36
    nx, ny = 265 * u, 490 * u
    src = [(40 * u, 245 * u, 0)]
    s = fdtd_amd.Solver(fdtd_amd.make_params(nx, ny, 1, src), ecoef=CON, kernel=B.KERNEL_MULTISTEP)
    o = oracle.Oracle(oracle.make_params(nx, ny, 1, src), ecoef=CON)
    specs, pgm = os.path.join(DATA, "mat_specs_riken"), os.path.join(DATA, "pgmdata")
    assert s.steps_per_pass == 8
    s.load_materials(specs, pgm, first_slice=800, upsample=u)
    table, _ = B.read_material_specs(specs)
    codes = B.read_pgm_codes(os.path.join(pgm, "v1_00800.pgm"), 265, 490)
    o.set_materials(np.repeat(np.repeat(codes, u, axis=0), u, axis=1)[None], table)
    assert s.steps_per_pass == 8          # the highest code of the map is 51: its table rows fit beside 8 steps
    s.step(steps)
    o.step(steps, threads=16)
    for fid in (B.HX, B.HY, B.DZ, B.EZ):
        got, want = s.field(fid), np.asarray(o.field(fid))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), B.FIELD_NAMES[fid]
        assert np.abs(got).max() > 0
        del got, want


# ------------------------------------------------------------------ command-line driver

def test_cli_runs_like_the_reference_and_writes_its_output_format(oracle, tmp_path):
    """./fdtd from a directory with data/input_params, data/mat_specs_riken and PGM slices:
    the reference's progress lines and {H,D,E}_field_{x,y,z}_%05d_z%05d.out files
    (fdtd.cu:1288-1494), values equal to the oracle's printed with " % 9.3E"."""
    import shutil
    import subprocess
    from conftest import ROOT
    nx, ny, nz = 21, 18, 5
    data = tmp_path / "data"
    (data / "pgmdata").mkdir(parents=True)
    (tmp_path / "output").mkdir()
    text = open(os.path.join(DATA, "input_params")).read()
    text = text.replace("nx_ny_nz: 100 100 100", f"nx_ny_nz: {nx} {ny} {nz}").replace(
        "source_location: 50 50 50", "source_location: 11 9 3")
    (data / "input_params").write_text(text)
    shutil.copy(os.path.join(DATA, "mat_specs_riken"), data / "mat_specs_riken")
    rng = np.random.default_rng(11)
    for iz in range(nz):
        codes = rng.choice([0, 1, 45, 48, 94], size=nx * ny)
        (data / "pgmdata" / f"v1_{iz + 1:05d}.pgm").write_text(
            f"P2\n# test\n{nx} {ny}\n94\n" + "\n".join(str(c) for c in codes) + "\n")
    exe = os.path.join(ROOT, "fdtd_amd", "fdtd")
    run = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "Reading parameters..." in run.stdout and "Running iteration 3" in run.stdout
    assert f"Field size:                 {nx:04d}x{ny:04d}x{nz:04d}" in run.stdout

    o = oracle.Oracle(oracle.load_params(str(data / "input_params")), ecoef=REF)
    o.load_materials(str(data / "mat_specs_riken"), str(data / "pgmdata"))
    names = {"H": (B.HX, B.HY, B.HZ), "D": (B.DX, B.DY, B.DZ), "E": (B.EX, B.EY, B.EZ)}
    for it in range(3):
        o.step(1)
        for fam, fids in names.items():
            for axis, fid in zip("xyz", fids):
                for iz in range(nz):
                    got = (tmp_path / "output" / f"{fam}_field_{axis}_{it:05d}_z{iz:05d}.out").read_text()
                    want = "".join("".join(" % 9.3E" % v for v in row) + "\n" for row in o.field(fid)[iz])
                    assert got == want, (fam, axis, it, iz)
    # the grid lines through the source (writeResults, fdtd.cu:1197-1285), every second iteration
    run = subprocess.run([exe, "--axis-lines", "--no-planes", "--output-every", "2"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    o = oracle.Oracle(oracle.load_params(str(data / "input_params")), ecoef=REF)
    o.load_materials(str(data / "mat_specs_riken"), str(data / "pgmdata"))
    sx, sy, sz = 10, 8, 2
    order = (B.DX, B.DY, B.DZ, B.HX, B.HY, B.HZ, B.EX, B.EY, B.EZ)
    for it in range(3):
        o.step(1)
        if it % 2:
            assert not (tmp_path / "output" / f"E_field_x_{it + 1:05d}.out").exists()
            continue
        for axis, n in zip("xyz", (nx, ny, nz)):
            want = ""
            for i in range(n):
                x, y, z = (i if axis == "x" else sx), (i if axis == "y" else sy), (i if axis == "z" else sz)
                want += " %3d %3d %3d" % (x + 1, y + 1, z + 1) + "".join(" % 9.3E" % o.field(f)[z, y, x] for f in order) + "\n"
            assert (tmp_path / "output" / f"E_field_{axis}_{it + 1:05d}.out").read_text() == want, (axis, it)
    # a missing parameter file is reported and fails the process, it does not crash
    bad = subprocess.run([exe, "--params", "data/none"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "Error" in bad.stdout


# ------------------------------------------------------------------ checkpoint / restart

@pytest.mark.parametrize("dims,kernel", [((300, 200, 1), B.KERNEL_AUTO), ((40, 30, 20), B.KERNEL_AUTO),
                                         ((300, 200, 1), MS3)])
def test_checkpoint_restart_continues_bit_exactly(oracle, tmp_path, dims, kernel):
    src = [(dims[0] // 2, dims[1] // 2, dims[2] // 2)]
    rng = np.random.default_rng(21)
    codes = rng.integers(0, len(util.RIKEN_LIKE), size=dims[::-1], dtype=np.uint8)

    def make():
        s = fdtd_amd.Solver(fdtd_amd.make_params(*dims, src), ecoef=CON, kernel=kernel)
        for code, row in enumerate(util.RIKEN_LIKE):
            s.set_material(code, *[float(v) for v in row])
        s.set_codes(codes)
        return s

    a = make()
    a.step(37)
    path = str(tmp_path / "state.ckp")
    a.checkpoint_save(path)
    a.step(41)
    b = make()
    b.checkpoint_load(path)
    assert b.steps_done == 37
    b.step(41)
    # the uninterrupted run, the restarted run and the oracle agree bit for bit
    o = oracle.Oracle(oracle.make_params(*dims, src), ecoef=CON)
    o.set_materials(codes, util.RIKEN_LIKE)
    o.step(78, threads=8)
    for fid in util.fields_of(a):
        want = np.asarray(o.field(fid))
        assert np.array_equal(a.field(fid).view(np.uint32), want.view(np.uint32)), B.FIELD_NAMES[fid]
        assert np.array_equal(b.field(fid).view(np.uint32), want.view(np.uint32)), B.FIELD_NAMES[fid]
    assert np.abs(a.field(B.EZ)).max() > 0
    other = fdtd_amd.Solver(fdtd_amd.make_params(dims[0] + 4, dims[1], dims[2], src), ecoef=CON)
    with pytest.raises(fdtd_amd.FdtdError, match="another shape"):
        other.checkpoint_load(path)
    with pytest.raises(fdtd_amd.FdtdError, match="open"):
        b.checkpoint_load(str(tmp_path / "missing"))
