"""Slabs with the library's own halo exchange, on one GPU: every rank is a context driven by
its own host thread and the messages go through the in-process transport (device-to-device
copies after a rendezvous of the threads) -- the same call sequence and streams as the RCCL transport:
rows next to a neighbour swept first on the second stream, pack -> send/recv -> unpack there
beside the sweep of the interior, join before the next pass.  The joined result must equal the
oracle on the whole grid bit for bit, with the overlap on and off."""
import threading

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import binding as B
import util

pytestmark = pytest.mark.gpu

CON = B.ECOEF_CONSISTENT
DEFAULTS = dict(sigma=0.1, eps_s=5.0, eps_i=2.0, tau_d=1e-11)
# row 0 = the defaults of the parameter file: a slab that never gets codes equals one full of code 0
TABLE = np.array([[0.1, 5.0, 2.0, 1e-11]] + [list(r) for r in util.RIKEN_LIKE[1:]], dtype=np.float32)


def run_ranks(fn, world):
    """fn(rank) on one thread per rank; the first exception of any rank is raised here."""
    errors = [None] * world

    def body(r):
        try:
            fn(r)
        except BaseException as e:      # noqa: BLE001 - re-raised below
            errors[r] = e

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in the exchange"
    for e in errors:
        if e is not None:
            raise e


def build_slabs(monkeypatch, dims, srcs, kernel, world, overlap, codes, default_slab):
    monkeypatch.setenv("FDTD_HALO_OVERLAP", "1" if overlap else "0")
    monkeypatch.setenv("FDTD_LOCAL_TIMEOUT", "20")     # a rank that fails must not keep its peers waiting for minutes
    nx, ny, nz = dims
    pa = fdtd_amd.make_params(nx, ny, nz, srcs, **DEFAULTS)
    parts = fdtd_amd.partition_rows(ny if nz == 1 else nz, world)
    group = B.LocalGroup(world)
    slabs = []
    for r, (b, n) in enumerate(parts):
        sv = fdtd_amd.Solver(pa, ecoef=CON, kernel=kernel, slab=(b, n))
        for code, row in enumerate(TABLE):     # every rank holds the whole material table ...
            sv.set_material(code, *[float(v) for v in row])
        if r != default_slab:                  # ... but that slab keeps the default material in every cell
            sv.set_codes(codes[b:b + n])
        sv.comm_init_local(group.handle, r, world)
        slabs.append(sv)
    return slabs, parts, group


def check_against_oracle(oracle, dims, srcs, slabs, codes, steps, axis):
    nx, ny, nz = dims
    o = oracle.Oracle(oracle.make_params(nx, ny, nz, srcs, **DEFAULTS), ecoef=CON)
    o.set_materials(codes if nz > 1 else codes[None], TABLE)
    o.step(steps, threads=8)
    for fid in util.fields_of(slabs[0]):
        got = np.concatenate([sv.field(fid) for sv in slabs], axis=axis)
        want = np.asarray(o.field(fid))
        util.assert_bits_equal(got, want, B.FIELD_NAMES[fid])
    assert np.abs(np.asarray(o.field(B.EZ))).max() > 0


MS3, MS8 = (B.KERNEL_MULTISTEP, 3), (B.KERNEL_MULTISTEP, 8)
TL3, TL8 = (B.KERNEL_TILE, 3), (B.KERNEL_TILE, 8)


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("kernel,world,ny", [(MS8, 2, 150), (MS8, 3, 200), (MS3, 3, 130), (MS8, 3, 40),
                                             (TL8, 2, 150), (TL8, 3, 200), (TL3, 3, 130), (TL8, 3, 40),
                                             (B.KERNEL_FUSED, 3, 100), (B.KERNEL_SPLIT, 2, 60)])
def test_2d_slabs_on_threads_equal_oracle(oracle, monkeypatch, kernel, world, ny, overlap):
    """Slab 0 holds the default material only and sits next to slabs with tissue codes: its
    halo rows hold the neighbour's materials and it has to take the table kernel for them; every
    rank advances the same number of steps per pass."""
    nx, steps = 500, 61
    rng = np.random.default_rng(17)
    codes = rng.integers(0, len(TABLE), size=(ny, nx), dtype=np.uint8)
    parts = fdtd_amd.partition_rows(ny, world)
    codes[:parts[0][1]] = 0
    b1 = parts[1][0]
    srcs = [(250, ny // 2, 0), (30, b1, 0), (470, b1 - 1, 0), (100, 3, 0)]
    slabs, parts, group = build_slabs(monkeypatch, (nx, ny, 1), srcs, kernel, world, overlap, codes, default_slab=0)
    try:
        run_2d(oracle, slabs, srcs, codes, nx, ny, b1, steps, world)
    finally:                        # streams and events must not outlive the test, pass or fail
        for sv in slabs:
            sv.close()
        group.close()


def run_2d(oracle, slabs, srcs, codes, nx, ny, b1, steps, world):
    assert len({sv.steps_per_pass for sv in slabs}) == 1
    probes = [(250, ny // 2, 0), (31, b1, 0), (469, b1 - 1, 0)]
    for sv in slabs:
        sv.set_probes(probes, capacity=steps)

    def rank(r):
        for n in (1, 20, 8, 32):                 # passes, remainder passes and single steps
            slabs[r].step(n)
        slabs[r].sync()

    run_ranks(rank, world)
    check_against_oracle(oracle, (nx, ny, 1), srcs, slabs, codes, steps, axis=1)
    # probe series: every probe is recorded by the rank that holds it
    o = oracle.Oracle(oracle.make_params(nx, ny, 1, srcs, **DEFAULTS), ecoef=CON)
    o.set_materials(codes[None], TABLE)
    want = []
    for _ in range(steps):
        o.step(1, threads=4)
        want.append([[o.field(f)[0, y, x] for f in (B.HX, B.HY, B.DZ, B.EZ)] for x, y, _ in probes])
    got = sum(sv.read_probes(0, steps) for sv in slabs) + np.float32(0)
    assert np.array_equal(got.view(np.uint32), (np.array(want, dtype=np.float32) + np.float32(0)).view(np.uint32))


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("kernel", [B.KERNEL_FUSED, B.KERNEL_SPLIT])
def test_3d_slabs_on_threads_equal_oracle(oracle, monkeypatch, kernel, overlap):
    nx, ny, nz, world, steps = 44, 23, 31, 3, 40
    rng = np.random.default_rng(18)
    codes = rng.integers(0, len(TABLE), size=(nz, ny, nx), dtype=np.uint8)
    parts = fdtd_amd.partition_rows(nz, world)
    codes[:parts[0][1]] = 0
    srcs = [(22, 11, 15), (5, 5, parts[1][0]), (6, 6, parts[1][0] - 1)]
    slabs, parts, group = build_slabs(monkeypatch, (nx, ny, nz), srcs, kernel, world, overlap, codes, default_slab=0)

    def rank(r):
        for n in (1, 25, 14):
            slabs[r].step(n)
        slabs[r].sync()

    try:
        run_ranks(rank, world)
        check_against_oracle(oracle, (nx, ny, nz), srcs, slabs, codes, steps, axis=0)
    finally:
        for sv in slabs:
            sv.close()
        group.close()


@pytest.mark.parametrize("dims,kernel", [((44, 23, 31), B.KERNEL_FUSED), ((300, 90, 1), MS3), ((300, 90, 1), MS8),
                                         ((300, 90, 1), TL8), ((300, 90, 1), B.KERNEL_FUSED)])
def test_rp_planes_from_loaded_materials_per_slab(oracle, monkeypatch, dims, kernel):
    """fdtd_setup_mur_from_materials on slabs: every slab derives the rp entries of the cells it
    holds (the only ones its Mur launches read) -- a multi-step slab also the x-face entries of the
    neighbours' rows it recomputes, from the material codes that arrive with the halos; the run
    equals the oracle with setupMurBoundary evaluated on the whole loaded grid."""
    nx, ny, nz = dims
    world, steps = 3, 36
    rng = np.random.default_rng(23)
    codes = rng.integers(0, len(TABLE), size=(nz, ny, nx) if nz > 1 else (ny, nx), dtype=np.uint8)
    srcs = [(nx // 2, ny // 2, nz // 2), (3, 3, 0 if nz == 1 else 2)]
    slabs, parts, group = build_slabs(monkeypatch, dims, srcs, kernel, world, True, codes, default_slab=-1)
    try:
        for sv in slabs:
            sv.setup_mur_from_materials()

        def rank(r):
            slabs[r].step(steps)
            slabs[r].sync()

        run_ranks(rank, world)
        o = oracle.Oracle(oracle.make_params(nx, ny, nz, srcs, **DEFAULTS), ecoef=CON)
        o.set_materials(codes if nz > 1 else codes[None], TABLE)
        o.setup_mur()
        assert len(set(np.round(o.rp(0), 3))) > 2          # the planes really vary
        o.step(steps, threads=8)
        for fid in util.fields_of(slabs[0]):
            got = np.concatenate([sv.field(fid) for sv in slabs], axis=0 if nz > 1 else 1)
            util.assert_bits_equal(got, np.asarray(o.field(fid)), B.FIELD_NAMES[fid])
    finally:
        for sv in slabs:
            sv.close()
        group.close()


@pytest.mark.timeout(900)
def test_2d_two_slabs_of_the_8_gpu_shape_65536x8192_deep_halo(oracle, monkeypatch):
    """What one rank of the 65536 x 65536 grid holds on 8 GPUs: two neighbouring slabs of
    65536 x 8192 (rows 24576.. and 32768.. of the grid, ranks 0 and 1 of a world of two), 8 steps
    per pass, deep-halo blocks of 6 x 8 x 65536 floats + 8 x 65536 code bytes (13.1 MB) through
    the in-process transport, overlap on, sources on both sides of the cut.  60 steps (7 passes
    of 8, one of 4): the window around the sources equals an oracle window bit for bit, nothing
    lies outside the light cone, and a slab takes its 49 bytes per cell plus O(faces) -- no
    plane of global extent.  Then a block of that size through RCCL (one-rank communicator)."""
    import ctypes
    import gc
    gc.collect()
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    if free_bytes() < 90e9:
        pytest.skip("needs 90 GB of free HBM")
    monkeypatch.setenv("FDTD_HALO_OVERLAP", "1")
    monkeypatch.setenv("FDTD_LOCAL_TIMEOUT", "60")
    n, rows, cut, steps, w = 65536, 8192, 32768, 60, 256
    sx = 40000
    srcs = [(sx, cut - 1, 0), (sx + 5, cut + 2, 0)]
    pa = fdtd_amd.make_params(n, n, 1, srcs)
    group = B.LocalGroup(2)
    before = free_bytes()
    slabs = [fdtd_amd.Solver(pa, ecoef=CON, kernel=MS8, slab=(cut - rows + r * rows, rows)) for r in range(2)]
    try:
        for r, sv in enumerate(slabs):
            sv.comm_init_local(group.handle, r, 2)
            assert sv.sweep_kernel_name == "fdtd2d_sweep_tb" and sv.steps_per_pass == 8
            assert sv.halo_count == 6 * 8 * n + 8 * n // 4
            sv.set_probes([(sx, cut - 1, 0), (sx, cut, 0)], capacity=steps)

        def rank(r):
            slabs[r].step(steps)
            slabs[r].sync()

        run_ranks(rank, 2)
        used = before - free_bytes()
        assert used < 2 * 27.5e9, used          # 49 B x 537 M cells = 26.3 GB per slab, halos and faces on top
        o = oracle.Oracle(oracle.make_params(w, w, 1, [(w // 2, w // 2, 0), (w // 2 + 5, w // 2 + 3, 0)]), ecoef=CON)
        o.step(steps, threads=8)
        h = w // 2
        for fid in (B.HX, B.HY, B.DZ, B.EZ):
            # window rows cut-1-h .. cut-1+h-1: the last h+1 rows of slab 0, the first h-1 of slab 1
            win = np.concatenate([slabs[0].window(fid, rows - h - 1, h + 1, sx - h, w),
                                  slabs[1].window(fid, 0, h - 1, sx - h, w)], axis=0)
            util.assert_bits_equal(win, np.asarray(o.field(fid)[0]), B.FIELD_NAMES[fid])
            assert np.any(win[:h + 1] != 0) and np.any(win[h + 1:] != 0)
            for sv in slabs:     # a frame around the light cone, and the far corners of both slabs
                for row0, x0 in ((0, 0), (0, n - w), (rows - w, 0), (rows - w, n - w), (rows // 2, sx - h),
                                 (rows - w if sv is slabs[1] else 0, sx - h), (rows - w if sv is slabs[0] else 0, sx - 2 * w),
                                 (rows - w if sv is slabs[0] else 0, sx + w + 5)):
                    assert not np.any(sv.window(fid, row0, w, x0, w)), (B.FIELD_NAMES[fid], row0, x0)
        got = slabs[0].read_probes(0, steps) + slabs[1].read_probes(0, steps)
        assert np.all(np.isfinite(got)) and np.all(got[-1][:, 3] != 0)
        # the same block through ncclSend/ncclRecv: slab 0 with a one-rank RCCL communicator sends
        # its packed lower boundary rows to itself
        slabs[0].comm_init(B.comm_unique_id(), 0, 1)
        slabs[0].comm_selftest()
    finally:
        for sv in slabs:
            sv.close()
        group.close()


def test_multi_step_slab_without_a_communicator_refuses_to_outrun_its_halos():
    s = fdtd_amd.Solver(fdtd_amd.make_params(300, 100, 1, [(150, 50, 0)]), ecoef=CON, kernel=MS3, slab=(0, 50))
    s.step(3)
    with pytest.raises(fdtd_amd.FdtdError, match="exchange the halos"):
        s.step(4)
