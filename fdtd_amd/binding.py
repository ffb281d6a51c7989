"""ctypes binding of libfdtd_amd.so (the C ABI in include/fdtd_amd.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MAX_SOURCES, MAX_JZ, MAX_CODES, MAX_PROBES = 16, 512, 256, 16
MODE_AUTO, MODE_3D, MODE_2D_TMZ = 0, 1, 2
ECOEF_REFERENCE, ECOEF_CONSISTENT = 0, 1
KERNEL_AUTO, KERNEL_SPLIT, KERNEL_FUSED, KERNEL_MULTISTEP, KERNEL_TILE = 0, 1, 2, 3, 4
HX, HY, HZ, DX, DY, DZ, EX, EY, EZ = range(9)
FIELD_NAMES = ["hx", "hy", "hz", "dx", "dy", "dz", "ex", "ey", "ez"]
DEFAULT_CODE = 255


class FdtdError(RuntimeError):
    pass


class Params(C.Structure):
    """fdtd_params: mirror of the reference's FdtdParams (fdtd.h:7-34)."""

    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
        ("iterations", C.c_int32),
        ("input_path", C.c_char * 1024), ("output_path", C.c_char * 1024),
        ("elements_per_wave", C.c_int32),
        ("wave_frequency", C.c_float), ("pulse_width", C.c_float),
        ("pulse_modulation_frequency", C.c_float),
        ("sources_count", C.c_int32), ("sources", C.c_int32 * (MAX_SOURCES * 3)),
        ("default_sigma", C.c_float), ("default_eps_s", C.c_float),
        ("default_eps_i", C.c_float), ("default_tau_d", C.c_float),
        ("jz_count", C.c_int32), ("jz", C.c_float * MAX_JZ),
        ("pi", C.c_float), ("c", C.c_float), ("timeskip", C.c_float), ("lam", C.c_float),
        ("dt", C.c_float), ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float),
        ("mu0", C.c_float), ("eps0", C.c_float),
    ]


class Options(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("ecoef", C.c_int32), ("device", C.c_int32), ("kernel", C.c_int32),
        ("slab_begin", C.c_int32), ("slab_count", C.c_int32), ("chunk_rows", C.c_int32),
        ("steps_per_pass", C.c_int32), ("reserved", C.c_int32 * 8),
    ]


# Every symbol include/fdtd_amd.h declares: (name, restype, argtypes)
_VP, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
_PP, _OP = C.POINTER(Params), C.POINTER(Options)
SYMBOLS = [
    ("fdtd_last_error", C.c_char_p, []),
    ("fdtd_options_default", None, [_OP]),
    ("fdtd_params_load", _I, [C.c_char_p, _PP]),
    ("fdtd_params_derive", _I, [_PP]),
    ("fdtd_setup_sources", _I, [_PP]),
    ("fdtd_read_material_specs", _I, [C.c_char_p, _VP, _VP]),
    ("fdtd_read_pgm_codes", _I, [C.c_char_p, _I, _I, _I, _I, _I, _VP]),
    ("fdtd_material_coefficients", _I, [_PP, _I, C.c_float, C.c_float, C.c_float, C.c_float, _VP, _VP]),
    ("fdtd_rp_value", C.c_float, [_PP, _I, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("fdtd_create", _I, [_PP, _OP, C.POINTER(_VP)]),
    ("fdtd_destroy", None, [_VP]),
    ("fdtd_load_materials", _I, [_VP, C.c_char_p, C.c_char_p, _I, _I]),
    ("fdtd_set_material", _I, [_VP, _I, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("fdtd_set_codes", _I, [_VP, _VP, _SZ]),
    ("fdtd_set_rp", _I, [_VP, _I, _VP, _SZ]),
    ("fdtd_get_rp", _I, [_VP, _I, _VP, _SZ]),
    ("fdtd_setup_mur_from_materials", _I, [_VP]),
    ("fdtd_step", _I, [_VP, _I]),
    ("fdtd_sync", _I, [_VP]),
    ("fdtd_steps_done", C.c_int64, [_VP]),
    ("fdtd_steps_per_pass", _I, [_VP]),
    ("fdtd_profile_begin", _I, [_VP, _I]),
    ("fdtd_profile_read", _I, [_VP, _VP, _I, C.POINTER(_I)]),
    ("fdtd_profile_read_launches", _I, [_VP, _VP, _VP, _I, C.POINTER(_I)]),
    ("fdtd_profile_waves_begin", _I, [_VP, C.c_int64]),
    ("fdtd_profile_waves_read", _I, [_VP, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    ("fdtd_sweep_work", _I, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("fdtd_sweep_kernel_name", C.c_char_p, [_VP]),
    ("fdtd_read_field", _I, [_VP, _I, _I, _VP, _SZ]),
    ("fdtd_write_field", _I, [_VP, _I, _I, _VP, _SZ]),
    ("fdtd_read_window", _I, [_VP, _I, _I, C.c_int64, C.c_int64, _I, _I, _VP]),
    ("fdtd_snapshot_async", _I, [_VP, _I, _VP, _SZ]),
    ("fdtd_snapshot_wait", _I, [_VP]),
    ("fdtd_snapshot_data", C.POINTER(C.c_float), [_VP]),
    ("fdtd_set_probes", _I, [_VP, _I, _VP, _I]),
    ("fdtd_probe_fields", _I, [_VP]),
    ("fdtd_read_probes", _I, [_VP, C.c_int64, _I, _VP, _SZ]),
    ("fdtd_checkpoint_save", _I, [_VP, C.c_char_p]),
    ("fdtd_checkpoint_load", _I, [_VP, C.c_char_p]),
    ("fdtd_halo_count", _SZ, [_VP]),
    ("fdtd_halo_get", _I, [_VP, _I, _VP, _SZ]),
    ("fdtd_halo_put", _I, [_VP, _I, _VP, _SZ]),
    ("fdtd_halo_get_noalloc", _I, [_VP, _I, _VP, _SZ]),
    ("fdtd_comm_unique_id", _I, [C.c_char_p]),
    ("fdtd_comm_init", _I, [_VP, C.c_char_p, _I, _I]),
    ("fdtd_comm_local_group", _I, [_I, C.POINTER(_VP)]),
    ("fdtd_comm_local_group_free", None, [_VP]),
    ("fdtd_comm_init_local", _I, [_VP, _VP, _I, _I]),
    ("fdtd_comm_selftest", _I, [_VP]),
]

_lib = None


def library_path():
    return os.path.join(_HERE, "libfdtd_amd.so")


def load_library():
    """Load the in-tree library.  There is no fallback: without the HIP
    library nothing in this package computes anything."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise FdtdError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C fdtd_amd/csrc`")
        lib = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise FdtdError(f"[{rc}] {load_library().fdtd_last_error().decode()}")


def params_load(path):
    """initParamsWithPath + setupSources (fdtd.cu:547-637, 1027-1082)."""
    p = Params()
    _check(load_library().fdtd_params_load(os.fsencode(path), C.byref(p)))
    return p


def make_params(nx, ny, nz, sources, sigma=0.0, eps_s=1.0, eps_i=1.0, tau_d=0.0,
                elements_per_wave=100, wave_frequency=3.0e9, pulse_width=0.5,
                pulse_modulation_frequency=0.0, iterations=3):
    """Parameters filled in by hand (sources: 0-based (x, y, z) triples)."""
    p = Params()
    p.nx, p.ny, p.nz = nx, ny, nz
    p.iterations = ((iterations - 1) // 3 + 1) * 3
    p.elements_per_wave = elements_per_wave
    p.wave_frequency = wave_frequency
    p.pulse_width = pulse_width
    p.pulse_modulation_frequency = pulse_modulation_frequency
    p.sources_count = len(sources)
    for i, s in enumerate(sources):
        for k in range(3):
            p.sources[i * 3 + k] = int(s[k])
    p.default_sigma, p.default_eps_s, p.default_eps_i, p.default_tau_d = sigma, eps_s, eps_i, tau_d
    lib = load_library()
    _check(lib.fdtd_params_derive(C.byref(p)))
    _check(lib.fdtd_setup_sources(C.byref(p)))
    return p


def read_material_specs(path):
    """(table[256, 4] of sigma, eps_s, eps_inf, tau_d; present[256]) from a mat_specs file."""
    table = np.zeros((MAX_CODES, 4), dtype=np.float32)
    present = np.zeros(MAX_CODES, dtype=np.uint8)
    _check(load_library().fdtd_read_material_specs(os.fsencode(path), table.ctypes.data, present.ctypes.data))
    return table, present.astype(bool)


def read_pgm_codes(path, nx, ny, upsample=1, y0=0, rows=None):
    rows = ny - y0 if rows is None else rows
    out = np.zeros((rows, nx), dtype=np.uint8)
    _check(load_library().fdtd_read_pgm_codes(os.fsencode(path), nx, ny, upsample, y0, rows, out.ctypes.data))
    return out


def material_coefficients(params, ecoef, sigma, eps_s, eps_i, tau_d):
    acd = np.zeros(3, dtype=np.float32)
    k = np.zeros(3, dtype=np.float64)
    _check(load_library().fdtd_material_coefficients(C.byref(params), ecoef, sigma, eps_s, eps_i, tau_d,
                                                     acd.ctypes.data, k.ctypes.data))
    return acd, k


def rp_value(params, x_face, sigma, eps_s, eps_i, tau_d):
    return float(load_library().fdtd_rp_value(C.byref(params), int(x_face), sigma, eps_s, eps_i, tau_d))


class Solver:
    """One fdtd_ctx: a whole grid or one slab of it on one GPU."""

    def __init__(self, params, mode=MODE_AUTO, ecoef=ECOEF_REFERENCE, kernel=KERNEL_AUTO, device=0,
                 slab=None, chunk_rows=0, steps_per_pass=0):
        self.lib = load_library()
        self.p = params
        opt = Options()
        self.lib.fdtd_options_default(C.byref(opt))
        if isinstance(kernel, tuple):            # (KERNEL_MULTISTEP, steps per pass)
            kernel, steps_per_pass = kernel
        opt.mode, opt.ecoef, opt.kernel, opt.device, opt.chunk_rows = mode, ecoef, kernel, device, chunk_rows
        opt.steps_per_pass = steps_per_pass
        if slab is not None:
            opt.slab_begin, opt.slab_count = slab
        self._ctx = _VP()
        _check(self.lib.fdtd_create(C.byref(params), C.byref(opt), C.byref(self._ctx)))
        self.is2d = (mode == MODE_2D_TMZ) or (mode == MODE_AUTO and params.nz == 1)
        gn = params.ny if self.is2d else params.nz
        self.slab = slab if slab is not None else (0, gn)
        self.shape = (1, self.slab[1], params.nx) if self.is2d else (self.slab[1], params.ny, params.nx)

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.fdtd_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    # -- materials and boundaries
    def load_materials(self, specs_path, pgm_dir, first_slice=1, upsample=1):
        _check(self.lib.fdtd_load_materials(self._ctx, os.fsencode(specs_path), os.fsencode(pgm_dir),
                                            first_slice, upsample))

    def set_material(self, code, sigma, eps_s, eps_i, tau_d):
        _check(self.lib.fdtd_set_material(self._ctx, code, sigma, eps_s, eps_i, tau_d))

    def set_codes(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        _check(self.lib.fdtd_set_codes(self._ctx, codes.ctypes.data, codes.size))

    def set_rp(self, face, data):
        data = np.ascontiguousarray(data, dtype=np.float32).ravel()
        _check(self.lib.fdtd_set_rp(self._ctx, face, data.ctypes.data, data.size))

    def rp(self, face):
        p = self.p
        n = [2 * p.ny * p.nz, p.nx * 2 * p.nz, p.nx * p.ny * 2][face % 3]
        out = np.empty(n, dtype=np.float32)
        _check(self.lib.fdtd_get_rp(self._ctx, face, out.ctypes.data, out.size))
        return out

    def setup_mur_from_materials(self):
        _check(self.lib.fdtd_setup_mur_from_materials(self._ctx))

    # -- stepping
    def step(self, nsteps=1):
        _check(self.lib.fdtd_step(self._ctx, nsteps))

    def sync(self):
        _check(self.lib.fdtd_sync(self._ctx))

    @property
    def steps_done(self):
        return self.lib.fdtd_steps_done(self._ctx)

    def profile_begin(self, capacity):
        _check(self.lib.fdtd_profile_begin(self._ctx, capacity))

    def profile_read(self, capacity):
        ms = np.zeros(capacity, dtype=np.float32)
        n = _I(0)
        _check(self.lib.fdtd_profile_read(self._ctx, ms.ctypes.data, capacity, C.byref(n)))
        return ms[:n.value]

    def profile_read_launches(self, capacity):
        """(milliseconds, time steps advanced) of every timed sweep launch."""
        ms = np.zeros(capacity, dtype=np.float32)
        steps = np.zeros(capacity, dtype=np.int32)
        n = _I(0)
        _check(self.lib.fdtd_profile_read_launches(self._ctx, ms.ctypes.data, steps.ctypes.data, capacity, C.byref(n)))
        return ms[:n.value], steps[:n.value]

    def profile_waves_begin(self, capacity):
        """Arm the wave timeline for the next launch of the multi-step sweep."""
        _check(self.lib.fdtd_profile_waves_begin(self._ctx, capacity))

    def profile_waves_read(self, capacity):
        """Rows {start, end (100 MHz ticks), strip | band << 32, flags} per launch position."""
        t = np.zeros((capacity, 4), dtype=np.uint64)
        n = C.c_int64(0)
        _check(self.lib.fdtd_profile_waves_read(self._ctx, t.ctypes.data, capacity, C.byref(n)))
        return t[:n.value]

    @property
    def steps_per_pass(self):
        return self.lib.fdtd_steps_per_pass(self._ctx)

    @property
    def sweep_work(self):
        """(waves, row iterations) of one launch of the main sweep over this slab."""
        w, it = C.c_int64(0), C.c_int64(0)
        _check(self.lib.fdtd_sweep_work(self._ctx, C.byref(w), C.byref(it)))
        return w.value, it.value

    @property
    def sweep_kernel_name(self):
        return self.lib.fdtd_sweep_kernel_name(self._ctx).decode()

    # -- results
    def field(self, fid, level=0):
        out = np.empty(self.shape, dtype=np.float32)
        _check(self.lib.fdtd_read_field(self._ctx, fid, level, out.ctypes.data, out.size))
        return out

    def window(self, fid, first_row, nrows, x0, ncols, level=0):
        """Rows first_row.. of the slab (3-D: row = z * ny + y), columns x0..: an [nrows][ncols] array."""
        out = np.empty((nrows, ncols), dtype=np.float32)
        _check(self.lib.fdtd_read_window(self._ctx, fid, level, first_row, nrows, x0, ncols, out.ctypes.data))
        return out

    def set_field(self, fid, data, level=0):
        data = np.ascontiguousarray(data, dtype=np.float32)
        if data.size != int(np.prod(self.shape)):
            raise FdtdError("set_field: wrong size")
        _check(self.lib.fdtd_write_field(self._ctx, fid, level, data.ctypes.data, data.size))

    def snapshot_async(self, fid, out=None):
        """Queue a snapshot; out=None keeps it in the library's pinned buffer (see snapshot_wait)."""
        if out is None:
            _check(self.lib.fdtd_snapshot_async(self._ctx, fid, None, int(np.prod(self.shape))))
        else:
            assert out.dtype == np.float32 and out.flags.c_contiguous
            _check(self.lib.fdtd_snapshot_async(self._ctx, fid, out.ctypes.data, out.size))

    def snapshot_wait(self):
        """Block until the queued snapshot is on the host; returns a view of the pinned
        staging buffer (valid until the next snapshot_async)."""
        _check(self.lib.fdtd_snapshot_wait(self._ctx))
        ptr = self.lib.fdtd_snapshot_data(self._ctx)
        return np.ctypeslib.as_array(ptr, shape=self.shape) if ptr else None

    def set_probes(self, xyz, capacity):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32).reshape(-1, 3)
        self._nprobes = xyz.shape[0]
        _check(self.lib.fdtd_set_probes(self._ctx, xyz.shape[0], xyz.ctypes.data, capacity))

    def read_probes(self, first_step, nsteps):
        nf = self.lib.fdtd_probe_fields(self._ctx)
        out = np.empty((nsteps, self._nprobes, nf), dtype=np.float32)
        _check(self.lib.fdtd_read_probes(self._ctx, first_step, nsteps, out.ctypes.data, out.size))
        return out

    def checkpoint_save(self, path):
        _check(self.lib.fdtd_checkpoint_save(self._ctx, os.fsencode(path)))

    def checkpoint_load(self, path):
        _check(self.lib.fdtd_checkpoint_load(self._ctx, os.fsencode(path)))

    # -- slabs
    @property
    def halo_count(self):
        return self.lib.fdtd_halo_count(self._ctx)

    def halo_get(self, side):
        out = np.empty(self.halo_count, dtype=np.float32)
        _check(self.lib.fdtd_halo_get(self._ctx, side, out.ctypes.data, out.size))
        return out

    def halo_get_noalloc(self, side):
        out = np.empty(self.halo_count, dtype=np.float32)
        _check(self.lib.fdtd_halo_get_noalloc(self._ctx, side, out.ctypes.data, out.size))
        return out

    def halo_put(self, side, data):
        data = np.ascontiguousarray(data, dtype=np.float32)
        _check(self.lib.fdtd_halo_put(self._ctx, side, data.ctypes.data, data.size))

    def comm_init(self, unique_id, rank, world):
        _check(self.lib.fdtd_comm_init(self._ctx, unique_id, rank, world))

    def comm_init_local(self, group, rank, world):
        _check(self.lib.fdtd_comm_init_local(self._ctx, group, rank, world))

    def comm_selftest(self):
        _check(self.lib.fdtd_comm_selftest(self._ctx))


def comm_unique_id():
    buf = C.create_string_buffer(128)
    _check(load_library().fdtd_comm_unique_id(buf))
    return buf.raw


class LocalGroup:
    """Rendezvous object of the in-process transport (ranks = contexts of this process, one
    host thread each)."""

    def __init__(self, world):
        self.world = world
        self.handle = _VP()
        _check(load_library().fdtd_comm_local_group(world, C.byref(self.handle)))

    def close(self):
        if self.handle:
            load_library().fdtd_comm_local_group_free(self.handle)
            self.handle = None
