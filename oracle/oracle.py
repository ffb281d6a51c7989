"""ctypes binding of the CPU oracle (oracle/libfdtd_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py -- never by the fdtd_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfdtd_oracle.so")

MAX_SOURCES = 16
MAX_JZ = 512
JZ_TABLE = 16384

ECOEF_REFERENCE, ECOEF_CONSISTENT = 0, 1
MODE_3D, MODE_2D_TMZ = 0, 1
HX, HY, HZ, DX, DY, DZ, EX, EY, EZ = range(9)
FIELD_NAMES = ["hx", "hy", "hz", "dx", "dy", "dz", "ex", "ey", "ez"]


class Params(C.Structure):
    """Mirror of orc_params (itself the mirror of FdtdParams, fdtd.h:7-34)."""

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
        ("jz_count", C.c_int32), ("jz", C.c_float * JZ_TABLE),
        ("pi", C.c_float), ("c", C.c_float), ("timeskip", C.c_float), ("lam", C.c_float),
        ("dt", C.c_float), ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float),
        ("mu0", C.c_float), ("eps0", C.c_float),
    ]


def build(force=False):
    """Compile the oracle in place (gcc, seconds)."""
    src = os.path.join(_HERE, "fdtd_oracle.c")
    hdr = os.path.join(_HERE, "fdtd_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libfdtd_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, ci = C.c_void_p, C.c_int
        PP = C.POINTER(Params)
        L.orc_params_load.argtypes = [C.c_char_p, PP]
        L.orc_params_derive.argtypes = [PP]
        L.orc_params_derive.restype = None
        L.orc_setup_sources.argtypes = [PP]
        L.orc_create.argtypes = [PP, ci, ci]
        L.orc_create.restype = vp
        L.orc_create_slab2d.argtypes = [PP, ci, ci, ci]
        L.orc_create_slab2d.restype = vp
        L.orc_destroy.argtypes = [vp]
        L.orc_destroy.restype = None
        L.orc_load_materials.argtypes = [vp, C.c_char_p, C.c_char_p, ci]
        L.orc_set_materials.argtypes = [vp, C.c_void_p, C.c_void_p, ci]
        L.orc_setup_mur.argtypes = [vp]
        L.orc_setup_mur.restype = None
        L.orc_set_rp.argtypes = [vp, ci, C.c_void_p, C.c_size_t]
        L.orc_rp.argtypes = [vp, ci, C.POINTER(C.c_size_t)]
        L.orc_rp.restype = C.POINTER(C.c_float)
        L.orc_step.argtypes = [vp, ci, ci]
        L.orc_step.restype = None
        L.orc_steps_done.argtypes = [vp]
        L.orc_field.argtypes = [vp, ci]
        L.orc_field.restype = C.POINTER(C.c_float)
        L.orc_field_level.argtypes = [vp, ci, ci]
        L.orc_field_level.restype = C.POINTER(C.c_float)
        L.orc_field_count.argtypes = [vp]
        L.orc_field_count.restype = C.c_size_t
        L.orc_material.argtypes = [vp, ci]
        L.orc_material.restype = C.POINTER(C.c_float)
        L.orc_cell_coefs.argtypes = [vp, C.c_size_t, C.c_void_p]
        L.orc_cell_coefs.restype = None
        L.orc_time_steps.argtypes = [vp, ci, ci]
        L.orc_time_steps.restype = C.c_double
        _lib = L
    return _lib


def load_params(path, setup_sources=True):
    p = Params()
    rc = lib().orc_params_load(os.fsencode(path), C.byref(p))
    if rc != 0:
        raise RuntimeError(f"orc_params_load({path}) = {rc}")
    if setup_sources:
        rc = lib().orc_setup_sources(C.byref(p))
        if rc != 0:
            raise RuntimeError(f"orc_setup_sources = {rc}")
    return p


def make_params(nx, ny, nz, sources, sigma=0.0, eps_s=1.0, eps_i=1.0, tau_d=0.0,
                elements_per_wave=100, wave_frequency=3.0e9, pulse_width=0.5,
                pulse_modulation_frequency=0.0, iterations=3):
    """Parameters filled in by hand; sources are 0-based (x, y, z) triples."""
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
    lib().orc_params_derive(C.byref(p))
    rc = lib().orc_setup_sources(C.byref(p))
    if rc != 0:
        raise RuntimeError(f"orc_setup_sources = {rc}")
    return p


class Oracle:
    """One oracle state.  Fields come back as numpy views of the C arrays."""

    def __init__(self, params, mode=None, ecoef=ECOEF_REFERENCE, slab=None):
        self.p = params
        self.mode = (MODE_2D_TMZ if params.nz == 1 else MODE_3D) if mode is None else mode
        self.slab = slab
        if slab is not None:
            gy0, nyl = slab
            self._s = lib().orc_create_slab2d(C.byref(params), ecoef, gy0, nyl)
        else:
            self._s = lib().orc_create(C.byref(params), self.mode, ecoef)
        if not self._s:
            raise RuntimeError("orc_create failed")

    def close(self):
        if self._s:
            lib().orc_destroy(self._s)
            self._s = None

    def __del__(self):
        self.close()

    @property
    def shape(self):
        """Storage shape (z, y, x); a 2-D state stores nyl + 2 rows (halo rows)."""
        p = self.p
        if self.mode == MODE_2D_TMZ:
            nyl = self.slab[1] if self.slab else p.ny
            return (1, nyl + 2, p.nx)
        return (p.nz, p.ny, p.nx)

    def _view(self, ptr):
        n = lib().orc_field_count(self._s)
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(self.shape)

    def field_storage(self, fid, level=0):
        return self._view(lib().orc_field_level(self._s, fid, level))

    def field(self, fid, level=0):
        """Field without halo rows: (nz, ny, nx) or, in 2-D, (1, nyl, nx)."""
        a = self.field_storage(fid, level)
        return a[:, 1:-1, :] if self.mode == MODE_2D_TMZ else a

    def set_field(self, fid, data, level=0):
        self.field(fid, level)[...] = np.asarray(data, dtype=np.float32).reshape(self.field(fid, level).shape)

    def load_materials(self, specs_path, pgm_dir, first_slice=1):
        rc = lib().orc_load_materials(self._s, os.fsencode(specs_path), os.fsencode(pgm_dir), first_slice)
        if rc != 0:
            raise RuntimeError(f"orc_load_materials = {rc}")

    def set_materials(self, codes, table):
        """codes: uint8 array over the cells (no halo rows); table: (ncodes, 4)
        rows of (sigma, eps_s, eps_i, tau_d)."""
        codes = np.asarray(codes, dtype=np.uint8)
        table = np.ascontiguousarray(table, dtype=np.float32)
        full = np.zeros(self.shape, dtype=np.uint8)
        if self.mode == MODE_2D_TMZ:
            full[:, 1:-1, :] = codes.reshape(1, self.shape[1] - 2, self.shape[2])
        else:
            full[...] = codes.reshape(self.shape)
        rc = lib().orc_set_materials(self._s, full.ctypes.data, table.ctypes.data, table.shape[0])
        if rc != 0:
            raise RuntimeError(f"orc_set_materials = {rc}")

    def material(self, which):
        """Per-cell sigma (0), eps_s (1), eps_i (2) or tau_d (3), halo rows dropped."""
        a = self._view(lib().orc_material(self._s, which))
        return a[:, 1:-1, :] if self.mode == MODE_2D_TMZ else a

    def cell_coefs(self, cell):
        out = np.zeros(4, dtype=np.float32)
        lib().orc_cell_coefs(self._s, cell, out.ctypes.data)
        return out

    def setup_mur(self):
        lib().orc_setup_mur(self._s)

    def rp(self, face):
        n = C.c_size_t()
        ptr = lib().orc_rp(self._s, face, C.byref(n))
        return np.ctypeslib.as_array(ptr, shape=(n.value,))

    def set_rp(self, face, data):
        data = np.ascontiguousarray(data, dtype=np.float32).ravel()
        if lib().orc_set_rp(self._s, face, data.ctypes.data, data.size) != 0:
            raise RuntimeError("orc_set_rp: wrong size")

    def step(self, nsteps=1, threads=1):
        lib().orc_step(self._s, nsteps, threads)

    def time_steps(self, nsteps, threads=1):
        return lib().orc_time_steps(self._s, nsteps, threads)

    @property
    def steps_done(self):
        return lib().orc_steps_done(self._s)
