"""TEST INFRASTRUCTURE.  ctypes wrapper of oracle/_ref/libfdtd_ref_host.so: the reference's own
host set-up functions -- initParamsWithPath (fdtd.cu:547-637), loadMaterials (fdtd.cu:875-939),
setupMurBoundary (fdtd.cu:942-1024), setupSources (fdtd.cu:1027-1082) -- compiled as C by
`make -C oracle ref_host` (see oracle/ref_host_driver.c).  CPU only.  Only tests and the fixture
generator under tests/golden/ may import this module."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libfdtd_ref_host.so")
_lib = None

INT_NAMES = ("nx", "ny", "nz", "iterations", "elements_per_wave", "sources_count")
FLOAT_NAMES = ("wave_frequency", "pulse_width", "pulse_modulation_frequency", "default_sigma", "default_eps_s",
               "default_eps_i", "default_tau_d", "pi", "c", "timeskip", "lam", "dt", "dx", "dy", "dz", "mu0", "eps0")
JZ_TABLE = 16384


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(PATH)
        vp = C.c_void_p
        L.refh_params_load.restype = vp
        L.refh_params_load.argtypes = [C.c_char_p, vp, vp, C.c_char_p, C.c_char_p, vp, C.c_int]
        L.refh_params_set.restype = None
        L.refh_params_set.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.refh_setup_sources.restype = None
        L.refh_setup_sources.argtypes = [vp, vp, C.c_int]
        L.refh_load_materials.restype = None
        L.refh_load_materials.argtypes = [vp, C.c_char_p, C.c_char_p, vp, vp, vp, vp]
        L.refh_setup_mur.restype = None
        L.refh_setup_mur.argtypes = [vp, vp, vp, vp, vp, vp]
        L.refh_params_free.restype = None
        L.refh_params_free.argtypes = [vp]
        _lib = L
    return _lib


class RefParams:
    """FdtdParams as initParamsWithPath(path) returns it."""

    def __init__(self, path):
        ints = np.zeros(8, dtype=np.int32)
        floats = np.zeros(17, dtype=np.float32)
        inp, out = C.create_string_buffer(1024), C.create_string_buffer(1024)
        src = np.zeros(16 * 3, dtype=np.int32)
        self._h = lib().refh_params_load(os.fsencode(path), ints.ctypes.data, floats.ctypes.data, inp, out,
                                         src.ctypes.data, 16)
        if not self._h:
            raise FileNotFoundError(path)
        for name, v in zip(INT_NAMES, ints):
            setattr(self, name, int(v))
        for name, v in zip(FLOAT_NAMES, floats):
            setattr(self, name, np.float32(v))
        self.input_path, self.output_path = inp.value, out.value
        self.sources = src[:3 * self.sources_count].copy()

    def set(self, nx, ny, nz, sigma, eps_s, eps_i, tau_d):
        """Extents and material defaults of a parameter set made by hand."""
        lib().refh_params_set(self._h, nx, ny, nz, sigma, eps_s, eps_i, tau_d)
        self.nx, self.ny, self.nz = nx, ny, nz

    def setup_sources(self):
        """params->jz after setupSources: all 16384 entries."""
        jz = np.zeros(JZ_TABLE, dtype=np.float32)
        lib().refh_setup_sources(self._h, jz.ctypes.data, JZ_TABLE)
        return jz

    def load_materials(self, specs_path, pgm_dir):
        """(sigma, eps_s, eps_i, tau_d) arrays of shape (nz, ny, nx) as loadMaterials fills them."""
        shape = (self.nz, self.ny, self.nx)
        arrs = [np.zeros(shape, dtype=np.float32) for _ in range(4)]
        lib().refh_load_materials(self._h, os.fsencode(specs_path), os.fsencode(pgm_dir), *[a.ctypes.data for a in arrs])
        return arrs

    def setup_mur(self, sigma, eps_s, eps_i, tau_d):
        """The six rp planes (x0, y0, z0, xEnd, yEnd, zEnd) of setupMurBoundary on the given per-cell arrays."""
        nx, ny, nz = self.nx, self.ny, self.nz
        sizes = [2 * ny * nz, nx * 2 * nz, nx * ny * 2] * 2
        planes = [np.zeros(n, dtype=np.float32) for n in sizes]
        ptrs = (C.c_void_p * 6)(*[p.ctypes.data for p in planes])
        mats = [np.ascontiguousarray(a, dtype=np.float32) for a in (sigma, eps_s, eps_i, tau_d)]
        lib().refh_setup_mur(self._h, *[a.ctypes.data for a in mats], ptrs)
        return planes

    def close(self):
        if self._h:
            lib().refh_params_free(self._h)
            self._h = None

    def __del__(self):
        self.close()
