"""TEST INFRASTRUCTURE.  ctypes wrapper of oracle/_ref/libfdtd_ref.so: the reference's own
kernels (fdtd_calculations.cu:32-428) built for gfx950 by `make -C oracle ref` (see
oracle/ref_driver.hip).  Needs a GPU to run.  Only tests may import this module."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libfdtd_ref.so")
_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
        _lib.fdtd_ref_run.restype = C.c_int
        _lib.fdtd_ref_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib.fdtd_ref_time.restype = C.c_float
        _lib.fdtd_ref_time.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    return _lib


def time_per_step(params, warm=3, nsteps=12):
    """Milliseconds per step of the reference's kernels on a vacuum grid of params' extents."""
    p = params
    dims = (C.c_int * 3)(p.nx, p.ny, p.nz)
    consts = (C.c_float * 6)(p.dt, p.dx, p.dy, p.dz, p.mu0, p.eps0)
    jz = (C.c_float * len(p.jz))(*p.jz)
    return float(lib().fdtd_ref_time(dims, consts, p.jz_count, jz, warm, nsteps))


def run_like(o, nsteps):
    """Advance a copy of the state of the 3-D oracle `o` by nsteps with the reference's kernels.
    Returns {field id: [array of level 0, level 1, level 2]} in the oracle's level convention
    (level 0 = newest) as of step o.steps_done + nsteps, H with one level."""
    from . import oracle as orc
    p = o.p
    n = p.nx * p.ny * p.nz
    it0 = o.steps_done
    # reference buffer b holds the oracle's level (it - 1 - b) mod 3
    level_of = lambda it, b: (it - 1 - b) % 3
    h = np.stack([np.array(o.field(orc.HX + k)).ravel() for k in range(3)]).astype(np.float32)
    e = np.stack([np.array(o.field(orc.EX + k, level_of(it0, b))).ravel() for b in range(3) for k in range(3)])
    d = np.stack([np.array(o.field(orc.DX + k, level_of(it0, b))).ravel() for b in range(3) for k in range(3)])
    e, d = np.ascontiguousarray(e, dtype=np.float32), np.ascontiguousarray(d, dtype=np.float32)
    # updateEField takes sigma, epsI, epsS, tauD (fdtd_calculations.h:34)
    mat = np.ascontiguousarray(np.stack([np.array(o.material(w)).ravel() for w in (0, 2, 1, 3)]), dtype=np.float32)
    rps = [np.ascontiguousarray(o.rp(f), dtype=np.float32) for f in range(6)]
    rp_ptr = (C.c_void_p * 6)(*[r.ctypes.data for r in rps])
    rp_n = (C.c_size_t * 6)(*[r.size for r in rps])
    dims = (C.c_int * 3)(p.nx, p.ny, p.nz)
    consts = (C.c_float * 6)(p.dt, p.dx, p.dy, p.dz, p.mu0, p.eps0)
    src = (C.c_int * (3 * max(1, p.sources_count)))(*list(p.sources)[:3 * p.sources_count])
    jz = (C.c_float * len(p.jz))(*p.jz)
    rc = lib().fdtd_ref_run(dims, consts, p.sources_count, src, p.jz_count, jz, h.ctypes.data, e.ctypes.data,
                            d.ctypes.data, mat.ctypes.data, rp_ptr, rp_n, it0, nsteps)
    if rc != 0:
        raise RuntimeError(f"fdtd_ref_run = {rc}")
    it1 = it0 + nsteps
    shape = (p.nz, p.ny, p.nx)
    out = {}
    for k in range(3):
        out[orc.HX + k] = [h[k].reshape(shape)]
        # level l sits in buffer (it1 - 1 - l) mod 3
        out[orc.EX + k] = [e[((it1 - 1 - l) % 3) * 3 + k].reshape(shape) for l in range(3)]
        out[orc.DX + k] = [d[((it1 - 1 - l) % 3) * 3 + k].reshape(shape) for l in range(3)]
    return out
