"""fdtd_amd -- MI355X-native FDTD field-update path.

The product is the C-ABI library `libfdtd_amd.so` (hand-written HIP kernels for
gfx950 plus C++ host code, see include/fdtd_amd.h) and the `fdtd` command-line
driver.  This package is the thin Python binding the tests and the benchmark
drive it through; it loads the in-tree library and fails loudly if it is absent.
"""
from .binding import (  # noqa: F401
    ECOEF_CONSISTENT, ECOEF_REFERENCE, FIELD_NAMES, KERNEL_AUTO, KERNEL_FUSED, KERNEL_SPLIT,
    MODE_2D_TMZ, MODE_3D, MODE_AUTO, FdtdError, Options, Params, Solver, library_path, load_library,
    make_params, params_load,
)
from .slab import SlabRunner, partition_rows  # noqa: F401
