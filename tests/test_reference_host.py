"""The input surface against the reference's OWN host functions: initParamsWithPath,
setupSources, loadMaterials and setupMurBoundary of fdtd.cu compiled as C (oracle/_ref/
libfdtd_ref_host.so, `make -C oracle ref_host`, built where /root/reference exists).  The oracle's
restatement (orc_params_load, orc_setup_sources, orc_load_materials, orc_setup_mur) and the
library's host code (fdtd_params_load, fdtd_read_material_specs, fdtd_read_pgm_codes, fdtd_rp_value)
must agree with them bit for bit.  CPU only.  The same values, recorded from these functions, are
kept in tests/golden/reference_host_vectors.json for machines without the reference
(tests/test_oracle_golden.py checks the oracle against that file)."""
import os

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import binding
from conftest import DATA
from oracle import reference_host as RH

pytestmark = pytest.mark.skipif(not RH.available(), reason="oracle/_ref/libfdtd_ref_host.so is built only where "
                                "the reference tree exists (make -C oracle ref_host)")

SHIPPED = os.path.join(DATA, "input_params")
SPECS = os.path.join(DATA, "mat_specs_riken")
VARIANTS = [dict(), dict(wave_freq="1.0e9", elements_per_wavelength="40", pulse_width="0.3"),
            dict(wave_freq="2.45e9", elements_per_wavelength="64", pulse_width="0.8", pulse_modulation_frequency="1.5"),
            dict(pulse_modulation_frequency="0.25", number_of_excitation_sources="3"),
            dict(fsigma="0.9", feps_s="52.7", feps_inf="4.0", ftau_d="8.4e-12", nx_ny_nz="37 29 23")]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def write_params(tmp_path, **changes):
    """data/input_params with the values of some keys replaced (and source lines added)."""
    out = []
    for line in open(SHIPPED):
        key = line.split(":")[0]
        if key in changes:
            line = f"{key}: {changes[key]}\n"
        out.append(line)
        if key == "source_location" and "number_of_excitation_sources" in changes:
            for i in range(int(changes["number_of_excitation_sources"]) - 1):
                out.append(f"source_location: {10 + i} {20 + i} {30 + i}\n")
    path = tmp_path / "input_params"
    path.write_text("".join(out))
    return str(path)


@pytest.mark.parametrize("changes", VARIANTS)
def test_params_and_source_waveform(oracle, tmp_path, changes):
    """initParamsWithPath + setupSources: every scalar, the sources, and all 16384 jz samples
    (the kernels see the first 512)."""
    path = write_params(tmp_path, **changes)
    ref = RH.RefParams(path)
    orc = oracle.load_params(path)
    mine = fdtd_amd.params_load(path)
    for name in RH.INT_NAMES:
        assert getattr(orc, name) == getattr(ref, name) == getattr(mine, name), name
    for name in RH.FLOAT_NAMES:
        want = np.float32(getattr(ref, name))
        assert np.float32(getattr(orc, name)).view(np.uint32) == want.view(np.uint32), name
        assert np.float32(getattr(mine, name)).view(np.uint32) == want.view(np.uint32), name
    n = 3 * ref.sources_count
    assert list(orc.sources[:n]) == list(ref.sources) == list(mine.sources[:n])
    assert orc.input_path == ref.input_path == mine.input_path
    assert orc.output_path == ref.output_path == mine.output_path
    jz = ref.setup_sources()
    assert np.abs(jz).max() > 0
    assert np.array_equal(bits(orc.jz[:]), bits(jz))
    assert np.array_equal(bits(mine.jz[:]), bits(jz[:binding.MAX_JZ]))
    ref.close()


def write_pgm(path, nx, ny, codes):
    with open(path, "w") as f:
        f.write(f"P2\n# test\n{nx} {ny}\n94\n")
        for row in np.asarray(codes).reshape(ny, nx):
            f.write(" ".join(str(int(c)) for c in row) + "\n")


@pytest.mark.parametrize("dims", [(265, 490, 3), (100, 120, 4)])
def test_load_materials_on_the_shipped_slice(oracle, tmp_path, dims):
    """loadMaterials on data/pgmdata/v1_00800.pgm (as slices 1..nz): at the file's own size and
    on a narrower grid, where the token stream is re-wrapped instead of cropped (fdtd.cu:922-931)."""
    nx, ny, nz = dims
    for iz in range(nz):
        os.symlink(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), tmp_path / f"v1_{iz + 1:05d}.pgm")
    ref = RH.RefParams(SHIPPED)
    ref.set(nx, ny, nz, 0.0, 1.0, 1.0, 0.0)
    want = ref.load_materials(SPECS, str(tmp_path))
    o = oracle.Oracle(oracle.make_params(nx, ny, nz, [(5, 5, 1)]), mode=oracle.MODE_3D)
    o.load_materials(SPECS, str(tmp_path))
    table, _ = binding.read_material_specs(SPECS)
    codes = binding.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), nx, ny)
    assert len(np.unique(codes)) > 1
    for which in range(4):
        assert np.array_equal(bits(o.material(which)), bits(want[which])), which
        for iz in range(nz):
            assert np.array_equal(bits(table[codes, which]), bits(want[which][iz])), which
    ref.close()


def test_load_materials_every_code_and_mur_planes(oracle, tmp_path):
    """Random slices over every code of mat_specs_riken (code 94, which overruns the reference's
    table, included), tissue on all six faces: loadMaterials, then setupMurBoundary on the loaded
    arrays -- all six planes, the x faces with their real-valued c1, c2 (fdtd.cu:959-960)."""
    nx, ny, nz = 23, 19, 7
    table, present = binding.read_material_specs(SPECS)
    rng = np.random.default_rng(42)
    all_codes = rng.choice(np.flatnonzero(present), size=(nz, ny, nx))
    for iz in range(nz):
        write_pgm(tmp_path / f"v1_{iz + 1:05d}.pgm", nx, ny, all_codes[iz])
    ref = RH.RefParams(SHIPPED)
    ref.set(nx, ny, nz, 0.0, 1.0, 1.0, 0.0)
    mats = ref.load_materials(SPECS, str(tmp_path))
    o = oracle.Oracle(oracle.make_params(nx, ny, nz, [(5, 5, 1)]))
    o.load_materials(SPECS, str(tmp_path))
    for which in range(4):
        assert np.array_equal(bits(o.material(which)), bits(mats[which])), which
        assert np.array_equal(bits(table[all_codes, which]), bits(mats[which])), which
    planes = ref.setup_mur(*mats)
    o.setup_mur()
    p = fdtd_amd.params_load(SHIPPED)
    for face in range(6):
        assert len(np.unique(planes[face])) > 10
        assert np.array_equal(bits(o.rp(face)), bits(planes[face])), face
    # the library's value function on the cells behind the plane entries
    rpx0 = planes[0].reshape(nz, ny, 2)
    rpyE = planes[4].reshape(nz, 2, nx)
    rpz0 = planes[2].reshape(2, ny, nx)
    for z, y, x in [(0, 0, 0), (3, 7, 1), (6, 18, 0)]:
        assert np.float32(binding.rp_value(p, 1, *[float(v) for v in table[all_codes[z, y, x]]])) == rpx0[z, y, x]
    for z, k, x in [(0, 0, 0), (4, 1, 11), (6, 1, 22)]:
        assert np.float32(binding.rp_value(p, 0, *[float(v) for v in table[all_codes[z, ny - 2 + k, x]]])) == rpyE[z, k, x]
    for k, y, x in [(0, 0, 0), (1, 9, 13)]:
        assert np.float32(binding.rp_value(p, 0, *[float(v) for v in table[all_codes[k, y, x]]])) == rpz0[k, y, x]
    ref.close()


@pytest.mark.parametrize("mat", [(0.0, 1.0, 1.0, 0.0), (0.9, 52.7, 4.0, 8.4e-12), (5.0, 80.0, 4.0, 9.0e-12),
                                 (1e-3, 3.0, 1.5, 2e-10)])
def test_mur_planes_from_the_defaults(oracle, mat):
    """What the reference actually runs: setupMurBoundary before loadMaterials (fdtd.cu:59 vs :65),
    on arrays that hold the fsigma/feps_s/feps_inf/ftau_d defaults in every cell."""
    nx, ny, nz = 9, 8, 7
    ref = RH.RefParams(SHIPPED)
    ref.set(nx, ny, nz, *mat)
    cells = [np.full((nz, ny, nx), v, dtype=np.float32) for v in mat]
    planes = ref.setup_mur(*cells)
    o = oracle.Oracle(oracle.make_params(nx, ny, nz, [(4, 4, 4)], sigma=mat[0], eps_s=mat[1], eps_i=mat[2], tau_d=mat[3]))
    p = fdtd_amd.params_load(SHIPPED)
    for face in range(6):
        assert np.array_equal(bits(o.rp(face)), bits(planes[face])), face
        assert np.float32(binding.rp_value(p, face % 3 == 0, *mat)) == planes[face][0]
    ref.close()
