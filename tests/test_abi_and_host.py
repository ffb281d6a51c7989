"""CPU-only checks of the shipped library: it loads, exports every symbol the
header declares, and its host-side input surface (parameter parse, source
waveform, material specs, PGM reader, E coefficients, rp values) agrees with
the oracle bit for bit.  No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import binding
from conftest import DATA, ROOT


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fdtd_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fdtd_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = fdtd_amd.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/fdtd_amd.h but not exported"
    bound = {name for name, _, _ in binding.SYMBOLS}
    assert declared == bound, declared ^ bound


def test_struct_layouts_match_header():
    assert C.sizeof(binding.Options) == 16 * 4
    # nx..iterations, two paths, 4 scalars, count + 48 sources, 4 defaults, count + 512 jz, 10 constants
    assert C.sizeof(binding.Params) == 16 + 2048 + 16 + 4 + 192 + 16 + 4 + 2048 + 40


def test_params_load_matches_oracle(oracle):
    path = os.path.join(DATA, "input_params")
    a, b = fdtd_amd.params_load(path), oracle.load_params(path)
    for name, _ in binding.Params._fields_:
        if name in ("jz", "sources", "jz_count"):
            continue
        assert getattr(a, name) == getattr(b, name), name
    assert list(a.sources[:3]) == list(b.sources[:3]) == [49, 49, 49]
    ja, jb = np.array(a.jz[:], dtype=np.float32), np.array(b.jz[:binding.MAX_JZ], dtype=np.float32)
    assert np.array_equal(ja.view(np.uint32), jb.view(np.uint32))     # bit-exact, all 512 samples


@pytest.mark.parametrize("kw", [
    dict(wave_frequency=1.0e9, elements_per_wave=40, pulse_width=0.3),
    dict(wave_frequency=2.45e9, elements_per_wave=64, pulse_width=0.8, pulse_modulation_frequency=1.5),
    dict(wave_frequency=3.0e9, elements_per_wave=100, pulse_width=0.5, pulse_modulation_frequency=0.25),
])
def test_source_waveform_matches_oracle(oracle, kw):
    a = fdtd_amd.make_params(30, 30, 30, [(15, 15, 15)], **kw)
    b = oracle.make_params(30, 30, 30, [(15, 15, 15)], **kw)
    for name in ("dt", "dx", "mu0", "eps0", "lam"):
        assert getattr(a, name) == getattr(b, name), name
    ja, jb = np.array(a.jz[:], dtype=np.float32), np.array(b.jz[:binding.MAX_JZ], dtype=np.float32)
    assert np.abs(jb).max() > 0
    assert np.array_equal(ja.view(np.uint32), jb.view(np.uint32))


def test_params_errors(tmp_path):
    with pytest.raises(fdtd_amd.FdtdError, match="cannot open"):
        fdtd_amd.params_load(str(tmp_path / "missing"))
    bad = tmp_path / "truncated"
    bad.write_text("nx_ny_nz: 10 10\n")
    with pytest.raises(fdtd_amd.FdtdError, match="nx_ny_nz"):
        fdtd_amd.params_load(str(bad))
    text = open(os.path.join(DATA, "input_params")).read().replace(
        "number_of_excitation_sources: 1", "number_of_excitation_sources: 17")
    many = tmp_path / "many"
    many.write_text(text)
    with pytest.raises(fdtd_amd.FdtdError, match="number_of_excitation_sources"):
        fdtd_amd.params_load(str(many))


def test_material_specs_and_pgm_match_oracle(oracle):
    specs = os.path.join(DATA, "mat_specs_riken")
    table, present = binding.read_material_specs(specs)
    assert present.sum() == 53 and present[94] and not present[17]
    assert np.all(table[~present] == 0)
    codes = binding.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), 265, 490)
    assert sorted(np.unique(codes).tolist()) == [0, 18, 31, 45, 48, 49, 51]
    # the oracle's loadMaterials: per-cell floats must be table[code]
    p = oracle.make_params(265, 490, 1, [(10, 10, 0)])
    o = oracle.Oracle(p)
    o.load_materials(specs, os.path.join(DATA, "pgmdata"), first_slice=800)
    for which in range(4):
        assert np.array_equal(o.material(which)[0], table[codes, which])
    # sequential-token semantics: a narrower grid re-wraps the stream (fdtd.cu:922-931)
    narrow = binding.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), 100, 50)
    assert np.array_equal(narrow.ravel(), codes.ravel()[:5000])
    # nearest-neighbour upsampling and row windows
    up = binding.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), 265 * 3, 490 * 3, upsample=3,
                                y0=600, rows=30)
    assert np.array_equal(up, np.repeat(np.repeat(codes, 3, axis=0), 3, axis=1)[600:630])
    with pytest.raises(fdtd_amd.FdtdError, match="upsample"):
        binding.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00800.pgm"), 100, 50, upsample=3)
    with pytest.raises(fdtd_amd.FdtdError, match="open"):
        binding.read_pgm_codes(os.path.join(DATA, "pgmdata", "v1_00001.pgm"), 10, 10)


def test_e_coefficients_and_rp_match_oracle(oracle):
    specs = os.path.join(DATA, "mat_specs_riken")
    table, present = binding.read_material_specs(specs)
    rows = [tuple(float(v) for v in table[c]) for c in np.nonzero(present)[0]]
    rows += [(0.0, 1.0, 1.0, 0.0), (5.0, 80.0, 4.0, 9.0e-12), (0.01, 2.5, 2.5, 0.0), (1e-3, 3.0, 1.5, 2e-10)]
    for freq in (3.0e9, 0.9e9):
        pa = fdtd_amd.make_params(16, 16, 16, [(8, 8, 8)], wave_frequency=freq)
        for ecoef in (0, 1):
            for (sg, es, ei, tau) in rows:
                po = oracle.make_params(4, 4, 4, [(2, 2, 2)], sigma=sg, eps_s=es, eps_i=ei, tau_d=tau,
                                        wave_frequency=freq)
                o = oracle.Oracle(po, ecoef=ecoef)
                a_, b_, c_, d_ = o.cell_coefs(0)
                acd, k = binding.material_coefficients(pa, ecoef, sg, es, ei, tau)
                assert (acd[0], acd[1], acd[2]) == (a_, c_, d_), (sg, es, ei, tau)
                dt, tf = np.float32(pa.dt), np.float32(tau)
                assert k[0] == 2.0 * float(dt + tf) and k[1] == 2.0 * float(dt) + 4.0 * float(tf) \
                    and k[2] == 2.0 * float(tf)
                if ecoef == 0:
                    assert binding.rp_value(pa, 1, sg, es, ei, tau) == float(o.rp(0)[0])
                    assert binding.rp_value(pa, 0, sg, es, ei, tau) == float(o.rp(1)[0])
                    assert float(o.rp(1)[0]) == float(o.rp(2)[0])


def test_partition_rows():
    assert fdtd_amd.partition_rows(10, 1) == [(0, 10)]
    assert fdtd_amd.partition_rows(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert fdtd_amd.partition_rows(65536, 8) == [(i * 8192, 8192) for i in range(8)]
    with pytest.raises(ValueError):
        fdtd_amd.partition_rows(5, 3)
