"""Register budget of the multi-step 2-D sweep, checked on the gfx950 assembly hipcc emits (no GPU
needed).  At 8 steps per pass the kernel sits a few registers below the 256 a wave may hold at two
waves per SIMD; a change that tips the plain marching loop -- the one nearly every wave runs --
into spilling costs a factor of two in pass time while every parity test stays green."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc here")
    out = tmp_path_factory.mktemp("isa") / "sweep2d_tb.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include",
                           f"-I{ROOT}/fdtd_amd/csrc", "-S", "--cuda-device-only", f"{ROOT}/fdtd_amd/csrc/sweep2d_tb.hip", "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


def kernel_body(s, k, one, plain, wide):
    name = re.findall(r"\.globl\s+(\S*fdtd2d_sweep_tbILi%dELb%dELb%dELb%dE\S*)" % (k, one, plain, wide), s)[0]
    i = s.index(name + ":")
    return s[i:s.index("; Occupancy:", i) + 40]


def marching_loops(body):
    """(first line, last line, lines) of every innermost loop that stores the six rows of a pass."""
    lines = body.splitlines()
    labels = {m.group(1): n for n, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = set()
    for n, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            loops.add((labels[m.group(1)], n))
    picked = []
    for a, b in sorted(loops, key=lambda t: t[1] - t[0]):
        seg = lines[a:b + 1]
        if sum(1 for l in seg if re.match(r"\s+global_store", l)) != 6:
            continue
        picked.append((a, b, seg))
    return picked


@pytest.mark.parametrize("variant", [(8, 1, 1, 0), (8, 0, 1, 1), (7, 0, 1, 0), (6, 1, 1, 0)])
def test_plain_marching_loop_keeps_its_state_in_registers(assembly, variant):
    body = kernel_body(assembly, *variant)
    assert int(re.search(r"; Occupancy: (\d+)", body).group(1)) >= 2          # two waves per SIMD
    loops = marching_loops(body)
    assert loops, "no marching loop found"
    # a marching loop loads six rows and stores six; the copy without guards is the shortest of them
    loops = [t for t in loops if sum(1 for l in t[2] if re.match(r"\s+global_load_dwordx4", l)) == 6]
    assert loops, "no loop with six row loads and six row stores"
    plain = min(loops, key=lambda t: sum(1 for l in t[2] if re.match(r"\s+v_", l)))
    spills = [l.strip() for l in plain[2] if re.match(r"\s+scratch_", l)]
    assert not spills, f"the plain marching loop of K={variant[0]} spills: {spills[:4]}"
