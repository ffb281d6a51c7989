#!/usr/bin/env python3
"""Instruction mix of the inner loops of the multi-step sweep, from the gfx950 assembly hipcc
emits (no GPU needed).  For each kernel instantiation: every loop (a label and the backward
branch to it), its instruction counts by class, and an issue-time estimate per iteration
(wave64 on a SIMD-32: fp32/integer vector instruction 2 cycles, fp64 vector instruction 4;
scalar, LDS and memory instructions issue from their own ports and are listed, not priced).
Writes profiles/isa_counts.json: the loop bench.py prices is the march of the waves that lie
inside every guard (the shortest loop that loads from global memory and writes to it)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fdtd_amd", "csrc", "sweep2d_tb.hip")


def assembly():
    out = "/tmp/fdtd_tb_isa.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           f"-I{ROOT}/include", f"-I{ROOT}/fdtd_amd/csrc", "-S", "--cuda-device-only", SRC, "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def classify(op):
    if op.startswith("v_"):
        return "valu_f64" if "f64" in op else "valu_f32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def loops_of(lines, start, end):
    """Innermost-or-not loops of one function: (label, first line, last line)."""
    labels = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    found = []
    for i in range(start, end):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i]) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            found.append((m.group(1), labels[m.group(1)], i))
    return found


def count(lines, a, b):
    c = {"valu_f32": 0, "valu_f64": 0, "lds": 0, "vmem": 0, "salu": 0, "other": 0, "vmem_load": 0, "vmem_store": 0}
    for i in range(a, b + 1):
        m = re.match(r"^\s+([a-z_0-9]+)\s", lines[i] + " ")
        if not m or lines[i].lstrip().startswith((".", ";")):
            continue
        op = m.group(1)
        c[classify(op)] += 1
        if op.startswith(("global_load", "buffer_load")):
            c["vmem_load"] += 1
        if op.startswith(("global_store", "buffer_store")):
            c["vmem_store"] += 1
    c["total"] = sum(c[k] for k in ("valu_f32", "valu_f64", "lds", "vmem", "salu", "other"))
    c["valu_issue_cycles"] = 2 * c["valu_f32"] + 4 * c["valu_f64"]
    return c


def main():
    lines = assembly()
    funcs = [(i, m.group(1)) for i, l in enumerate(lines)
             if (m := re.match(r"^(_ZN4fdtd\S*fdtd2d_sweep_tb\S*):", l))]
    result = {}
    for n, (start, name) in enumerate(funcs):
        end = next((i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        m = re.search(r"ILi(\d+)ELb(\d)ELb(\d)ELb(\d)E", name)
        kind = "one" if m.group(2) == "1" else ("table48" if m.group(4) == "1" else "table16")
        key = f"K{m.group(1)}_{kind}_{'plain' if m.group(3) == '1' else 'hinted'}"
        # A marching loop stores six rows and runs the E update of K stages x 4 cells (10 fp64
        # instructions each); the loops that fill the pipeline store nothing.  The plain copy is the
        # smallest such loop, the copy of the strips on the x faces the next one with six stores, the
        # fully guarded copy (sources, probes, Mur on the y faces: more stores, inner loops over the
        # source and probe lists -- its count is the span of its outermost back edge) the smallest with more.
        k = int(m.group(1))
        cands = []
        for label, a, b in loops_of(lines, start, end):
            c = count(lines, a, b)
            if c["vmem_load"] >= 6 and c["vmem_store"] >= 6 and c["valu_f64"] >= 24 * k:
                cands.append(c)
        if not cands:
            continue
        six = sorted((c for c in cands if c["vmem_store"] == 6 and c["valu_f64"] >= 40 * k), key=lambda c: c["total"])
        more = sorted((c for c in cands if c["vmem_store"] > 6), key=lambda c: c["total"])
        if not six:
            continue
        result[key] = {"inner_march_per_row_iteration": six[0]}
        xface = [c for c in six if c["total"] > 1.2 * six[0]["total"]]
        if xface:
            result[key]["x_face_march_per_row_iteration"] = xface[0]
        if more:
            result[key]["guarded_march_per_row_iteration"] = more[0]
    out = os.path.join(ROOT, "profiles", "isa_counts.json")
    json.dump(result, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(result.items()):
        c = v["inner_march_per_row_iteration"]
        print(f"{k}: {c['total']} instructions per row iteration (fp32/int VALU {c['valu_f32']}, fp64 VALU {c['valu_f64']}, "
              f"LDS {c['lds']}, memory {c['vmem']}, scalar {c['salu']}) -> {c['valu_issue_cycles']} VALU issue cycles")
    print("wrote", out)


if __name__ == "__main__":
    sys.exit(main())
