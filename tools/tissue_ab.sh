#!/bin/bash
# Tissue map (config 3) against vacuum on a grid of the same extents, and the scalar-row path of the
# table variant on and off; with build/old_lib/libfdtd_amd.so present (a build of an earlier commit)
# the same workloads on it too.  Run through gpurun from the repo root (the tree there is a scratch copy).
set -u
out=gpurun_out/tissue_ab; mkdir -p $out
run() { # name, env, args
    name=$1; shift; envs=$1; shift
    env $envs timeout -k 10 200 python bench.py "$@" --steps 200 --warmup 16 --no-cpu-baseline --no-one-step-leg > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; exit 1; }
    python - <<PY
import json
d=json.load(open("$out/$name.json")); r=d["roofline"]
print("%-28s %8.1f Gcell-updates/s  pass %.3f ms  frac %.3f  K=%d" % ("$name", d["value"]/1e3, r["kernel_avg_ms"], r["frac"], r["steps_per_launch"]))
PY
}
run tissue_scalar FDTD_TB_SCALAR_ROWS=1 --tissue 32 &&
run tissue_lds FDTD_TB_SCALAR_ROWS=0 --tissue 32 &&
run vacuum_same_extents X=1 --nx 8480 --ny 15680 &&
run vacuum_16384 X=1 || exit 1
if [ -f build/old_lib/libfdtd_amd.so ]; then
    cp fdtd_amd/libfdtd_amd.so /tmp/new_lib.so && cp build/old_lib/libfdtd_amd.so fdtd_amd/libfdtd_amd.so &&
    run old_tissue X=1 --tissue 32 &&
    run old_vacuum_same_extents X=1 --nx 8480 --ny 15680 &&
    run old_vacuum_16384 X=1
    cp /tmp/new_lib.so fdtd_amd/libfdtd_amd.so
fi
