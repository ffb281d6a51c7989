#!/bin/bash
# A/B of library builds on the GPU box: the in-tree library and every build/variants/*/libfdtd_amd.so
# on a list of workloads (bench.py arguments, one per line on stdin).  The tree on the box is a scratch copy.
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/lib_ab; mkdir -p $out
cp fdtd_amd/libfdtd_amd.so /tmp/base_lib.so
mapfile -t loads
for lib in base $(ls build/variants 2>/dev/null); do
  if [ $lib = base ]; then cp /tmp/base_lib.so fdtd_amd/libfdtd_amd.so; else cp build/variants/$lib/libfdtd_amd.so fdtd_amd/libfdtd_amd.so; fi
  for args in "${loads[@]}"; do
    [ -z "$args" ] && continue
    timeout -k 10 200 python bench.py $args --no-cpu-baseline --no-one-step-leg > $out/run.json 2> $out/run.err || { echo "$lib $args failed"; tail -3 $out/run.err; cp /tmp/base_lib.so fdtd_amd/libfdtd_amd.so; exit 1; }
    python - "$lib" "$args" <<'PY'
import json,sys
d=json.load(open("gpurun_out/lib_ab/run.json")); r=d["roofline"]
print("%-10s %-60s %8.1f Gcell-updates/s  pass %.4f ms  frac %.3f  K=%d" % (sys.argv[1], sys.argv[2], d["value"]/1e3, r["kernel_avg_ms"], r["frac"], r["steps_per_launch"]), flush=True)
PY
  done
done
cp /tmp/base_lib.so fdtd_amd/libfdtd_amd.so
