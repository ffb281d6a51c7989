#!/bin/bash
# Mid-sized grids after the cheaper pipeline fill: rows that go in short bands at the end of a launch
# (FDTD_TB_TAIL) x height of those bands (FDTD_TB_TAIL_ROWS), around the library's n/8 and rows/3.
cd $GRAFT_REPO_ROOT
out=gpurun_out/tailscan2; mkdir -p $out
while read n rows; do
  for tail in auto 0 $((n/16)) $((n/8)) $((n/4)) $((n/2)); do
    for tr in $((rows/4)) $((rows/3)) $((rows/2)); do
      if [ $tail = auto ]; then unset FDTD_TB_TAIL FDTD_TB_TAIL_ROWS; [ $tr != $((rows/3)) ] && continue; else export FDTD_TB_TAIL=$tail FDTD_TB_TAIL_ROWS=$tr; fi
      [ $tail = 0 ] && [ $tr != $((rows/3)) ] && continue
      timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --nx $n --ny $n --steps 240 --warmup 16 > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; exit 1; }
      python - $n $tail $tr <<'PY'
import json,sys
d=json.load(open("gpurun_out/tailscan2/run.json")); r=d["roofline"]
print("%5s^2 tail %5s tail_rows %3s: pass %.4f ms  %7.1f Gcell/s" % (sys.argv[1],sys.argv[2],sys.argv[3],r["kernel_avg_ms"],d["value"]/1e3), flush=True)
PY
    done
  done
done <<'LIST'
4096 24
6144 52
8192 96
LIST
