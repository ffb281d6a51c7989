#!/bin/bash
# Steps per pass and kernel choice around the boundaries of the automatic selection (tiled sweep below
# 9 M cells, 6 steps per pass below 14 M).  One line per run.
cd $GRAFT_REPO_ROOT
out=gpurun_out/kscan; mkdir -p $out
run() {
  local label=$1; shift
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg "$@" > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; exit 1; }
  python - "$label" <<'PY'
import json,sys
d=json.load(open("gpurun_out/kscan/run.json")); r=d["roofline"]
print("%-34s pass %.4f ms  %7.1f Gcell/s  frac %.3f  %s K=%s" % (sys.argv[1],r["kernel_avg_ms"],d["value"]/1e3,r["frac"],d["config"]["kernel"],r["steps_per_launch"]), flush=True)
PY
}
for n in 2560 2896 3072 3328 3584 3840; do
  run "$n^2 auto" --nx $n --ny $n --steps 240 --warmup 24
  run "$n^2 multistep K=6" --nx $n --ny $n --steps 240 --warmup 24 --kernel multistep --steps-per-pass 6
  run "$n^2 multistep K=7" --nx $n --ny $n --steps 280 --warmup 28 --kernel multistep --steps-per-pass 7
  run "$n^2 multistep K=8" --nx $n --ny $n --steps 240 --warmup 24 --kernel multistep --steps-per-pass 8
done
