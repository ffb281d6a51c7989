#!/bin/bash
# After the cheaper pipeline fill: band height (in rounds of the 2048 wave slots) and steps per pass
# again, from the tiled sweep's sizes up to the headline's.  One line per run.
cd $GRAFT_REPO_ROOT
out=gpurun_out/fillscan; mkdir -p $out
run() { # nx ny label args...
  local nx=$1 ny=$2 label=$3; shift 3
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --nx $nx --ny $ny "$@" > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; exit 1; }
  python - $nx $ny "$label" <<'PY'
import json,sys
d=json.load(open("gpurun_out/fillscan/run.json")); r=d["roofline"]
print("%5sx%-5s %-28s pass %.4f ms  %7.1f Gcell/s  frac %.3f  %s K=%s" % (sys.argv[1],sys.argv[2],sys.argv[3],r["kernel_avg_ms"],d["value"]/1e3,r["frac"],d["config"]["kernel"],r["steps_per_launch"]), flush=True)
PY
}
while read nx ny steps rounds_list; do
  unset FDTD_TB_ROWS
  run $nx $ny auto --steps $steps --warmup 16
  strips=$(( ( (nx + 3) / 4 + 59 ) / 60 ))
  for K in 6 8; do
    for rounds in $rounds_list; do
      r=$(python3 -c "print(max(8, int(round($ny * $strips / (2048 * $rounds) / 4)) * 4))")
      [ $r -gt 240 ] && continue
      export FDTD_TB_ROWS=$r
      run $nx $ny "K=$K rounds=$rounds rows=$r" --steps $((K*30)) --warmup $((2*K)) --kernel multistep --steps-per-pass $K
    done
  done
done <<'LIST'
2048 2048 240 0.5 0.7 1.0 1.45
3072 3072 240 0.7 1.0 1.2 1.45 1.9
4096 4096 240 1.0 1.2 1.45 1.9 2.4
6144 6144 240 1.2 1.45 1.9 2.4 3.0
8192 8192 240 1.45 1.9 2.4 3.0 3.6
LIST
