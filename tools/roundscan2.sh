#!/bin/bash
# Rounds of the 2048 wave slots a pass of the multi-step sweep is cut into (FDTD_TB_ROUNDS overrides
# the library's rule; the 176-row cap stays), from 4096^2 to the tissue map.  One line per run.
cd $GRAFT_REPO_ROOT
out=gpurun_out/roundscan2; mkdir -p $out
while read label steps args; do
  for rounds in auto 1.45 1.9 2.3 2.9; do
    if [ $rounds = auto ]; then unset FDTD_TB_ROUNDS; else export FDTD_TB_ROUNDS=$rounds; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg $args --steps $steps --warmup 16 > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; exit 1; }
    python - "$label" $rounds <<'PY'
import json,sys
d=json.load(open("gpurun_out/roundscan2/run.json")); r=d["roofline"]
print("%-14s rounds=%-5s pass %.4f ms  %7.1f Gcell/s  frac %.3f K=%s" % (sys.argv[1],sys.argv[2],r["kernel_avg_ms"],d["value"]/1e3,r["frac"],r["steps_per_launch"]), flush=True)
PY
  done
done <<'LIST'
4608^2 240 --nx 4608 --ny 4608
5632^2 240 --nx 5632 --ny 5632
7168^2 240 --nx 7168 --ny 7168
9216^2 160 --nx 9216 --ny 9216
11264^2 160 --nx 11264 --ny 11264
12288^2 160 --nx 12288 --ny 12288
14336^2 160 --nx 14336 --ny 14336
16384x8192 160 --nx 16384 --ny 8192
8192x16384 160 --nx 8192 --ny 16384
tissue 160 --tissue 32
LIST
