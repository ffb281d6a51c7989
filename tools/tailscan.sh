#!/bin/bash
# 16384^2: band height x rows in short bands at the end of the launch x height of those bands.
cd $GRAFT_REPO_ROOT
out=gpurun_out/midscan; mkdir -p $out
for rows in 160 176 192; do for tail in 1024 2048 3072 4096; do for tr in 16 32 48; do
  export FDTD_TB_ROWS=$rows FDTD_TB_TAIL=$tail FDTD_TB_TAIL_ROWS=$tr
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --steps 160 --warmup 16 > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; exit 1; }
  python - $rows $tail $tr <<'PY'
import json,sys
d=json.load(open("gpurun_out/midscan/run.json")); r=d["roofline"]
print("rows %s tail %s tail_rows %s: pass %.4f ms  %7.1f Gcell/s" % (sys.argv[1],sys.argv[2],sys.argv[3],r["kernel_avg_ms"],d["value"]/1e3), flush=True)
PY
done; done; done
