#!/bin/bash
# Band-height landscape of the multi-step sweep on mid-sized grids: for each grid and number of steps
# per pass, every band height of a list (0 = the library's choice).
# Prints pass time, waves per launch and the time per row iteration of the longest wave.
cd $GRAFT_REPO_ROOT
out=gpurun_out/midscan; mkdir -p $out
run() { # nx ny K rows
  if [ $4 = 0 ]; then unset FDTD_TB_ROWS; else export FDTD_TB_ROWS=$4; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --nx $1 --ny $2 --steps $((8*$3*4)) --warmup $((2*$3)) --kernel multistep --steps-per-pass $3 > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; return 1; }
  python - $1 $2 $3 $4 <<'PY'
import json,sys
nx,ny,K,rows=map(int,sys.argv[1:5])
d=json.load(open("gpurun_out/midscan/run.json")); r=d["roofline"]
ms=r["kernel_avg_ms"]
print("%5dx%-5d K=%d rows=%3d pass %.4f ms  %7.1f Gcell/s  frac %.3f" % (nx,ny,K,rows,ms,d["value"]/1e3,r["frac"]), flush=True)
PY
}
while read nx ny K cands; do
  for r in $cands; do run $nx $ny $K $r || exit 1; done
done <<'LIST'
4096 4096 6 0 16 20 24 28 32 36 40 48
4096 4096 8 0 16 20 24 28 32 36 40 48
5120 5120 8 0 24 28 32 36 40 48 56
6144 6144 8 0 32 40 44 48 52 56 64 72
8192 8192 8 0 48 56 64 72 80 96 112
10240 10240 8 0 80 96 112 128 144
12288 12288 8 0 96 112 128 144 160 176
LIST
