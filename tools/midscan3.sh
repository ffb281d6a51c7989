#!/bin/bash
# Pass time of the multi-step sweep against the number of (equal) waves of a launch: where it steps
# up is the number of waves the GPU holds at once.
cd $GRAFT_REPO_ROOT
out=gpurun_out/midscan; mkdir -p $out
run() { # nx ny K rows
  export FDTD_TB_ROWS=$4 FDTD_TB_TAIL=0
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --nx $1 --ny $2 --steps $((8*$3*4)) --warmup $((2*$3)) --kernel multistep --steps-per-pass $3 > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; return 1; }
  python - $1 $2 $3 $4 <<'PY'
import json,sys
nx,ny,K,rows=map(int,sys.argv[1:5])
d=json.load(open("gpurun_out/midscan/run.json")); r=d["roofline"]
strips=((nx+3)//4+59)//60
bands=(ny+rows-1)//rows
ms=r["kernel_avg_ms"]
print("%5dx%-5d K=%d rows=%3d waves %5d  pass %.4f ms  %7.1f Gcell/s  frac %.3f  pass/(iters of a wave) %.2f us" % (nx,ny,K,rows,strips*bands,ms,d["value"]/1e3,r["frac"], ms*1e3/(rows+2*K-1)), flush=True)
PY
}
for bands in 16 26 29 30 32 36 44 52 59 60 64 72 80 89 90 96 120 128; do run 16384 $((64*bands)) 8 64 || exit 1; done
