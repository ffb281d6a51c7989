#!/bin/bash
# Time per row iteration of a tall wave of the multi-step sweep against row pitch and waves in flight.
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
print("%5dx%-5d K=%d rows=%3d waves %5d  pass %.4f ms  %7.1f Gcell/s  frac %.3f  us/iter %.2f" % (nx,ny,K,rows,strips*bands,ms,d["value"]/1e3,r["frac"], ms*1e3/(rows+2*K-1)), flush=True)
PY
}
while read nx ny K rows; do run $nx $ny $K $rows || exit 1; done <<'LIST'
16384 1024 8 128
16384 1024 8 64
16384 1024 8 32
16384 2048 8 128
16384 2048 8 64
16384 4096 8 128
16384 4096 8 176
16384 8192 8 128
16384 8192 8 256
16384 16384 8 256
16384 16384 8 512
4096 16384 8 128
4096 16384 8 256
4096 16384 8 512
4096 32768 8 512
4080 4096 8 128
4100 4096 8 128
3840 4096 8 128
16384 1024 4 128
16384 1024 2 128
16384 1024 6 128
LIST
