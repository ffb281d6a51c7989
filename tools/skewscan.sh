#!/bin/bash
# Row pitch skew (FDTD_PITCH_SKEW x 256 bytes) against band height: is the band-height landscape of the
# multi-step sweep a matter of band strides that are multiples of a large power of two?
cd $GRAFT_REPO_ROOT
out=gpurun_out/midscan; mkdir -p $out
run() { # nx ny K rows skew
  if [ $4 = 0 ]; then unset FDTD_TB_ROWS; else export FDTD_TB_ROWS=$4; fi
  export FDTD_PITCH_SKEW=$5
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --nx $1 --ny $2 --steps $((8*$3*4)) --warmup $((2*$3)) --kernel multistep --steps-per-pass $3 > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; return 1; }
  python - $1 $2 $3 $4 $5 <<'PY'
import json,sys
nx,ny,K,rows,skew=map(int,sys.argv[1:6])
d=json.load(open("gpurun_out/midscan/run.json")); r=d["roofline"]
print("%5dx%-5d K=%d rows=%3d skew=%d pass %.4f ms  %7.1f Gcell/s  frac %.3f" % (nx,ny,K,rows,skew,r["kernel_avg_ms"],d["value"]/1e3,r["frac"]), flush=True)
PY
}
while read nx ny K cands; do
  for skew in 0 1 3; do
    for r in $cands; do run $nx $ny $K $r $skew || exit 1; done
  done
done <<'LIST'
5120 5120 8 28 32 36 40
6144 6144 8 44 48 52
16384 16384 8 0 128 176
LIST
