#!/bin/bash
# Band height of the multi-step sweep expressed in rounds of the 2048 wave slots: rows per wave =
# rows x strips / (2048 x rounds), for several values of `rounds` on each grid.
cd $GRAFT_REPO_ROOT
out=gpurun_out/midscan; mkdir -p $out
while read nx ny K; do
  strips=$(( ( (nx + 3) / 4 + 59 ) / 60 ))
  for rounds in 1.15 1.3 1.45 1.6 1.8 2.0 2.3 2.7; do
    r=$(python3 -c "print(max(8, int(round($ny * $strips / (2048 * $rounds) / 4)) * 4))")
    [ $r -gt 200 ] && continue
    export FDTD_TB_ROWS=$r
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --nx $nx --ny $ny --steps $((8*K*4)) --warmup $((2*K)) --kernel multistep --steps-per-pass $K > $out/run.json 2>$out/run.err || { tail -3 $out/run.err; exit 1; }
    python - $nx $ny $K $r $rounds <<'PY'
import json,sys
nx,ny,K,rows=map(int,sys.argv[1:5])
d=json.load(open("gpurun_out/midscan/run.json")); r=d["roofline"]
print("%5dx%-5d K=%d rounds=%s rows=%3d pass %.4f ms  %7.1f Gcell/s  frac %.3f" % (nx,ny,K,sys.argv[5],rows,r["kernel_avg_ms"],d["value"]/1e3,r["frac"]), flush=True)
PY
  done
done <<'LIST'
3072 3072 6
3584 3584 6
4096 4096 8
4608 4608 8
5632 5632 8
7168 7168 8
9216 9216 8
4096 8192 8
8192 4096 8
11264 11264 8
14336 14336 8
16384 8192 8
LIST
