# band height of the multi-step sweep on mid-sized grids: the rule's choice (rows=auto) against its neighbours
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "rows=${FDTD_TB_ROWS:-auto} $1 $2 $3 $4" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-40s %9.2f us/step %8.1f Gcell/s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3), flush=True)
PY
}
while read nx ny cands; do
  for r in $cands; do
    if [ $r = 0 ]; then unset FDTD_TB_ROWS; else export FDTD_TB_ROWS=$r; fi
    run --nx $nx --ny $ny --steps 480 --warmup 48 --kernel multistep --steps-per-pass 8 || exit 1
  done
done <<'LIST'
3072 3072 0 16 24 32
5120 5120 0 24 32 40 48 56
7168 7168 0 40 48 56 64 72
4096 8192 0 32 40 48 56 64
8192 4096 0 32 40 48 56 64
10240 10240 0 80 96 112 128
LIST
