# mid-sized grids: bands of one height, as many waves as fit the GPU at once (2048) or a multiple
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "rows=${FDTD_TB_ROWS:-auto} tail=${FDTD_TB_TAIL:-auto} $1 $2 $3 $4" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-50s %9.2f us/step %8.1f Gcell/s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3), flush=True)
PY
}
while read nx ny cands; do
  unset FDTD_TB_ROWS FDTD_TB_TAIL
  run --nx $nx --ny $ny --steps 480 --warmup 48 --kernel multistep --steps-per-pass 8 || exit 1
  export FDTD_TB_TAIL=0
  for r in $cands; do
    export FDTD_TB_ROWS=$r
    run --nx $nx --ny $ny --steps 480 --warmup 48 --kernel multistep --steps-per-pass 8 || exit 1
  done
done <<'LIST'
3072 3072 20 21 24 40
4096 4096 32 36 37 38 40 74
5120 5120 28 55 56 58 60
6144 6144 39 40 42 78 80
8192 8192 47 48 70 71 72 140
LIST
