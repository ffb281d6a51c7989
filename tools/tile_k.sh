# tiled sweep: steps per launch x tile height on grids of 1 M .. 8 M cells
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "rows=${FDTD_TILE_ROWS:-auto} $*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-75s %9.2f us/step %8.1f Gcell/s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3), flush=True)
PY
}
for n in ${SIZES:-1024 2048 2896}; do
  for k in ${KS:-4 6 8}; do
    for r in ${ROWS:-32 64 642}; do
      FDTD_TILE_ROWS=$r run --nx $n --ny $n --steps 960 --warmup 48 --kernel tile --steps-per-pass $k || exit 1
    done
  done
done
