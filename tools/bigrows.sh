# band height of the multi-step sweep at 16384^2 (and the tissue map) around the library's choice
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "rows=${FDTD_TB_ROWS:-auto} $*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-50s %9.2f us/step %8.1f Gcell/s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3), flush=True)
PY
}
for r in 0 224 240 256 272 288 304; do
  if [ $r = 0 ]; then unset FDTD_TB_ROWS; else export FDTD_TB_ROWS=$r; fi
  run --steps 240 --warmup 16 || exit 1
done
for r in 0 136 144 152 160 176 192; do
  if [ $r = 0 ]; then unset FDTD_TB_ROWS; else export FDTD_TB_ROWS=$r; fi
  run --tissue 32 --steps 140 --warmup 14 || exit 1
done
