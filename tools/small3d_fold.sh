# small 3-D grids with the Mur faces inside the sweep (1) and as a launch of their own (0)
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$FDTD_3D_FOLD_MUR $*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-60s %9.2f us/step %8.1f Gcell/s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3), flush=True)
PY
}
for n in "50 50 50" "100 100 100" "128 128 128" "265 490 8" "90 110 130"; do set -- $n
  for f in 0 1; do
    export FDTD_3D_FOLD_MUR=$f
    run --nx $1 --ny $2 --nz $3 --steps 1000 --warmup 50 --kernel fused || exit 1
  done
done
