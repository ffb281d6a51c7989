cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline --kernel tile "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$TAG $*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-60s %9.2f us/step %8.1f Gcell/s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3), flush=True)
PY
}
for n in 768 1024 2048 4096; do for r in 32 64; do TAG="rows $r" FDTD_TILE_ROWS=$r run --nx $n --ny $n --steps 960 --warmup 48 || exit 1; done; done
