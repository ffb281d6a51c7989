cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-70s %9.2f us/step %8.1f Gcell/s  %s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3, d["config"]["kernel"]), flush=True)
PY
}
for n in "100 100 100" "64 64 64" "200 200 200"; do set -- $n
  run --nx $1 --ny $2 --nz $3 --steps 2000 --warmup 50 --kernel fused &&
  run --nx $1 --ny $2 --nz $3 --steps 2000 --warmup 50 --kernel split || exit 1
done
