cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-70s %9.2f us/step %8.1f Gcell/s  %s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3, d["config"]["kernel"]), flush=True)
PY
}
for n in "265 490" "1024 1024" "2048 2048" "3072 3072"; do set -- $n
  run --nx $1 --ny $2 --steps 4000 --warmup 80 --kernel tile || exit 1
done
run --nx 3072 --ny 3072 --steps 480 --warmup 48 --kernel multistep &&
run --nx 4096 --ny 4096 --steps 480 --warmup 48 --kernel tile
