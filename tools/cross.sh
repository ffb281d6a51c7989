cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-70s %9.2f us/step %8.1f Gcell/s  %s x%d"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3, d["config"]["kernel"], d["config"]["steps_per_pass"]), flush=True)
PY
}
for n in 1024 1536 2048 3072 4096; do
  run --nx $n --ny $n --steps 960 --warmup 48 --kernel tile &&
  run --nx $n --ny $n --steps 960 --warmup 48 --kernel multistep &&
  run --nx $n --ny $n --steps 960 --warmup 48 --kernel multistep --steps-per-pass 8 &&
  run --nx $n --ny $n --steps 960 --warmup 48 --kernel multistep --steps-per-pass 4 || exit 1
done
