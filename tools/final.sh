# The headline measurements, one line each (run through gpurun from the repo root).
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
r=d.get("roofline",{})
print("%-75s %8.1f Gcell/s %9.4f ms/step  kernel %.4f ms  frac %.3f  %s"%(sys.argv[1], d["value"]/1e3, d["ms_per_step"], r.get("kernel_avg_ms",0), r.get("frac",0), d["config"]["kernel"]), flush=True)
PY
}
run --steps 200 --warmup 16 --no-one-step-leg &&
run --steps 20 --warmup 5 --no-one-step-leg &&
run --steps 200 --warmup 16 --kernel fused &&
run --steps 400 --warmup 16 --no-one-step-leg --probes 16 &&
run --steps 400 --warmup 16 --no-one-step-leg --snapshot-every 100 --probes 16 &&
run --steps 2000 --warmup 16 --no-one-step-leg --snapshot-every 100 --probes 16 &&
run --tissue 32 --steps 160 --warmup 16 &&
run --tissue 32 --steps 100 --warmup 10 --kernel fused &&
run --nx 65536 --ny 65536 --steps 48 --warmup 8 --no-one-step-leg &&
run --nx 65536 --ny 4096 --steps 96 --warmup 8 --no-one-step-leg &&
run --nx 8192 --ny 8192 --steps 120 --warmup 12 --no-one-step-leg &&
run --nx 4096 --ny 4096 --steps 240 --warmup 16 --no-one-step-leg &&
run --nx 512 --ny 512 --nz 512 --steps 100 --warmup 5 &&
run --nx 265 --ny 490 --nz 1687 --steps 40 --warmup 5
