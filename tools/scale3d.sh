cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-60s %8.1f Gcell/s %8.4f ms/step kernel %.4f ms frac %.3f"%(sys.argv[1], d["value"]/1e3, d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["roofline"]["frac"]), flush=True)
PY
}
run --nx 512 --ny 512 --nz 128 --steps 100 --warmup 5 && run --nx 512 --ny 512 --nz 256 --steps 100 --warmup 5 && run --nx 512 --ny 512 --nz 512 --steps 100 --warmup 5 && run --nx 512 --ny 512 --nz 1024 --steps 50 --warmup 5 && run --nx 1024 --ny 1024 --nz 256 --steps 50 --warmup 5 && run --nx 256 --ny 256 --nz 2048 --steps 50 --warmup 5
