cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --steps 200 --warmup 16 --kernel multistep --steps-per-pass 8 "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-40s %8.1f Gcell/s kernel %.4f ms"%(sys.argv[1], d["value"]/1e3, d["roofline"]["kernel_avg_ms"]), flush=True)
PY
}
run --nx 8192 --ny 32768 && run --nx 16384 --ny 16384 && run --nx 32768 --ny 8192 && run --nx 65536 --ny 4096 && run --nx 131072 --ny 2048 && run --nx 32768 --ny 32768 && run --nx 16384 --ny 32768 && run --nx 32768 --ny 16384
