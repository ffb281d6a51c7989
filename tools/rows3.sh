cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-step-leg $EXTRA > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-60s %8.1f Gcell/s kernel %.4f ms"%(sys.argv[1], d["value"]/1e3, d["roofline"]["kernel_avg_ms"]), flush=True)
PY
}
for cfg in "--tissue 32 --steps 140 --warmup 14" "--nx 32768 --ny 8192 --steps 200 --warmup 16" "--nx 8192 --ny 32768 --steps 200 --warmup 16" "--nx 12288 --ny 12288 --steps 200 --warmup 16" "--nx 8192 --ny 8192 --steps 192 --warmup 16 --kernel multistep --steps-per-pass 8"  "--nx 8192 --ny 8192 --steps 192 --warmup 12"; do
  for r in 64 96 128 160 176; do EXTRA="$cfg" FDTD_TB_ROWS=$r run "$cfg rows $r" || exit 1; done; done
EXTRA="--nx 65536 --ny 65536 --steps 48 --warmup 8" FDTD_TB_ROWS=176 run "65536^2 rows 176"
