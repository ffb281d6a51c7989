cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-step-leg $EXTRA > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-60s %8.1f Gcell/s kernel %.4f ms"%(sys.argv[1], d["value"]/1e3, d["roofline"]["kernel_avg_ms"]), flush=True)
PY
}
for cfg in "--steps 200 --warmup 16" "--tissue 32 --steps 140 --warmup 14" "--nx 12288 --ny 12288 --steps 200 --warmup 16" "--nx 8192 --ny 8192 --steps 192 --warmup 12" "--nx 6144 --ny 6144 --steps 192 --warmup 12" "--nx 5120 --ny 5120 --steps 192 --warmup 12" "--nx 5120 --ny 5120 --steps 192 --warmup 16 --kernel tile"; do
  EXTRA="$cfg" run "$cfg (rule)" || exit 1; done
EXTRA="--nx 8192 --ny 8192 --steps 192 --warmup 12" FDTD_TB_ROWS=48 run "8192^2 rows 48"
EXTRA="--nx 8192 --ny 8192 --steps 192 --warmup 12" FDTD_TB_ROWS=80 run "8192^2 rows 80"
EXTRA="--nx 8192 --ny 8192 --steps 192 --warmup 16 --kernel multistep --steps-per-pass 8" run "8192^2 K=8 (rule)"
EXTRA="--tissue 32 --steps 140 --warmup 14" FDTD_TB_ROWS=80 run "tissue rows 80"
EXTRA="--tissue 32 --steps 140 --warmup 14" FDTD_TB_ROWS=112 run "tissue rows 112"
EXTRA="--nx 6144 --ny 6144 --steps 192 --warmup 12" FDTD_TB_ROWS=48 run "6144^2 rows 48"
EXTRA="--nx 6144 --ny 6144 --steps 192 --warmup 16 --kernel tile" run "6144^2 tile"
