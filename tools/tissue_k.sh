cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-step-leg $EXTRA > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-60s %8.1f Gcell/s kernel %.4f ms x%d"%(sys.argv[1], d["value"]/1e3, d["roofline"]["kernel_avg_ms"], d["roofline"]["steps_per_launch"]), flush=True)
PY
}
EXTRA="--tissue 32 --steps 168 --warmup 14" run "tissue K=7 (default)"
EXTRA="--tissue 32 --steps 168 --warmup 16 --steps-per-pass 8" run "tissue K=8 (3 waves per block)"
EXTRA="--tissue 32 --steps 168 --warmup 16 --steps-per-pass 8" FDTD_TB_ROWS=64 run "tissue K=8 rows 64"
EXTRA="--tissue 32 --steps 168 --warmup 12 --steps-per-pass 6" run "tissue K=6"
