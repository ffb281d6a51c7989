cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --steps 200 --warmup 16 $EXTRA > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-40s %8.1f Gcell/s kernel %.4f ms"%(sys.argv[1], d["value"]/1e3, d["roofline"]["kernel_avg_ms"]), flush=True)
PY
}
for r in 144 160 176 192; do for t in "2048 32" "2048 48" "3072 48" "3072 64" "4096 48"; do set -- $t
  FDTD_TB_ROWS=$r FDTD_TB_TAIL=$1 FDTD_TB_TAIL_ROWS=$2 run "rows $r tail $1 in $2" || exit 1; done; done
