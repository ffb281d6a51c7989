cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-step-leg --steps 200 --warmup 16 > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-60s %8.1f Gcell/s kernel %.4f ms"%(sys.argv[1], d["value"]/1e3, d["roofline"]["kernel_avg_ms"]), flush=True)
PY
}
run default
FDTD_TB_MAP=1 run "XCD panels"
FDTD_TB_INTERLEAVE=0 run "arrays apart"
FDTD_TB_PLAIN_LOADS=0 run "nontemporal loads"
FDTD_TB_TAIL=0 run "no tail"
FDTD_TB_TAIL=1024 run "tail 1024"
FDTD_TB_TAIL=4096 FDTD_TB_TAIL_ROWS=88 run "tail 4096 in 88"
FDTD_TB_TAIL=2048 FDTD_TB_TAIL_ROWS=24 run "tail 2048 in 24"
