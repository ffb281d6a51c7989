# 16384^2, multi-step sweep at 6, 7 and 8 steps per pass
cd $GRAFT_REPO_ROOT
for k in 6 7 8; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --steps $((k*24)) --warmup $((k*2)) --kernel multistep --steps-per-pass $k > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sm.json"))
print("K=$k %8.1f Gcell/s kernel %.4f ms"%(d["value"]/1e3, d["roofline"]["kernel_avg_ms"]), flush=True)
PY
done
