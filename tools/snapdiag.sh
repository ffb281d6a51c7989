set -e
cd $GRAFT_REPO_ROOT
for a in "--probes 1" "--probes 16" "--probes 1 --snapshot-every 100" "--probes 16 --snapshot-every 100" "--probes 1 --snapshot-every 200"; do
  timeout -k 10 120 python bench.py --steps 400 --warmup 16 --no-one-step-leg --no-cpu-baseline $a > gpurun_out/sd.json
  python - "$a" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sd.json"))
print(sys.argv[1], "| %.1f Gcell/s  %.4f ms/step kernel %.4f ms |"%(d["value"]/1e3,d["ms_per_step"],d["roofline"]["kernel_avg_ms"]), d["config"]["output"][-70:], flush=True)
PY
done
