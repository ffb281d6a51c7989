#!/bin/bash
# Steps per pass of the multi-step sweep on grids around the 9 M / 14 M cell thresholds, and the tiled sweep beside it.
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-70s %10.2f us/step %8.1f Gcell/s  %s x%d"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3, d["config"]["kernel"], d["config"]["steps_per_pass"]), flush=True)
PY
}
for n in 2560 2896 3072 3584 3840 4096 4608; do
  for k in 6 7 8; do run --nx $n --ny $n --steps 480 --warmup 48 --kernel multistep --steps-per-pass $k || exit 1; done
  [ $n -le 3584 ] && { run --nx $n --ny $n --steps 480 --warmup 48 --kernel tile --steps-per-pass 6 || exit 1; }
done
