#!/bin/bash
# Timing of the multi-step sweep under its tuning switches (run on the GPU box from the repo root).
# Usage: tools/tb_variants.sh "<bench args>" VAR=val[,VAR=val...] ...   (one run per argument; "-" = defaults)
args=$1
shift
for v in "$@"; do
    envs=()
    [ "$v" != "-" ] && IFS=',' read -ra envs <<< "$v"
    out=$(env "${envs[@]}" timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-one-step-leg 2>/dev/null) || { echo "$v: failed"; exit 1; }
    echo "$v: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.1f Gcell/s  kernel %.3f ms  %s" % (d["value"]/1e3, r["kernel_avg_ms"], d["config"]["kernel"]))')"
done
