#!/bin/bash
# Where the waves of the multi-step sweep spend their cycles: two passes of SQ counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/diag_sq
rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$i" -- python bench.py ${DIAG_ARGS:---steps 48 --warmup 8} --no-cpu-baseline --no-one-step-leg > "$out/bench_$i.json" 2> "$out/pmc_$i.err" || { grep -v "rocprofv3\]" "$out/pmc_$i.err" | tail -5; exit 1; }
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
c = defaultdict(list)
for path in glob.glob("gpurun_out/diag_sq/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "fdtd2d_sweep_tb" in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(c):
    print(f"{k:28s} {sum(c[k])/len(c[k]):18.1f}  (n={len(c[k])})")
PY
