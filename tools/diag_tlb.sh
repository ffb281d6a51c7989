#!/bin/bash
# Address translation and memory-side stall counters of the multi-step sweep.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/diag_tlb
rm -rf $out; mkdir -p $out
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$i" -- python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-one-step-leg > "$out/bench_$i.json" 2> "$out/pmc_$i.err" || { grep -v "rocprofv3\]" "$out/pmc_$i.err" | tail -5; exit 1; }
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
c = defaultdict(list)
for path in glob.glob("gpurun_out/diag_tlb/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "fdtd2d_sweep_tb" in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(c):
    print(f"{k:40s} {sum(c[k])/len(c[k]):18.1f}  (n={len(c[k])})")
PY
