#!/bin/bash
# Diagnostic PMC passes (one counter group per run) for one bench.py configuration.
set -u
tag=$1; shift
args="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_REQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"; do
    i=$((i+1))
    timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$i" -- python bench.py $args > "$out/bench_$i.json" 2> "$out/pmc_$i.err" || { grep -v "rocprofv3\]" "$out/pmc_$i.err" | tail -5; }
done
