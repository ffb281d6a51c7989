#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the
# repo root): one kernel-trace pass with --stats, then one PMC pass per counter group
# (FETCH_SIZE and WRITE_SIZE do not fit into one pass; no trace domains beside --pmc).
# Output goes to gpurun_out/<tag>/; copy the summaries into profiles/ afterwards.
set -u
tag=${1:-r01}
shift || true
args=${*:---steps 48 --warmup 8 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# (without the 21 empty launches with which a context loads the instantiations of the multi-step sweep:
# they carry the kernels' names and would enter the per-kernel averages)
export FDTD_TB_PRELOAD=0
out=gpurun_out/$tag
mkdir -p "$out"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py $args > "$out/bench_trace.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    name=$(echo "$grp" | tr ' ' '+')
    timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -- python bench.py $args > "$out/bench_$name.json" 2> "$out/pmc_$name.err" || { tail -5 "$out/pmc_$name.err"; exit 1; }
done
find "$out" -name "*.csv" | head -40
