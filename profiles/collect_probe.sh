#!/bin/bash
# rocprofv3 evidence for the 3-D two-step feasibility probe (tools/sweep3d_tb_probe): one kernel-trace
# pass with --stats, then one PMC pass per counter group.  Output under gpurun_out/<tag>/ in the
# layout profiles/summarize.py reads.  Usage (through gpurun): profiles/collect_probe.sh <tag> <probe arguments>
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag
mkdir -p "$out"
export PROBE_PREFETCH=0 PROBE_XCD=1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- tools/sweep3d_tb_probe "$@" > "$out/probe_trace.txt" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo "$grp" | tr ' ' '+')
    timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -- tools/sweep3d_tb_probe "$@" > "$out/probe_$name.txt" 2> "$out/pmc_$name.err" || { tail -5 "$out/pmc_$name.err"; exit 1; }
done
cat "$out/probe_trace.txt"
