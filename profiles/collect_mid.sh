#!/bin/bash
# Mid-sized 2-D grids: for the band height the library chooses and one neighbour on each side, a
# rocprofv3 kernel trace, the SQ wave counters, the TCP->TCC read latency, and the wave timeline of one
# pass (tools/wave_timeline.py).  Output under gpurun_out/r03_mid/<grid>_rows<r>/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export FDTD_TB_PRELOAD=0 # (no empty launches under the kernels' names in the per-kernel averages)
while read nx ny k rows; do
  for r in $rows; do
    out=gpurun_out/r03_mid/${nx}x${ny}_rows$r
    mkdir -p $out
    if [ $r = auto ]; then unset FDTD_TB_ROWS; else export FDTD_TB_ROWS=$r; fi
    args="--nx $nx --ny $ny --steps $((8*k*3)) --warmup $((2*k)) --kernel multistep --steps-per-pass $k --no-cpu-baseline --no-one-step-leg"
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py $args > $out/bench_trace.json 2> $out/trace.err || { tail -3 $out/trace.err; exit 1; }
    timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $out/pmc_sq -- python bench.py $args > $out/bench_sq.json 2> $out/pmc_sq.err || { tail -3 $out/pmc_sq.err; exit 1; }
    timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum --output-format csv -d $out/pmc_tcp -- python bench.py $args > $out/bench_tcp.json 2> $out/pmc_tcp.err || { tail -3 $out/pmc_tcp.err; exit 1; }
    timeout -k 10 100 python tools/wave_timeline.py $nx $ny $k $([ $r = auto ] && echo 0 || echo $r) > $out/timeline.json 2> $out/timeline.txt || { tail -3 $out/timeline.txt; exit 1; }
    echo "$out done"
  done
done <<'LIST'
4096 4096 8 auto 24 32
6144 6144 8 auto 44 64
LIST
