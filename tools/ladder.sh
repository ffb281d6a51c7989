# What FDTD_KERNEL_AUTO delivers across grid sizes, 2-D and 3-D: one line per size.
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$1 $2 $3 $4 $5 $6" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-34s %10.2f us/step %8.1f Gcell/s  %s x%d"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3, d["config"]["kernel"], d["config"]["steps_per_pass"]), flush=True)
PY
}
while read nx ny steps; do
  run --nx $nx --ny $ny --steps $steps --warmup 48 || exit 1
done <<'LIST'
265 490 4800
512 512 4800
768 768 4800
1024 1024 2400
1536 1536 2400
2048 2048 960
2896 2896 960
3072 3072 480
4096 4096 480
5120 5120 480
6144 6144 480
8192 8192 480
12288 12288 240
16384 16384 240
LIST
while read nx ny nz steps; do
  run --nx $nx --ny $ny --nz $nz --steps $steps --warmup 20 || exit 1
done <<'LIST'
32 32 32 2000
64 64 64 2000
100 100 100 2000
128 128 128 1000
200 200 200 400
256 256 256 200
512 512 512 60
265 490 1687 30
LIST
