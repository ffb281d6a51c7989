# the multi-step sweep over mid-sized and odd-shaped grids with the band height the library chooses
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --no-profile --no-cpu-baseline "$@" > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; return 1; }
  python - "$1 $2 $3 $4" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sm.json"))
print("%-30s %9.2f us/step %8.1f Gcell/s %s"%(sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e3, d["config"]["kernel"]), flush=True)
PY
}
while read nx ny extra; do
  run --nx $nx --ny $ny --steps 480 --warmup 48 $extra || exit 1
done <<'LIST'
3072 3072
3584 3584
3584 3584 --steps-per-pass 8
2896 2896
2896 2896 --kernel tile
4096 4096
5120 5120
6144 6144
7168 7168
8192 8192
9216 9216
10240 10240
12288 12288
4096 8192
8192 4096
2048 16384
16384 2048
16384 16384
LIST
