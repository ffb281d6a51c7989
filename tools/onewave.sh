# the multi-step sweep with one workgroup (four waves) per CU instead of two: what a second wave per SIMD is worth
cd $GRAFT_REPO_ROOT
for k in 6 8; do for pad in 0; do  # (49152: needs the FDTD_TB_LDS_PAD launch knob, removed again)
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-one-step-leg --steps $((k*16)) --warmup $((k*2)) --kernel multistep --steps-per-pass $k > gpurun_out/sm.json 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sm.json"))
print("K=$k pad=$pad %8.1f Gcell/s kernel %.4f ms"%(d["value"]/1e3, d["roofline"]["kernel_avg_ms"]), flush=True)
PY
done; done
