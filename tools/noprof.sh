cd $GRAFT_REPO_ROOT
for a in "" "--no-profile"; do for cfg in "--steps 200 --warmup 16" "--nx 8192 --ny 8192 --steps 192 --warmup 12" "--tissue 32 --steps 140 --warmup 14"; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-step-leg $cfg $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg $a: %.1f Gcell/s %.4f ms/step'%(d['value']/1e3, d['ms_per_step']))"; done; done
