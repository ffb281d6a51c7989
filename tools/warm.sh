cd $GRAFT_REPO_ROOT
for w in 5 40 200 800; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-step-leg --steps 20 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('warmup', d['warmup'], '%.1f Gcell/s'%(d['value']/1e3), [ (l['steps'], round(l['avg_ms'],3)) for l in d['roofline']['launches']])"; done
