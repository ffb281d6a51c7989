#!/usr/bin/env python3
"""The marching loops (those that store six rows) of one instantiation of the multi-step sweep in an
assembly file: instruction counts by class and spill traffic.  Usage: tb_loops.py file.s [K ONE PLAIN WIDE]"""
import re
import sys
s = open(sys.argv[1]).read()
tpl = tuple(sys.argv[2:6]) if len(sys.argv) > 5 else ("8", "1", "1", "0")
name = [x for x in re.findall(r'\.globl\s+(\S*fdtd2d_sweep_tbILi%sELb%sELb%sELb%sE\S*)' % tpl, s)][0]
i = s.index(name + ':'); j = s.index('; Occupancy:', i); body = s[i:j].splitlines()
labels = {}
for k, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = k
loops = set()
for k, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < k:
        loops.add((labels[m.group(1)], k))
seen = []
for a, b in sorted(loops, key=lambda t: t[1] - t[0]):
    seg = body[a:b + 1]
    cnt = lambda pat: sum(1 for l in seg if re.match(r'\s+' + pat, l))
    if cnt('global_store') < 6 or any(a >= x and b <= y for x, y in seen):
        continue
    if any(x >= a and y <= b for x, y in seen):
        continue
    seen.append((a, b))
    print(f"loop at {a}-{b}: valu {cnt('v_')} (f64 {cnt('v_[a-z0-9_]*f64')}) salu {cnt('s_')} lds {cnt('ds_')} "
          f"gload {cnt('global_load')} gstore {cnt('global_store')} scratch {cnt('scratch_')} waitcnt {cnt('s_waitcnt')} readlane {cnt('v_readlane')}")
