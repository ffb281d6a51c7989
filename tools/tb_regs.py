#!/usr/bin/env python3
"""Registers, scratch and LDS of every instantiation of the multi-step sweep, from an assembly file
(hipcc -S --cuda-device-only sweep2d_tb.hip)."""
import re
import sys
s = open(sys.argv[1]).read()
for n in re.findall(r'\.globl\s+(\S*fdtd2d_sweep_tb\S*)', s):
    i = s.index(n + ':'); j = s.index('; Occupancy:', i); body = s[i:j + 20]
    tpl = re.search(r'fdtd2d_sweep_tbILi(\d)ELb(\d)ELb(\d)ELb(\d)E', n).groups()
    g = lambda k: re.search(r'; ' + k + r': (\d+)', body).group(1)
    print('K,ONE,PLAIN,WIDE', tpl, 'vgpr', g('NumVgprs'), 'sgpr', g('TotalNumSgprs'), 'scratch', g('ScratchSize'), 'occ', g('Occupancy'),
          'lds', g('LDSByteSize'), 'v_instr', len(re.findall(r'^\s+v_', body, re.M)), 'lines', body.count('\n'))
