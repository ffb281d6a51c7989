#!/usr/bin/env python3
"""Builds tests/golden/body_codes_265x490x1687.u8.xz: the material codes of the reference's whole voxel
body model -- its 1687 PGM slices data/pgmdata/v1_00001.pgm .. v1_01687.pgm (265 x 490 each, 481 MB of
text), read with loadMaterials' own token semantics (fdtd.cu:909-936: six header tokens, then nx * ny
values in sequence, x fastest) -- as one uint8 array [z][y][x], xz-compressed (3.7 MB).  Data, not code.
Runs only where the reference tree exists:  python tests/golden/make_body_codes.py [/root/reference]"""
import lzma
import os
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
NX, NY, NZ = 265, 490, 1687
vol = np.zeros((NZ, NY, NX), dtype=np.uint8)
for iz in range(NZ):
    toks = open(os.path.join(ref, "data", "pgmdata", "v1_%05d.pgm" % (iz + 1))).read().split()
    assert toks[0] == "P2", toks[:6]
    vol[iz] = np.array(toks[6:6 + NX * NY], dtype=np.uint8).reshape(NY, NX)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "body_codes_265x490x1687.u8.xz")
with open(out, "wb") as f:
    f.write(lzma.compress(vol.tobytes(), preset=6))
print(out, os.path.getsize(out), "bytes; codes", np.unique(vol).tolist())
