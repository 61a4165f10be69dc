#!/usr/bin/env python3
"""Cut the two 24x24 crops used by the reference's only live test (libstereo-odometry/tests/
computeSAD8_unittest.cpp:20-41: L(646,263) vs R(624+ix,263+iy) on tests/0L.png / 0R.png) and record the 3x3 SAD
table computed with numpy.  Run in the build container only (needs /root/reference); the .npz is committed."""
import numpy as np
from PIL import Image
ref = "/root/reference/libstereo-odometry/tests/"
L = np.array(Image.open(ref + "0L.png").convert("L")); R = np.array(Image.open(ref + "0R.png").convert("L"))
assert L.shape == (600, 800)
lx, ly, rx, ry, h = 646, 263, 624, 263, 12
cl = L[ly - h:ly + h, lx - h:lx + h].copy(); cr = R[ry - h:ry + h, rx - h:rx + h].copy()
tab = np.zeros((3, 3), np.int64)
for iy in (-1, 0, 1):
    for ix in (-1, 0, 1):
        a = L[ly - 3:ly + 5, lx - 3:lx + 5].astype(int); b = R[ry + iy - 3:ry + iy + 5, rx + ix - 3:rx + ix + 5].astype(int)
        tab[iy + 1, ix + 1] = np.abs(a - b).sum()
print(tab)
np.savez_compressed(__file__.replace("make_sad8_kat.py", "sad8_kat.npz"), left=cl, right=cr, table=tab, half=h)
