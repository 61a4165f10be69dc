#!/opt/conda/bin/python3.9
"""Copies OpenCV's learned ORB pair table (bit_pattern_31_: 256 x [x0 y0 x1 y1]) out of the scikit-image tree of the build container
into tests/golden/orb_bit_pattern_31.txt as plain integers.  It is DATA that both libraries publish (cv::ORB uses it at
stage2_detect.cpp:482-493 through cv::ORB::create; scikit-image's ORB loads the same 256 rows); tools/gen_orb_tables.py turns it into
include/svo_orb_tables.h.  Run:  /opt/conda/bin/python3.9 tests/golden/make_orb_pattern.py"""
import os
import numpy as np
import skimage.feature

HERE = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(skimage.feature.__file__), "orb_descriptor_positions.txt")
P = np.loadtxt(src).astype(int)
assert P.shape == (256, 4) and P.min() == -13 and P.max() == 12 and tuple(P[0]) == (8, -3, 9, 5) and tuple(P[1]) == (4, 2, 7, -12)
with open(os.path.join(HERE, "orb_bit_pattern_31.txt"), "w") as f:
    f.write("# OpenCV's learned ORB test-pair table bit_pattern_31_ (256 x [x0 y0 x1 y1], patch 31), as shipped by scikit-image %s\n" % skimage.__version__)
    f.write("# (skimage/feature/orb_descriptor_positions.txt); copied as DATA by tests/golden/make_orb_pattern.py.  Consumed by\n")
    f.write("# tools/gen_orb_tables.py -> include/svo_orb_tables.h (oracle + k_describe).\n")
    for r in P:
        f.write("%d %d %d %d\n" % tuple(r))
print("wrote orb_bit_pattern_31.txt")
