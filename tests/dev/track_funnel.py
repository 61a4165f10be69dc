#!/usr/bin/env python3
"""Where stage 4's candidates go, level by level (CPU oracle; a development script, not a test).

For a few frames of a synthetic 1280x960 stream it prints svo_result.track_stats and, for the pairings that survive the joint
collision filter, how many of each ORB pyramid level the two fundamental-matrix masks keep -- the measured explanation of why a
1.0-pixel epipolar gate (stage4_match_consecutive.cpp:202, 237: cv::findFundamentalMat(p1, p2, FM_RANSAC, 1.0, 0.99)) tracks
a few hundred of ~1200 pairings: a keypoint of level l sits on a grid of 1.2^l pixels, and two frames round it differently.
usage: python tests/dev/track_funnel.py [planes|relief] [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereo_vo_amd.synth import SyntheticStereoWorld
from stereo_vo_amd.abi import north_star_params, TS_NAMES
from oracle import oracle as O

scene = sys.argv[1] if len(sys.argv) > 1 else "planes"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = SyntheticStereoWorld(1280, 960, 800.0, 0.12, seed=0, n_frames=nf, scene=scene, scene_seed=0)
p = north_star_params(O.default_params(), orb_nfeats=2000)
o = O.Oracle(p); cam = w.camera()
prev = None
for t in range(nf):
    L, R = [x.numpy() for x in w.render(t)]
    r = o.process(L, R, cam)
    kl, _ = o.keypoints(0, 0); m = o.matches(0); tr = o.tracked()
    print("frame %d: %s" % (t, dict(zip(TS_NAMES, r.track_stats))))
    if prev is not None and len(tr):
        pkl, pm = prev
        lev_all = pkl["octave"][pm["queryIdx"]]                       # level of every previous pairing's left keypoint
        lev_trk = pkl["octave"][pm["queryIdx"][tr["first"]]]
        print("   level:            " + " ".join("%5d" % l for l in range(8)))
        print("   previous pairings " + " ".join("%5d" % int((lev_all == l).sum()) for l in range(8)))
        print("   tracked           " + " ".join("%5d" % int((lev_trk == l).sum()) for l in range(8)))
        print("   kept              " + " ".join("%4.0f%%" % (100.0 * (lev_trk == l).sum() / max(1, (lev_all == l).sum())) for l in range(8)))
    prev = (kl.copy(), m.copy())
