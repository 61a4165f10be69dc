#!/usr/bin/env python3
"""Freeze a small end-to-end run of the oracle (inputs + outputs) as a regression fixture.

The reference pins nothing for stages 2-5, so these vectors are minted by the oracle itself:
they guard the frozen algorithm against accidental change and give the GPU tests fixed inputs that do not
depend on torch's float rendering.  Run from the repo root: python tests/golden/make_oracle_golden.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from stereo_vo_amd.synth import SyntheticStereoWorld
from stereo_vo_amd.abi import north_star_params
from oracle import oracle as O

W, H, F, NF = 256, 192, 180.0, 4
w = SyntheticStereoWorld(W, H, F, 0.12, seed=7, n_frames=NF, noise_sigma=1.5)
p = north_star_params(O.default_params(), orb_nfeats=220)
o = O.Oracle(p)
cam = w.camera()
out = {"W": W, "H": H, "F": F, "baseline": 0.12, "cx": w.cx, "cy": w.cy, "orb_nfeats": 220, "oracle_version": O.version()}
for t in range(NF):
    L, R = [x.numpy() for x in w.render(t)]
    r = o.process(L, R, cam)
    out["L%d" % t], out["R%d" % t] = L, R
    for side in (0, 1):
        k, d = o.keypoints(0, side)
        out["kps%d_%d" % (side, t)], out["desc%d_%d" % (side, t)] = k, d
        out["rowidx%d_%d" % (side, t)] = o.row_index(0, side)
    out["matches%d" % t] = o.matches(0)
    out["mrow%d" % t] = o.matches_row_index(0)
    out["tracked%d" % t] = o.tracked()
    out["residual%d" % t] = o.residuals()
    out["outliers%d" % t] = o.outliers()
    out["pose%d" % t] = np.array(r.outPose)
    out["delta%d" % t] = np.array(r.delta)
    out["scalars%d" % t] = np.array([r.num_it, r.num_it_final, r.valid, r.error_code, r.tracked_feats_from_last_frame,
                                     r.detected_left[0], r.detected_right[0], r.stereo_matches[0], r.n_outliers, r.n_residual])
    out["track_stats%d" % t] = np.array(list(r.track_stats))
    print(t, out["scalars%d" % t], out["track_stats%d" % t])
np.savez_compressed(os.path.join(os.path.dirname(__file__), "oracle_small_seq.npz"), **out)
