#!/usr/bin/env python3
"""First-contact diagnostic for the GPU box: runs the small golden sequence through the HIP path and prints where
(if anywhere) it departs from the oracle, stage by stage.  Not a test; tests/ holds the assertions."""
import os
os.environ.setdefault("SVO_DEBUG_MODE", "9")    # raw (pre-NMS) keypoints carry angles / descriptors only in this mode
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stereo_vo_amd import hip
from stereo_vo_amd.abi import StereoCamera, north_star_params
from oracle import oracle as O

g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_small_seq.npz"))
W, H = int(g["W"]), int(g["H"])
cam = StereoCamera.simple(float(g["F"]), float(g["cx"]), float(g["cy"]), float(g["baseline"]), W, H)
p = north_star_params(hip.default_params(), orb_nfeats=int(g["orb_nfeats"]))
ctx = hip.Context(n_lanes=2, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15, kernel_times=True)
ctx.set_params(p); ctx.set_camera(cam)
orc = O.Oracle(p)
for t in range(4):
    L, R = g["L%d" % t], g["R%d" % t]
    ctx.process_host([(L, R), (R[:, ::-1].copy(), L[:, ::-1].copy())])
    r = ctx.result(0)
    ro = orc.process(L, R, cam)
    print("frame", t, "status", ctx.status_word(0))
    if t == 0:
        lw, lh, sc = O.pyramid_sizes(W, H, 8)
        prev = L
        for l in range(1, 8):
            ref = O.resize(prev, lw[l], lh[l]); got = ctx.level(0, 0, l)
            print("  level", l, got.shape, "mismatch px:", int((got != ref).sum()))
            prev = ref
    for side in (0, 1):
        img = R if side else L
        kr, dr = ctx.raw_keypoints(0, side)
        ko, do = O.orb_detect(img, int(1.5 * int(g["orb_nfeats"])), 8, 20)
        same = len(kr) == len(ko) and kr.tobytes() == ko.tobytes()
        print("  raw side", side, len(kr), len(ko), "kps equal:", same, "desc equal:", len(kr) == len(ko) and bool((dr == do).all()))
        if not same and len(kr) == len(ko):
            for f in kr.dtype.names:
                bad = np.where(kr[f] != ko[f])[0]
                if len(bad): print("    field", f, "differs at", len(bad), "first", bad[:5], kr[f][bad[:3]], ko[f][bad[:3]])
        k, d = ctx.keypoints(0, 0, side); k2, d2 = orc.keypoints(0, side)
        print("  final side", side, len(k), len(k2), "equal:", k.tobytes() == k2.tobytes() and bool((d == d2).all()))
    m, m2 = ctx.matches(0), orc.matches(0)
    print("  matches", len(m), len(m2), "equal:", m.tobytes() == m2.tobytes())
    tr, tr2 = ctx.tracked(0), orc.tracked()
    print("  tracked", len(tr), len(tr2), "equal:", tr.tobytes() == tr2.tobytes())
    print("  result gpu", r.valid, r.error_code, r.num_it, r.num_it_final, r.n_outliers, r.n_residual, np.round(np.array(r.outPose), 6))
    print("  result cpu", ro.valid, ro.error_code, ro.num_it, ro.num_it_final, ro.n_outliers, ro.n_residual, np.round(np.array(ro.outPose), 6))
    if r.valid and ro.valid:
        print("  pose |diff|", np.abs(np.array(r.outPose) - np.array(ro.outPose)).max(), "outliers equal:", bool((ctx.outliers(0) == orc.outliers()).all()) if r.n_outliers == ro.n_outliers else False)
        rr, rr2 = ctx.residuals(0), orc.residuals()
        fin = rr2 < 1e300
        print("  residual max rel diff", np.abs(rr[fin] - rr2[fin]).max() if len(rr) == len(rr2) and fin.any() else None)
print({k: (round(v[0], 3), v[1]) for k, v in ctx.kernel_times().items()})
