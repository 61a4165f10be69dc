#!/usr/bin/env python3
"""Mint tests/golden/oracle_extras.npz: frozen vectors of the operators added after the first golden sequence --
stage 1 (grey conversion + rectification), adaptive NMS, getProjectedCoords / pose -> delta.  Inputs are seeded and
small; outputs come from the CPU oracle (oracle/svo_oracle.c).  Run from the repo root: python tests/golden/make_oracle_extras.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from stereo_vo_amd.abi import StereoCamera, keypoint_dtype, dmatch_dtype


def inputs():
    rng = np.random.default_rng(20260929)
    h, w = 48, 64
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    xn, yn = (xx - w / 2) / (w / 2), (yy - h / 2) / (w / 2)
    r2 = xn * xn + yn * yn
    mx = (xn * (1 + 0.11 * r2) * (w / 2) + w / 2 + 1.37).astype(np.float32)
    my = (yn * (1 + 0.11 * r2) * (w / 2) + h / 2 - 2.21).astype(np.float32)
    n = 300
    k = np.zeros(n, keypoint_dtype)
    k["x"] = rng.integers(0, 4 * w, n).astype(np.float32) / 4; k["y"] = rng.integers(0, 4 * h, n).astype(np.float32) / 4
    k["response"] = np.round(rng.normal(0, 1, n), 2).astype(np.float32)          # ties and negative responses on purpose
    k["size"] = 31; k["class_id"] = -1
    kl = np.zeros(40, keypoint_dtype); kr = np.zeros(40, keypoint_dtype)
    kl["x"] = rng.uniform(50, 590, 40).astype(np.float32); kl["y"] = rng.uniform(40, 440, 40).astype(np.float32)
    kr["x"] = kl["x"] - rng.uniform(4, 40, 40).astype(np.float32); kr["y"] = kl["y"]
    m = np.zeros(40, dmatch_dtype); m["queryIdx"] = np.arange(40); m["trainIdx"] = np.arange(40)
    tracked = np.where(rng.random(40) < 0.4, 3, -1).astype(np.int32)
    pose = np.array([0.21, -0.05, 0.33, 0.04, -0.015, 0.02])
    return bgr, mx, my, k, kl, kr, m, tracked, pose


def outputs(bgr, mx, my, k, kl, kr, m, tracked, pose):
    cam = StereoCamera.simple(400.0, 320.0, 240.0, 0.12, 640, 480)
    return dict(grey=O.prepare(bgr), rect=O.prepare(bgr, mx, my), rect_grey_in=O.prepare(bgr[..., 1].copy(), mx, my),
                anms_all=O.anms_copy(k, 10000), anms_100=O.anms_copy(k, 100), anms_r3=O.anms_copy(k, 10000, 3.0),
                delta=O.pose_to_delta(pose), proj=O.projected_coords(m, kl, kr, tracked, cam, pose))


if __name__ == "__main__":
    out = outputs(*inputs())
    out["oracle_version"] = np.array(O.version())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_extras.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.shape(v) for k, v in out.items()})
