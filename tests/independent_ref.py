"""A SECOND, INDEPENDENT reading of the published algorithms the oracle had to freeze (TEST INFRASTRUCTURE).

The reference delegates these to OpenCV / MRPT, whose sources are not in /root/reference, so
oracle/svo_oracle.c restates them from the papers.  This module restates them AGAIN, in numpy / scipy, written from the
same papers and NOT from svo_oracle.c -- different formulations on purpose (dense array arithmetic, library SVD and
rotations, float Gaussian, exact atan2) -- so that tests/test_independent_reading.py can say "two independent readings
agree" instead of "the oracle equals itself":

  Rosten & Drummond 2006   FAST-9/16 segment test, corner score = largest threshold that keeps the corner, 3x3 NMS
  Harris & Stephens 1988   corner measure det(M) - k tr(M)^2 on a 7x7 window of Sobel gradients (k = 0.04), as ORB ranks with it
  Rosin 1999 / Rublee 2011 intensity-centroid orientation over a circular patch of radius 15
  Rublee et al. 2011       steered BRIEF on a sigma = 2 smoothed patch, angle quantised to 12 degrees (sec. 4.2)
  Hartley 1997             normalised 8-point algorithm; symmetric distance to the epipolar lines
  bilinear resampling with pixel centres at half-integers (what cv::resize INTER_LINEAR documents)
  rotation vector <-> matrix <-> yaw/pitch/roll with scipy.spatial.transform

Only the 256 BRIEF test pairs are shared data (include/svo_orb_tables.h, angle bin 0): they are a table, not an algorithm.
"""
import os
import re

import numpy as np
from scipy import ndimage
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Bresenham circle of radius 3, clockwise from 12 o'clock (Rosten's numbering 1..16)
CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def fast9_score_map(img, th):
    """Score of every pixel at least 3 px from the border: the largest t for which >= 9 contiguous circle pixels are all
    brighter than I + t or all darker than I - t; 0 where that t is below th (not a corner at threshold th)."""
    I = img.astype(np.int32)
    H, W = I.shape
    c = I[3:H - 3, 3:W - 3]
    ring = np.stack([I[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for dx, dy in CIRCLE])       # 16 x h x w
    out = np.zeros_like(c)
    for sign in (1, -1):
        d = sign * (ring - c[None])                          # how much brighter (darker) each circle pixel is
        dd = np.concatenate([d, d[:8]])                      # wrap the circle
        best = np.full(c.shape, -10 ** 6, np.int32)
        for s in range(16):
            best = np.maximum(best, dd[s:s + 9].min(axis=0))   # the weakest pixel of the arc bounds the threshold
        out = np.maximum(out, best)
    # all 9 strictly beyond I +- t  <=>  t < best, so the largest such t is best - 1; a corner at th needs best > th
    score = np.where(out > th, out - 1, 0)
    full = np.zeros((H, W), np.int32)
    full[3:H - 3, 3:W - 3] = score
    return full


def nms3x3(score, border):
    """strict 3x3 maxima of a score map, at least `border` px from the image border"""
    H, W = score.shape
    eq = np.zeros_like(score)
    pad = np.pad(score, 1)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                eq += (pad[1 + dy:1 + dy + H, 1 + dx:1 + dx + W] >= score).astype(score.dtype)
    keep = (score > 0) & (eq == 0)
    keep[:border] = False; keep[H - border:] = False; keep[:, :border] = False; keep[:, W - border:] = False
    return keep


def harris_map(img, k=0.04, block=7):
    """Harris measure of every pixel: M = sum over a block x block window of [Ix^2 IxIy; IxIy Iy^2], Sobel gradients,
    normalised like the 8-bit OpenCV implementation ORB ranks with (gradient scale 1 / (4 * block * 255))."""
    I = img.astype(np.int64)
    sx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int64)
    Ix = ndimage.correlate(I, sx, mode="constant")
    Iy = ndimage.correlate(I, sx.T, mode="constant")
    box = np.ones((block, block), np.int64)
    a = ndimage.correlate(Ix * Ix, box, mode="constant")
    b = ndimage.correlate(Iy * Iy, box, mode="constant")
    c = ndimage.correlate(Ix * Iy, box, mode="constant")
    s = 1.0 / (4.0 * block * 255.0)
    a, b, c = a.astype(np.float64), b.astype(np.float64), c.astype(np.float64)
    return (a * b - c * c - k * (a + b) ** 2) * s ** 4


def disc_mask(radius=15):
    """pixels whose centre lies within radius + 1/2 of the patch centre"""
    v, u = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    return (u * u + v * v) < (radius + 0.5) ** 2


def ic_angle_deg(img, x, y, radius=15):
    """orientation of the intensity centroid, degrees in [0, 360), image axes (x right, y down)"""
    P = img[y - radius:y + radius + 1, x - radius:x + radius + 1].astype(np.float64) * disc_mask(radius)
    v, u = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    a = np.degrees(np.arctan2((v * P).sum(), (u * P).sum()))
    return a + 360.0 if a < 0 else a


def brief_pairs():
    """the 256 test pairs (x1 y1 x2 y2) of cv::ORB: OpenCV's learned table, from the committed plain-integer copy of the file
    scikit-image ships (tests/golden/make_orb_pattern.py) -- not from the header the oracle compiles"""
    rows = [[int(v) for v in l.split()] for l in open(os.path.join(ROOT, "tests", "golden", "orb_bit_pattern_31.txt")) if l.strip() and not l.startswith("#")]
    return np.array(rows, np.int64).reshape(256, 4)


def steered_brief(img, x, y, angle_deg, pairs):
    """256-bit descriptor, 32 bytes, bit i of byte i // 8 (LSB first) = smoothed I(p_i) < smoothed I(q_i); the pairs are
    rotated by the CONTINUOUS angle (double precision) and rounded to pixels (half away from zero, as scikit-image does)"""
    th = np.radians(angle_deg)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])

    def rnd(v):
        return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int64)
    p = rnd(pairs[:, 0:2] @ R.T)
    q = rnd(pairs[:, 2:4] @ R.T)
    patch = img[y - 24:y + 25, x - 24:x + 25].astype(np.float64)
    g = np.exp(-np.arange(-3, 4) ** 2 / (2 * 2.0 ** 2)); g /= g.sum()
    sm = ndimage.correlate1d(ndimage.correlate1d(patch, g, axis=0, mode="mirror"), g, axis=1, mode="mirror")
    a = sm[24 + p[:, 1], 24 + p[:, 0]]
    c = sm[24 + q[:, 1], 24 + q[:, 0]]
    bits = (a < c).astype(np.uint8)
    margin = np.abs(a - c)                                   # how decisive each test is (a float blur vs an 8-bit one)
    # how close the nearest of the four rotated coordinates of each pair lies to a rounding tie
    fr = np.abs(np.c_[pairs[:, 0:2] @ R.T, pairs[:, 2:4] @ R.T])
    tie = np.min(np.abs(fr - np.floor(fr) - 0.5), axis=1)
    return np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").reshape(32), margin, tie


def eight_point(p1, p2, rank2=True):
    """Hartley's normalised 8-point algorithm: F with x2^T F x1 = 0, unit Frobenius norm; rank2: the smallest singular value of
    the normalised solution is zeroed before denormalising (Hartley 1997 section 3.2; what cv::findFundamentalMat's models are)"""
    def norm(p):
        c = p.mean(0)
        s = np.sqrt(2.0) / np.mean(np.linalg.norm(p - c, axis=1))
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    T1, T2 = norm(p1), norm(p2)
    h1 = (np.c_[p1, np.ones(len(p1))] @ T1.T); h2 = (np.c_[p2, np.ones(len(p2))] @ T2.T)
    A = np.stack([np.kron(b, a) for a, b in zip(h1, h2)])
    F0 = np.linalg.svd(A)[2][-1].reshape(3, 3)
    if rank2:
        U, S, Vt = np.linalg.svd(F0)
        F0 = U @ np.diag([S[0], S[1], 0.0]) @ Vt
    F = T2.T @ F0 @ T1
    return F / np.linalg.norm(F)


def seven_point(p1, p2):
    """The 7-point algorithm as the textbooks give it (Hartley & Zisserman 11.1.2; cv::findFundamentalMat's minimal solver): null space
    of the 7 x 9 constraint matrix by SVD, F = a F1 + (1 - a) F2, the cubic det F = 0 by numpy's companion-matrix roots.  Real
    roots only; unit Frobenius norm each.  (Normalised coordinates for conditioning: the solution set does not depend on them.)"""
    def norm(p):
        c = p.mean(0)
        s = np.sqrt(2.0) / np.mean(np.linalg.norm(p - c, axis=1))
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    T1, T2 = norm(p1), norm(p2)
    h1 = (np.c_[p1, np.ones(7)] @ T1.T); h2 = (np.c_[p2, np.ones(7)] @ T2.T)
    A = np.stack([np.kron(b, a) for a, b in zip(h1, h2)])
    V = np.linalg.svd(A)[2]
    F1, F2 = V[7].reshape(3, 3), V[8].reshape(3, 3)
    ls = np.array([-1.0, 0.0, 1.0, 2.0])
    co = np.polyfit(ls, [np.linalg.det(l * F1 + (1 - l) * F2) for l in ls], 3)          # a cubic through four of its values: exact
    out = []
    for r in np.roots(co):
        if abs(r.imag) < 1e-9 * max(1.0, abs(r.real)):
            F = T2.T @ (r.real * F1 + (1 - r.real) * F2) @ T1
            out.append(F / np.linalg.norm(F))
    return out


def symmetric_epipolar_sq(F, p1, p2):
    """max of the two squared point-to-epipolar-line distances"""
    h1 = np.c_[p1, np.ones(len(p1))]; h2 = np.c_[p2, np.ones(len(p2))]
    l2 = h1 @ F.T; l1 = h2 @ F
    d2 = (np.sum(l2 * h2, 1) ** 2) / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
    d1 = (np.sum(l1 * h1, 1) ** 2) / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
    return np.maximum(d1, d2)


def bilinear_resize(img, dw, dh):
    """area-preserving grid: destination pixel centre (i + 1/2) maps to source coordinate (i + 1/2) * src / dst - 1/2"""
    H, W = img.shape
    xs = np.clip((np.arange(dw) + 0.5) * W / dw - 0.5, 0, W - 1)
    ys = np.clip((np.arange(dh) + 0.5) * H / dh - 0.5, 0, H - 1)
    out = ndimage.map_coordinates(img.astype(np.float64), np.meshgrid(ys, xs, indexing="ij"), order=1, mode="nearest")
    return out


def delta_to_pose(delta):
    """Delta = (rotation vector, translation) maps previous-frame points into the current frame; the pose of the current
    frame seen from the previous one is its inverse, as x y z yaw pitch roll with R = Rz(yaw) Ry(pitch) Rx(roll)"""
    R = Rotation.from_rotvec(delta[:3]).as_matrix()
    Ri = R.T
    t = -Ri @ delta[3:]
    ypr = Rotation.from_matrix(Ri).as_euler("ZYX")
    return np.r_[t, ypr]


def project(lmk, cam, delta):
    """left / right pixels of landmarks given in the previous left-camera frame, after the motion Delta"""
    R = Rotation.from_rotvec(delta[:3]).as_matrix()
    X = lmk @ R.T + delta[3:]
    ul = cam.l_fx * X[:, 0] / X[:, 2] + cam.l_cx; vl = cam.l_fy * X[:, 1] / X[:, 2] + cam.l_cy
    ur = cam.r_fx * (X[:, 0] - cam.baseline) / X[:, 2] + cam.r_cx; vr = cam.r_fy * X[:, 1] / X[:, 2] + cam.r_cy
    return np.stack([ul, vl, ur, vr], 1)
