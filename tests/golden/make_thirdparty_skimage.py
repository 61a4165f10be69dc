#!/opt/conda/bin/python3.9
"""Third-party cross-check vectors: scikit-image 0.18.3 (an independent implementation of the published FAST / oFAST / normalised
8-point algorithms; NOT the reference's dependency -- that is OpenCV, which this image lacks) run on seeded inputs.

scikit-image is only present in this build container (an Anaconda tree under /opt/conda, python3.9); it cannot travel, so its
outputs are committed as data in tests/golden/thirdparty_skimage.npz and tests/test_oracle_thirdparty.py compares the oracle with
them.  This narrows "parity unpinned"  for the pieces the two implementations define identically:
  * the FAST-9/16 segment test at a given threshold (corner set, and through several thresholds the corner score);
  * the intensity-centroid orientation over the radius-15 disc (same umax table as cv::ORB);
  * brute-force Hamming matching (match_descriptors, first minimum) on 256-bit descriptors with planted ties;
  * one x1/1.2 bilinear pyramid step (skimage.transform.resize, order 1, half-pixel centres: cv::resize INTER_LINEAR's geometry) to +-1
    grey level away from the borders (float weights against the oracle's 11-bit fixed point, other border rule);
  * the fundamental matrix of exactly eight correspondences (Hartley's normalised 8-point algorithm with rank-2 enforcement; since
    oracle version 4 the oracle's RANSAC solves seven points per sample, as cv::findFundamentalMat does: on EXACT correspondences the
    8-point matrix must be one of the 7-point models, tests/test_oracle_thirdparty.py)
    (scikit-image scales to an RMS distance of sqrt 2, cv::findFundamentalMat and the oracle to a MEAN distance of sqrt 2: the
    null vector of eight exact correspondences is the same, the rank-2 projection of noisy ones is taken in slightly different
    coordinates -- hence a tight comparison on exact sets and a loose one on noisy sets).
  * (round 4) the 256 steered BRIEF tests: skimage.feature.orb_cy._orb_loop -- OpenCV's learned pair table rotated by the continuous
    angle and rounded, cv::ORB's computeOrbDescriptor -- on an already blurred image at given positions and angles;
  * (round 4) the RANKING of the Harris measure (k = 0.04) at FAST corners (skimage.feature.corner_harris: Gaussian window, where
    cv::ORB sums a 7 x 7 box -- values differ, order mostly agrees).
Run:  /opt/conda/bin/python3.9 tests/golden/make_thirdparty_skimage.py
"""
import os
import warnings
warnings.filterwarnings("ignore")
import numpy as np
import skimage
from skimage.feature import corner_fast, corner_orientations
from skimage.feature.orb import OFAST_MASK
from skimage.transform import FundamentalMatrixTransform, resize
from skimage.feature import match_descriptors

HERE = os.path.dirname(os.path.abspath(__file__))


def textured(seed, h=240, w=320):
    """rectangles of random grey levels + a little noise, box-blurred once: corners of every contrast"""
    rng = np.random.RandomState(seed)
    img = np.full((h, w), 110.0)
    for _ in range(160):
        y0, x0 = rng.randint(0, h - 8), rng.randint(0, w - 8)
        hh, ww = rng.randint(4, 60), rng.randint(4, 60)
        img[y0:y0 + hh, x0:x0 + ww] = rng.randint(0, 256)
    img += rng.normal(0, 3.0, img.shape)
    k = np.ones(3) / 3.0
    img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, img)
    img = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, img)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


out = {"skimage_version": np.array(skimage.__version__)}
THS = [7, 12, 20, 33, 50, 80]
out["thresholds"] = np.array(THS)
for s in range(3):
    img = textured(1000 + s)
    out["img%d" % s] = img
    for t in THS:
        # corner <=> a 9-arc with every pixel > c + t (or < c - t).  skimage works on img / 255 in float64 with a strict compare;
        # t + 0.5 keeps the comparison half a grey level away from every representable difference, so rounding cannot flip it
        resp = corner_fast(img, n=9, threshold=(t + 0.5) / 255.0)
        out["fast%d_th%d" % (s, t)] = np.packbits(resp > 0)
    # orientation of the strongest corners at th = 20 (31 x 31 disc mask = cv::ORB's umax table), radians, atan2(m01, m10)
    resp = corner_fast(img, n=9, threshold=20.5 / 255.0)
    ys, xs = np.nonzero(resp[20:-20, 20:-20] > 0)
    order = np.argsort(-resp[20:-20, 20:-20][ys, xs], kind="stable")[:200]
    corners = np.ascontiguousarray(np.stack([ys[order] + 20, xs[order] + 20], axis=1)).astype(np.intp)
    out["ori_corners%d" % s] = corners.astype(np.int32)
    out["ori_angles%d" % s] = corner_orientations(img, corners, np.ascontiguousarray(OFAST_MASK.astype(np.uint8))).astype(np.float64)

# eight correspondences of a synthetic two-view geometry (pixel coordinates as float32, like the tracker's), 40 sets
rng = np.random.RandomState(7)
F_sets, P1, P2 = [], [], []
for k in range(40):
    X = np.stack([rng.uniform(-4, 4, 8), rng.uniform(-3, 3, 8), rng.uniform(4, 20, 8)], axis=1)
    f, cx, cy = 800.0, 640.0, 480.0
    ang = rng.uniform(-0.05, 0.05, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
    Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
    Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = rng.uniform(-0.3, 0.3, 3)
    X2 = X @ R.T + t
    noise = 0.3 if k >= 20 else 0.0         # sets 0..19 exact (up to the float32 rounding of the coordinates), 20..39 with 0.3 px of noise
    p1 = np.stack([f * X[:, 0] / X[:, 2] + cx, f * X[:, 1] / X[:, 2] + cy], axis=1) + rng.normal(0, 1.0, (8, 2)) * noise
    p2 = np.stack([f * X2[:, 0] / X2[:, 2] + cx, f * X2[:, 1] / X2[:, 2] + cy], axis=1) + rng.normal(0, 1.0, (8, 2)) * noise
    p1 = p1.astype(np.float32); p2 = p2.astype(np.float32)
    tf = FundamentalMatrixTransform()
    assert tf.estimate(p1.astype(np.float64), p2.astype(np.float64))      # dst^T F src = 0
    F_sets.append(tf.params); P1.append(p1); P2.append(p2)
out["f8_p1"] = np.array(P1); out["f8_p2"] = np.array(P2); out["f8_F"] = np.array(F_sets)

# brute-force Hamming: for every query the FIRST train row of minimum distance (numpy argmin), as cv::BFMatcher::match
rng = np.random.RandomState(11)
q = rng.randint(0, 2, (300, 256)).astype(bool); t = rng.randint(0, 2, (400, 256)).astype(bool)
t[399] = t[3]; q[0] = t[3]; t[17] = t[5]; q[1] = t[5]; q[1, 0] ^= True                   # exact duplicates and a tie at distance 1
m = match_descriptors(q, t, metric="hamming", cross_check=False)
assert (m[:, 0] == np.arange(300)).all()
out["ham_q"] = np.packbits(q, axis=1, bitorder="little"); out["ham_t"] = np.packbits(t, axis=1, bitorder="little"); out["ham_idx"] = m[:, 1].astype(np.int32)

# one pyramid step: level 1 of image 0 (size by the oracle's rule round(dim / 1.2), passed in), bilinear, no anti-aliasing
img = out["img0"]
dh, dw = int(np.rint(np.float32(img.shape[0]) / np.float32(1.2))), int(np.rint(np.float32(img.shape[1]) / np.float32(1.2)))
r = resize(img.astype(np.float64), (dh, dw), order=1, mode="edge", anti_aliasing=False, preserve_range=True)
out["resize_src"] = img; out["resize_dst"] = r.astype(np.float32)
# ---- round 4: the steered BRIEF tests and the Harris ranking -------------------------------------------------------------
# skimage.feature.orb_cy._orb_loop applies OpenCV's learned pair table (the same 256 rows as bit_pattern_31_) rotated by the
# CONTINUOUS angle -- spc = round(cos * x - sin * y), spr = round(sin * x + cos * y): cv::ORB's computeOrbDescriptor with the two
# pattern columns named (row, column) -- to whatever image it is given.  Here: an already blurred 8-bit image, 500 positions per
# image, angles in degrees drawn as float32 (what a keypoint carries); scikit-image gets the radians the oracle forms from
# them, angle * (float)(pi / 180) in single precision, as a double.
from skimage.feature.orb_cy import _orb_loop
from skimage.feature import corner_harris
from scipy import ndimage
rng = np.random.RandomState(23)
for s in range(3):
    img = out["img%d" % s]
    bl = np.clip(np.rint(ndimage.gaussian_filter(img.astype(np.float64), 2.0, truncate=1.5)), 0, 255).astype(np.uint8)
    h, w = bl.shape
    ys = rng.randint(19, h - 19, 500); xs = rng.randint(19, w - 19, 500)
    deg = rng.uniform(0.0, 360.0, 500).astype(np.float32)
    deg[:8] = np.array([0.0, 90.0, 180.0, 270.0, 45.0, 359.99997, 12.0, 0.5], np.float32)
    rad = (deg * np.float32(0.017453292)).astype(np.float64)          # the single-precision product, exactly, as a double
    d = _orb_loop(np.ascontiguousarray(bl.astype(np.float64)), np.ascontiguousarray(np.stack([ys, xs], 1).astype(np.intp)), np.ascontiguousarray(rad))
    out["brief_img%d" % s] = bl
    out["brief_yx%d" % s] = np.stack([ys, xs], 1).astype(np.int32)
    out["brief_deg%d" % s] = deg
    out["brief_desc%d" % s] = np.packbits(np.asarray(d).astype(bool), axis=1, bitorder="little")
    # Harris measure det - 0.04 trace^2 at the FAST corners of the image (skimage: Sobel derivatives, GAUSSIAN window sigma 2 where
    # cv::ORB sums a 7 x 7 box): the values are on another scale and the windows differ, the RANKING is what the two share
    resp = corner_fast(img, n=9, threshold=20.5 / 255.0)
    cy, cx = np.nonzero(resp[24:-24, 24:-24] > 0)
    pick = rng.permutation(len(cy))[:400]
    cy, cx = cy[pick] + 24, cx[pick] + 24
    H = corner_harris(img.astype(np.float64), method="k", k=0.04, sigma=2.0)
    out["harris_yx%d" % s] = np.stack([cy, cx], 1).astype(np.int32)
    out["harris_val%d" % s] = H[cy, cx].astype(np.float64)

np.savez_compressed(os.path.join(HERE, "thirdparty_skimage.npz"), **out)
print("wrote thirdparty_skimage.npz:", {k: v.shape for k, v in out.items() if k.startswith(("ori_angles", "f8_F"))})
