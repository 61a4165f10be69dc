"""The oracle against a THIRD-PARTY implementation of the same published algorithms: scikit-image 0.18.3, run in the build container
by tests/golden/make_thirdparty_skimage.py (it cannot travel; its outputs are committed as tests/golden/thirdparty_skimage.npz).
Not the reference's dependency (OpenCV, absent here), so this narrows "parity unpinned", it does not lift it -- DESIGN.md section 3."""
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def tp(golden_dir):
    return np.load(os.path.join(golden_dir, "thirdparty_skimage.npz"))


def test_fast9_corner_sets_and_scores_match_skimage(tp):
    """FAST-9/16 segment test: the same corner set as skimage.feature.corner_fast(n=9) at six thresholds, both from a score map
    computed AT that threshold and from the scores of the lowest one (score = largest threshold at which the pixel is still a
    corner, cv::FAST's cornerScore), on three textured images."""
    ths = [int(t) for t in tp["thresholds"]]
    total = 0
    for s in range(3):
        img = tp["img%d" % s]
        h, w = img.shape
        base = O.fast_score_map(img, ths[0])
        for t in ths:
            want = np.unpackbits(tp["fast%d_th%d" % (s, t)])[: h * w].reshape(h, w).astype(bool)
            got_at = O.fast_score_map(img, t) > 0
            got_from_scores = base >= t
            # both leave the 3-pixel border alone
            assert not want[:3].any() and not want[-3:].any() and not want[:, :3].any() and not want[:, -3:].any()
            assert (got_at == want).all(), (s, t, int((got_at != want).sum()))
            assert (got_from_scores == want).all(), (s, t, int((got_from_scores != want).sum()))
            total += int(want.sum())
    assert total > 5000          # the images are not empty of corners


def test_orientation_matches_skimage_intensity_centroid(tp):
    """oFAST orientation: atan2(m01, m10) over the 31x31 disc (skimage.feature.corner_orientations with ORB's own mask) against the
    oracle's degrees; the oracle's atan2 is a 7th-order polynomial (cv::fastAtan2's accuracy class, ~0.03 degrees)."""
    worst = 0.0
    for s in range(3):
        img = tp["img%d" % s]
        for (y, x), a in zip(tp["ori_corners%d" % s], tp["ori_angles%d" % s]):
            want = np.degrees(a) % 360.0
            got = O.orb_angle(img, int(x), int(y))
            d = abs(got - want); d = min(d, 360.0 - d)
            worst = max(worst, d)
            assert d < 0.05, (s, int(x), int(y), got, want)
    assert worst > 0.0


def test_eight_point_fundamental_matrix_matches_skimage(tp):
    """Exactly eight correspondences: every RANSAC sample is those eight, so the model the oracle returns is its normalised
    8-point solution with rank-2 enforcement; skimage.transform.FundamentalMatrixTransform.estimate is the same Hartley
    algorithm through two SVDs.  Same matrix up to scale (dst^T F src = 0 in both), rank 2 in both."""
    checked = 0
    for k, (p1, p2, Fs) in enumerate(zip(tp["f8_p1"], tp["f8_p2"], tp["f8_F"])):
        cnt, mask, F, best_h, n_used = O.ransac_fundamental(p1, p2)
        if cnt < 8:
            continue             # noise can leave a pair beyond 1 px of its epipolar line: the model is still the eight-point one
        A, B = F / np.linalg.norm(F), Fs / np.linalg.norm(Fs)
        if np.sum(A * B) < 0:
            B = -B
        # sets 0..19 are exact correspondences (float32 coordinates): the null vector does not depend on the normalisation.
        # sets 20..39 carry 0.3 px of noise: scikit-image normalises to an RMS distance of sqrt 2, the oracle (as
        # cv::findFundamentalMat) to a mean distance, so the rank-2 projection is taken in slightly different coordinates
        assert np.abs(A - B).max() < (2e-6 if k < 20 else 5e-3), (k, np.abs(A - B).max())
        sv = np.linalg.svd(F, compute_uv=False)
        assert sv[2] < 1e-9 * sv[0]
        checked += 1
    assert checked >= 25
    # and the epipolar constraint holds for the oracle's model on its own sample
    cnt, mask, F, _, _ = O.ransac_fundamental(tp["f8_p1"][0], tp["f8_p2"][0])
    x1 = np.c_[tp["f8_p1"][0].astype(np.float64), np.ones(8)]; x2 = np.c_[tp["f8_p2"][0].astype(np.float64), np.ones(8)]
    l = x1 @ F.T
    d = np.abs(np.sum(x2 * l, axis=1)) / np.hypot(l[:, 0], l[:, 1])
    assert d.max() < 1.0


def test_brute_force_hamming_matches_skimage_first_minimum(tp):
    """first minimum of the Hamming distance over the train set, ties and exact duplicates planted (cv::BFMatcher::match's rule)"""
    idx, dist = O.hamming_bf(tp["ham_q"], tp["ham_t"])
    assert (idx == tp["ham_idx"]).all()
    assert idx[0] == 3 and dist[0] == 0 and idx[1] == 5 and dist[1] == 1


def test_one_pyramid_step_matches_skimage_bilinear(tp):
    """x1/1.2 with half-pixel centres: skimage's float bilinear against the oracle's 11-bit fixed-point weights, away from the
    borders (clamping differs there): within one grey level everywhere, identical on the large majority of the pixels"""
    src, want = tp["resize_src"], tp["resize_dst"]
    dh, dw = want.shape
    got = O.resize(src, dw, dh).astype(np.float32)
    d = np.abs(got - want)[2:-2, 2:-2]
    assert d.max() <= 1.0, d.max()
    assert (np.abs(got - np.rint(want))[2:-2, 2:-2] == 0).mean() > 0.9
