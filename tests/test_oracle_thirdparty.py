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


def test_seven_point_models_contain_skimage_fundamental_matrix(tp):
    """scikit-image has no minimal solver; its FundamentalMatrixTransform.estimate is Hartley's 8-point algorithm.  For eight EXACT
    correspondences (float32 pixel coordinates) the true matrix satisfies all eight constraints, so it is one of the one-or-three
    matrices the 7-point algorithm returns for the first seven of them -- the oracle's Gauss-Jordan / Newton version of
    cv::findFundamentalMat's run7Point must therefore reproduce scikit-image's matrix (up to scale and the rounding of the
    coordinates) among its models, every model must be rank 2 and pass through its seven points; findFundamentalMat on all eight
    is the LMedS path, on the sixteen of the set taken twice the RANSAC, which explains them all from an early sample."""
    checked = 0
    for k in range(20):
        p1, p2, Fs = tp["f8_p1"][k], tp["f8_p2"][k], tp["f8_F"][k]
        B = Fs / np.linalg.norm(Fs)
        models = O.seven_point(p1[:7], p2[:7])
        assert len(models) in (1, 3)
        d = min(min(np.abs(F / np.linalg.norm(F) - B).max(), np.abs(F / np.linalg.norm(F) + B).max()) for F in models)
        assert d < 2e-4, (k, d)
        x1 = np.c_[p1[:7].astype(np.float64), np.ones(7)]; x2 = np.c_[p2[:7].astype(np.float64), np.ones(7)]
        for F in models:
            sv = np.linalg.svd(F, compute_uv=False)
            assert sv[2] < 1e-12 * sv[0]
            l = x1 @ F.T
            assert (np.abs(np.sum(x2 * l, axis=1)) / np.hypot(l[:, 0], l[:, 1])).max() < 1e-6
        # eight points are LMedS's (oracle v6: `npoints >= 15` is the RANSAC's condition): its 300 samples all run, the median of eight
        # errors is one of the seven fitted residuals (rounding noise), so the mask threshold is its floor of 0.001 px -- the winning
        # sample always passes, the eighth point only where the float32 rounding of its coordinates allows
        cnt, mask, F, best_h, n_used = O.ransac_fundamental(p1, p2)
        assert cnt in (7, 8) and n_used == 300 and 0 <= best_h < 300
        # the RANSAC proper, from 15 points on: the eight exact correspondences twice over (coincident pairs are not collinear TRIPLES
        # unless the sample's last point is one of them: such samples are drawn again) explain themselves from an early sample
        cnt, mask, F, best_h, n_used = O.ransac_fundamental(np.r_[p1, p1], np.r_[p2, p2])
        assert cnt == 16 and mask.all() and n_used <= 3
        checked += 1
    assert checked == 20


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


def test_steered_brief_matches_skimage_orb_loop(tp):
    """The 256 binary tests of cv::ORB (computeOrbDescriptor): OpenCV's learned pair table, rotated by the keypoint's CONTINUOUS angle
    and rounded to pixels, on a blurred image.  skimage.feature.orb_cy._orb_loop is that loop (same table, same rotation) in double
    precision with round-half-away; the oracle follows OpenCV's single precision and cvRound (half to even).  1500 positions x 256
    tests: every bit equal, except pairs one of whose four rotated coordinates lies within 1e-5 px of a rounding tie (there the
    float / double products and the two tie rules may round differently; counted, a handful)."""
    rows = [[int(v) for v in l.split()] for l in open(os.path.join(os.path.dirname(__file__), "golden", "orb_bit_pattern_31.txt")) if l.strip() and not l.startswith("#")]
    P = np.array(rows, np.float64)
    total, near_tie, cardinal_ok = 0, 0, 0
    for s in range(3):
        bl, yx, deg, want = tp["brief_img%d" % s], tp["brief_yx%d" % s], tp["brief_deg%d" % s], tp["brief_desc%d" % s]
        for i, ((y, x), a) in enumerate(zip(yx, deg)):
            got = O.steered_brief(bl, int(x), int(y), a)
            diff = np.unpackbits(got ^ want[i], bitorder="little").astype(bool)
            total += 256
            if not diff.any():
                cardinal_ok += int(i < 4)
                continue
            th = float(np.float32(a) * np.float32(0.017453292))
            c, sn = np.cos(th), np.sin(th)
            co = np.abs(np.stack([P[:, 0] * c - P[:, 1] * sn, P[:, 0] * sn + P[:, 1] * c, P[:, 2] * c - P[:, 3] * sn, P[:, 2] * sn + P[:, 3] * c], 1))
            tie = np.min(np.abs(co - np.floor(co) - 0.5), axis=1)
            assert not (diff & (tie > 1e-5)).any(), (s, i, float(a), np.nonzero(diff)[0], tie[diff])
            near_tie += int(diff.sum())
    assert total == 3 * 500 * 256 and near_tie < 40, near_tie
    assert cardinal_ok >= 9          # 0 / 90 / 180 / 270 degrees: exact integer coordinates on both sides


def test_harris_ranking_agrees_with_skimage(tp):
    """cv::ORB ranks a level's corners by HarrisResponses (7 x 7 box window, Sobel-like 3 x 3 gradients, k = 0.04).
    skimage.feature.corner_harris(method='k', k=0.04, sigma=2) is the same measure under a Gaussian window and another overall
    scale, so the VALUES differ; the ORDER of 400 FAST corners per image must agree closely (Spearman rank correlation), and
    what the oracle puts in its best third is, nearly all of it, in skimage's best half."""
    for s in range(3):
        img, yx, want = tp["img%d" % s], tp["harris_yx%d" % s], tp["harris_val%d" % s]
        got = np.array([O.harris(img, int(x), int(y)) for (y, x) in yx], np.float64)
        ra, rb = np.argsort(np.argsort(got)), np.argsort(np.argsort(want))
        rho = np.corrcoef(ra, rb)[0, 1]
        assert rho > 0.9, (s, rho)
        n = len(got)
        top_mine = set(np.argsort(-got)[: n // 3].tolist()); top_theirs = set(np.argsort(-want)[: n // 2].tolist())
        assert len(top_mine & top_theirs) >= 0.95 * len(top_mine), (s, len(top_mine & top_theirs), len(top_mine))
        # sign convention and scale sanity: a positive response is a corner in both
        assert ((got > 0) == (want > 0)).mean() > 0.9


def test_per_level_quota_is_the_published_geometric_split():
    """cv::ORB's nfeaturesPerLevel (no scikit-image counterpart: its ORB keeps the best n over all levels): n (1 - f) / (1 - f^L) f^l
    with f = 1 / 1.2 in single precision, cvRound per level, the last level takes the remainder.  Re-derived here in numpy float32."""
    for n, L in ((500, 8), (2000, 8), (1350, 8), (3300, 3), (77, 5), (2000, 1)):
        f = np.float32(1.0 / 1.2)
        nd = np.float32(n) * (np.float32(1.0) - f) / (np.float32(1.0) - np.float32(float(f) ** L))
        want, tot = [], 0
        for l in range(L - 1):
            q = int(np.rint(nd)); want.append(q); tot += q
            nd = np.float32(nd * f)
        want.append(max(n - tot, 0))
        assert O.level_quota(n, L) == want, (n, L, O.level_quota(n, L), want)
        assert sum(want) == n
