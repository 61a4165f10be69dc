"""Two independent readings of the published algorithms agree: the oracle's frozen stand-ins for what the reference
delegates to OpenCV / MRPT / Eigen (oracle/svo_oracle.c) against tests/independent_ref.py (numpy / scipy, written from
the same papers, not from the oracle).  This does not pin the oracle to the REAL OpenCV -- nothing in this image can
 -- but it turns "equal to ourselves" into "equal to a second, differently formulated implementation".
Integer quantities are compared exactly; where the papers leave a discretisation open (8-bit vs float Gaussian,
polynomial vs exact atan2) the comparison states its tolerance."""
import numpy as np
import pytest

import independent_ref as IR
from oracle import oracle as O
from stereo_vo_amd.abi import StereoCamera
from stereo_vo_amd.synth import SyntheticStereoWorld


@pytest.fixture(scope="module")
def scene():
    w = SyntheticStereoWorld(640, 480, 400.0, 0.12, seed=5, n_frames=2)
    return [x.numpy() for x in w.render(1)]


def test_fast_score_and_nms_exact(scene):
    rng = np.random.RandomState(1)
    for img, th in ((scene[0][:200, :260], 20), (scene[1][100:300, 300:560], 35), (rng.randint(0, 256, (120, 150)).astype(np.uint8), 10),
                    (np.clip(rng.randint(0, 4, (90, 90)) * 80, 0, 255).astype(np.uint8), 20)):
        img = np.ascontiguousarray(img)
        mine = IR.fast9_score_map(img, th)
        theirs = O.fast_score_map(img, th).astype(np.int32)
        assert (mine == theirs).all(), ("FAST score", th, int((mine != theirs).sum()))
        assert (mine > 0).sum() > 20
    # 3x3 NMS + 31-px border on a full image: the candidate set of the detector (every level-0 ORB keypoint is one of them)
    img = scene[0]
    keep = IR.nms3x3(IR.fast9_score_map(img, 20), 31)
    k, _ = O.orb_detect(img, 400, 1, 20)
    assert len(k) > 200
    assert all(keep[int(p["y"]), int(p["x"])] for p in k)
    kf, _ = O.fast_orb_detect(img, 20)                 # cv::FAST stand-in: ALL maxima, row-major, response = score
    ys, xs = np.nonzero(keep)
    assert len(kf) == len(ys) and (kf["x"].astype(int) == xs).all() and (kf["y"].astype(int) == ys).all()
    assert (kf["response"].astype(int) == IR.fast9_score_map(img, 20)[ys, xs]).all()


def test_harris_response_and_ranking(scene):
    img = scene[0]
    k, _ = O.orb_detect(img, 300, 1, 20)
    H = IR.harris_map(img)
    mine = H[k["y"].astype(int), k["x"].astype(int)]
    assert np.allclose(mine, k["response"], rtol=2e-5, atol=1e-12)
    # ORB keeps the quota best by Harris among the 2 * quota best by FAST score: the kept ones out-rank the dropped ones
    s = IR.fast9_score_map(img, 20); keep = IR.nms3x3(s, 31)
    ys, xs = np.nonzero(keep)
    # ... "2 * quota best" as KeyPointsFilter::retainBest means it: with everything that ties with the last of them
    order = np.lexsort((ys * img.shape[1] + xs, -s[ys, xs]))
    boundary = s[ys[order[599]], xs[order[599]]]
    order = [i for i in order if s[ys[i], xs[i]] >= boundary]
    assert len(order) >= 600
    cand = set(zip(xs[order].tolist(), ys[order].tolist()))
    got = set(zip(k["x"].astype(int).tolist(), k["y"].astype(int).tolist()))
    assert got <= cand and len(got) == 300
    dropped = [H[y, x] for (x, y) in cand - got]
    assert min(mine) >= max(dropped) - 1e-12


def test_orientation_and_steered_brief(scene):
    img = scene[0]
    k, d = O.orb_detect(img, 300, 1, 20)
    pairs = IR.brief_pairs()
    assert pairs.shape == (256, 4) and np.abs(pairs).max() <= 13 and tuple(pairs[0]) == (8, -3, 9, 5)
    ang = np.array([IR.ic_angle_deg(img, int(p["x"]), int(p["y"])) for p in k])
    da = np.abs((ang - k["angle"] + 180.0) % 360.0 - 180.0)
    assert da.max() < 0.5, da.max()                          # polynomial atan2 of the oracle (cv::fastAtan2): ~0.3 degrees off the exact one
    # the descriptor is steered by the keypoint's OWN angle (cv::ORB: kpt.angle, the polynomial one), continuously: the second
    # reading rotates in double precision with exact cos / sin and a float Gaussian; a bit may differ only where the two blurs
    # (float vs 8-bit taps) cannot separate the two samples, or where a rotated coordinate sits on a rounding tie
    n_cmp, wrong_decisive, agree = 0, 0, []
    for p, desc in zip(k, d):
        mine, margin, tie = IR.steered_brief(img, int(p["x"]), int(p["y"]), float(p["angle"]), pairs)
        diff = np.unpackbits(mine ^ desc, bitorder="little").astype(bool)
        agree.append(1.0 - diff.mean())
        wrong_decisive += int((diff & (margin > 1.5) & (tie > 1e-4)).sum())   # a float blur and an 8-bit blur differ by < 1 grey level
        n_cmp += 1
    assert n_cmp >= 300
    assert wrong_decisive == 0
    assert np.mean(agree) > 0.97, np.mean(agree)


def test_pyramid_level_is_half_pixel_centred_bilinear(scene):
    img = scene[0]
    lw, lh, sc = O.pyramid_sizes(640, 480, 4)
    assert (lw[1], lh[1]) == (533, 400) and abs(sc[1] - 1.2) < 1e-6
    prev = img
    for l in range(1, 4):
        got = O.resize(prev, lw[l], lh[l]).astype(np.float64)
        mine = IR.bilinear_resize(prev, lw[l], lh[l])
        # oracle v7 = cv::resize's 8-bit path: the 11-bit weights of each axis (<= 0.125), the row sums cut to 1/128 grey level, each of the two
        # weighted rows cut DOWN to a quarter grey level before they are added (up to -0.5 together), then one rounding of the quarters
        # (+-0.5): the result leans low -- within (-1.13, +0.63] of the float bilinear value, -0.12 on average, and equal to the rounded
        # float value on 87 % of the pixels (the single-rounding form of versions <= 6 stayed within +-0.63)
        d = got - mine
        assert -1.13 < d.min() and d.max() <= 0.63 and np.abs(d).mean() < 0.30 and -0.2 < d.mean() < -0.05
        assert (got == np.floor(mine + 0.5)).mean() > 0.85
        prev = got.astype(np.uint8)


def test_seven_point_and_epipolar_distance():
    rng = np.random.RandomState(3)
    K = np.array([[800.0, 0, 640], [0, 800.0, 480], [0, 0, 1]])
    X = np.c_[rng.uniform(-4, 4, 400), rng.uniform(-3, 3, 400), rng.uniform(4, 20, 400)]
    R = IR.Rotation.from_rotvec([0.01, -0.03, 0.005]).as_matrix(); t = np.array([0.05, -0.01, 0.3])
    x1 = X @ K.T; x1 = x1[:, :2] / x1[:, 2:]
    Y = X @ R.T + t; x2 = Y @ K.T; x2 = x2[:, :2] / x2[:, 2:]
    x1, x2 = x1.astype(np.float32), x2.astype(np.float32)
    # the minimal solver (oracle version 4: cv::findFundamentalMat's RANSAC solves SEVEN points per sample): for 40 samples of seven
    # correspondences -- exact ones and noisy ones -- the oracle's Gauss-Jordan null space + Newton / deflation roots give the same
    # one or three matrices as the SVD null space + companion-matrix roots of the independent reading, each of rank 2 and each
    # through all seven points
    n_three = 0
    for k in range(40):
        idx = rng.choice(400, 7, replace=False)
        a, b = x1[idx].copy(), x2[idx].copy()
        if k >= 20:
            a += rng.normal(0, 0.7, a.shape).astype(np.float32); b += rng.normal(0, 0.7, b.shape).astype(np.float32)
        mine = IR.seven_point(a.astype(np.float64), b.astype(np.float64))
        theirs = O.seven_point(a, b)
        assert len(theirs) == len(mine) and len(mine) in (1, 3), (k, len(theirs), len(mine))
        n_three += len(mine) == 3
        for F in theirs:
            Fo = F / np.linalg.norm(F)
            assert min(min(np.abs(Fo - Fm).max(), np.abs(Fo + Fm).max()) for Fm in mine) < 1e-7, k
            sv = np.linalg.svd(Fo, compute_uv=False)
            assert sv[2] < 1e-12 * sv[0]                                  # rank 2 by construction: a root of det F = 0
            assert IR.symmetric_epipolar_sq(Fo, a.astype(np.float64), b.astype(np.float64)).max() < 1e-10
    assert 5 <= n_three <= 38
    # exactly 7 correspondences: every sample of the schedule is those seven, the first one already explains them all
    cnt, mask, F, bh, nu = O.ransac_fundamental(x1[:7], x2[:7])
    assert cnt == 7 and mask.all() and bh == 0
    assert any(np.abs(np.abs(F / np.linalg.norm(F)) - np.abs(Fm)).max() < 1e-7 for Fm in IR.seven_point(x1[:7].astype(np.float64), x2[:7].astype(np.float64)))
    # RANSAC on 400 points with 30 % gross outliers: the inlier mask is exactly "symmetric epipolar distance <= 1 px"
    # under the returned model, and the true correspondences are found
    bad = rng.choice(400, 120, replace=False)
    x2o = x2.copy(); x2o[bad] += rng.uniform(20, 80, (120, 2)).astype(np.float32) * rng.choice([-1, 1], (120, 2))
    cnt, mask, F, bh, nu = O.ransac_fundamental(x1, x2o)
    e = IR.symmetric_epipolar_sq(F, x1.astype(np.float64), x2o.astype(np.float64))
    clear = np.abs(e - 1.0) > 1e-6
    assert ((e <= 1.0) == mask.astype(bool))[clear].all() and cnt == int(mask.sum())
    good = np.setdiff1d(np.arange(400), bad)
    assert mask[good].mean() > 0.9 and mask[bad].mean() < 0.1
    # the schedule stopped where the 0.99-confidence rule says, well below the 1000-hypothesis cap
    w7 = (cnt / 400.0) ** 7
    assert nu <= np.ceil(np.log(0.01) / np.log(1.0 - w7)) + 1 and 7 <= nu < 1000


def test_pose_conventions_against_scipy():
    rng = np.random.RandomState(9)
    cam = StereoCamera.simple(800.0, 639.5, 479.5, 0.12, 1280, 960)
    for _ in range(30):
        delta = np.r_[rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.5, 0.5, 3)]
        assert np.allclose(O.delta_to_pose(delta), IR.delta_to_pose(delta), atol=1e-12)
        assert np.allclose(O.pose_to_delta(O.delta_to_pose(delta)), delta, atol=1e-10)
        lm = np.c_[rng.uniform(-3, 3, 50), rng.uniform(-2, 2, 50), rng.uniform(3, 25, 50)]
        pix, jac = O.project(lm, cam, delta)
        assert np.allclose(pix, IR.project(lm, cam, delta), rtol=0, atol=2e-3)          # reference stores pixels as float32 (S5:188-195)
        eps = 1e-6                                                                     # analytic Jacobian vs central differences of the independent projection
        for j in range(6):
            dp = np.zeros(6); dp[j] = eps
            num = (IR.project(lm, cam, delta + dp) - IR.project(lm, cam, delta - dp)) / (2 * eps)
            # the reference writes dr22/dw3 with (w2^2 + w3^2) where the true derivative has (w1^2 + w2^2) (S5:162); the oracle
            # keeps it, hence the looser bound on the w3 column
            tol = 5e-3 if j == 2 else 2e-4
            assert np.abs(jac[:, :, j] - num).max() < tol * max(1.0, np.abs(num).max()), (j, np.abs(jac[:, :, j] - num).max())


def test_per_level_feature_split(scene):
    """cv::ORB hands level l about nfeatures (1 - f) / (1 - f^L) f^l keypoints, f = 1 / 1.2, rounded level by level in single
    precision, the last level taking the remainder (Rublee et al. 2011, sec. 6.1 "scale pyramid"; OpenCV's computeKeyPoints).
    On an image with corners to spare the detector's per-level counts ARE that split."""
    rng = np.random.RandomState(3)
    img = (rng.rand(480, 640) * 255).astype(np.uint8)              # white noise: thousands of FAST corners on every level
    for nfeatures, nlevels in ((500, 8), (1000, 5), (300, 3), (77, 2), (50, 1)):
        k, _ = O.orb_detect(img, nfeatures, nlevels, 20)
        got = [int((k["octave"] == l).sum()) for l in range(nlevels)]
        f = np.float32(1.0 / 1.2)
        nd = np.float32(nfeatures) * (np.float32(1) - f) / (np.float32(1) - np.float32(float(f) ** nlevels))
        want, total = [], 0
        for l in range(nlevels - 1):
            n = int(np.rint(nd)); want.append(n); total += n; nd = np.float32(nd * f)
        want.append(max(nfeatures - total, 0))
        assert got == want and sum(got) == nfeatures, (nfeatures, nlevels, got, want)


def test_pyramid_level_sizes():
    """level l of cv::ORB's pyramid: scale = (float) 1.2^l, size = round(dimension / scale) (OpenCV's getScale / cvRound)"""
    for w, h in ((1280, 960), (1241, 376), (640, 480), (2048, 1536), (417, 311)):
        lw, lh, sc = O.pyramid_sizes(w, h, 8)
        for l in range(8):
            s = np.float32(1.2 ** l)
            assert sc[l] == float(s), (w, h, l)
            assert (lw[l], lh[l]) == (int(np.rint(np.float32(w) / s)), int(np.rint(np.float32(h) / s))), (w, h, l, lw[l], lh[l])


def test_ransac_samples_are_cv_rng_under_getsubset_rules():
    """Oracle v5 draws the minimal samples as OpenCV does (VERDICT r04 missing #2; S4:202, 237 call cv::findFundamentalMat(FM_RANSAC)):
    cv::RNG is a multiply-with-carry generator -- state <- (uint32)state * 4164903690 + (state >> 32), output (uint32)state -- seeded
    (uint64)-1 by every RANSACPointSetRegistrator::run; getSubset takes rng.uniform(0, count) = next() % count, draws a repeated index
    again, and starts an attempt over when checkSubset finds the LAST point collinear with two earlier ones in either image.
    A literal Python reading (big integers, no C) against the oracle's sample lists, for point counts from 8 to 700 and for a point set
    with planted collinear triples and coincident points; plus the generator's defining lag-1 identity as an independent check of the table."""
    A, M32 = 4164903690, (1 << 32) - 1

    def stream():
        st = (1 << 64) - 1
        while True:
            st = (st & M32) * A + (st >> 32)
            yield st & M32

    raw = O.cv_rng_raw(4096)
    g = stream()
    assert [int(x) for x in raw[:512]] == [next(g) for _ in range(512)]
    # the multiply-with-carry identity: a * x_n + c_n = c_{n+1} * 2^32 + x_{n+1}; with state = c << 32 | x that is how the C code steps,
    # and it makes the sequence a Lehmer generator modulo a * 2^32 - 1: state_{n+1} * 2^32 == state_n  (mod a * 2^32 - 1)
    st, mod = (1 << 64) - 1, A * (1 << 32) - 1
    for x in raw[:64]:
        nxt = (st & M32) * A + (st >> 32)
        assert (nxt << 32) % mod == st % mod and (nxt & M32) == int(x)
        st = nxt

    def collinear(p, idx):
        # haveCollinearPoints: Point2f differences are taken in FLOAT, then widened (`double dx1 = ptr[j].x - ptr[i].x;`, oracle v7)
        i = 6
        assert p.dtype == np.float32
        for j in range(i):
            dx1 = float(p[idx[j], 0] - p[idx[i], 0]); dy1 = float(p[idx[j], 1] - p[idx[i], 1])
            for k in range(j):
                dx2 = float(p[idx[k], 0] - p[idx[i], 0]); dy2 = float(p[idx[k], 1] - p[idx[i], 1])
                if abs(dx2 * dy1 - dy2 * dx1) <= np.finfo(np.float32).eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                    return True
        return False

    def samples(p1, p2, count):
        g, out, n = stream(), [], len(p1)
        max_attempts = 1000 if n <= 14 else 10000      # getSubset's maxAttempts: LMedS's run takes the default, the RANSAC's passes 10000 (oracle v7)
        while len(out) < count:
            for _ in range(max_attempts):
                idx = []
                while len(idx) < 7:
                    v = next(g) % n
                    if v not in idx:
                        idx.append(v)
                if not (collinear(p1, idx) or collinear(p2, idx)):
                    break
            else:
                break                                   # `if (!found) break;`: the run ends with the samples it has
            out.append(idx)
        return np.array(out, np.int32).reshape(-1, 7)

    rng = np.random.RandomState(17)
    n_rejected = 0
    for n in (8, 9, 14, 15, 37, 300, 700):
        p1 = rng.uniform(40, 1200, (n, 2)).astype(np.float32); p2 = (p1 + rng.normal(0, 3, (n, 2))).astype(np.float32)
        if n in (37, 300):      # planted degeneracies: a run of points on one line, two coincident pairs
            p1[5:14, 1] = p1[5, 1]; p2[5:14] = p1[5:14]
            p1[20] = p1[21]; p2[20] = p2[21]
        want = samples(p1, p2, 120)
        got = O.ransac_samples(p1, p2, 120)
        assert got.shape == want.shape and (got == want).all(), n
        g, free = stream(), 0          # how many attempts the degeneracies cost (the plain orbit, without checkSubset)
    # the rejection really happens on the planted sets: the unfiltered draw sequence differs from the filtered one
    p1 = rng.uniform(40, 1200, (12, 2)).astype(np.float32); p1[:9, 0] = 100.0
    p2 = p1.copy()
    got = O.ransac_samples(p1, p2, 5)
    for s in got:
        assert not collinear(p1, list(s))
    # every point of image 1 on one line: no attempt ever passes, getSubset gives up (after 1000 attempts for LMedS's 12 points, 10000 for the
    # RANSAC's 40) and the run ends without a sample -- literal reading and oracle alike
    for n in (12, 40):
        q1 = np.c_[rng.uniform(50, 600, n), np.full(n, 77.0)].astype(np.float32); q2 = rng.uniform(50, 600, (n, 2)).astype(np.float32)
        assert len(samples(q1, q2, 3)) == 0 and len(O.ransac_samples(q1, q2, 3)) == 0
    # sub-pixel coordinates (scaled octaves) whose float differences round: the FLOAT subtraction decides, as in the oracle
    q1 = (rng.uniform(40, 1200, (60, 2)) / 1.2 ** 3).astype(np.float32); q2 = (q1 * np.float32(1.0001) + np.float32(0.3)).astype(np.float32)
    q1[10:30, 1] = q1[10, 1] + np.arange(20, dtype=np.float32) * np.float32(1e-5)           # a nearly horizontal run: cross products at the threshold
    assert (O.ransac_samples(q1, q2, 200) == samples(q1, q2, 200)).all()
    # fewer than 8 points: no sampling at all (exactly 7 take findFundamentalMat's direct path)
    assert len(O.ransac_samples(p1[:7], p2[:7], 3)) == 0


def test_eight_to_fourteen_points_run_lmeds():
    """Oracle v6: cv::findFundamentalMat(FM_RANSAC) runs the RANSAC only from 15 points on (`npoints >= 15`); 8...14 points go to the LMedS
    registrator (S4:202, 237, 684, 696 pass whatever the tracker has left).  A literal Python reading of LMeDSPointSetRegistrator::run --
    300 samples (RANSACUpdateNumIters(0.99, 0.45, 7, 1000)), the same generator and getSubset, per model the float errors
    (float)max(d1^2 s1, d2^2 s2), their median = the element of rank n / 2, the first strictly smaller median wins,
    sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median) >= 0.001, mask = err <= (float)sigma^2 -- against the oracle, bit for bit
    (Python floats are IEEE doubles without contraction, as the oracle is compiled).  Samples and models come from the oracle's entry
    points that the tests above pin independently (cv::RNG / getSubset reading; SVD + companion-matrix solver)."""
    import math
    assert int(round(math.log(0.01) / math.log(1.0 - 0.55 ** 7))) == 300

    def err32(F, a, b):
        F = [float(v) for v in F.reshape(9)]
        x1, y1, x2, y2 = float(a[0]), float(a[1]), float(b[0]), float(b[1])
        A = (F[0] * x1 + F[1] * y1) + F[2]; B = (F[3] * x1 + F[4] * y1) + F[5]; Cc = (F[6] * x1 + F[7] * y1) + F[8]
        with np.errstate(all="ignore"):
            s2 = np.float64(1.0) / np.float64(A * A + B * B); d2 = (x2 * A + y2 * B) + Cc
            A = (F[0] * x2 + F[3] * y2) + F[6]; B = (F[1] * x2 + F[4] * y2) + F[7]; Cc = (F[2] * x2 + F[5] * y2) + F[8]
            s1 = np.float64(1.0) / np.float64(A * A + B * B); d1 = (x1 * A + y1 * B) + Cc
            e1, e2 = (d1 * d1) * s1, (d2 * d2) * s2
            return np.float32(e2 if e1 < e2 else e1)                      # std::max(e1, e2)

    def lmeds(p1, p2):
        n = len(p1)
        smp = O.ransac_samples(p1, p2, 300)
        best, best_k, Fb = None, -1, None
        for k, idx in enumerate(smp):
            for F in O.seven_point(p1[idx], p2[idx]):
                e = np.array([err32(F, p1[i], p2[i]) for i in range(n)], np.float32)
                med = np.sort(e.view(np.int32))[n // 2:n // 2 + 1].view(np.float32)[0]      # nth_element over the bit patterns as ints
                if np.isfinite(med) and (best is None or float(med) < best):
                    best, best_k, Fb = float(med), k, F
        if best is None:
            return 0, np.zeros(n, bool), -1, len(smp)
        sigma = max(2.5 * 1.4826 * (1.0 + 5.0 / (n - 7)) * math.sqrt(best), 0.001)
        t = np.float32(sigma * sigma)
        mask = np.array([err32(Fb, p1[i], p2[i]) <= t for i in range(n)])
        return int(mask.sum()), mask, best_k, len(smp)

    f, cx, cy = 700.0, 600.0, 180.0
    seen_filter_on = 0
    for n in range(8, 15):
        for seed in range(3):
            r = np.random.RandomState(100 * n + seed)
            X = np.c_[r.uniform(-3, 3, n), r.uniform(-1, 1, n), r.uniform(4, 12, n)]
            p1 = np.c_[f * X[:, 0] / X[:, 2] + cx, f * X[:, 1] / X[:, 2] + cy].astype(np.float32)
            Y = X + np.array([0.1, 0.02, -0.4])
            p2 = np.c_[f * Y[:, 0] / Y[:, 2] + cx, f * Y[:, 1] / Y[:, 2] + cy]
            p2[:seed] += r.uniform(-30, 30, (seed, 2))                    # 0, 1 or 2 gross outliers
            p2 = (p2 + r.normal(0, 0.2, p2.shape)).astype(np.float32)
            want = lmeds(p1, p2)
            cnt, mask, F, bh, nu = O.ransac_fundamental(p1, p2)
            assert (cnt, bh, nu) == (want[0], want[2], want[3]) and (mask.astype(bool) == want[1]).all(), (n, seed, cnt, bh, nu, want)
            assert nu == 300
            # below 14 points the element of rank n / 2 <= 6 is one of the seven errors the minimal sample fits exactly: the median is
            # rounding noise, sigma its floor of 0.001 px, and the mask the sample itself -- seven "inliers", under the eight S4:205 asks for
            if n <= 13:
                assert cnt == 7
            seen_filter_on += cnt >= 8
    assert seen_filter_on >= 1                                             # 14 points: the median is a real residual and the mask can pass S4:205
    # 15 points: the RANSAC again (the budget shrinks with the first good model)
    r = np.random.RandomState(5)
    X = np.c_[r.uniform(-3, 3, 15), r.uniform(-1, 1, 15), r.uniform(4, 12, 15)]
    p1 = np.c_[f * X[:, 0] / X[:, 2] + cx, f * X[:, 1] / X[:, 2] + cy].astype(np.float32)
    Y = X + np.array([0.1, 0.02, -0.4])
    p2 = np.c_[f * Y[:, 0] / Y[:, 2] + cx, f * Y[:, 1] / Y[:, 2] + cy].astype(np.float32)
    cnt, mask, F, bh, nu = O.ransac_fundamental(p1, p2)
    assert cnt == 15 and nu < 300


def test_stop_rule_is_ransac_update_num_iters_for_every_count():
    """The RANSAC's budget after a model with cnt inliers of n is cv::RANSACUpdateNumIters(0.99, ep = (n - cnt) / n, 7, maxIters) =
    cvRound(log(1 - 0.99) / log(1 - pow(1 - ep, 7))) capped at maxIters.  OpenCV evaluates that with libm; the oracle (and, verbatim, the
    kernels) with its own logarithm from IEEE +, -, *, / and w^7 by multiplication so that both sides get the same bits.  What matters is
    the INTEGER: the literal libm expression against the oracle's for every (cnt, n) with n <= 320 under the full budget (all 4.5 million pairs up to n = 3000 were compared once, in C: none differs), and under
    shrunken budgets (the rule is applied to the budget left) for a sample of pairs up to n = 4000."""
    import math
    DBL_MIN = 2.2250738585072014e-308

    def literal(cnt, n, max_iters):
        ep = min(max(float(n - cnt) / n, 0.0), 1.0)
        num = max(1.0 - 0.99, DBL_MIN)
        denom = 1.0 - math.pow(1.0 - ep, 7)
        if denom < DBL_MIN:
            return 0
        num, denom = math.log(num), math.log(denom)
        if denom >= 0 or -num >= max_iters * (-denom):
            return max_iters
        q = num / denom
        return int(math.floor(q + 0.5)) if abs(q - round(q)) != 0.5 else int(2 * round(q / 2))      # cvRound: to nearest, ties to even

    checked = 0
    for n in range(8, 321):
        for cnt in range(7, n + 1):
            assert O.ransac_niters(cnt, n, 1000) == literal(cnt, n, 1000), (cnt, n)
            checked += 1
    rng = np.random.RandomState(4)
    for _ in range(20000):
        n = int(rng.randint(8, 4001)); cnt = int(rng.randint(7, n + 1)); k = int(rng.randint(1, 1001))
        assert O.ransac_niters(cnt, n, k) == literal(cnt, n, k), (cnt, n, k)
    assert checked > 45000
    assert O.ransac_niters(7, 8, 1000) == literal(7, 8, 1000) and O.ransac_niters(300, 300, 1000) == 0      # all inliers: the loop ends


def test_pyramid_step_is_cv_resize_8bit_linear_literal_reading():
    """Oracle v7: a pyramid level is cv::resize(prev, INTER_LINEAR) on 8-bit data as OpenCV 2.4 / 3.x compute it.  A literal
    numpy reading of that code path, written from the same recollection but sharing no code with the C oracle: tap positions and weights in
    FLOAT (`fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx`), the two weights rounded separately to shorts
    (`saturate_cast<short>(w * 2048)`, half to even), horizontal pass in int, and the uchar specialisation of the vertical pass with its
    two-step rounding: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  Byte for byte, over the level sizes of every
    benchmarked image size and on noise."""
    def table(src, dst):
        scale = 1.0 / (float(dst) / float(src))
        fx = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        sx = np.floor(fx.astype(np.float64)).astype(np.int64)
        fx = fx - sx.astype(np.float32)                                    # float32 - float32
        lo, hi = sx < 0, sx >= src - 1
        fx[lo | hi] = np.float32(0); sx[lo] = 0; sx[hi] = src - 1
        a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)      # numpy rint: half to even, as cvRound
        a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
        return sx, a0, a1

    def cv_resize(img, dw, dh):
        H, W = img.shape
        xi, a0, a1 = table(W, dw); yi, b0, b1 = table(H, dh)
        im = img.astype(np.int64)
        x1 = np.minimum(xi + 1, W - 1)
        Hrow = im[:, xi] * a0[None, :] + im[:, x1] * a1[None, :]           # HResizeLinear: every source row once
        y1 = np.minimum(yi + 1, H - 1)
        S0, S1 = Hrow[yi], Hrow[y1]
        v = (((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2
        assert v.min() >= 0 and v.max() <= 255
        return v.astype(np.uint8)

    rng = np.random.RandomState(6)
    n_off_weights = 0
    for (W, H) in ((640, 480), (1280, 960), (1241, 376), (97, 65)):
        lw, lh, _ = O.pyramid_sizes(W, H, 8)
        img = rng.randint(0, 256, (H, W)).astype(np.uint8)
        if W == 640:
            img = scene_image()
        prev = img
        for l in range(1, 8):
            xi, a0, a1 = table(prev.shape[1], int(lw[l]))
            n_off_weights += int((a0 + a1 != 2048).sum())
            got, want = O.resize(prev, int(lw[l]), int(lh[l])), cv_resize(prev, int(lw[l]), int(lh[l]))
            assert got.shape == want.shape and (got == want).all(), (W, H, l, int((got != want).sum()))
            prev = got
    # the tables themselves through the oracle's hook: same first taps, same weight pairs
    for src, dst in ((1280, 1067), (960, 800), (1067, 889), (376, 313), (51, 43)):
        idx, w01 = O.resize_table(src, dst)
        sx, a0, a1 = table(src, dst)
        assert (idx == sx).all() and ((w01 & 0xFFFF) == a0).all() and ((w01 >> 16) == a1).all()
    assert n_off_weights == 0        # (on these sizes the separately rounded weights always add up to 2048; the code does not rely on it)


def scene_image():
    w = SyntheticStereoWorld(640, 480, 400.0, 0.12, seed=5, n_frames=2)
    return w.render(1)[0].numpy()


def test_seven_point_with_a_vanishing_cubic_coefficient():
    """Oracle v7: run7Point hands det(lambda g + f2) to cv::solveCubic, which has a branch for a leading coefficient that is EXACTLY zero (then
    a quadratic, or a linear equation).  Integer coordinates under a pure image translation reach it: up to version 6 the oracle divided by
    zero there (NaN models, no inliers).  Now: the matrix g itself (the root at infinity of this parametrisation) first, then the finite
    roots -- and for a pure translation in x one of them is the exact answer [t]_x.  Every model lies in the null space of the seven
    constraints and is singular."""
    cases = [(np.array([[-1, 3, -1, 0, 3, 2, -1], [3, 1, -1, 3, -3, 1, -3]], np.float32).T, np.array([1.0, 0.0], np.float32))]
    rng = np.random.RandomState(0)
    hits = 0
    for trial in range(4000):
        if trial < len(cases):
            p1, shift = cases[trial]
        else:
            p1 = rng.randint(-3, 4, (7, 2)).astype(np.float32); shift = np.array([rng.randint(1, 4), 0], np.float32)
        p2 = p1 + shift
        F = O.seven_point(p1, p2)
        if len(F) != 2:
            continue
        hits += 1
        assert np.isfinite(F).all()
        h1, h2 = np.c_[p1, np.ones(7)].astype(np.float64), np.c_[p2, np.ones(7)].astype(np.float64)
        found_exact = False
        for M in F:
            sc = np.abs(M).max()
            assert sc > 0
            assert np.abs(np.einsum("ni,ij,nj->n", h2, M, h1)).max() < 1e-9 * sc          # x2^T F x1 = 0 for the seven pairs
            assert abs(np.linalg.det(M / sc)) < 1e-9                                       # singular
            E = M / sc
            found_exact |= bool(np.abs(E - E[2, 1] * np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]])).max() < 1e-9 and abs(E[2, 1]) > 0.5)
        assert found_exact                                                                  # the epipolar geometry of a pure x translation
    assert hits >= 20, hits
