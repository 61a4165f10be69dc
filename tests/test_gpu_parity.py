"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): keypoints, descriptors, pairings and tracked pairs bit-exact; poses within
1e-4 rad / 1e-3 m.  Every test here needs a real MI355X: run with `pytest -m gpu`."""
import ctypes as C
import os
import numpy as np
import pytest

from stereo_vo_amd import hip
from stereo_vo_amd.abi import StereoCamera, north_star_params, keypoint_dtype, dmatch_dtype, index_pair_dtype
from stereo_vo_amd.synth import SyntheticStereoWorld

pytestmark = pytest.mark.gpu

POSE_TOL_M, POSE_TOL_RAD = 1e-3, 1e-4


def O():
    from oracle import oracle
    return oracle


def assert_same_frame(ctx, lane, orc, r, ro, tag):
    for side in (0, 1):
        k, d = ctx.keypoints(lane, 0, side)
        ko, do = orc.keypoints(0, side)
        assert len(k) == len(ko), (tag, side, len(k), len(ko))
        assert k.tobytes() == ko.tobytes(), (tag, "keypoints", side)
        assert (d == do).all(), (tag, "descriptors", side)
    assert ctx.matches(lane).tobytes() == orc.matches(0).tobytes(), (tag, "pairings")
    assert ctx.tracked(lane).tobytes() == orc.tracked().tobytes(), (tag, "tracked pairs")
    assert (r.valid, r.error_code) == (ro.valid, ro.error_code), (tag, r.valid, r.error_code, ro.valid, ro.error_code)
    assert (r.detected_left[0], r.detected_right[0], r.stereo_matches[0]) == (ro.detected_left[0], ro.detected_right[0], ro.stereo_matches[0])
    assert (r.n_residual, r.n_outliers) == (ro.n_residual, ro.n_outliers), tag
    assert list(r.track_stats) == list(ro.track_stats), (tag, "stage-4 pass-through counters", list(r.track_stats), list(ro.track_stats))
    if ro.valid:
        dp = np.abs(np.array(r.outPose) - np.array(ro.outPose))
        assert dp[:3].max() < POSE_TOL_M and dp[3:].max() < POSE_TOL_RAD, (tag, dp)
        assert r.tracked_feats_from_last_frame == ro.tracked_feats_from_last_frame
        assert (ctx.outliers(lane) == orc.outliers()).all(), (tag, "inlier list")
        a, b = ctx.residuals(lane), orc.residuals()
        fin = b < 1e300
        assert ((a < 1e300) == fin).all() and np.allclose(a[fin], b[fin], rtol=1e-6, atol=1e-9), (tag, "residuals")


def load_small(golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_small_seq.npz"))
    cam = StereoCamera.simple(float(g["F"]), float(g["cx"]), float(g["cy"]), float(g["baseline"]), int(g["W"]), int(g["H"]))
    p = north_star_params(hip.default_params(), orb_nfeats=int(g["orb_nfeats"]))
    return g, cam, p


def test_small_sequence_against_golden_and_oracle(golden_dir):
    g, cam, p = load_small(golden_dir)
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(4):
        ctx.process_host([(g["L%d" % t], g["R%d" % t])])
        r, ro = ctx.result(0), orc.process(g["L%d" % t], g["R%d" % t], cam)
        assert_same_frame(ctx, 0, orc, r, ro, "t=%d" % t)
        # and against the committed vectors directly
        for side in (0, 1):
            k, d = ctx.keypoints(0, 0, side)
            assert k.tobytes() == g["kps%d_%d" % (side, t)].tobytes() and (d == g["desc%d_%d" % (side, t)]).all()
        assert ctx.matches(0).tobytes() == g["matches%d" % t].tobytes()
        assert ctx.tracked(0).tobytes() == g["tracked%d" % t].tobytes()
        assert np.allclose(np.array(r.outPose), g["pose%d" % t], atol=1e-6)
        assert ctx.status_word(0) == 0
    ctx.close()


def test_pyramid_and_detector_stage_outputs(golden_dir, monkeypatch):
    # debug mode 9 = the detector's own order (describe every detected keypoint, then the reference's NMS), the only
    # mode in which the pre-NMS list carries angles and descriptors; the product default describes the survivors only
    monkeypatch.setenv("SVO_DEBUG_MODE", "9")
    g, cam, p = load_small(golden_dir)
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    ctx.process_host([(g["L0"], g["R0"])], hip.RUN_DETECT)
    lw, lh, _ = O().pyramid_sizes(int(g["W"]), int(g["H"]), 8)
    prev = g["L0"]
    for l in range(1, 8):
        ref = O().resize(prev, lw[l], lh[l])
        assert (ctx.level(0, 0, l) == ref).all(), "level %d" % l
        prev = ref
    for side, img in ((0, g["L0"]), (1, g["R0"])):
        k, d = ctx.raw_keypoints(0, side)
        ko, do = O().orb_detect(img, int(1.5 * int(g["orb_nfeats"])), 8, 20)
        assert k.tobytes() == ko.tobytes() and (d == do).all()
    # ... and that order gives the same final lists as the default one
    monkeypatch.delenv("SVO_DEBUG_MODE")
    ctx2 = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx2.set_params(p); ctx2.set_camera(cam)
    ctx2.process_host([(g["L0"], g["R0"])], hip.RUN_DETECT)
    ctx.process_host([(g["L0"], g["R0"])], hip.RUN_DETECT | hip.FLAG_REPEAT)
    for side in (0, 1):
        (k1, d1), (k2, d2) = ctx.keypoints(0, 0, side), ctx2.keypoints(0, 0, side)
        assert k1.tobytes() == k2.tobytes() and (d1 == d2).all()
    ctx.close(); ctx2.close()


@pytest.mark.parametrize("mode", ["50", "51", "53", "54"])        # (the A/B-only forms 14 and 52 live in libsvo_hip_ab.so: test_kernel_launch_knobs_do_not_change_results)
def test_every_ransac_kernel_form_gives_the_oracle_models(golden_dir, monkeypatch, mode):
    """k_ransac_hyp (16 lanes per hypothesis, the one-stream form: debug mode 50) and k_ransac_hyp_thread (one thread per
    hypothesis, the many-lane form: 51) both repeat the oracle's eight_point operation for operation: same fundamental
    matrices, hence same inlier counts, masks, tracked pairs and stage-4 counters, whichever form the lane count selects.
    Likewise the inlier counts: sixteen hypotheses per block on the VALU (14), matrix-core tiles of 4 hypotheses x 4 line
    components (52), of 16 hypotheses x 16 pairs with the numerator on the matrix cores too (53: the many-lane default), and
    the latter with every verdict replayed through the oracle's own expression (54)."""
    monkeypatch.setenv("SVO_DEBUG_MODE", mode)
    g, cam, p = load_small(golden_dir)
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(4):
        ctx.process_host([(g["L%d" % t], g["R%d" % t])])
        assert_same_frame(ctx, 0, orc, ctx.result(0), orc.process(g["L%d" % t], g["R%d" % t], cam), "mode %s t=%d" % (mode, t))
    ctx.close()
    w, h = 1280, 960
    world = SyntheticStereoWorld(w, h, 800.0, 0.12, seed=321, n_frames=3)
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=2000)
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=4096)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(3):
        L, R = (x.numpy() for x in world.render(t))
        ctx.process_host([(L, R)])
        assert_same_frame(ctx, 0, orc, ctx.result(0), orc.process(L, R, cam), "mode %s full size t=%d" % (mode, t))
    ctx.close()


def test_noise_images_overflow_the_fast_tile_list():
    """White noise lets most positions through k_fast's cardinal pre-test: far more than the 1024-entry LDS list of a 64x56
    tile holds, so the survivors beyond it are scored and suppressed by the threads that found them (the overflow path of
    fast_tile).  Same corners, keypoints, descriptors, pairings as the oracle; second frame = speculated thresholds."""
    w, h = 640, 480
    rng = np.random.RandomState(77)
    cam = StereoCamera.simple(500.0, w / 2.0, h / 2.0, 0.12, w, h)
    p = north_star_params(hip.default_params(), orb_nfeats=1000)
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=2048, max_cand=1 << 18)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(2):
        L = rng.randint(0, 256, (h, w)).astype(np.uint8)
        R = np.roll(L, -7, axis=1).copy()
        if t == 1:
            L[100:300, 200:500] = 128; R[100:300, 193:493] = 128                    # a flat patch: tiles with and without overflow
        ctx.process_host([(L, R)])
        assert_same_frame(ctx, 0, orc, ctx.result(0), orc.process(L, R, cam), "noise t=%d" % t)
        assert ctx.status_word(0) == 0
    ctx.close()


@pytest.mark.parametrize("w,h,f,cx,cy,B,nfe", [(1280, 960, 800.0, None, None, 0.12, 2000), (1241, 376, 718.856, 607.19, 185.22, 0.537, 900)])
def test_full_size_streams_match_oracle(w, h, f, cx, cy, B, nfe):
    """BASELINE.json configs[1] (1280x960, ~2000 kps) and configs[2] (KITTI-00 shape), 2 lanes x 3 frames."""
    worlds = [SyntheticStereoWorld(w, h, f, B, seed=100 + s, n_frames=3, cx=cx, cy=cy) for s in range(2)]
    cam = worlds[0].camera()
    p = north_star_params(hip.default_params(), orb_nfeats=nfe)
    ctx = hip.Context(n_lanes=2, max_w=w, max_h=h, max_kps=4096)
    ctx.set_params(p); ctx.set_camera(cam)
    orcs = [O().Oracle(p) for _ in range(2)]
    for t in range(3):
        pairs = [tuple(x.numpy() for x in wd.render(t)) for wd in worlds]
        ctx.process_host(pairs)
        for lane in range(2):
            ro = orcs[lane].process(pairs[lane][0], pairs[lane][1], cam)
            assert_same_frame(ctx, lane, orcs[lane], ctx.result(lane), ro, "%dx%d lane=%d t=%d" % (w, h, lane, t))
            assert ctx.status_word(lane) == 0
            if t > 0 and w == 1280:
                assert ro.detected_left[0] > 1500 and ro.tracked_feats_from_last_frame > 100
    ctx.close()


def test_hamming_match_against_oracle_and_properties():
    ctx = hip.Context(n_lanes=1, max_w=64, max_h=64, max_kps=64)
    rng = np.random.RandomState(0)
    for nq, nt in [(1, 1), (3, 255), (255, 256), (257, 257), (1000, 3), (2025, 2025), (5000, 3000)]:
        q = rng.randint(0, 256, (nq, 32)).astype(np.uint8)
        t = rng.randint(0, 256, (nt, 32)).astype(np.uint8)
        if nt > 10:          # planted ties and exact duplicates: the FIRST minimum must win (cv::BFMatcher::match)
            t[nt - 1] = t[2]; q[0] = t[2]
            t[7] = t[5]
            if nq > 1: q[1] = t[5]; q[1, 0] ^= 1
        idx, dist = ctx.hamming_match(q, t)
        io, do = O().hamming_bf(q, t)
        assert (idx == io).all() and (dist == do).all(), (nq, nt)
    idx, dist = ctx.hamming_match(q[:4], np.zeros((0, 32), np.uint8))
    assert (idx == -1).all()
    # full size beyond what the oracle covers quickly: size-independent properties on a sample of queries
    nq = nt = 30000
    q = rng.randint(0, 256, (nq, 32)).astype(np.uint8); t = rng.randint(0, 256, (nt, 32)).astype(np.uint8)
    idx, dist = ctx.hamming_match(q, t)
    pc = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for i in rng.choice(nq, 40, replace=False):
        dd = pc[q[i][None, :] ^ t].sum(1)
        assert dist[i] == dd.min() and idx[i] == int(np.argmin(dd))
    # self-match: every row's nearest neighbour in its own set is itself at distance 0
    idx, dist = ctx.hamming_match(t[:20000], t[:20000])
    assert (dist == 0).all() and (idx == np.arange(20000)).all()
    ctx.close()


def _tracks(cam, delta, n=400, seed=3, noise=0.0, n_out=0):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_units import _synthetic_tracks
    return _synthetic_tracks(cam, delta, n=n, seed=seed, noise=noise, n_out=n_out)


def test_change_in_pose_against_oracle():
    """getChangeInPose boundary (common.cpp:355-413): stage 5 alone on caller arrays."""
    cam = StereoCamera.simple(800.0, 639.5, 479.5, 0.12, 1280, 960)
    p = north_star_params(hip.default_params())
    for delta, noise, n_out, n in [(np.array([0.004, -0.009, 0.002, 0.02, -0.01, -0.25]), 0.0, 0, 400),
                                   (np.array([-0.01, 0.02, 0.005, -0.05, 0.02, 0.4]), 0.3, 40, 1500),
                                   (np.array([0.0, 0.001, 0.0, 0.0, 0.0, -0.1]), 0.1, 0, 6)]:
        t, m, pl, pr, cl, cr = _tracks(cam, delta, n=n, noise=noise, n_out=n_out)
        ctx = hip.Context(n_lanes=1, max_w=1280, max_h=960, max_kps=2048)
        ctx.set_params(p)
        orc = O().Oracle(p)
        for rep in range(2):       # second call exercises the m_last_computed_pose warm start
            v, r, resid, outl = ctx.change_in_pose(t, m, m, pl, pr, cl, cr, cam)
            vo, ro, resid_o, outl_o = orc.change_in_pose(t, m, m, pl, pr, cl, cr, cam)
            assert v == vo and (r.n_residual, r.n_outliers, r.error_code) == (ro.n_residual, ro.n_outliers, ro.error_code)
            if vo:
                assert np.abs(np.array(r.delta) - np.array(ro.delta)).max() < 1e-7
                dp = np.abs(np.array(r.outPose) - np.array(ro.outPose))
                assert dp[:3].max() < POSE_TOL_M and dp[3:].max() < POSE_TOL_RAD
                assert (outl == outl_o).all()
                fin = resid_o < 1e300
                assert np.allclose(resid[fin], resid_o[fin], rtol=1e-6, atol=1e-9)
                assert abs(r.num_it - ro.num_it) <= 1 and abs(r.num_it_final - ro.num_it_final) <= 1
        # custom initial pose has priority and is not stored (S5:504-505, 720)
        q = p.copy(); q.use_custom_initial_pose = 1
        ctx.set_params(q); orc.set_params(q)
        v, r, _, _ = ctx.change_in_pose(t, m, m, pl, pr, cl, cr, cam, init6=delta * 0.9)
        vo, ro, _, _ = orc.change_in_pose(t, m, m, pl, pr, cl, cr, cam, init6=delta * 0.9)
        assert v == vo and (not vo or np.abs(np.array(r.delta) - np.array(ro.delta)).max() < 1e-7)
        ctx.close()


def test_recovery_rule_and_first_frame(golden_dir):
    """a19: voecFirstIteration on the first frame; after voecBadTracking the previous frame is kept (P:86-89)."""
    g, cam, p = load_small(golden_dir)
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    blank = np.full_like(g["L0"], 128)
    seq = [(g["L0"], g["R0"], False), (blank, blank, False), (g["L1"], g["R1"], False), (g["L2"], g["R2"], True), (g["L3"], g["R3"], False)]
    for i, (L, R, rep) in enumerate(seq):
        ctx.process_host([(L, R)], hip.RUN_ALL | (hip.FLAG_REPEAT if rep else 0))
        r, ro = ctx.result(0), orc.process(L, R, cam, repeat=rep)
        assert_same_frame(ctx, 0, orc, r, ro, "step %d" % i)
        kp, _ = ctx.keypoints(0, 1, 0); kpo, _ = orc.keypoints(1, 0)
        assert kp.tobytes() == kpo.tobytes(), "previous-frame keypoints at step %d" % i
    assert [4, 5][0] == 4
    ctx.close()


def test_precomputed_data_bypass(golden_dir):
    """P:131-162 / P:219-251: caller-supplied features and pairings, then stages 4-5 only."""
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    for which, t in ((1, 1), (0, 2)):
        for side in (0, 1):
            ctx.put_features(0, which, side, g["kps%d_%d" % (side, t)], g["desc%d_%d" % (side, t)], W, H)
        ctx.put_matches(0, which, g["matches%d" % t])
    ctx.run_stages(hip.RUN_TRACK | hip.RUN_OPTIMIZE)
    assert ctx.tracked(0).tobytes() == g["tracked2"].tobytes()
    r = ctx.result(0)
    # frame 2 of the golden run started from frame 1's pose; here the warm start is zero: same optimum within tolerance
    dp = np.abs(np.array(r.outPose) - g["pose2"])
    assert r.valid and dp[:3].max() < 5e-3 and dp[3:].max() < 5e-4
    # stage 3 alone on put features reproduces the pairings
    ctx.run_stages(hip.RUN_MATCH)
    assert ctx.matches(0).tobytes() == g["matches2"].tobytes()
    # precomputed_matches_ID (H:218, P:233-244): the IDs come back and m_last_match_ID becomes their maximum
    ids = np.arange(len(g["matches2"]), dtype=np.int32)[::-1] * 3 + 7
    ctx.put_match_ids(0, 0, ids)
    assert list(ctx.match_ids(0, 0)) == list(ids)
    ctx.close()


@pytest.mark.parametrize("ifm_method", [0, 1])
def test_few_candidates_reach_the_sampler_edge_cases(golden_dir, ifm_method):
    """cv::findFundamentalMat where the point count is tiny (VERDICT r04 missing #2): the previous frame's pairings are thinned
    until a handful of candidates reaches it -- fewer than 7 (no model), exactly 7 (the direct path: whole mask set, below the 8 inliers
    S4:205 asks for), 8..14 (oracle v6: the LMedS registrator -- 300 samples, medians, its own mask threshold; many repeated draws per
    attempt: the tabulated attempts of small n) and a few dozen (the RANSAC).  Stage 4 alone on caller-supplied lists (P:131-162):
    tracked pairs AND the eight stage counters against the oracle's stage 4.  Both trackers: brute force (ifmDescBF, S4:88-329) and the
    windowed search over the current pairings' row index (ifmDescWin, S4:435-738; its F-matrix filter is S4:682-705)."""
    g, cam, p = load_small(golden_dir)
    p.ifm_method = ifm_method; p.ifm_win_w = 20; p.ifm_win_h = 30
    W, H = int(g["W"]), int(g["H"])

    def pairings_row_index(m, kl):                                        # matches_lr_row_index (S3:425-445) of a thinned list
        ri, idx = np.zeros(H + 1, np.int64), 0
        for y in range(H):
            ri[y] = idx
            while idx < len(m) and kl[m[idx]["queryIdx"]]["y"] <= np.float32(y):
                idx += 1
        ri[H] = len(m)
        return ri
    assert (pairings_row_index(g["matches2"], g["kps0_2"]) == g["mrow2"]).all()
    rng = np.random.RandomState(23)
    pm_all, cm = g["matches1"], g["matches2"]
    seen, lmeds_inliers = set(), []
    ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    zeros = np.zeros(H + 1, np.int64)
    for keep in (6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 24, 32, 48, 64):
        for rep in range(6):
            sel = np.sort(rng.choice(len(pm_all), min(keep, len(pm_all)), replace=False))
            pm = np.ascontiguousarray(pm_all[sel])
            for side in (0, 1):
                ctx.put_features(0, 1, side, g["kps%d_1" % side], g["desc%d_1" % side], W, H)
                ctx.put_features(0, 0, side, g["kps%d_2" % side], g["desc%d_2" % side], W, H)
            ctx.put_matches(0, 1, pm); ctx.put_matches(0, 0, cm)
            ctx.run_stages(hip.RUN_TRACK)
            want, want_ts = O().track(p, p.orb_max_distance, g["kps0_1"], g["desc0_1"], g["kps1_1"], g["desc1_1"], pm, pairings_row_index(pm, g["kps0_1"]) if ifm_method else zeros,
                                      g["kps0_2"], g["desc0_2"], g["kps1_2"], g["desc1_2"], cm, np.ascontiguousarray(g["mrow2"], np.int64) if ifm_method else zeros, W, H, stats=True)
            got = ctx.tracked(0)
            assert got.tobytes() == want.tobytes(), (keep, rep, len(got), len(want))
            ts = ctx.result(0).track_stats
            # candidates, collision survivors, inliers left / right, samples visited left / right, both masks, tracked: the oracle's, all eight
            assert [int(v) for v in ts[:8]] == [int(v) for v in want_ts], (keep, rep, list(ts[:8]), list(want_ts))
            seen.add(min(int(ts[1]), 40))                                # candidates that reached the RANSAC (collision survivors)
            if 8 <= ts[1]:
                assert ts[4] > 0 and ts[5] > 0                           # samples were drawn and visited on both sides
            if 8 <= ts[1] <= 14:
                assert (ts[4], ts[5]) == (300, 300)                      # LMedS (oracle v6): its fixed budget, never shortened
                lmeds_inliers.append((int(ts[1]), int(ts[2]), int(ts[3])))
            if ts[1] == 7:
                assert (ts[2], ts[3], ts[4], ts[5]) == (7, 7, 0, 0)      # direct path: seven "inliers", no sample visited
    ctx.close()
    assert 7 in seen and any(8 <= n <= 14 for n in seen) and any(n < 7 for n in seen) and any(n >= 15 for n in seen), sorted(seen)
    # below 14 points LMedS's median is one of the seven fitted residuals: the mask is the sample (seven, under S4:205's eight); at 14 it can pass
    assert all(7 <= l <= 9 and 7 <= r <= 9 for n, l, r in lmeds_inliers if n <= 13) and len(lmeds_inliers) >= 6, lmeds_inliers


def line_scene(n, ys, d, outlier_frac, seed, W=640, H=480, disp=20, off_line=True):
    """Stage-4 input lists built by hand: n pairings whose keypoints sit on the rows `ys` at INTEGER x; the current frame is the previous
    one shifted by d pixels in x (pure translation: every correspondence satisfies F = [t]_x exactly); a fraction of the pairings is
    moved somewhere else in the current frame (outliers).  Every keypoint has a random descriptor of its own, so the brute-force tracker
    pairs pairing k of the previous frame with its own continuation."""
    rng = np.random.RandomState(seed)
    per = max(1, n // len(ys))
    pts = []
    for y in ys:
        xs = rng.choice(np.arange(60, W - 60), per, replace=False)
        pts += [(float(x), float(y)) for x in np.sort(xs)]
    pts = np.array(pts, np.float32)
    n = len(pts)

    def kps(xy):
        k = np.zeros(len(xy), keypoint_dtype)
        k["x"], k["y"], k["size"], k["response"], k["class_id"] = xy[:, 0], xy[:, 1], 31.0, 1.0, -1
        return k
    desc, descr = rng.randint(0, 256, (n, 32)).astype(np.uint8), rng.randint(0, 256, (n, 32)).astype(np.uint8)
    m = np.zeros(n, dmatch_dtype); m["queryIdx"] = np.arange(n); m["trainIdx"] = np.arange(n); m["distance"] = 10.0
    cur = pts + np.array([d, 0], np.float32)
    out = rng.rand(n) < outlier_frac
    if off_line:
        cur[out, 1] = rng.randint(40, H - 40, int(out.sum())).astype(np.float32); cur[out, 0] = rng.randint(60, W - 60, int(out.sum())).astype(np.float32)
    else:
        yi = np.searchsorted(np.array(ys, np.float32), pts[:, 1])
        cur[out, 1] = np.array(ys, np.float32)[(yi[out] + 1) % len(ys)]
    order = np.lexsort((cur[:, 0], cur[:, 1]))                             # the current lists are row-sorted too
    curL = cur[order]
    sh = np.array([disp, 0], np.float32)
    return dict(pkl=kps(pts), pkr=kps(pts - sh), pdl=desc, pdr=descr, pm=m, ckl=kps(curL), ckr=kps(curL - sh), cdl=desc[order], cdr=descr[order], cm=m.copy(), W=W, H=H)


def test_collinear_point_sets_continue_the_sampler_past_its_table(golden_dir):
    """ADVICE r05 (medium) + oracle v7.  Keypoints along a few image rows at integer coordinates: most of cv::findFundamentalMat's samples
    are rejected by checkSubset (the last point is collinear with two earlier ones), so a lane needs far more ATTEMPTS than the 1152 a
    row of the host-built table holds -- the device now continues cv::RNG from the state the row ended in (rs_schedule_block) instead of
    ending the schedule early.  Cases: two / three rows with half the pairings thrown off the rows (hundreds of samples visited, most
    all-inlier attempts rejected), every point on ONE row (no sample in getSubset's 10000 attempts: the run ends as OpenCV's does, no
    model), few points on two rows (LMedS: getSubset's default 1000 attempts, oracle v7), and exact pure translations on an integer grid
    (the 7-point cubic loses its leading coefficient: cv::solveCubic's quadratic branch, oracle v7).  Stage 4 alone on caller-supplied
    lists: tracked pairs and all eight stage counters equal the oracle's, and SVO_ST_INTERNAL stays down."""
    _, cam, p = load_small(golden_dir)
    visited_max, n_cases, ended_without_model = 0, 0, 0
    ctx = hip.Context(n_lanes=1, max_w=640, max_h=480, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    zeros = np.zeros(481, np.int64)
    cases = [([100, 300], 120, 0.5, True), ([100, 200, 300], 120, 0.5, True), ([100, 300], 80, 0.7, True), ([100, 300], 200, 0.3, True), ([100], 100, 0.5, True),
             ([100, 300], 120, 0.5, False), ([100, 300], 14, 0.0, True), ([100, 300], 12, 0.2, True), ([100, 300], 10, 0.0, False), ([60, 100, 300, 420], 200, 0.6, True),
             ([90, 91, 92, 93, 94, 95, 96, 97, 98, 99, 100, 101], 96, 0.2, True)]
    for ys, n, of, off in cases:
        for seed in range(3):
            s = line_scene(n, ys, 4.0, of, seed, off_line=off)
            for side, (k1, d1, k0, d0) in enumerate(((s["pkl"], s["pdl"], s["ckl"], s["cdl"]), (s["pkr"], s["pdr"], s["ckr"], s["cdr"]))):
                ctx.put_features(0, 1, side, k1, d1, s["W"], s["H"])
                ctx.put_features(0, 0, side, k0, d0, s["W"], s["H"])
            ctx.put_matches(0, 1, s["pm"]); ctx.put_matches(0, 0, s["cm"])
            ctx.run_stages(hip.RUN_TRACK)
            want, want_ts = O().track(p, p.orb_max_distance, s["pkl"], s["pdl"], s["pkr"], s["pdr"], s["pm"], zeros, s["ckl"], s["cdl"], s["ckr"], s["cdr"], s["cm"], zeros, s["W"], s["H"], stats=True)
            got, ts = ctx.tracked(0), ctx.result(0).track_stats
            assert [int(v) for v in ts[:8]] == [int(v) for v in want_ts], (ys, n, of, off, seed, list(ts[:8]), list(want_ts))
            assert got.tobytes() == want.tobytes(), (ys, n, of, off, seed, len(got), len(want))
            assert not (ctx.status_word(0) & 8), (ys, n, of, off, seed, "SVO_ST_INTERNAL: the sampler ran out of attempts")
            visited_max = max(visited_max, int(ts[4]), int(ts[5])); n_cases += 1
            ended_without_model += int(ts[1] >= 15 and ts[4] == 0 and ts[5] == 0)
    ctx.close()
    # two rows, half the points off them: ~11 % of the all-on-the-rows attempts pass, yet several hundred samples were visited -- more than a
    # table row of 1152 attempts can yield at that pass rate is not asserted (the pass rate is the data's), the equality above is the check
    assert n_cases == 33 and visited_max >= 500 and ended_without_model >= 3, (n_cases, visited_max, ended_without_model)


def test_armed_post_event_is_disarmed_by_a_failing_call(golden_dir):
    """svo_record_after_post arms ONE call.  When that call leaves early -- here with SVO_ERR_STATE: a post-processing call on a
    context whose geometry was never set up -- the event is still recorded (a waiter is released) and disarmed, so that a later, unrelated
    call cannot record it at the wrong point: re-recorded by the test itself behind a long kernel queue, it must stay where the test put it."""
    import torch
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    ev = torch.cuda.Event()
    L = hip.lib()
    assert L.svo_record_after_post(ctx.h, C.c_void_p(ev.cuda_event)) == 0
    rc = L.svo_process(ctx.h, None, C.c_uint32(hip.RUN_DETECT_POST | hip.RUN_MATCH | hip.FLAG_NO_SHIFT))      # no frame has set the geometry up yet: SVO_ERR_STATE
    assert rc < 0
    ev.synchronize()                                     # recorded by the failing call: returns at once instead of hanging
    # a full frame now (it runs the post-processing): the event is no longer armed, so it is NOT re-recorded behind this frame
    ctx.process_host([(g["L0"], g["R0"])])
    assert ev.query()                                    # still the old record: complete although the frame may be in flight
    ctx.wait()
    # and the armed event of a SUCCESSFUL call is recorded by that call
    ev2 = torch.cuda.Event()
    assert L.svo_record_after_post(ctx.h, C.c_void_p(ev2.cuda_event)) == 0
    ctx.process_host([(g["L1"], g["R1"])])
    ev2.synchronize()
    ctx.wait()
    orc = O().Oracle(p)
    for t in (0, 1):
        ro = orc.process(g["L%d" % t], g["R%d" % t], cam)
    assert_same_frame(ctx, 0, orc, ctx.result(0), ro, "after the armed calls")
    ctx.close()


def test_device_resident_images_and_lane_independence(golden_dir):
    import torch
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    ctx = hip.Context(n_lanes=3, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    keep = []
    for t in range(3):
        Ld = torch.from_numpy(g["L%d" % t]).cuda().contiguous(); Rd = torch.from_numpy(g["R%d" % t]).cuda().contiguous()
        blank = torch.full_like(Ld, 100)
        keep += [Ld, Rd, blank]
        torch.cuda.synchronize()
        # lanes 0 and 2 see the same stream, lane 1 sees a blank stream: lanes must not influence each other
        ctx.process_device([(Ld.data_ptr(), Rd.data_ptr()), (blank.data_ptr(), blank.data_ptr()), (Ld.data_ptr(), Rd.data_ptr())], W, H, W)
        ro = orc.process(g["L%d" % t], g["R%d" % t], cam)
        for lane in (0, 2):
            assert_same_frame(ctx, lane, orc, ctx.result(lane), ro, "lane %d t=%d" % (lane, t))
        rb = ctx.result(1)
        assert rb.detected_left[0] == 0 and rb.error_code == (4 if t == 0 else 5)
    ctx.close()


def test_edge_inputs():
    p = north_star_params(hip.default_params(), orb_nfeats=100)
    cam = StereoCamera.simple(100.0, 31.5, 31.5, 0.1, 64, 64)
    ctx = hip.Context(n_lanes=1, max_w=128, max_h=128, max_kps=256, max_cand=4096)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    rng = np.random.RandomState(9)
    # 64x64: every level is narrower than two border widths + 1 -> at most a handful of candidates
    for shape in ((64, 64), (65, 97), (128, 128)):
        L = rng.randint(0, 256, shape).astype(np.uint8); R = np.roll(L, -3, axis=1)
        ctx.reset()
        orc2 = O().Oracle(p)
        for rep in range(2):
            ctx.process_host([(L, R)])
            r, ro = ctx.result(0), orc2.process(L, R, cam)
            assert_same_frame(ctx, 0, orc2, r, ro, str(shape))
    ctx.close()


def test_cpp_estimator_demo_matches_oracle_chain(tmp_path):
    """The C++ host mirror (rso::CStereoOdometryEstimator over the C-ABI) driven like the reference's demo
    (demo-main.cpp:210-253) on BASELINE.json configs[0] (20-frame 640x480), against the oracle chained in Python."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from make_sequence import write_sequence
    from stereo_vo_amd.synth import pose6_to_matrix
    seq, out = str(tmp_path / "seq.svoseq"), str(tmp_path / "camera_pose.txt")
    world = write_sequence(seq, 640, 480, 400.0, 0.12, seed=11, n_frames=20)
    exe = os.path.join(root, "tools", "demo_stereo_odometry")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    subprocess.check_call([exe, seq, out, "500"])
    got = np.loadtxt(out)
    # the same run configured as the reference's demo is: from its application INI (demo-main.cpp:154-167, --opt)
    ini, out2 = tmp_path / "demo-stereo-odometry-config.ini", str(tmp_path / "camera_pose_ini.txt")
    ini.write_text("[DETECT]\ndetect_method = 0\norb_nfeats = 500   // ORB\n[MATCH]\nmatch_method = 0\nmax_y_diff = 1.0\nenable_robust_1to1_match = true\n"
                   "orb_max_distance = 60\n[IF-MATCH]\nif_match_method = 0\n[GENERAL]\nvo_use_matches_ids = false\n")
    subprocess.check_call([exe, seq, out2, "--opt", str(ini)])
    assert open(out2).read() == open(out).read()
    orc = O().Oracle(north_star_params(hip.default_params(), orb_nfeats=500))
    pose = np.eye(4); rows = []
    from stereo_vo_amd.synth import _rot_zyx
    for t in range(20):
        L, R = [x.numpy() for x in world.render(t)]
        r = orc.process(L, R, world.camera())
        if r.valid:
            pose = pose @ pose6_to_matrix(np.array(r.outPose))
        pitch = np.arctan2(-pose[2, 0], np.hypot(pose[0, 0], pose[1, 0]))
        rows.append([pose[0, 3], pose[1, 3], pose[2, 3], np.arctan2(pose[1, 0], pose[0, 0]), pitch, np.arctan2(pose[2, 1], pose[2, 2])])
    want = np.array(rows)
    assert got.shape == want.shape == (20, 6)
    assert np.abs(got - want).max() < 2e-3      # the file keeps 3 decimals (D:251)
    assert np.abs(got[-1, :3] - world.poses[19][:3, 3]).max() < 0.35   # and the chain follows the generator's ground truth


def test_row_index_tables_match_oracle(golden_dir):
    """a5 / a8: pyr_feats_index (stage2_detect.cpp:103-129) and matches_lr_row_index (stage3:425-445) on device."""
    g, cam, p = load_small(golden_dir)
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(2):
        ctx.process_host([(g["L%d" % t], g["R%d" % t])])
        orc.process(g["L%d" % t], g["R%d" % t], cam)
        for side in (0, 1):
            assert (ctx.row_index(0, 0, side) == orc.row_index(0, side)).all()
        assert (ctx.matches_row_index(0, 0) == orc.matches_row_index(0)).all()
    ctx.close()


@pytest.mark.parametrize("w,h,nfe,noct", [(640, 480, 600, 3), (1280, 960, 3300, 3)])
def test_fast_orb_multi_octave_matches_oracle(w, h, nfe, noct):
    """a2 (stage2_detect.cpp:502-515) + nOctaves x1/2 pyramid: FAST + ORB describe per octave, per-octave pairing and
    tracking, stage 5 on the merged octaves (S5:419-461) -- BASELINE.json configs[4]'s operator mix."""
    from stereo_vo_amd.abi import DM_FAST_ORB
    world = SyntheticStereoWorld(w, h, 400.0 * w / 640.0, 0.12, seed=21, n_frames=3)
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=nfe)
    p.detect_method = DM_FAST_ORB; p.nOctaves = noct
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=4096, max_octaves=noct)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(3):
        L, R = [x.numpy() for x in world.render(t)]
        ctx.process_host([(L, R)])
        r, ro = ctx.result(0), orc.process(L, R, cam)
        assert ctx.status_word(0) == 0
        assert r.n_octaves == ro.n_octaves == noct
        for o in range(noct):
            for side in (0, 1):
                k, d = ctx.keypoints(0, 0, side, o); ko, do = orc.keypoints(0, side, o)
                assert len(k) == len(ko) and k.tobytes() == ko.tobytes() and (d == do).all(), (t, o, side, len(k), len(ko))
                assert (ctx.row_index(0, 0, side, o) == orc.row_index(0, side, o)).all()
            assert ctx.matches(0, 0, o).tobytes() == orc.matches(0, o).tobytes(), (t, o)
            assert ctx.tracked(0, o).tobytes() == orc.tracked(o).tobytes(), (t, o)
            assert (r.detected_left[o], r.detected_right[o], r.stereo_matches[o]) == (ro.detected_left[o], ro.detected_right[o], ro.stereo_matches[o])
        assert (r.valid, r.error_code, r.n_residual, r.n_outliers) == (ro.valid, ro.error_code, ro.n_residual, ro.n_outliers)
        if ro.valid:
            dp = np.abs(np.array(r.outPose) - np.array(ro.outPose))
            assert dp[:3].max() < POSE_TOL_M and dp[3:].max() < POSE_TOL_RAD
            assert (ctx.outliers(0) == orc.outliers()).all()
    assert ro.valid and ro.tracked_feats_from_last_frame > 20
    ctx.close()


@pytest.mark.parametrize("match_method,ifm_method,one2one,ydiff", [(1, 0, 1, 1.0), (0, 1, 1, 1.0), (1, 1, 0, 2.0), (1, 1, 1, 0.0)])
def test_row_by_row_and_window_variants_match_oracle(match_method, ifm_method, one2one, ydiff):
    """a7 smDescRbR (stage3:185-419) and a10 ifmDescWin (stage4:435-738), with the reference's quirks, vs the oracle."""
    w, h = 640, 480
    world = SyntheticStereoWorld(w, h, 400.0, 0.12, seed=31, n_frames=3)
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=700)
    p.match_method = match_method; p.ifm_method = ifm_method; p.enable_robust_1to1_match = one2one; p.max_y_diff = ydiff
    p.ifm_win_w = 20; p.ifm_win_h = 30; p.minimum_ORB_response = 1e-5
    ctx = hip.Context(n_lanes=2, max_w=w, max_h=h, max_kps=2048)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(3):
        L, R = [x.numpy() for x in world.render(t)]
        ctx.process_host([(L, R), (L, R)])
        ro = orc.process(L, R, cam)
        for lane in (0, 1):
            assert_same_frame(ctx, lane, orc, ctx.result(lane), ro, "mm=%d ifm=%d t=%d lane=%d" % (match_method, ifm_method, t, lane))
            assert (ctx.matches_row_index(lane, 0) == orc.matches_row_index(0)).all()
    if ydiff > 0:
        assert ro.stereo_matches[0] > 50 and ro.tracked_feats_from_last_frame > 10
    else:
        assert ro.stereo_matches[0] == 0        # max_y_diff = 0 -> empty right window (appendix A #10)
    ctx.close()


def test_match_ids_and_keyframe_counter(golden_dir):
    """a11: vo_use_matches_ids bookkeeping (S3:172-173, S4:268-305, 743-751, P:254-267) against the oracle."""
    g, cam, p = load_small(golden_dir)
    p = p.copy(); p.vo_use_matches_ids = 1
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(4):
        if t == 3:
            ctx.reset_ids(0); orc.L.svo_oracle_reset_ids(orc.h)
        ctx.process_host([(g["L%d" % t], g["R%d" % t])])
        r, ro = ctx.result(0), orc.process(g["L%d" % t], g["R%d" % t], cam)
        assert (ctx.match_ids(0, 0) == orc.match_ids(0)).all() and len(ctx.match_ids(0, 0)) == ro.stereo_matches[0], t
        if t > 0:
            assert (ctx.match_ids(0, 1) == orc.match_ids(1)).all(), t
        assert r.tracked_feats_from_last_KF == ro.tracked_feats_from_last_KF, (t, r.tracked_feats_from_last_KF, ro.tracked_feats_from_last_KF)
        if t == 1:
            ctx.set_this_frame_as_kf(0); orc.L.svo_oracle_set_this_frame_as_kf(orc.h)
    assert ro.tracked_feats_from_last_KF > 0
    ctx.close()


def test_config5_high_res_three_octaves():
    """BASELINE.json configs[4]: 2048x1536, FAST+ORB on a 3-octave x1/2 pyramid, ~5000 keypoints per image in total
    (orb_nfeats = 3300 -> 2828 + 1414 + 707), pseudo-Huber kernel (kernel_param 3): parity with the oracle."""
    from stereo_vo_amd.abi import DM_FAST_ORB
    w, h = 2048, 1536
    world = SyntheticStereoWorld(w, h, 1280.0, 0.12, seed=41, n_frames=2)
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=3300)
    p.detect_method = DM_FAST_ORB; p.nOctaves = 3; p.use_robust_kernel = 1; p.kernel_param = 3.0
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=4096, max_octaves=3, max_cand=1 << 18)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(2):
        L, R = [x.numpy() for x in world.render(t)]
        ctx.process_host([(L, R)])
        r, ro = ctx.result(0), orc.process(L, R, cam)
        assert ctx.status_word(0) == 0
        tot = 0
        for o in range(3):
            for side in (0, 1):
                k, d = ctx.keypoints(0, 0, side, o); ko, do = orc.keypoints(0, side, o)
                assert k.tobytes() == ko.tobytes() and (d == do).all(), (t, o, side, len(k), len(ko))
            assert ctx.matches(0, 0, o).tobytes() == orc.matches(0, o).tobytes()
            assert ctx.tracked(0, o).tobytes() == orc.tracked(o).tobytes()
            tot += r.detected_left[o]
        assert tot > 4000
        assert (r.valid, r.error_code) == (ro.valid, ro.error_code)
        if ro.valid:
            dp = np.abs(np.array(r.outPose) - np.array(ro.outPose))
            assert dp[:3].max() < POSE_TOL_M and dp[3:].max() < POSE_TOL_RAD
    assert ro.valid
    ctx.close()


def _distortion_maps(w, h, k1=0.08, shift=(2.3, -1.7), rot=0.01):
    """A smooth radial + rigid rectification-like map (float32 source coordinates per output pixel), some of it
    pointing outside the image."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    xn, yn = (xx - w / 2) / (w / 2), (yy - h / 2) / (w / 2)
    r2 = xn * xn + yn * yn
    xd, yd = xn * (1 + k1 * r2), yn * (1 + k1 * r2)
    c, s = np.cos(rot), np.sin(rot)
    mx = (c * xd - s * yd) * (w / 2) + w / 2 + shift[0]
    my = (s * xd + c * yd) * (w / 2) + h / 2 + shift[1]
    return mx.astype(np.float32), my.astype(np.float32)


@pytest.mark.parametrize("w,h,bgr,device", [(320, 240, False, False), (317, 243, True, False), (640, 480, True, True), (1241, 376, False, True)])
def test_stage1_grey_and_rectify_on_device(w, h, bgr, device):
    """Stage 1 on the device (svo_set_rectify_map, SVO_FLAG_BGR_IMAGES) against the oracle's svo_oracle_prepare: the
    prepared level-0 images bit-exact, and the whole frame computed from them identical to the oracle's on the
    oracle-prepared images."""
    import torch
    rng = np.random.default_rng(7)
    W0, H0 = (w + 7) // 8 * 8, (h + 7) // 8 * 8                 # the renderer wants friendly sizes; odd shapes are crops
    world = SyntheticStereoWorld(W0, H0, 300.0 * w / 640.0, 0.12, seed=3, n_frames=2, device=torch.device("cpu"))
    cam = StereoCamera.simple(300.0 * w / 640.0, w / 2.0, h / 2.0, 0.12, w, h)
    p = north_star_params(hip.default_params(), orb_nfeats=300)
    ctx = hip.Context(n_lanes=2, max_w=w, max_h=h, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    maps = [_distortion_maps(w, h), _distortion_maps(w, h, k1=-0.05, shift=(-3.25, 0.5), rot=-0.004)]
    ctx.set_rectify_map(0, 0, *maps[0]); ctx.set_rectify_map(0, 1, *maps[1])       # lane 0 rectifies, lane 1 does not
    orcs = [O().Oracle(p), O().Oracle(p)]
    for t in range(2):
        L, R = [x.numpy()[:h, :w] for x in world.render(t)]
        if bgr:       # a colour image whose grey value is not just one of its channels
            L = np.stack([L, np.roll(L, 3, 1), 255 - L // 2], -1); R = np.stack([R, np.roll(R, 2, 0), 255 - R // 2], -1)
        L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
        if device:
            tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
            ctx.process_device([(tl.data_ptr(), tr.data_ptr())] * 2, w, h, L.strides[0], hip.RUN_ALL | (hip.FLAG_BGR_IMAGES if bgr else 0))
        else:
            ctx.process_host([(L, R)] * 2)
        want = [[O().prepare(L, *maps[0]), O().prepare(R, *maps[1])], [O().prepare(L), O().prepare(R)]]
        for lane in range(2):
            for side in range(2):
                assert (ctx.level(lane, side, 0) == want[lane][side]).all(), (t, lane, side)
            r, ro = ctx.result(lane), orcs[lane].process(want[lane][0], want[lane][1], cam)
            assert_same_frame(ctx, lane, orcs[lane], r, ro, "t=%d lane=%d" % (t, lane))
    # clearing the map turns rectification off again
    ctx.set_rectify_map(0, 0, None, None); ctx.set_rectify_map(0, 1, None, None)
    L, R = [np.ascontiguousarray(x.numpy()[:h, :w]) for x in world.render(0)]
    ctx.process_host([(L, R)] * 2)
    assert (ctx.level(0, 0, 0) == L).all() and (ctx.level(1, 1, 0) == R).all()


@pytest.mark.parametrize("w,h,nfe", [(640, 480, 500), (1280, 960, 2000)])
def test_adaptive_nms_matches_oracle(w, h, nfe):
    """nmsMethod = nmsmAdaptive (stage2_detect.cpp:599-606, 141-215) on the device against the oracle, whole frames."""
    import torch
    world = SyntheticStereoWorld(w, h, 800.0 * w / 1280.0, 0.12, seed=11, n_frames=3, device=torch.device("cpu"))
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=nfe)
    p.nmsMethod = 1
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=4096)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(3):
        L, R = [x.numpy() for x in world.render(t)]
        ctx.process_host([(L, R)])
        r, ro = ctx.result(0), orc.process(L, R, cam)
        assert_same_frame(ctx, 0, orc, r, ro, "anms t=%d" % t)
    assert r.detected_left[0] > nfe // 2


def test_state_save_load_resumes_a_stream(golden_dir, tmp_path):
    """saveStateToFile / loadStateFromFile (common.cpp:475-543, 261-350): the file written through the C-ABI reads back
    through the independent Python reader with the lists the getters return, and a fresh context that loads it continues
    the stream exactly like the one that kept running (and like the oracle)."""
    from stereo_vo_amd.state_file import read_state
    g, cam, p = load_small(golden_dir)
    p.vo_use_matches_ids = 1
    W, H = int(g["W"]), int(g["H"])
    a = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    a.set_params(p); a.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(3):
        a.process_host([(g["L%d" % t], g["R%d" % t])]); a.result(0)
        orc.process(g["L%d" % t], g["R%d" % t], cam)
    path = str(tmp_path / "vo_state.bin")
    a.save_state(0, path)
    s = read_state(path)
    for which, name in ((1, "pre"), (0, "cur")):
        for side, sn in ((0, "left"), (1, "right")):
            k, d = a.keypoints(0, which, side)
            assert s[name][sn][0].tobytes() == k.tobytes() and (s[name][sn][1] == d).all()
        assert s[name]["matches"].tobytes() == a.matches(0, which).tobytes()
        assert list(s[name]["ids"]) == list(a.match_ids(0, which))
    assert s["npyr"] == 1 and s["num_tracked_last_frame"] == a.result(0).tracked_feats_from_last_frame
    b = hip.Context(n_lanes=2, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)       # a different context shape on purpose
    b.set_params(p); b.set_camera(cam)
    b.process_host([(g["L0"], g["R0"])] * 2); b.wait()                                 # lane 1 gets unrelated history first
    b.load_state(1, path)
    a.process_host([(g["L3"], g["R3"])]); b.process_host([(g["L3"], g["R3"])] * 2)
    ra, rb, ro = a.result(0), b.result(1), orc.process(g["L3"], g["R3"], cam)
    assert_same_frame(a, 0, orc, ra, ro, "kept running")
    # the file does not carry m_last_computed_pose (neither does the reference's), so the resumed lane starts its
    # Gauss-Newton from the identity: lists identical, pose equal to within the tolerance rather than bit for bit
    for side in (0, 1):
        assert b.keypoints(1, 0, side)[0].tobytes() == a.keypoints(0, 0, side)[0].tobytes()
    assert b.matches(1).tobytes() == a.matches(0).tobytes() and b.tracked(1).tobytes() == a.tracked(0).tobytes()
    assert (rb.valid, rb.error_code, rb.tracked_feats_from_last_frame) == (ra.valid, ra.error_code, ra.tracked_feats_from_last_frame)
    dp = np.abs(np.array(rb.outPose) - np.array(ra.outPose))
    assert dp[:3].max() < POSE_TOL_M and dp[3:].max() < POSE_TOL_RAD, dp
    assert list(b.match_ids(1, 0)) == list(a.match_ids(0, 0)) == list(orc.match_ids(0))
    # malformed input is refused, the lane is left as it was
    bad = str(tmp_path / "bad.bin"); open(bad, "wb").write(open(path, "rb").read()[:100])
    with pytest.raises(Exception):
        b.load_state(0, bad)


def test_projected_coords_match_oracle(golden_dir):
    """getProjectedCoords (H:175-182, common.cpp:415-466) through the C-ABI against the oracle, bit for bit."""
    g, cam, p = load_small(golden_dir)
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    for t in range(2):
        ctx.process_host([(g["L%d" % t], g["R%d" % t])])
    r = ctx.result(0)
    kl, _ = ctx.keypoints(0, 1, 0); kr, _ = ctx.keypoints(0, 1, 1); m = ctx.matches(0, 1)
    tracked = np.full(len(m), -1, np.int32); tracked[ctx.tracked(0)["first"]] = 1
    for pose in (list(r.outPose), [0.3, -0.1, 0.5, 0.2, -0.05, 0.1], [0, 0, 0, 0, 0, 0]):
        a, b = ctx.projected_coords(m, kl, kr, tracked, cam, pose), O().projected_coords(m, kl, kr, tracked, cam, pose)
        assert a.shape == b.shape == (int((tracked == -1).sum()), 4) and a.tobytes() == b.tobytes()
    assert ctx.projected_coords(m[:0], kl, kr, tracked[:0], cam, [0] * 6).shape == (0, 4)


def test_split_detect_post_stage_equals_one_call(golden_dir):
    """SVO_FLAG_DETECT_NO_POST + SVO_RUN_DETECT_POST (the NMS / row-sort block scheduled with stages 3-5, as bench.py does
    on its overlap stream) gives exactly what one SVO_RUN_ALL call gives."""
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    a = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    b = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    for c in (a, b):
        c.set_params(p); c.set_camera(cam)
    for t in range(4):
        a.process_host([(g["L%d" % t], g["R%d" % t])])
        b.process_host([(g["L%d" % t], g["R%d" % t])], hip.RUN_DETECT | hip.FLAG_DETECT_NO_POST)
        b.run_stages(hip.RUN_DETECT_POST | hip.RUN_MATCH | hip.RUN_TRACK | hip.RUN_OPTIMIZE)
        ra, rb = a.result(0), b.result(0)
        for side in (0, 1):
            assert a.keypoints(0, 0, side)[0].tobytes() == b.keypoints(0, 0, side)[0].tobytes()
        assert a.matches(0).tobytes() == b.matches(0).tobytes() and a.tracked(0).tobytes() == b.tracked(0).tobytes()
        assert (ra.valid, ra.error_code) == (rb.valid, rb.error_code) and list(ra.outPose) == list(rb.outPose)


def test_scheduling_hooks_streams_and_selective_timing(golden_dir):
    """svo_set_stream + the split detect stage on two HIP streams ordered by events (what bench.py's pipelined schedule
    does) reproduces the single-stream results; svo_kernel_times_select restricts the event spans to one kernel."""
    import torch
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    a = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    b = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15, kernel_times=True, stream=s1.cuda_stream)
    for c in (a, b):
        c.set_params(p); c.set_camera(cam)
    imgs = [(torch.from_numpy(g["L%d" % t]).cuda(), torch.from_numpy(g["R%d" % t]).cuda()) for t in range(4)]
    torch.cuda.synchronize()
    det, rest = torch.cuda.Event(), torch.cuda.Event()
    for t in range(4):
        a.process_host([(g["L%d" % t], g["R%d" % t])])
        if t:
            s1.wait_event(rest)
        b.set_stream(s1.cuda_stream)
        b.process_device([(imgs[t][0].data_ptr(), imgs[t][1].data_ptr())], W, H, W, hip.RUN_DETECT)
        det.record(s1); s2.wait_event(det)
        b.set_stream(s2.cuda_stream)
        b.run_stages(hip.RUN_MATCH | hip.RUN_TRACK | hip.RUN_OPTIMIZE)
        rest.record(s2)
        if t == 1:
            s2.synchronize(); b.wait(); b.kernel_times_select("fast"); b.kernel_times_reset()
    ra, rb = a.result(0), b.result(0)
    assert a.keypoints(0, 0, 0)[0].tobytes() == b.keypoints(0, 0, 0)[0].tobytes() and a.tracked(0).tobytes() == b.tracked(0).tobytes()
    assert (ra.valid, list(ra.outPose)) == (rb.valid, list(rb.outPose))
    kt = {k: v for k, v in b.kernel_times().items() if v[1] > 0}
    assert list(kt) == ["fast"] and kt["fast"][1] == 2
    b.kernel_times_select(None)


@pytest.mark.parametrize("B", [64, hip.MAX_LANES])
def test_sixty_four_lanes_each_match_their_own_oracle(B):
    """The batched configuration bench.py runs (64 estimators per context) and the largest context the ABI takes (SVO_MAX_LANES): every
    kernel's lane indexing, XCD-aware tile orders and the MFMA matcher's per-lane tiles, checked on four lanes against independent oracles."""
    import torch
    W, H = 640, 480
    worlds = [SyntheticStereoWorld(W, H, 400.0, 0.12, seed=100 + s, n_frames=3, device=torch.device("cuda"), scene_seed=s % 4) for s in range(B)]
    cam = worlds[0].camera()
    p = north_star_params(hip.default_params(), orb_nfeats=500)
    ctx = hip.Context(n_lanes=B, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    probe = (0, 17, B - 24, B - 1)
    orcs = {l: O().Oracle(p) for l in probe}
    for t in range(3):
        fr = [w.render(t) for w in worlds]
        torch.cuda.synchronize()
        ctx.process_device([(a.data_ptr(), b.data_ptr()) for a, b in fr], W, H, W)
        res = ctx.results()
        for l in probe:
            ro = orcs[l].process(fr[l][0].cpu().numpy(), fr[l][1].cpu().numpy(), cam)
            assert_same_frame(ctx, l, orcs[l], res[l], ro, "t=%d lane=%d" % (t, l))
    assert sum(1 for r in res if r.valid) >= B - 2


def test_change_in_pose_leaves_the_live_lanes_alone(golden_dir):
    """getChangeInPose works on temporaries (common.cpp:362-400): calling it between two frames must not disturb the
    estimator's previous / current frame; only m_last_computed_pose (the warm start, S5:506-507, 720-721) is shared --
    exactly as in the oracle, which shares it the same way."""
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    ctx = hip.Context(n_lanes=2, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orcs = [O().Oracle(p), O().Oracle(p)]
    cam2 = StereoCamera.simple(800.0, 639.5, 479.5, 0.12, 1280, 960)
    t_, m_, pl, pr, cl, cr = _tracks(cam2, np.array([0.004, -0.009, 0.002, 0.02, -0.01, -0.25]), n=300, noise=0.2, n_out=10)
    for t in range(4):
        pair = (g["L%d" % t], g["R%d" % t]); mirrored = (g["R%d" % t][:, ::-1].copy(), g["L%d" % t][:, ::-1].copy())
        ctx.process_host([pair, mirrored])
        for lane, fr in ((0, pair), (1, mirrored)):
            ro = orcs[lane].process(fr[0], fr[1], cam)
            assert_same_frame(ctx, lane, orcs[lane], ctx.result(lane), ro, "t=%d lane=%d" % (t, lane))
        if t in (1, 2):
            before = [(ctx.keypoints(l, w_, s)[0].tobytes(), ctx.matches(l, w_).tobytes()) for l in (0, 1) for w_ in (0, 1) for s in (0, 1)]
            res_before = [bytes(ctx.result(l)) for l in (0, 1)]
            v, r, resid, outl = ctx.change_in_pose(t_, m_, m_, pl, pr, cl, cr, cam2)
            vo, ro, resid_o, outl_o = orcs[0].change_in_pose(t_, m_, m_, pl, pr, cl, cr, cam2)
            assert v == vo and (r.n_residual, r.n_outliers) == (ro.n_residual, ro.n_outliers) and (outl == outl_o).all()
            assert np.abs(np.array(r.delta) - np.array(ro.delta)).max() < 1e-7
            after = [(ctx.keypoints(l, w_, s)[0].tobytes(), ctx.matches(l, w_).tobytes()) for l in (0, 1) for w_ in (0, 1) for s in (0, 1)]
            assert before == after
            assert res_before == [bytes(ctx.result(l)) for l in (0, 1)]          # the live result records too
    ctx.close()


def test_speculative_fast_threshold_and_its_fallback_match_oracle(golden_dir):
    """k_select carries a per-(image, level) FAST threshold from frame to frame and k_fast_redo starts a level over at
    the caller's threshold when the speculation found too few corners.  A sequence whose corner count collapses and
    recovers (textured -> low contrast -> textured -> blank -> textured ...) drives both paths; every list must equal
    the oracle's, which knows no speculation."""
    import torch
    W, H = 640, 480
    w = SyntheticStereoWorld(W, H, 400.0, 0.12, seed=21, n_frames=6)
    cam = w.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=500)
    ctx = hip.Context(n_lanes=2, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 16)
    ctx.set_params(p); ctx.set_camera(cam)
    orcs = [O().Oracle(p), O().Oracle(p)]
    def low(img, k):      # contrast / k around mid-grey: most corners fall below the FAST threshold
        return (128.0 + (img.astype(np.float32) - 128.0) / k).round().clip(0, 255).astype(np.uint8)
    n_redo_like = 0
    for t, k in enumerate([1, 1, 4, 1, 1.5, 0, 1, 2.5, 1, 1]):
        L, R = [x.numpy() for x in w.render(t % 6)]
        if k == 0: L, R = np.full_like(L, 128), np.full_like(R, 128)
        elif k != 1: L, R = low(L, k), low(R, k)
        other = (R[:, ::-1].copy(), L[:, ::-1].copy()) if t % 3 else (L, R)        # lane 1 sees a different stream
        ctx.process_host([(L, R), other])
        for lane, fr in ((0, (L, R)), (1, other)):
            ro = orcs[lane].process(fr[0], fr[1], cam)
            assert_same_frame(ctx, lane, orcs[lane], ctx.result(lane), ro, "t=%d lane=%d k=%s" % (t, lane, k))
            assert ctx.status_word(lane) == 0
        n_redo_like += int(k not in (1,))
    assert n_redo_like >= 4
    ctx.close()


def test_get_values_and_per_octave_precomputed_bypass():
    """getValues in one synchronisation (svo_get_values) equals the individual getters; the precomputed-data bypass
    (P:131-162, 219-251) per OCTAVE: a FAST+ORB context with two octaves fed another context's lists octave by octave
    tracks and optimises to the same result."""
    from stereo_vo_amd.abi import DM_FAST_ORB
    W, H = 640, 480
    w = SyntheticStereoWorld(W, H, 400.0, 0.12, seed=31, n_frames=3)
    cam = w.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=600)
    p.detect_method = DM_FAST_ORB; p.nOctaves = 2; p.vo_use_matches_ids = 1
    a = hip.Context(n_lanes=2, max_w=W, max_h=H, max_kps=2048, max_cand=1 << 16, max_octaves=2)
    b = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=2048, max_cand=1 << 16, max_octaves=2)
    for c in (a, b):
        c.set_params(p); c.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(3):
        L, R = [x.numpy() for x in w.render(t)]
        a.process_host([(L, R), (R[:, ::-1].copy(), L[:, ::-1].copy())])
        ro = orc.process(L, R, cam)
        ra = a.result(0)
        assert ra.status == 0 and ra.n_octaves == 2
        for o in range(2):
            kl, dl, kr, dr, m, ids = a.values(0, 0, o)
            assert kl.tobytes() == a.keypoints(0, 0, 0, o)[0].tobytes() and (dl == a.keypoints(0, 0, 0, o)[1]).all()
            assert kr.tobytes() == a.keypoints(0, 0, 1, o)[0].tobytes() and (dr == a.keypoints(0, 0, 1, o)[1]).all()
            assert m.tobytes() == a.matches(0, 0, o).tobytes() and (ids == a.match_ids(0, 0, o)).all()
            assert kl.tobytes() == orc.keypoints(0, 0, o)[0].tobytes() and m.tobytes() == orc.matches(0, o).tobytes()
            assert len(kl) == ra.detected_left[o] and len(m) == ra.stereo_matches[o] and len(kl) > 50
            if t:
                pk = a.values(0, 1, o)
                assert pk[0].tobytes() == a.keypoints(0, 1, 0, o)[0].tobytes() and pk[4].tobytes() == a.matches(0, 1, o).tobytes()
            # hand this octave's lists to the other context
            if o == 0:
                b.L.svo_process(b.h, None, 0)                  # the shift of P:86-100
            b.put_features(0, 0, 0, kl, dl, W, H, octave=o); b.put_features(0, 0, 1, kr, dr, W, H, octave=o)
            b.put_matches(0, 0, m, octave=o); b.put_match_ids(0, 0, ids, octave=o)
        b.run_stages(hip.RUN_TRACK | hip.RUN_OPTIMIZE)
        rb = b.result(0)
        assert (rb.valid, rb.error_code) == (ra.valid, ra.error_code) == (ro.valid, ro.error_code)
        assert [rb.detected_left[o] for o in range(2)] == [ra.detected_left[o] for o in range(2)] and [rb.stereo_matches[o] for o in range(2)] == [ra.stereo_matches[o] for o in range(2)]
        for o in range(2):
            assert b.tracked(0, o).tobytes() == a.tracked(0, o).tobytes() == orc.tracked(o).tobytes()
        if ra.valid:
            assert np.abs(np.array(rb.outPose) - np.array(ra.outPose)).max() < 1e-9
    a.close(); b.close()


def test_capacity_overflow_is_reported_in_the_result_record(golden_dir):
    """svo_result.status carries the capacity bits (the reference has no caps, stage2_detect.cpp:461-464): a context too
    small for the requested keypoints says so in every result record instead of leaving it to a debug probe."""
    g, cam, p = load_small(golden_dir)
    q = p.copy(); q.orb_nfeats = 220
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=64, max_cand=1 << 15)
    ctx.set_camera(cam)
    try:
        ctx.set_params(q)                # 330 keypoints asked of the detector, 64 slots: refused when the parameters are loaded
        ctx.process_host([(g["L0"], g["R0"])])
        r = ctx.result(0)
        assert r.status & 2 and ctx.status_word(0) & 2
    except hip.SvoError as e:          # or the geometry is refused outright: also an explicit error, not a silent cut
        assert "capacity" in str(e)
    ctx.close()
    # a request that FITS when the parameters are loaded and overflows at run time: FAST+ORB without the NMS keeps every corner
    # (S2:613-614: no cap to check beforehand), far more than 128 slots here -- bit 2 in the record and in the status word
    from stereo_vo_amd.abi import DM_FAST_ORB
    q2 = p.copy(); q2.detect_method = DM_FAST_ORB; q2.nOctaves = 1; q2.non_maximal_suppression = 0
    small = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=128, max_cand=1 << 15)
    small.set_params(q2); small.set_camera(cam)
    small.process_host([(g["L0"], g["R0"])])
    assert small.result(0).status & 2 and small.status_word(0) & 2
    # a feature count that is not positive is refused outright (it used to wrap through size_t)
    q3 = p.copy(); q3.orb_nfeats = 0
    with pytest.raises(hip.SvoError):
        small.set_params(q3)
    small.close()
    ok = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ok.set_params(p); ok.set_camera(cam)
    ok.process_host([(g["L0"], g["R0"])])
    assert ok.result(0).status == 0
    ok.close()


def test_graph_replay_matches_plain_launches(golden_dir):
    """svo_use_graphs: a frame's kernel sequence captured into a hipGraph and replayed.  Host frames and device frames, the
    repeat flag (a different flag word: its own graph), a threshold change (its own graph) and a parameter change (graphs
    dropped) all give the oracle's lists."""
    import torch
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    for device_images in (False, True):
        ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
        ctx.set_params(p); ctx.set_camera(cam); ctx.use_graphs(True)
        orc = O().Oracle(p)
        seq = [(0, False, None), (1, False, None), (1, True, None), (2, False, None), (3, False, 50), (2, False, None), (1, False, None), (0, False, None)]
        for i, (t, rep, orbth) in enumerate(seq):
            L, R = g["L%d" % t], g["R%d" % t]
            if orbth is not None:
                ctx.set_orb_threshold(orbth); orc.set_orb_threshold(orbth)
            flags = hip.RUN_ALL | (hip.FLAG_REPEAT if rep else 0)
            if device_images:
                dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
                torch.cuda.synchronize()
                ctx.process_device([(dl.data_ptr(), dr.data_ptr())], W, H, W, flags)
            else:
                ctx.process_host([(L, R)], flags)
            r = ctx.result(0)
            ro = orc.process(L, R, cam, repeat=rep)
            assert_same_frame(ctx, 0, orc, r, ro, "graph dev=%s i=%d" % (device_images, i))
        q = p.copy(); q.orb_nfeats = 150
        ctx.set_params(q); orc.set_params(q)
        for t in (1, 2):
            ctx.process_host([(g["L%d" % t], g["R%d" % t])])
            assert_same_frame(ctx, 0, orc, ctx.result(0), orc.process(g["L%d" % t], g["R%d" % t], cam), "graph after set_params t=%d" % t)
        ctx.close()


@pytest.mark.parametrize("graphs", [False, True])
def test_adaptive_nms_after_fast_orb_matches_oracle(graphs):
    """stage2_detect.cpp:599-606 applies m_adaptive_non_max_sup to whatever detector ran: nmsmAdaptive on the FAST+ORB
    detector's output (every FAST corner of every x1/2 octave), two octaves, three frames, two lanes.  With graphs: the first
    frame of this mode needs a lazily allocated scratch buffer, which must exist before the capture begins."""
    from stereo_vo_amd.abi import DM_FAST_ORB
    W, H = 640, 480
    w = SyntheticStereoWorld(W, H, 400.0, 0.12, seed=12, n_frames=3)
    cam = w.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=400)
    p.detect_method = DM_FAST_ORB; p.nOctaves = 2; p.nmsMethod = 1
    ctx = hip.Context(n_lanes=2, max_w=W, max_h=H, max_kps=2048, max_cand=1 << 16, max_octaves=2)
    ctx.set_params(p); ctx.set_camera(cam)
    if graphs:
        ctx.use_graphs(True)
    orcs = [O().Oracle(p), O().Oracle(p)]
    for t in (0, 1, 2, 1, 0) if graphs else range(3):          # with graphs: both ring slots captured, then replayed
        L, R = [x.numpy() for x in w.render(t)]
        frames = [(L, R), (R[:, ::-1].copy(), L[:, ::-1].copy())]
        ctx.process_host(frames)
        for lane in range(2):
            ro = orcs[lane].process(frames[lane][0], frames[lane][1], cam)
            r = ctx.result(lane)
            for o in range(2):
                for side in (0, 1):
                    k, d = ctx.keypoints(lane, 0, side, o)
                    ko, do = orcs[lane].keypoints(0, side, o)
                    assert len(k) == len(ko) and k.tobytes() == ko.tobytes() and (d == do).all(), (t, lane, o, side, len(k), len(ko))
                assert ctx.matches(lane, 0, o).tobytes() == orcs[lane].matches(0, o).tobytes()
                assert ctx.tracked(lane, o).tobytes() == orcs[lane].tracked(o).tobytes()
            assert (r.valid, r.error_code) == (ro.valid, ro.error_code)
            if ro.valid:
                dp = np.abs(np.array(r.outPose) - np.array(ro.outPose))
                assert dp[:3].max() < POSE_TOL_M and dp[3:].max() < POSE_TOL_RAD
            assert ctx.status_word(lane) == 0 and r.detected_left[0] > 100
    ctx.close()


def test_five_thousand_keypoints_in_one_octave():
    """The reference has no keypoint cap (stage2_detect.cpp:461-464: orb_nfeats is free).  A single-octave ORB run asking
    for 5000 keypoints (7500 before the NMS) on 2048x1536 needs lists above 4096 entries: max_kps = 8192, where the NMS and
    Gauss-Newton kernels keep their sort / hash arrays in global scratch.  Three frames against the oracle, every list."""
    W, H = 2048, 1536
    w = SyntheticStereoWorld(W, H, 1280.0, 0.12, seed=51, n_frames=3)
    cam = w.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=5000)
    ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=8192, max_cand=1 << 18)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(3):
        L, R = [x.numpy() for x in w.render(t)]
        ctx.process_host([(L, R)])
        r, ro = ctx.result(0), orc.process(L, R, cam)
        assert_same_frame(ctx, 0, orc, r, ro, "5000 kps t=%d" % t)
        assert r.status == 0 and r.detected_left[0] > 4096, (r.status, r.detected_left[0])
    assert r.valid and r.tracked_feats_from_last_frame > 300
    ctx.close()
    # the same request against a 4096-entry context is refused outright (capacity), not cut silently
    small = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=4096, max_cand=1 << 18)
    small.set_camera(cam)
    with pytest.raises(hip.SvoError, match="capacity.*max_kps is 4096"):
        small.set_params(p)              # svo_set_params itself says so (SVO_ERR_CAPACITY + the numbers), before any frame
    small.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SVO_FUZZ_SEEDS", "12")))))
def test_random_parameter_sets_match_oracle(seed):
    """Parameter combinations nobody wrote a dedicated test for: every documented key of the path (SURVEY.md 8b "Config keys")
    drawn at random in ORB mode -- pyramid depth, feature count, NMS on / off / adaptive and its cell size, FAST threshold,
    response floor, both stereo matchers with and without the 1-to-1 rule, both trackers, RANSAC on / off, robust kernel on /
    off, iteration limits, warm start, match IDs -- four frames each, every list bit-exact, poses within tolerance."""
    rng = np.random.RandomState(1000 + seed)
    w, h = [(640, 480), (512, 384), (800, 600), (417, 311)][seed % 4]
    world = SyntheticStereoWorld(w + (-w) % 8, h + (-h) % 8, 400.0 * w / 640.0, 0.12, seed=100 + seed, n_frames=4)
    cam = StereoCamera.simple(400.0 * w / 640.0, w / 2.0, h / 2.0, 0.12, w, h)
    p = north_star_params(hip.default_params(), orb_nfeats=int(rng.choice([60, 200, 500, 900])))
    p.orb_nlevels = int(rng.choice([1, 2, 3, 5, 8]))
    p.non_maximal_suppression = int(rng.rand() < 0.75)
    p.nmsMethod = int(rng.rand() < 0.3)
    p.min_distance = int(rng.choice([2, 3, 4, 7, 10]))
    p.initial_FAST_threshold = int(rng.choice([5, 12, 20, 35]))
    p.minimum_ORB_response = float(rng.choice([0.0, 0.0, 1e-4]))
    p.match_method = int(rng.rand() < 0.4)
    p.enable_robust_1to1_match = int(rng.rand() < 0.6)
    p.max_y_diff = float(rng.choice([0.0, 1.0, 2.0, 3.5]))
    p.orb_max_distance = float(rng.choice([30.0, 60.0, 90.0]))
    p.ifm_method = int(rng.rand() < 0.4)
    p.ifm_win_w = int(rng.choice([5, 15, 40])); p.ifm_win_h = int(rng.choice([5, 15, 40]))
    p.filter_fund_matrix = int(rng.rand() < 0.8)
    p.use_robust_kernel = int(rng.rand() < 0.6)
    p.kernel_param = float(rng.choice([1.0, 3.0, 10.0]))
    p.initial_max_iters = int(rng.choice([1, 3, 10])); p.max_iters = int(rng.choice([1, 20, 100]))
    p.max_incr_cost = int(rng.choice([0, 3]))
    p.residual_threshold = float(rng.choice([2.0, 10.0, 50.0]))
    p.bad_tracking_th = int(rng.choice([5, 30]))
    p.use_previous_pose_as_initial = int(rng.rand() < 0.5)
    p.vo_use_matches_ids = int(rng.rand() < 0.5)
    tag = "seed %d: %dx%d levels %d nfeats %d nms %d/%d md %d th %d match %d/%d ydiff %.1f ifm %d F %d robust %d" % (
        seed, w, h, p.orb_nlevels, p.orb_nfeats, p.non_maximal_suppression, p.nmsMethod, p.min_distance, p.initial_FAST_threshold,
        p.match_method, p.enable_robust_1to1_match, p.max_y_diff, p.ifm_method, p.filter_fund_matrix, p.use_robust_kernel)
    # a level's 2 x quota list must fit the selection kernels (2048 entries; 4096 for contexts with max_kps > 4096): few levels x many features
    max_kps = 8192 if (p.orb_nlevels <= 2 and p.orb_nfeats >= 500) else 2048
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=max_kps, max_cand=1 << 16)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(4):
        L, R = [np.ascontiguousarray(x.numpy()[:h, :w]) for x in world.render(t)]
        ctx.process_host([(L, R)])
        r, ro = ctx.result(0), orc.process(L, R, cam)
        assert_same_frame(ctx, 0, orc, r, ro, "%s t=%d" % (tag, t))
        if p.vo_use_matches_ids:
            assert (ctx.match_ids(0, 0) == orc.match_ids(0)).all(), (tag, t, "match ids")
    ctx.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SVO_FUZZ_SEEDS", "10")))))
def test_random_parameter_sets_fast_orb_match_oracle(seed):
    """The same for detect_method = FAST + ORB on the x1/2 octave pyramid (stage2_detect.cpp:502-515): 1-3 octaves, the
    threshold limits that drive the reference's FAST-threshold adaptation, both NMS methods or none, both matchers / trackers."""
    from stereo_vo_amd.abi import DM_FAST_ORB
    rng = np.random.RandomState(2000 + seed)
    w, h = [(640, 480), (800, 600), (512, 384)][seed % 3]
    world = SyntheticStereoWorld(w, h, 400.0 * w / 640.0, 0.12, seed=200 + seed, n_frames=4)
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=int(rng.choice([150, 400, 900])))
    p.detect_method = DM_FAST_ORB
    p.nOctaves = int(rng.choice([1, 2, 3]))
    p.non_maximal_suppression = int(rng.rand() < 0.75)
    p.nmsMethod = int(rng.rand() < 0.3)
    p.min_distance = int(rng.choice([2, 3, 5, 8]))
    p.initial_FAST_threshold = int(rng.choice([8, 20, 40]))
    p.fast_min_th = int(rng.choice([3, 5])); p.fast_max_th = int(rng.choice([30, 60]))
    p.match_method = int(rng.rand() < 0.4)
    p.enable_robust_1to1_match = int(rng.rand() < 0.6)
    p.max_y_diff = float(rng.choice([1.0, 2.0]))
    p.orb_min_th = int(rng.choice([20, 30])); p.orb_max_th = int(rng.choice([60, 100]))
    p.ifm_method = int(rng.rand() < 0.4)
    p.use_robust_kernel = int(rng.rand() < 0.6)
    p.vo_use_matches_ids = int(rng.rand() < 0.5)
    tag = "seed %d: %dx%d octaves %d nfeats %d nms %d/%d md %d th %d match %d ifm %d" % (
        seed, w, h, p.nOctaves, p.orb_nfeats, p.non_maximal_suppression, p.nmsMethod, p.min_distance, p.initial_FAST_threshold, p.match_method, p.ifm_method)
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=4096, max_cand=1 << 17, max_octaves=3)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(4):
        L, R = [x.numpy() for x in world.render(t)]
        ctx.process_host([(L, R)])
        r, ro = ctx.result(0), orc.process(L, R, cam)
        # without NMS the reference keeps every FAST corner (S2:613-614), the context as many as its lists hold (max_kps >> octave):
        # a frame that exceeds them must say so in the result record, and is then no parity case
        if not p.non_maximal_suppression and any(max(ro.detected_left[o], ro.detected_right[o]) > max(64, 4096 >> o) for o in range(p.nOctaves)):
            assert r.status & 2, (tag, t, "list capacity exceeded without SVO_ST_KPS_OVERFLOW")
            ctx.close()
            return
        assert r.status == 0, (tag, t, r.status)
        assert (r.valid, r.error_code, r.n_octaves) == (ro.valid, ro.error_code, ro.n_octaves), (tag, t)
        for o in range(p.nOctaves):
            for side in (0, 1):
                k, d = ctx.keypoints(0, 0, side, octave=o)
                ko, do = orc.keypoints(0, side, octave=o)
                assert k.tobytes() == ko.tobytes() and (d == do).all(), (tag, t, "keypoints", o, side)
            assert ctx.matches(0, 0, octave=o).tobytes() == orc.matches(0, octave=o).tobytes(), (tag, t, "pairings", o)
            assert ctx.tracked(0, octave=o).tobytes() == orc.tracked(octave=o).tobytes(), (tag, t, "tracked", o)
        assert ctx.fast_threshold() == orc.fast_threshold(), (tag, t, "FAST threshold")
        if ro.valid:
            dp = np.abs(np.array(r.outPose) - np.array(ro.outPose))
            assert dp[:3].max() < POSE_TOL_M and dp[3:].max() < POSE_TOL_RAD, (tag, t, dp)
    ctx.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SVO_FUZZ_SEEDS", "6")))))
def test_random_call_sequences_match_oracle(seed):
    """What a caller may do BETWEEN frames, in random order, mirrored on the oracle: the next frame, the same frame again with
    request.repeat (P:86-89), a featureless frame (voecBadTracking -> the recovery rule keeps the older frame, P:326-330), new
    FAST / ORB thresholds (H:531, 538), resetIds (H:684), setThisFrameAsKF (H:675-683), a fresh estimator (svo_reset).  Every
    list, the match IDs of both slots, the key-frame counter and both thresholds after every call."""
    rng = np.random.RandomState(3000 + seed)
    w, h = 640, 480
    world = SyntheticStereoWorld(w, h, 400.0, 0.12, seed=300 + seed, n_frames=16)
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=int(rng.choice([300, 600])))
    p.vo_use_matches_ids = 1
    p.ifm_method = int(rng.rand() < 0.3); p.match_method = int(rng.rand() < 0.3)
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=2048, max_cand=1 << 16)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    blank = np.full((h, w), 90, np.uint8)
    t = 0; last = None; log = []; have_pairings = False
    for step in range(14):
        op = rng.choice(["next", "next", "next", "repeat", "blank", "fast_th", "orb_th", "reset_ids", "kf", "reset"])
        log.append(op)
        if op in ("fast_th", "orb_th"):
            v = int(rng.choice([2, 10, 25, 45, 80, 200]))
            if op == "fast_th": ctx.set_fast_threshold(v); orc.set_fast_threshold(v)
            else: ctx.set_orb_threshold(v); orc.set_orb_threshold(v)
            assert (ctx.fast_threshold(), ctx.orb_threshold()) == (orc.fast_threshold(), orc.orb_threshold()), (seed, log)
            continue
        if op == "reset_ids": ctx.reset_ids(0); orc.L.svo_oracle_reset_ids(orc.h); continue
        if op == "kf":
            if have_pairings: ctx.set_this_frame_as_kf(0); orc.L.svo_oracle_set_this_frame_as_kf(orc.h)
            else:
                with pytest.raises(hip.SvoError): ctx.set_this_frame_as_kf(0)          # nothing to make a key frame of
            continue
        if op == "reset":                       # the thresholds are the context's, not the lane's: carried over by hand
            ctx.reset(0); orc = O().Oracle(p); last = None; have_pairings = False
            orc.set_fast_threshold(ctx.fast_threshold()); orc.set_orb_threshold(ctx.orb_threshold())
            continue
        repeat = op == "repeat" and last is not None
        if op == "blank": L = R = blank
        elif repeat: L, R = last
        else:
            L, R = [x.numpy() for x in world.render(t)]; t += 1
        last = (L, R)
        ctx.process_host([(L, R)], hip.RUN_ALL | (hip.FLAG_REPEAT if repeat else 0))
        r, ro = ctx.result(0), orc.process(L, R, cam, repeat=repeat)
        assert_same_frame(ctx, 0, orc, r, ro, "seed %d %s" % (seed, log))
        have_pairings = ro.stereo_matches[0] > 0
        assert (ctx.match_ids(0, 0) == orc.match_ids(0)).all(), (seed, log, "ids cur")
        assert r.tracked_feats_from_last_KF == ro.tracked_feats_from_last_KF, (seed, log)
        assert (ctx.fast_threshold(), ctx.orb_threshold()) == (orc.fast_threshold(), orc.orb_threshold()), (seed, log)
    ctx.close()


def test_foreign_stream_may_be_destroyed_after_the_caller_switched_away(golden_dir):
    """svo_set_stream takes raw hipStream_t handles.  A C caller may enqueue frames on its own stream, synchronise it, DESTROY
    it and switch back to the context's stream: later waits / getters / svo_destroy must not touch the dead handle (they wait
    on context-owned events recorded behind the work instead)."""
    rt = C.CDLL("libamdhip64.so")
    g, cam, p = load_small(golden_dir)
    ctx = hip.Context(n_lanes=1, max_w=int(g["W"]), max_h=int(g["H"]), max_kps=1024, max_cand=1 << 15)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O().Oracle(p)
    for t in range(4):
        st = C.c_void_p()
        assert rt.hipStreamCreateWithFlags(C.byref(st), C.c_uint(1)) == 0             # hipStreamNonBlocking
        ctx.set_stream(st.value)
        ctx.process_host([(g["L%d" % t], g["R%d" % t])])
        assert rt.hipStreamSynchronize(st) == 0
        ctx.set_stream(None)
        assert rt.hipStreamDestroy(st) == 0
        r = ctx.result(0)                                                             # svo_wait inside: must not synchronise `st`
        ro = orc.process(g["L%d" % t], g["R%d" % t], cam)
        assert_same_frame(ctx, 0, orc, r, ro, "destroyed stream t=%d" % t)
    ctx.close()


def test_hand_over_record_of_another_layout_is_refused(golden_dir):
    """svo_import_frame checks the record's header (magic, version, max_kps, max_h, octaves, lanes): a record exported by a
    differently configured context has other offsets; it is not unpacked, the lane's status word says so, and the importer's
    own state stays usable."""
    import torch
    g, cam, p = load_small(golden_dir)
    W, H = int(g["W"]), int(g["H"])
    a = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    b = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=2048, max_cand=1 << 15)
    for c in (a, b):
        c.set_params(p); c.set_camera(cam)
        c.process_host([(g["L0"], g["R0"])]); c.process_host([(g["L1"], g["R1"])])
    nb = max(a.handover_bytes(), b.handover_bytes())
    blob = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()              # torch fills the buffer on ITS stream: it must be through before the export on the context's
    a.export_frame(blob.data_ptr(), nb); a.wait()
    before = b.keypoints(0, 1, 0)[0].tobytes()                                        # b's previous frame
    b.import_frame(blob.data_ptr(), nb); b.wait()
    assert b.status_word(0) & 4 and b.result(0).status & 4
    assert b.keypoints(0, 1, 0)[0].tobytes() == before
    # a matching record is taken, and garbage is refused as well
    a.import_frame(blob.data_ptr(), nb); a.wait()
    assert a.status_word(0) & 4 == 0
    blob.fill_(0x5A)
    a.import_frame(blob.data_ptr(), nb); a.wait()
    assert a.status_word(0) & 4
    # b carries on as if nothing had happened (the flag is per frame: cleared by the next detect)
    orc = O().Oracle(p)
    for t in (0, 1, 2):
        ro = orc.process(g["L%d" % t], g["R%d" % t], cam)
    b.process_host([(g["L2"], g["R2"])])
    assert_same_frame(b, 0, orc, b.result(0), ro, "after a refused import")
    a.close(); b.close()


@pytest.mark.parametrize("env", [{"SVO_DESC_KPW": "1"}, {"SVO_DESC_KPW": "3", "SVO_DESC_TL": "0"}, {"SVO_DESC_KPW": "64"}, {"SVO_HAM_SPLITS": "1"}, {"SVO_HAM_SPLITS": "7"}, {"SVO_HAM_FP4": "0"}, {"SVO_HAM_FP4": "0", "SVO_HAM_SPLITS": "7"},
                                 {"SVO_DEBUG_MODE": "53", "SVO_RC_SPLIT": "4,2,3"}, {"SVO_DEBUG_MODE": "53", "SVO_RC_SPLIT": "1"},
                                 {"SVO_NMS_NT": "512"}, {"SVO_REST_PRIO": "3"}, {"SVO_RS_C0": "32"}, {"SVO_RS_C0": "160"}, {"SVO_RS_C0": "160", "SVO_DEBUG_MODE": "53"}, {"SVO_RS_C0": "320", "SVO_DEBUG_MODE": "53"}, {"SVO_RS_C0": "320", "SVO_DEBUG_MODE": "51"},
                                 {"SVO_DEBUG_MODE": "14"}, {"SVO_DEBUG_MODE": "52"}, {"SVO_DEBUG_MODE": "14", "SVO_RS_C0": "160"}, {"SVO_TIMELINE": "1"}])
def test_kernel_launch_knobs_do_not_change_results(golden_dir, env):
    """The launch-shape knobs the library reads from the environment once per process (keypoints per wave of k_describe and where
    its Gaussian operands live, train splits of k_hamming, pair splits of the matrix-core RANSAC count with its ticket protocol, the chunk ends of the sample schedule -- 320 / 320 by default for a handful of lanes, 32 / 160 in the batched shapes --) select other code paths of the same arithmetic: the committed small
    sequence in a fresh process under each setting, every list against the golden vectors (= the oracle's), bit for bit."""
    import subprocess, sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from stereo_vo_amd import hip\n"
        "from stereo_vo_amd.abi import StereoCamera, north_star_params\n"
        "g = np.load(os.path.join(%r, 'oracle_small_seq.npz'))\n"
        "cam = StereoCamera.simple(float(g['F']), float(g['cx']), float(g['cy']), float(g['baseline']), int(g['W']), int(g['H']))\n"
        "p = north_star_params(hip.default_params(), orb_nfeats=int(g['orb_nfeats']))\n"
        "ctx = hip.Context(n_lanes=1, max_w=int(g['W']), max_h=int(g['H']), max_kps=1024, max_cand=1 << 15)\n"
        "ctx.set_params(p); ctx.set_camera(cam)\n"
        "for t in range(4):\n"
        "    ctx.process_host([(g['L%%d' %% t], g['R%%d' %% t])])\n"
        "    r = ctx.result(0)\n"
        "    for side in (0, 1):\n"
        "        k, d = ctx.keypoints(0, 0, side)\n"
        "        assert k.tobytes() == g['kps%%d_%%d' %% (side, t)].tobytes() and (d == g['desc%%d_%%d' %% (side, t)]).all(), ('features', t, side)\n"
        "    assert ctx.matches(0).tobytes() == g['matches%%d' %% t].tobytes(), ('pairings', t)\n"
        "    assert ctx.tracked(0).tobytes() == g['tracked%%d' %% t].tobytes(), ('tracked', t)\n"
        "    assert np.allclose(np.array(r.outPose), g['pose%%d' %% t], atol=1e-6) and ctx.status_word(0) == 0\n"
        "ctx.close(); print('same')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), golden_dir)
    # The A/B-only kernel forms (the int8 matcher, the RANSAC count forms 14 / 52: csrc/svo_kernels.h) are not in the product library since
    # round 6: the child loads libsvo_hip_ab.so for them (SVO_HIP_LIB) -- and the PRODUCT library must refuse the knob, loudly.
    ab = env.get("SVO_HAM_FP4") == "0" or env.get("SVO_DEBUG_MODE") in ("14", "52")
    full = dict(os.environ, **env)
    if ab:
        refused = subprocess.run([sys.executable, "-c", code], env=full, capture_output=True, text=True, timeout=600)
        assert refused.returncode != 0 and "A/B kernel form" in refused.stderr, (env, refused.stdout[-400:], refused.stderr[-800:])
        full["SVO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stereo_vo_amd", "libsvo_hip_ab.so")
    out = subprocess.run([sys.executable, "-c", code], env=full, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "same" in out.stdout, (env, out.stdout[-400:], out.stderr[-1200:])
