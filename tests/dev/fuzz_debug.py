#!/usr/bin/env python3
"""GPU box: FAST+ORB fuzz seeds of tests/test_gpu_parity.py, first differences printed (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stereo_vo_amd import hip
from stereo_vo_amd.abi import north_star_params, DM_FAST_ORB
from stereo_vo_amd.synth import SyntheticStereoWorld
from oracle import oracle as O
for seed in [int(a) for a in sys.argv[1:]]:
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
    print("seed", seed, (w, h), "oct", p.nOctaves, "nf", p.orb_nfeats, "nms", p.non_maximal_suppression, p.nmsMethod, "md", p.min_distance, "th", p.initial_FAST_threshold, p.fast_min_th, p.fast_max_th, "mm", p.match_method, "ifm", p.ifm_method)
    ctx = hip.Context(n_lanes=1, max_w=w, max_h=h, max_kps=4096, max_cand=1 << 17, max_octaves=3)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O.Oracle(p)
    for t in range(4):
        L, R = [x.numpy() for x in world.render(t)]
        ctx.process_host([(L, R)])
        r, ro = ctx.result(0), orc.process(L, R, cam)
        print("  t", t, "valid", r.valid, ro.valid, "err", r.error_code, ro.error_code, "status", ctx.status_word(0), "fast_th", ctx.fast_threshold(), orc.fast_threshold())
        bad = False
        for o in range(p.nOctaves):
            for side in (0, 1):
                k, d = ctx.keypoints(0, 0, side, octave=o); ko, do = orc.keypoints(0, side, octave=o)
                same = len(k) == len(ko) and k.tobytes() == ko.tobytes() and (d == do).all()
                if not same:
                    bad = True
                    print("    oct", o, "side", side, "n", len(k), len(ko))
                    n = min(len(k), len(ko))
                    for i in range(n):
                        if k[i].tobytes() != ko[i].tobytes() or not (d[i] == do[i]).all():
                            print("      first diff at", i, k[i], ko[i], "desc equal", bool((d[i] == do[i]).all())); break
                    # set comparison
                    sk = set((float(a["x"]), float(a["y"])) for a in k); so = set((float(a["x"]), float(a["y"])) for a in ko)
                    print("      positions only-hip", len(sk - so), "only-oracle", len(so - sk))
            mm = ctx.matches(0, 0, octave=o).tobytes() == orc.matches(0, octave=o).tobytes()
            tt = ctx.tracked(0, octave=o).tobytes() == orc.tracked(octave=o).tobytes()
            if not (mm and tt): print("    oct", o, "matches", mm, "tracked", tt); bad = True
        if bad: break
    ctx.close()
