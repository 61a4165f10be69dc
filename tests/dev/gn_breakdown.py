#!/usr/bin/env python3
"""dev probe (GPU box): where k_gauss_newton's time goes for ONE stream -- SVO_DEBUG_MODE 10 = return after the set-up,
11 = one iteration per phase, 0 = the whole kernel; prints us per launch and the mean iteration count."""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from stereo_vo_amd import hip
    from stereo_vo_amd.abi import north_star_params
    from stereo_vo_amd.synth import SyntheticStereoWorld
    W, H = 1280, 960
    dev = torch.device("cuda", 0)
    w = SyntheticStereoWorld(W, H, 800.0, 0.12, seed=0, n_frames=6, device=dev)
    frames = [w.render(t) for t in range(6)]
    p = north_star_params(hip.default_params(), orb_nfeats=2000)
    ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=4096, kernel_times=True)
    ctx.set_params(p); ctx.set_camera(w.camera())
    sched = [0, 1, 2, 3, 4, 5, 4, 3, 2, 1]
    its, trk = [], []
    for i in range(40):
        if i == 10: ctx.wait(); ctx.kernel_times_reset()
        L, R = frames[sched[i % 10]]
        ctx.process_device([(L.data_ptr(), R.data_ptr())], W, H, W)
        if i >= 10:
            r = ctx.result(0); its.append(r.num_it + r.num_it_final); trk.append(r.tracked_feats_from_last_frame)
    ctx.wait()
    kt = ctx.kernel_times()
    out = {k: round(1000 * v[0] / max(1, v[1]), 2) for k, v in kt.items() if v[1] > 0 and k in ("gauss_newton", "nms_rowsort", "track_finalize", "track_filter")}
    out["mean_iters"] = sum(its) / len(its); out["mean_tracked"] = sum(trk) / len(trk)
    print(json.dumps(out))
else:
    for nt in sys.argv[1:] or ["384"]:
        for dm in ("0", "64", "65", "66", "67", "10", "11", "60", "61", "62", "63"):
            env = dict(os.environ, SVO_DEBUG_MODE=dm, SVO_GN_NT=nt)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            print("NT", nt, "debug", dm, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
