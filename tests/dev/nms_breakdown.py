#!/usr/bin/env python3
"""dev probe (GPU box): k_nms_rowsort's phases for ONE stream through its early-return debug modes
(21 after the key build, 22 after the response sort, 26 after the cell table, 23 after the grid NMS, 24 after the survivor compaction,
25 after the row sort, 0 the whole kernel)."""
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
    for i in range(30):
        if i == 8: ctx.wait(); ctx.kernel_times_reset()
        L, R = frames[[0, 1, 2, 3, 4, 5, 4, 3, 2, 1][i % 10]]
        ctx.process_device([(L.data_ptr(), R.data_ptr())], W, H, W, hip.RUN_DETECT)
    ctx.wait()
    kt = ctx.kernel_times()
    print(json.dumps({k: round(1000 * v[0] / max(1, v[1]), 2) for k, v in kt.items() if v[1] > 0}))
else:
    for dm in ("0", "21", "22", "26", "23", "24", "25"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SVO_DEBUG_MODE=dm), capture_output=True, text=True)
        print("debug", dm, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
