#!/usr/bin/env python3
"""Development probe (GPU box): stage-4 counters of the HIP path against the oracle for chosen stream seeds, with the many-lane kernel
forms (a 9-lane context fed nine copies of one stream).  usage: python tests/dev/ransac_repro.py [seed ...]   (SVO_DEBUG_MODE honoured)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from stereo_vo_amd import hip
from stereo_vo_amd.abi import north_star_params, TS_NAMES
from stereo_vo_amd.synth import SyntheticStereoWorld
from oracle import oracle as O

seeds = [int(a) for a in sys.argv[1:]] or [127]
W, H, NF, LANES = 1280, 960, 3, 9
dev = torch.device("cuda", 0)
p = north_star_params(hip.default_params(), orb_nfeats=2000)
bad = 0
for seed in seeds:
    w = SyntheticStereoWorld(W, H, 800.0, 0.12, seed=seed, n_frames=NF, device=dev, scene_seed=seed % 4)
    frames = [w.render(t) for t in range(NF)]
    torch.cuda.synchronize()
    cam = w.camera()
    ctx = hip.Context(n_lanes=LANES, max_w=W, max_h=H, max_kps=4096)
    ctx.set_params(p); ctx.set_camera(cam)
    orc = O.Oracle(p)
    for t in range(NF):
        L, R = frames[t]
        ctx.process_device([(L.data_ptr(), R.data_ptr())] * LANES, W, H, W)
        res = ctx.results()
        ro = orc.process(L.cpu().numpy(), R.cpu().numpy(), cam)
        for lane in (0, LANES - 1):
            r = res[lane]
            same = list(r.track_stats) == list(ro.track_stats) and ctx.tracked(lane).tobytes() == orc.tracked().tobytes()
            if not same:
                bad += 1
                print("seed %d frame %d lane %d MISMATCH" % (seed, t, lane))
                print("   gpu", dict(zip(TS_NAMES, r.track_stats)), "tracked", r.tracked_feats_from_last_frame)
                print("   cpu", dict(zip(TS_NAMES, ro.track_stats)), "tracked", ro.tracked_feats_from_last_frame)
    ctx.close(); orc.close()
print("seeds", seeds, "mismatching (frame, lane) pairs:", bad, "debug mode", os.environ.get("SVO_DEBUG_MODE", "0"))
