#!/usr/bin/env python3
"""GPU box: time of k_gauss_newton per frame and iterations per frame for one stream (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_vo_amd import hip
from stereo_vo_amd.abi import north_star_params
from stereo_vo_amd.synth import SyntheticStereoWorld
B = int(os.environ.get("LANES", "1")); W, H, F = 1280, 960, 24
dev = torch.device("cuda", 0)
worlds = [SyntheticStereoWorld(W, H, 800.0, 0.12, seed=s, n_frames=F, device=dev) for s in range(B)]
frames = [[w.render(t) for t in range(F)] for w in worlds]
ctx = hip.Context(n_lanes=B, max_w=W, max_h=H, max_kps=4096, kernel_times=True, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_params(north_star_params(hip.default_params(), orb_nfeats=2000)); ctx.set_camera(worlds[0].camera())
its = []; trk = []
for t in range(F):
    if t == 4: ctx.kernel_times_reset()
    ctx.process_device([(frames[l][t][0].data_ptr(), frames[l][t][1].data_ptr()) for l in range(B)], W, H, W)
    r = ctx.result(0)
    if t >= 4: its.append(r.num_it + r.num_it_final); trk.append(r.tracked_feats_from_last_frame)
kt = ctx.kernel_times()
g = kt["gauss_newton"]
print("gn ms/frame %.4f  iterations/frame %.2f  us/iteration %.2f  tracked %.0f" % (g[0] / g[1], sum(its) / len(its), 1e3 * g[0] / g[1] / (sum(its) / len(its)), sum(trk) / len(trk)),
      {k: round(v[0] / max(1, v[1]), 4) for k, v in kt.items() if v[1]})
