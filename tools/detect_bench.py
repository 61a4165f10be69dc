#!/usr/bin/env python3
"""Micro-benchmark of stage 2 only (tuning aid): prints per-kernel ms for B lanes of 1280x960."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_vo_amd import hip
from stereo_vo_amd.abi import north_star_params
from stereo_vo_amd.synth import SyntheticStereoWorld
B = int(os.environ.get("LANES", "16")); W, H = 1280, 960
flags = int(os.environ.get("FLAGS", str(hip.RUN_ALL)))
dev = torch.device("cuda", 0)
worlds = [SyntheticStereoWorld(W, H, 800.0, 0.12, seed=s, n_frames=2, device=dev, scene_seed=s % 2) for s in range(B)]
frames = [[w.render(t) for t in range(2)] for w in worlds]
torch.cuda.synchronize()
ctx = hip.Context(n_lanes=B, max_w=W, max_h=H, max_kps=4096, kernel_times=True, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_params(north_star_params(hip.default_params(), orb_nfeats=2000)); ctx.set_camera(worlds[0].camera())
for i in range(3): ctx.process_device([(frames[l][i % 2][0].data_ptr(), frames[l][i % 2][1].data_ptr()) for l in range(B)], W, H, W, flags)
ctx.kernel_times_reset()
t0 = time.perf_counter()
N = 10
for i in range(N): ctx.process_device([(frames[l][i % 2][0].data_ptr(), frames[l][i % 2][1].data_ptr()) for l in range(B)], W, H, W, flags)
ctx.wait(); dt = time.perf_counter() - t0
kt = ctx.kernel_times()
print("mode", os.environ.get("SVO_DEBUG_MODE", "0"), "ms/step %.3f" % (1e3 * dt / N), {k: round(v[0] / max(1, v[1]), 3) for k, v in kt.items() if v[1]})
