#!/usr/bin/env python3
"""One stream, one context (the reference's own usage: one estimator per camera rig): frames/s and ms per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_vo_amd import hip
from stereo_vo_amd.abi import north_star_params
from stereo_vo_amd.synth import SyntheticStereoWorld
W, H = 1280, 960
dev = torch.device("cuda", 0)
w = SyntheticStereoWorld(W, H, 800.0, 0.12, seed=0, n_frames=6, device=dev)
frames = [w.render(t) for t in range(6)]
torch.cuda.synchronize()
ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=4096, kernel_times=False, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_params(north_star_params(hip.default_params(), orb_nfeats=2000)); ctx.set_camera(w.camera())
sched = [0, 1, 2, 3, 4, 5, 4, 3, 2, 1]
def step(i):
    L, R = frames[sched[i % 10]]
    ctx.process_device([(L.data_ptr(), R.data_ptr())], W, H, W)
for i in range(10): step(i)
ctx.wait()
N = 200
t0 = time.perf_counter()
for i in range(N): step(i)
ctx.wait(); dt = time.perf_counter() - t0
print("pipelined enqueue: %.3f ms/frame, %.0f pairs/s" % (1e3 * dt / N, N / dt))
t0 = time.perf_counter()
for i in range(N): step(i); r = ctx.result(0)
dt = time.perf_counter() - t0
print("result read back every frame: %.3f ms/frame, %.0f pairs/s (valid %d)" % (1e3 * dt / N, N / dt, r.valid))
