#!/usr/bin/env python3
"""Experiment: NCTX contexts x LANES lanes on separate HIP streams (do the latency-bound per-lane kernels of one
context overlap the throughput kernels of the other?).  Prints pairs/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_vo_amd import hip
from stereo_vo_amd.abi import north_star_params
from stereo_vo_amd.synth import SyntheticStereoWorld
NCTX = int(os.environ.get("NCTX", "2")); B = int(os.environ.get("LANES", "32")); W, H = 1280, 960
F = 4
dev = torch.device("cuda", 0)
worlds = [SyntheticStereoWorld(W, H, 800.0, 0.12, seed=s, n_frames=F, device=dev, scene_seed=s % 4) for s in range(B * NCTX)]
frames = [[w.render(t) for t in range(F)] for w in worlds]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev) for _ in range(NCTX)]
ctxs = []
for k in range(NCTX):
    c = hip.Context(n_lanes=B, max_w=W, max_h=H, max_kps=4096, kernel_times=False, stream=streams[k].cuda_stream)
    c.set_params(north_star_params(hip.default_params(), orb_nfeats=2000)); c.set_camera(worlds[0].camera())
    ctxs.append(c)
sched = [0, 1, 2, 3, 2, 1]
def step(i):
    t = sched[i % 6]
    for k, c in enumerate(ctxs):
        c.process_device([(frames[k * B + l][t][0].data_ptr(), frames[k * B + l][t][1].data_ptr()) for l in range(B)], W, H, W, hip.RUN_ALL)
for i in range(6): step(i)
for c in ctxs: c.wait()
N = 30
t0 = time.perf_counter()
for i in range(N): step(6 + i)
for c in ctxs: c.wait()
dt = time.perf_counter() - t0
print("nctx %d lanes %d: %.3f ms/step, %.0f pairs/s" % (NCTX, B, 1e3 * dt / N, NCTX * B * N / dt))
