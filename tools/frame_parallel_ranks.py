#!/usr/bin/env python3
"""Frame-parallelism within ONE stereo stream ACROSS RANKS: frame t is owned by rank t % world.  Each rank
runs stages 2-3 of its frames as soon as it can, receives the previous frame's hand-over record from rank - 1 with
torch.distributed recv (ncclRecv over xGMI when the backend is nccl and every rank has its own GPU), imports it, runs
stages 4-5, exports its own record and sends it to rank + 1.  Rank 0 also runs the same sequence through ONE context
sequentially and checks that every frame's pose and counts are IDENTICAL (the record carries the warm start too).

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/frame_parallel_ranks.py [--frames 12]
Test hooks: FP_FORCE_DEVICE=0 puts every rank on device 0, FP_BACKEND=gloo stages the record through host memory (RCCL
refuses two ranks on one device); the product path for several GPUs is the nccl one."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereo_vo_amd import hip
from stereo_vo_amd.abi import Result, north_star_params
from stereo_vo_amd.synth import SyntheticStereoWorld


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12); ap.add_argument("--width", type=int, default=640); ap.add_argument("--height", type=int, default=480)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("FP_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("FP_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))
    W, H, N = a.width, a.height, a.frames
    w = SyntheticStereoWorld(W, H, 400.0 * W / 640.0, 0.12, seed=77, n_frames=N, device=dev)
    cam = w.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=500)
    frames = [w.render(t) for t in range(N)]
    torch.cuda.synchronize()
    st = torch.cuda.current_stream(dev)
    ctx = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15, device=local, stream=st.cuda_stream)
    ctx.set_params(p); ctx.set_camera(cam)
    nb = ctx.handover_bytes()
    blob_in, blob_out = torch.zeros(nb, dtype=torch.uint8, device=dev), torch.zeros(nb, dtype=torch.uint8, device=dev)
    mine = {}
    dist.barrier()
    t0 = time.perf_counter()
    for t in range(N):
        if t % world != rank:
            continue
        L, R = frames[t]
        ctx.process_device([(L.data_ptr(), R.data_ptr())], W, H, W, hip.RUN_DETECT | hip.RUN_MATCH)
        if t > 0:
            src = (rank - 1) % world
            if backend == "nccl":
                dist.recv(blob_in, src=src)
            else:
                host = torch.empty(nb, dtype=torch.uint8); dist.recv(host, src=src); blob_in.copy_(host)
            ctx.import_frame(blob_in.data_ptr(), nb)
        ctx.run_stages(hip.RUN_TRACK | hip.RUN_OPTIMIZE)
        ctx.export_frame(blob_out.data_ptr(), nb)
        if t + 1 < N:
            dst = (rank + 1) % world
            if backend == "nccl":
                dist.send(blob_out, dst=dst)
            else:
                torch.cuda.synchronize(); dist.send(blob_out.cpu(), dst=dst)
        r = ctx.result(0)
        mine[t] = (int(r.valid), int(r.error_code), list(r.outPose), int(r.tracked_feats_from_last_frame), int(r.stereo_matches[0]), int(r.detected_left[0]))
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank == 0:
        merged = {}
        for d in allr: merged.update(d)
        seq = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15, device=local)
        seq.set_params(p); seq.set_camera(cam)
        same = True
        for t in range(N):
            L, R = frames[t]
            seq.process_device([(L.data_ptr(), R.data_ptr())], W, H, W)
            r = seq.result(0)
            ref = (int(r.valid), int(r.error_code), list(r.outPose), int(r.tracked_feats_from_last_frame), int(r.stereo_matches[0]), int(r.detected_left[0]))
            same = same and ref == merged[t]
        seq.close()
        print(json.dumps({"frame_parallel_ranks": world, "backend": backend, "frames": N, "identical_to_sequential": bool(same),
                          "valid_frames": sum(v[0] for v in merged.values()), "ms_per_frame": round(1e3 * dt / N, 3), "handover_bytes": nb}))
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
