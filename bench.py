#!/usr/bin/env python3
"""bench.py -- stereo pairs/sec of the HIP hot path (stages 2-5 of processNewImagePair) on synthetic streams.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU.  A "step" advances every lane (independent stereo stream) of every rank by
one frame: value = N * lanes * K / t, t = max over ranks of the barrier-bracketed wall time of the K steps.
All frames are rendered and resident in HBM before the timed region; nothing is copied host->device inside it.
Workload at N=1: BASELINE.json configs[1] -- 1280x960 synthetic stereo streams, ~2000 ORB keypoints per image
(orb_nfeats=2000, 8 levels), BF left-right matching, BF tracking, robust Gauss-Newton; 192 streams per GPU held by three
contexts of 64 (the per-stream latency-bound kernels of stages 3-5 of one context overlap the throughput kernels of
stage 2 of the next on a second HIP stream; kernel times below are per launch = per context of 64 streams).
Streams shard by independent stream across ranks with no data-path collective ("weak" scaling); the per-frame
result records are all-gathered over RCCL as in configs[3] (512 B-class, latency only).

Rank 0 prints ONE JSON line with the `roofline` of the dominant kernel (HIP events recorded around every kernel on
the stream it runs on) and the `cpu_baseline` (the CPU oracle on a bounded sample of the same workload, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from stereo_vo_amd import hip  # noqa: E402
from stereo_vo_amd.abi import Result, north_star_params  # noqa: E402
from stereo_vo_amd.synth import SyntheticStereoWorld  # noqa: E402

import ctypes as C  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def lane_seeds(rank, world_size, lanes):
    """Stream ids owned by `rank`: stream s -> rank s // lanes (contiguous blocks), SURVEY.md 8e."""
    assert 0 <= rank < world_size
    return [rank * lanes + i for i in range(lanes)]


def frame_schedule(step, n_frames):
    """Ping-pong 0..F-1..0.. so that consecutive steps are always consecutive poses of the trajectory."""
    if n_frames <= 1:
        return 0
    period = 2 * (n_frames - 1)
    k = step % period
    return k if k < n_frames else period - k


def gather_records(local, world_size):
    """All-gather of the per-rank result records (uint8 tensor [lanes, sizeof(svo_result)]) -> [world*lanes, ...]."""
    if world_size == 1:
        return local
    import torch.distributed as dist
    parts = [torch.empty_like(local) for _ in range(world_size)]
    dist.all_gather(parts, local)
    return torch.cat(parts, dim=0)


def reduce_max(value, device, world_size):
    if world_size == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def algorithmic_bytes(kernel, n_img, lv, n_kps, n_match, n_track):
    """ALGORITHMIC bytes one launch of `kernel` must move (SURVEY.md 8d split per kernel).
    lv = [(w, h)] per pyramid level."""
    px = [w * h for (w, h) in lv]
    if kernel == "fast":            # read every level once, 4 B per emitted corner
        return n_img * sum(px)
    if kernel == "resize":          # per launch (one level): read level l-1 + write level l; averaged over the 7 launches
        return n_img * sum(px[l - 1] + px[l] for l in range(1, len(px))) / max(1, len(px) - 1)
    if kernel == "describe":        # 37x37 window read + 28 B keypoint + 32 B descriptor written per keypoint slot
        return n_img * n_kps * 1.5 * (37 * 37 + 60)
    if kernel == "select":
        return n_img * n_kps * 3 * (4 + 81)
    if kernel == "nms_rowsort":
        return n_img * n_kps * 1.5 * 2 * 60
    if kernel == "hamming_lr":      # both descriptor sets read once, one packed word written per query
        return (n_img // 2) * (2 * n_kps * 32 + n_kps * 4)
    if kernel == "hamming_track":
        return (n_img // 2) * 2 * (2 * n_match * 32 + n_match * 4)
    if kernel == "gauss_newton":    # 40 B per tracked pair per iteration, ~12 iterations
        return (n_img // 2) * n_track * 40 * 12
    return (n_img // 2) * (n_match * 16 + n_track * 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--lanes", type=int, default=192, help="independent stereo streams per GPU (all contexts together)")
    ap.add_argument("--contexts", type=int, default=3, help="contexts per GPU, each on its own HIP stream with lanes/contexts streams: the latency-bound per-stream kernels of one overlap the throughput kernels of the other")
    ap.add_argument("--frames", type=int, default=6, help="distinct frames rendered per stream (played ping-pong)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=960)
    ap.add_argument("--orb-nfeats", type=int, default=2000)
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of one stream timed on the CPU oracle (0 = skip)")
    ap.add_argument("--schedule", default="pipelined", choices=["pipelined", "free"], help="pipelined: detect phases of the contexts serialised, stages 3-5 overlap the next context's detect; free: contexts run unsynchronised")
    ap.add_argument("--post-on-rest", type=int, default=0, help="1: the NMS / row-sort block of stage 2 runs on the overlap stream with stages 3-5")
    ap.add_argument("--det-priority", default="low", choices=["low", "high"])
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5"],
                    help="BASELINE.json configs[1] (default, the metric's configuration), configs[2] KITTI shape, configs[4] 2048x1536 FAST+ORB 3 octaves")
    args = ap.parse_args()
    detect_fast_orb, n_octaves, kitti = False, 1, False
    if args.workload == "config3":
        args.width, args.height, args.orb_nfeats, kitti = 1241, 376, 900, True
    elif args.workload == "config5":
        args.width, args.height, args.orb_nfeats, detect_fast_orb, n_octaves = 2048, 1536, 3300, True, 3
        args.lanes = min(args.lanes, 64); args.contexts = min(args.contexts, 2)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (one-GPU boxes): BENCH_FORCE_DEVICE puts every rank on that device, BENCH_DIST_BACKEND=gloo replaces RCCL,
    # so that the N > 1 code path (sharding, event ordering, all-gather, MAX-time) can be exercised without N GPUs
    if os.environ.get("BENCH_FORCE_DEVICE"):
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    W, H, B, F = args.width, args.height, args.lanes, args.frames
    focal = 718.856 if kitti else 800.0 * W / 1280.0
    baseline = 0.537 if kitti else 0.12
    cxy = dict(cx=607.19, cy=185.22) if kitti else {}
    seeds = lane_seeds(rank, world, B)
    # four scenes shared by the streams (textures are the slow part to mint), one trajectory per stream
    worlds = [SyntheticStereoWorld(W, H, focal, baseline, seed=s, n_frames=F, device=dev, scene_seed=s % 4, **cxy) for s in seeds]
    frames = [[w.render(t) for t in range(F)] for w in worlds]          # [lane][t] -> (L, R) uint8 on device
    cam = worlds[0].camera()
    torch.cuda.synchronize()

    p = north_star_params(hip.default_params(), orb_nfeats=args.orb_nfeats)
    if detect_fast_orb:
        from stereo_vo_amd.abi import DM_FAST_ORB
        p.detect_method = DM_FAST_ORB; p.nOctaves = n_octaves; p.use_robust_kernel = 1; p.kernel_param = 3.0
    NC = max(1, args.contexts)
    assert B % NC == 0 and B // NC <= 64, "--lanes must split evenly over --contexts, at most 64 streams per context"
    Bc = B // NC
    streams = [torch.cuda.Stream(dev) for _ in range(NC)]
    ctxs = []
    for k in range(NC):
        c_ = hip.Context(n_lanes=Bc, max_w=W, max_h=H, max_kps=4096, device=local_rank, kernel_times=True, stream=streams[k].cuda_stream,
                         max_octaves=n_octaves, max_cand=(1 << 18) if W * H > 2000000 else (1 << 17))
        c_.set_params(p); c_.set_camera(cam)
        ctxs.append(c_)
    rec = torch.zeros((B, C.sizeof(Result)), dtype=torch.uint8, device=dev)
    done = [torch.cuda.Event() for _ in range(NC)]

    # Schedule ("pipelined"): ONE normal-priority stream carries the detect phases (stage 2: resize / FAST / select /
    # describe / NMS -- the throughput kernels) of all contexts back to back; ONE high-priority stream carries the rest
    # of each frame (stages 3-5: mostly per-stream, latency-bound kernels), which therefore overlaps the next context's
    # detect phase without being starved by it.  Events: rest(k) waits for detect(k); the next detect of context k waits
    # for rest(k) (stage 4 reads the feature slot that detection overwrites next).
    s_det = torch.cuda.Stream(dev, priority=-1 if args.det_priority == "high" else 0)
    s_rest = torch.cuda.Stream(dev, priority=0 if args.det_priority == "high" else -1)
    det_done = [torch.cuda.Event() for _ in range(NC)]
    rest_done = [torch.cuda.Event() for _ in range(NC)]
    REST = hip.RUN_MATCH | hip.RUN_TRACK | hip.RUN_OPTIMIZE | (hip.RUN_DETECT_POST if args.post_on_rest else 0)
    state = {"first": True}
    gathered = torch.cuda.Event()
    pipelined = NC > 1 and args.schedule == "pipelined"

    def step(i):
        t = frame_schedule(i, F)
        for k, c_ in enumerate(ctxs):
            ptrs = [(frames[k * Bc + l][t][0].data_ptr(), frames[k * Bc + l][t][1].data_ptr()) for l in range(Bc)]
            if pipelined:
                if not state["first"]:
                    s_det.wait_event(rest_done[k])
                c_.set_stream(s_det.cuda_stream)
                c_.process_device(ptrs, W, H, W, hip.RUN_DETECT | (hip.FLAG_DETECT_NO_POST if args.post_on_rest else 0))
                det_done[k].record(s_det)
                s_rest.wait_event(det_done[k])
                c_.set_stream(s_rest.cuda_stream)
                c_.run_stages(REST)
                c_.copy_results_async(rec[k * Bc:(k + 1) * Bc].data_ptr(), Bc * C.sizeof(Result))
                rest_done[k].record(s_rest)
            else:
                c_.process_device(ptrs, W, H, W)
                c_.copy_results_async(rec[k * Bc:(k + 1) * Bc].data_ptr(), Bc * C.sizeof(Result))
        state["first"] = False
        if world > 1:          # the all-gather runs on torch's current stream: it waits for every context's last work
            for k in range(NC):
                if pipelined:
                    torch.cuda.current_stream(dev).wait_event(rest_done[k])
                else:
                    done[k].record(streams[k]); torch.cuda.current_stream(dev).wait_event(done[k])
        out = gather_records(rec, world)
        if world > 1:          # the next step's result copies must not overwrite `rec` while the all-gather still reads it
            gathered.record(torch.cuda.current_stream(dev))
            (s_rest if pipelined else torch.cuda.current_stream(dev)).wait_event(gathered)
            if not pipelined:
                for k in range(NC):
                    streams[k].wait_event(gathered)
        return out

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    def pooled_times():      # launches of all contexts pooled: ms and launch counts add up, a launch covers Bc streams
        acc = {}
        for c_ in ctxs:
            for kname, v in c_.kernel_times().items():
                a_ = acc.setdefault(kname, [0.0, 0]); a_[0] += v[0]; a_[1] += v[1]
        return acc

    # The warm-up steps time every kernel (the per-kernel table and the choice of the roofline kernel); the timed region
    # keeps the two events per launch only around that one kernel, whose duration it has to measure live.
    for c_ in ctxs:
        c_.wait()
    kt_warm = pooled_times()
    detect_kernels = ("resize", "fast", "select", "describe") + (() if args.post_on_rest else ("nms_rowsort",))
    pool = [k for k in kt_warm if kt_warm[k][1] > 0 and (k in detect_kernels or not pipelined)]
    dom = max(pool, key=lambda k: kt_warm[k][0] / kt_warm[k][1]) if pool else "fast"
    for c_ in ctxs:
        c_.kernel_times_select(dom)
        c_.kernel_times_reset()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    allrec = None
    for i in range(args.steps):
        allrec = step(args.warmup + i)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    dt = reduce_max(dt, dev, world)

    kt = pooled_times()
    results = []
    for c_ in ctxs:
        results += c_.results()
    n_valid = sum(1 for r in results if r.valid)
    mean_kps = float(np.mean([r.detected_left[0] for r in results]))
    mean_match = float(np.mean([r.stereo_matches[0] for r in results]))
    mean_track = float(np.mean([r.tracked_feats_from_last_frame for r in results]))
    assert allrec is not None and allrec.shape[0] == world * B

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / dt
        lw, lh = [], []
        if detect_fast_orb:       # FAST+ORB works on the x1/2 octave images (S1:80-83), not on ORB's x1/1.2 levels
            for o in range(n_octaves):
                lw.append(W >> o); lh.append(H >> o)
        else:
            for l in range(8):
                sf = np.float32(1.2 ** l)
                lw.append(int(np.rint(np.float32(W) / sf))); lh.append(int(np.rint(np.float32(H) / sf)))
        lv = list(zip(lw, lh))
        def table(tt):
            return {k: {"ms_per_launch": v[0] / max(1, v[1]) / (7 if k == "resize" else 1), "ms_per_step": v[0] / max(1, v[1]), "launches": int(v[1]) * (7 if k == "resize" else 1)}
                    for k, v in tt.items() if v[1] > 0}
        per_kernel_warm = table(kt_warm)          # every kernel, from the warm-up steps
        per_kernel = table(kt)                    # the roofline kernel, live in the timed region
        if dom not in per_kernel:
            per_kernel[dom] = per_kernel_warm[dom]
        abytes = algorithmic_bytes(dom, 2 * Bc, lv, mean_kps, mean_match, mean_track)
        achieved = abytes / (per_kernel[dom]["ms_per_launch"] * 1e-3) / 1e9
        # HBM traffic of that kernel from the committed PMC passes (tools/pmc_traffic.py: L2 memory-side read / write
        # requests counted by size in separate rocprofv3 --pmc runs of this same command), scaled from the profiled
        # lane count to this run's; None when the profile does not cover this workload
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            kk = "k_" + ("hamming" if dom.startswith("hamming") else dom)
            if pm.get("workload") == args.workload and kk in pm["read_bytes"]:
                traffic = int((pm["read_bytes"][kk] + pm["write_bytes"].get(kk, 0)) * Bc / pm["lanes"])
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(abytes), "avg_launch_ms": round(per_kernel[dom]["ms_per_launch"], 4),
                    "streams_per_launch": Bc,
                    "note": ("dominant kernel of the detect stream; stage 3-5 kernels run on the overlap stream and their spans are time-shared, not exclusive" if pipelined else "single stream: every span is an exclusive duration")}
        # end-to-end algorithmic traffic of the whole path (SURVEY.md 8d formula), for the DESIGN.md table
        P = sum(a * b for a, b in lv) / float(W * H)
        pair_bytes = 2 * (3 * P - 1) * W * H + 2 * mean_kps * 60 + 16 * mean_match + 40 * mean_track * 12
        cpu_baseline = None
        pose_rmse = None
        if world == 1 and args.cpu_frames > 0:
            from oracle import oracle as O      # checker / baseline only; never on the product path
            orc = O.Oracle(p)
            host = [(frames[0][t][0].cpu().numpy(), frames[0][t][1].cpu().numpy()) for t in range(F)]
            orc.process(host[0][0], host[0][1], cam)
            c0 = time.perf_counter()
            cpu_poses = []
            for i in range(args.cpu_frames):
                t = frame_schedule(1 + i, F)
                ro = orc.process(host[t][0], host[t][1], cam)
                cpu_poses.append((ro.valid, list(ro.outPose)))
            cdt = time.perf_counter() - c0
            # BASELINE.json's metric is "pairs/sec + pose RMSE vs CPU ref": the same frames of stream 0 through a fresh
            # one-stream HIP context (untimed), pose by pose against the oracle's
            chk = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=4096, device=local_rank, max_octaves=n_octaves,
                              max_cand=(1 << 18) if W * H > 2000000 else (1 << 17))
            chk.set_params(p); chk.set_camera(cam)
            err_t, err_r, n_cmp, n_flag = [], [], 0, 0
            for i in range(-1, args.cpu_frames):
                t = frame_schedule(1 + i, F)
                chk.process_device([(frames[0][t][0].data_ptr(), frames[0][t][1].data_ptr())], W, H, W)
                rg = chk.result(0)
                if i < 0:
                    continue
                ok_c, pc = cpu_poses[i]
                n_flag += int(bool(rg.valid) != bool(ok_c))
                if rg.valid and ok_c:
                    dpz = np.array(rg.outPose) - np.array(pc)
                    err_t.append(float(np.sum(dpz[:3] ** 2))); err_r.append(float(np.sum(dpz[3:] ** 2))); n_cmp += 1
            chk.close()
            pose_rmse = {"translation_m": float(np.sqrt(np.mean(err_t))) if err_t else None, "rotation_rad": float(np.sqrt(np.mean(err_r))) if err_r else None,
                         "frames": n_cmp, "valid_flag_mismatches": n_flag, "tolerance": "1e-3 m / 1e-4 rad per frame (tests/test_gpu_parity.py)"}
            cpu_baseline = {"value": round(args.cpu_frames / cdt, 3), "unit": "stereo pairs/s", "cores": 1, "kind": "port",
                            "sample": "%d frames of stream 0 (%dx%d, orb_nfeats=%d) on the single-threaded C oracle, host has %d cores"
                                      % (args.cpu_frames, W, H, args.orb_nfeats, os.cpu_count() or 0)}
        line = {
            "metric": "stereo pairs/sec @%d\u00d7%d" % (W, H), "value": round(value, 2), "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f64", "data": "synthetic",
            "config": {"workload": "%s: %dx%d synthetic stereo streams, %s, orb_nfeats %d (~%d kps/image in octave 0 after NMS), BF match, BF track, robust GN; %d independent streams per GPU in %d contexts on separate HIP streams, one frame per stream per step"
                                   % (args.workload, W, H, "FAST+ORB on %d x1/2 octaves" % n_octaves if detect_fast_orb else "ORB x 8 levels", args.orb_nfeats, int(mean_kps), B, NC),
                       "lanes_per_gpu": B, "contexts_per_gpu": NC, "lanes_per_context": Bc, "frames_per_stream": F, "schedule": args.schedule if NC > 1 else "single stream", "parallelism": "streams sharded across %d GPU(s), result all-gather per step" % world},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "pose_rmse_vs_cpu": pose_rmse,
            "path_hbm_frac": round(pair_bytes * value / 1e9 / HBM_PEAK_GBS, 5),
            "algorithmic_bytes_per_pair": int(pair_bytes),
            "valid_last_step": "%d/%d" % (n_valid, B),
            "mean_kps": round(mean_kps, 1), "mean_matches": round(mean_match, 1), "mean_tracked": round(mean_track, 1),
            "kernels_ms_per_context_step": {k: round(v["ms_per_step"], 4) for k, v in per_kernel_warm.items()},
            "kernels_ms_note": "all kernels: HIP-event spans of the %d warm-up steps; roofline kernel: spans of the timed region" % args.warmup,
        }
        print(json.dumps(line))
    for c_ in ctxs:
        c_.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
