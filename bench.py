#!/usr/bin/env python3
"""bench.py -- stereo pairs/sec of the HIP hot path (stages 2-5 of processNewImagePair) on synthetic streams.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU.  A "step" advances every lane (independent stereo stream) of every rank by
one frame: value = N * lanes * K / t, t = max over ranks of the barrier-bracketed wall time of the K steps.
All frames are rendered and resident in HBM before the timed region; nothing is copied host->device inside it.
Every stream sees a NEW frame at every step of the run (round 4): --trajectories T trajectories of F frames each are rendered
through a long "street" world (stereo_vo_amd/synth.py), stream s plays trajectory s % T from frame 7 (s // T) on, and F is
chosen so that no stream ever wraps (F >= 200).  The speculative FAST threshold of k_fast is therefore always applied to a frame
it has not seen: `fast_redo_rate` reports how often it failed, `scene_cuts` times the same batch with a cut every 20 frames.
Workload at N=1: BASELINE.json configs[1] -- 1280x960 synthetic stereo streams, ~2000 ORB keypoints per image
(orb_nfeats=2000, 8 levels), BF left-right matching, BF tracking, robust Gauss-Newton; 192 streams per GPU held by two
contexts of 96 since round 6 (three of 64 in rounds 2-5: 69.9 k against 70.8 k pairs/s in one call, profiles/r06_contexts_sweep.txt) (the per-stream latency-bound kernels of stages 3-5 of one context overlap the throughput kernels of
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

# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams that share one never overlap.  Which
# streams share depends on every stream the process has made (torch's included): the two-context shape of configs[4] ran at 23.7 k or at
# 36.7 k pairs/s depending on nothing else (profiles/r06_config5_hw_queues.txt).  Eight queues give every stream of the schedules here one of
# its own; the variable must be in the environment before the HIP runtime initialises, hence here, and only if the caller has not chosen.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from stereo_vo_amd import hip  # noqa: E402
from stereo_vo_amd.abi import Result, TS_NAMES, north_star_params  # noqa: E402
from stereo_vo_amd.synth import SyntheticStereoWorld  # noqa: E402
from stereo_vo_amd.pipeline import StreamBatch  # noqa: E402

import ctypes as C  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
PMC_PROFILE = {"config2": "r06_pmc.json", "config3": "r06_pmc_config3.json", "config5": "r06_pmc_config5.json"}
# svo_debug_timeline's kinds (stereo_vo_amd/csrc/svo_device.h, TL_*), and which of them the detect stream carries in the default split
TIMELINE_KINDS = ["begin_frame", "resize", "fast", "select", "harris", "select_sort", "nms_rowsort", "describe", "hamming", "match_lr_filter", "track_filter",
                  "ransac_schedule", "ransac_hyp", "ransac_count", "track_finalize", "match_ids", "gauss_newton", "other"]
TIMELINE_DETECT = {"resize", "fast", "select", "harris", "select_sort"}


def timeline_report(ctxs, post_on_rest, csv_path=""):
    """SVO_TIMELINE=1: what was really in flight during the steps since the tables were last cleared.  Every launch
    left the hull [first wave in, last wave out] of its blocks on the device-wide 100 MHz wall clock; a launch belongs to queue 0 (the
    detect stream) or to queue 1 + k (context k's stage 3-5 stream).  Returns: the share of the busy time with 1, 2, 3 ... queues
    executing; which kernel is alone on the chip, by time; per queue the time it executes, and the gaps between the end of one of its
    kernels and the start of the next; per kernel the mean hull (to set beside rocprofv3's durations)."""
    NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
    recs = []
    for k, c in enumerate(ctxs):
        _, a = c.timeline()
        for s_, kind, aux in zip(*np.nonzero(a[:, :, :, 0] != NONE)):
            t0, t1 = int(a[s_, kind, aux, 0]), int(a[s_, kind, aux, 1])
            if t1 <= t0 or kind >= len(TIMELINE_KINDS):
                continue
            name = TIMELINE_KINDS[kind]
            on_detect = (name in TIMELINE_DETECT or (name == "begin_frame" and aux == 0) or (not post_on_rest and name in ("nms_rowsort", "describe")))
            recs.append((0 if on_detect else 1 + k, k, name, int(aux), int(s_), t0, t1))
    if not recs:
        return {"error": "no records: was the library built with the timeline scope and the context created under SVO_TIMELINE=1?"}
    if csv_path:
        with open(csv_path, "w") as f:
            f.write("queue,context,kernel,aux,frame_mod16,t0_us,t1_us\n")
            base = min(r[5] for r in recs)
            for r in sorted(recs, key=lambda r: r[5]):
                f.write("%d,%d,%s,%d,%d,%.2f,%.2f\n" % (r[0], r[1], r[2], r[3], r[4], (r[5] - base) / 100.0, (r[6] - base) / 100.0))
    nq = 1 + len(ctxs)
    ev = []
    for i, r in enumerate(recs):
        ev.append((r[5], 1, i)); ev.append((r[6], -1, i))
    ev.sort(key=lambda e: (e[0], e[1]))
    active = [set() for _ in range(nq)]
    hist = [0.0] * (nq + 1)
    alone, pair = {}, {}
    last = ev[0][0]
    for t, d, i in ev:
        busy = [q for q in range(nq) if active[q]]
        dt_ = t - last
        hist[len(busy)] += dt_
        if dt_ > 0 and len(busy) == 1:
            for j in active[busy[0]]:
                alone[recs[j][2]] = alone.get(recs[j][2], 0.0) + dt_
        if dt_ > 0 and len(busy) == 2:
            key = " + ".join(sorted({recs[j][2] for q in busy for j in active[q]}))
            pair[key] = pair.get(key, 0.0) + dt_
        last = t
        (active[recs[i][0]].add if d > 0 else active[recs[i][0]].discard)(i)
    span = ev[-1][0] - ev[0][0]
    busy_t = sum(hist[1:])
    per_queue = {}
    for q in range(nq):
        rs = sorted((r for r in recs if r[0] == q), key=lambda r: r[5])
        run = sum(r[6] - r[5] for r in rs)
        gaps = [b[5] - a_[6] for a_, b in zip(rs, rs[1:]) if b[5] > a_[6]]
        per_queue["detect" if q == 0 else "rest%d" % (q - 1)] = {"launches": len(rs), "executing_frac_of_span": round(run / span, 4),
                                                                "gap_us_mean": round(float(np.mean(gaps)) / 100.0, 2) if gaps else None,
                                                                "gap_us_median": round(float(np.median(gaps)) / 100.0, 2) if gaps else None,
                                                                "gaps_over_20us": int(sum(1 for g in gaps if g > 2000))}
    hull = {}
    for r in recs:
        key = r[2] if r[2] not in ("resize", "ransac_hyp", "ransac_count", "hamming", "select", "fast", "begin_frame") else "%s[%d]" % (r[2], r[3])
        hull.setdefault(key, []).append((r[6] - r[5]) / 100.0)
    top = lambda d, n: {k: round(v / busy_t, 4) for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:n]}
    return {"launches": len(recs), "span_ms": round(span / 1e5, 3), "idle_frac": round(hist[0] / span, 4),
            "queues_executing_frac_of_busy": {str(k): round(hist[k] / busy_t, 4) for k in range(1, nq + 1)},
            "mean_queues_executing": round(sum(k * hist[k] for k in range(1, nq + 1)) / busy_t, 3),
            "alone_on_chip_by_kernel_frac_of_busy": top(alone, 8), "two_in_flight_pairs_frac_of_busy": top(pair, 8),
            "per_queue": per_queue, "hull_us_mean": {k: round(float(np.mean(v)), 1) for k, v in sorted(hull.items())},
            "note": "in-kernel wall-clock stamps (s_memrealtime, 10 ns) of an UNPROFILED run; hull = first sampled wave in .. last sampled wave out of a launch (grids above 1024 blocks: every 32nd block and the last 32); queue 0 = the detect stream, 1 + k = context k's stage 3-5 stream"}
    # ONE committed counter summary per workload (tools/pmc_passes.py writes it); a line never quotes an older file silently


def lane_seeds(rank, world_size, lanes):
    """Stream ids owned by `rank`: stream s -> rank s // lanes (contiguous blocks), SURVEY.md 8e."""
    assert 0 <= rank < world_size
    return [rank * lanes + i for i in range(lanes)]


def frame_schedule(step, n_frames):
    """Ping-pong 0..F-1..0.. so that consecutive steps are always consecutive poses of a trajectory (short test sequences, the host-fed ring)."""
    if n_frames <= 1:
        return 0
    period = 2 * (n_frames - 1)
    k = step % period
    return k if k < n_frames else period - k


PHASE_STRIDE = 7       # frames between two streams that play the same trajectory


def frames_needed(lanes, trajectories, total_steps):
    """Frames per trajectory such that stream l = (trajectory l % T, first frame PHASE_STRIDE * (l // T)) never wraps in total_steps steps."""
    phases = (lanes + trajectories - 1) // trajectories
    return max(200, (phases - 1) * PHASE_STRIDE + total_steps + 1)


def lane_frame(lane, step, trajectories, n_frames):
    """(trajectory, frame) stream `lane` of this rank is shown at `step`: consecutive frames of its trajectory, never the same twice
    within n_frames - phase steps (past that it plays the trajectory backwards: a test hook, the timed runs never get there)."""
    j, k = lane % trajectories, PHASE_STRIDE * (lane // trajectories) + step
    if n_frames <= 1:
        return j, 0
    return j, frame_schedule(k, n_frames)


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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--lanes", type=int, default=192, help="independent stereo streams per GPU (all contexts together)")
    ap.add_argument("--contexts", type=int, default=2, help="contexts per GPU, each on its own HIP stream with lanes/contexts streams: the latency-bound per-stream kernels of one overlap the throughput kernels of the other")
    ap.add_argument("--frames", type=int, default=0, help="frames rendered per trajectory; 0 (default) = as many as it takes for no stream ever to see a frame twice (>= 200)")
    ap.add_argument("--trajectories", type=int, default=8, help="distinct camera trajectories rendered per rank; stream s plays trajectory s %% T from frame 7 * (s // T) on")
    ap.add_argument("--long-steps", type=int, default=100, help="N=1: when --steps is smaller than this, one more timed leg of this many steps (after --warmup untimed ones, estimators reset, the same streams from their first frame on: no stream sees a frame twice inside the leg), reported as `long_run` -- so that a short --steps run still carries a measurement over a few hundred milliseconds; 0 = skip")
    ap.add_argument("--cut-steps", type=int, default=60, help="steps of the scene-cut leg (every stream jumps to another trajectory every 20 frames, estimators reset as an application would); N=1 only; 0 = skip")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=960)
    ap.add_argument("--orb-nfeats", type=int, default=2000)
    ap.add_argument("--cpu-frames", type=int, default=128, help="upper bound on the frames per probe stream replayed on the CPU oracle (baseline timing + parity probe; 0 = skip)")
    ap.add_argument("--host-fed-steps", type=int, default=16, help="steps of the host-fed leg (page-locked host frames uploaded per step on the contexts' copy streams; 0 = skip); N=1 only")
    ap.add_argument("--single-stream", type=int, default=1, help="1: also time ONE stream alone (plain launches, hipGraph replay, frames dealt to 2 / 3 contexts); N=1, config2 only")
    ap.add_argument("--exclusive", type=int, default=1, help="1: after the timed region, time the roofline kernel again with one context alone on the GPU (roofline.exclusive); 0 = skip (profiling runs)")
    ap.add_argument("--dump-records", default="", help="test hook: every rank writes its own and the gathered result records of the last step to <path>.rank<r>.npz")
    ap.add_argument("--schedule", default="pipelined", choices=["pipelined", "free"], help="pipelined: detect phases of the contexts serialised, stages 3-5 overlap the next context's detect; free: contexts run unsynchronised")
    ap.add_argument("--post-on-rest", type=int, default=1, help="0: all of stage 2 on the detect stream; 1 (default): the NMS / row-sort block of stage 2 and the description run on the overlap stream(s) with stages 3-5; 2: on a third stream of its own; 3: as 1, together with the per-level selection (the detect stream keeps pyramid + FAST only)")
    ap.add_argument("--det-priority", default="high", choices=["low", "high"], help="which side of the pipelined schedule gets the high HIP stream priority: 'high' (default) = the detect stream, the serial chain of a step once the NMS, the description and stages 3-5 run on one stream per context (66.4 k against 64.9 k pairs/s); 'low' = the stage 3-5 streams (the better choice when ONE stream carries all of them: 53.3 k vs 47.1 k in round 2)")
    ap.add_argument("--det-streams", type=int, default=1, help="HIP streams the detect phases of the contexts alternate over (pipelined schedule)")
    ap.add_argument("--rest-streams", type=int, default=0, help="HIP streams stages 3-5 of the contexts alternate over (pipelined schedule); 0 = one per context")
    ap.add_argument("--detect-ahead", type=int, default=1, help="1 (default): a context's detector of frame t + 1 starts once the description of frame t has read the detector's scratch (it overlaps that context's own stages 3-5); 0: it waits for the whole frame t (rounds 1-3)")
    ap.add_argument("--max-kps", type=int, default=4096, help="keypoint capacity per image of the contexts (svo_config.max_kps)")
    ap.add_argument("--scene", default="street", choices=["street", "planes", "relief"], help="synthetic scene type (stereo_vo_amd/synth.py): street (default, round 4) = ground + far wall + a facade every 4.5 m along a world as long as the trajectory; planes = wall + ground + facades (rounds 1-3: a ~1 m world, needs --frames 6); relief = planes plus 28 billboards at 4..22 m")
    ap.add_argument("--relief-lanes", type=int, default=16, help="N=1, config2: streams of the extra leg on the OTHER scene type (reported as `other_scene`: pass-through counters and pose error against ground truth beside the timed scene's); 0 = skip")
    ap.add_argument("--other-workloads", type=int, default=1, help="1: after everything else, short timed legs of the other single-GPU configurations (BASELINE.json configs[2] KITTI shape, configs[4] 2048x1536 FAST+ORB) as sub-processes, reported as `other_workloads`; N=1, config2 only")
    ap.add_argument("--timeline", type=int, default=0, help="N: create the contexts under SVO_TIMELINE=1 and, after the timed region, run N (<= 14) more steps whose launches are reported as `timeline` (what was in flight, from wall-clock stamps inside the kernels); the stamps cost a little, so a line with --timeline is not the headline")
    ap.add_argument("--timeline-csv", default="", help="with --timeline: also write every launch hull of those steps to this CSV")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5"],
                    help="BASELINE.json configs[1] (default, the metric's configuration), configs[2] KITTI shape, configs[4] 2048x1536 FAST+ORB 3 octaves")
    args = ap.parse_args()
    detect_fast_orb, n_octaves, kitti = False, 1, False
    if args.workload == "config3":
        args.width, args.height, args.orb_nfeats, kitti = 1241, 376, 900, True
    elif args.workload == "config5":
        args.width, args.height, args.orb_nfeats, detect_fast_orb, n_octaves = 2048, 1536, 3300, True, 3
        args.lanes = min(args.lanes, 64); args.contexts = min(args.contexts, 2)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: this process becomes the launcher of its own N ranks (one per GPU)
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d (or plainly, without WORLD_SIZE set: bench.py then launches its own ranks)" % (args.gpus, world, args.gpus))
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

    W, H, B = args.width, args.height, args.lanes
    T = max(1, min(args.trajectories, B))
    total_steps = args.warmup + args.steps
    long_steps = args.long_steps if (world == 1 and args.long_steps > args.steps) else 0
    plan_steps = max(total_steps + min(max(args.timeline, 0), 14), args.warmup + long_steps)           # frames / pointer tables cover the longer of the legs
    F = args.frames if args.frames > 0 else frames_needed(B, T, plan_steps)
    focal = 718.856 if kitti else 800.0 * W / 1280.0
    baseline = 0.537 if kitti else 0.12
    cxy = dict(cx=607.19, cy=185.22) if kitti else {}
    seeds = lane_seeds(rank, world, B)
    # T trajectories per rank through four scenes (textures are the slow part to mint); every stream = (trajectory, first frame)
    worlds = [SyntheticStereoWorld(W, H, focal, baseline, seed=rank * T + j, n_frames=F, device=dev, scene_seed=j % 4, scene=args.scene, noise_on_device=True, **cxy) for j in range(T)]
    t_render = time.perf_counter()
    frames = [[w.render(t) for t in range(F)] for w in worlds]          # [trajectory][t] -> (L, R) uint8 on device
    cam = worlds[0].camera()
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t_render
    t_render_end = time.perf_counter()

    def frame_of(lane, step):
        j, t = lane_frame(lane, step, T, F)
        return frames[j][t]

    def ptrs_at_step(step):
        out = []
        for l in range(B):
            L_, R_ = frame_of(l, step)
            out.append((L_.data_ptr(), R_.data_ptr()))
        return out

    p = north_star_params(hip.default_params(), orb_nfeats=args.orb_nfeats)
    if detect_fast_orb:
        from stereo_vo_amd.abi import DM_FAST_ORB
        p.detect_method = DM_FAST_ORB; p.nOctaves = n_octaves; p.use_robust_kernel = 1; p.kernel_param = 3.0
    NC = max(1, args.contexts)
    if args.timeline > 0:
        os.environ["SVO_TIMELINE"] = "1"
    batch = StreamBatch(p, cam, W, H, B, NC, device=local_rank, schedule=args.schedule, post_on_rest=("own" if args.post_on_rest == 2 else ("select" if args.post_on_rest == 3 else bool(args.post_on_rest))),
                        det_priority=args.det_priority, kernel_times=True, max_octaves=n_octaves, det_streams=args.det_streams, rest_streams=args.rest_streams, max_kps=args.max_kps, detect_ahead=bool(args.detect_ahead))
    Bc, pipelined, ctxs = batch.Bc, batch.pipelined, batch.ctxs
    ptrs_by_step = [ptrs_at_step(i) for i in range(plan_steps)]          # built ahead: the timed loop only indexes it

    frames_by_step = [batch.prepare(q) for q in ptrs_by_step]            # the ctypes frame tables too: the host's share of a step is then the launches
    gather_done = [torch.cuda.Event(), torch.cuda.Event()]

    def step(i):
        """One frame of every stream of this rank (stereo_vo_amd/pipeline.py), then the all-gather of the result records.
        N > 1: the records alternate between two buffers, so that step i + 1 of this rank does not wait for the all-gather of step i
        (which needs every rank to have finished step i): only the buffer's previous use, two steps back, has to be over."""
        if world > 1:
            batch.flip_records()
            if i >= 2:
                batch.hold_for(gather_done[i & 1])       # the all-gather that read this buffer two steps ago
        batch.step_prepared(frames_by_step[i])
        if world > 1:          # the all-gather runs on torch's current stream: it waits for every context's last work
            batch.make_wait(torch.cuda.current_stream(dev))
        out = gather_records(batch.rec, world)
        if world > 1:
            gather_done[i & 1].record(torch.cuda.current_stream(dev))
        return out

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def run_long_leg(order_note):
        """`--long-steps` timed steps of the same batch, reset, every stream from its first frame on (see --long-steps); with the final
        state checked against the oracle.  Returns the `long_run` block of the line."""
        long_run = None
        try:
            batch.reset()
            for i in range(args.warmup):
                step(i)
            torch.cuda.synchronize()
            for c_ in ctxs:
                c_.redo_count(reset=True)
            tl = time.perf_counter()
            for i in range(long_steps):
                step(args.warmup + i)
            torch.cuda.synchronize()
            tl = time.perf_counter() - tl
            redo_l = sum(c_.redo_count(reset=True) for c_ in ctxs)
            res_l = batch.results()
            lr_parity = None
            if args.cpu_frames > 0:
                try:
                    lr_parity = final_state_parity(batch, p, cam, res_l, lambda g: [frame_of(g, i) for i in range(args.warmup + long_steps)], args.warmup + long_steps)
                except Exception as e:
                    lr_parity = {"error": str(e)}
            long_run = {"parity": lr_parity, "pairs_per_s": round(B * long_steps / tl, 2), "ms_per_step": round(1e3 * tl / long_steps, 4), "steps": long_steps, "warmup": args.warmup,
                        "timed_region_s": round(tl, 4), "valid_last_step": "%d/%d" % (sum(1 for r in res_l if r.valid), B),
                        "fast_redo_rate": round(redo_l / float(max(1, long_steps * 2 * B * (n_octaves if detect_fast_orb else 8))), 6),
                        "note": "the same batch, reset, every stream from its first frame on (%d distinct consecutive frames per stream), timed like `value`; `value` itself is the --steps run the contract asks for" % (args.warmup + long_steps)}
        except Exception as e:
            long_run = {"error": str(e)}
        if isinstance(long_run, dict):
            long_run["order"] = order_note
        return long_run

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    # The warm-up steps time every kernel (the per-kernel table and the choice of the roofline kernel); the timed region
    # keeps the two events per launch only around that one kernel, whose duration it has to measure live.
    for c_ in ctxs:
        c_.wait()
    kt_warm = batch.pooled_kernel_times()
    detect_kernels = ("resize", "fast", "select") + (() if args.post_on_rest else ("describe", "nms_rowsort"))      # what the detect stream carries
    pool = [k for k in kt_warm if kt_warm[k][1] > 0 and (k in detect_kernels or not pipelined)]
    dom = max(pool, key=lambda k: kt_warm[k][0] / kt_warm[k][1] / (7 if k == "resize" else 1)) if pool else "fast"      # per LAUNCH: the resize span covers seven
    for c_ in ctxs:
        c_.kernel_times_select(dom)
        c_.kernel_times_reset()
    for c_ in ctxs:
        c_.redo_count(reset=True)                 # (waits for the context: the warm-up is over)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    allrec = None
    for i in range(args.steps):
        allrec = step(args.warmup + i)
    t_enqueue = time.perf_counter() - t0          # the host's share: every launch of the timed steps is enqueued (not: executed) by now
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    dt_own = dt                                   # this rank's own clock around the K steps; `value` uses the MAX over ranks
    dt = reduce_max(dt, dev, world)

    kt = batch.pooled_kernel_times()
    redo_pairs = sum(c_.redo_count(reset=True) for c_ in ctxs)
    results = batch.results()
    timeline = None
    if args.timeline > 0 and world == 1:
        nt_ = min(args.timeline, 14, plan_steps - total_steps)
        if nt_ <= 2:
            timeline = {"error": "--timeline needs at least 3 steps"}
        else:
            for i in range(2):                                   # back to the steady state after the getters above
                step(total_steps + i)
            for c_ in ctxs:
                c_.timeline(reset=True)
            for i in range(2, nt_):
                step(total_steps + i)
            torch.cuda.synchronize()
            timeline = timeline_report(ctxs, bool(args.post_on_rest), args.timeline_csv)
            timeline["steps"] = nt_ - 2
    n_valid = sum(1 for r in results if r.valid)
    mean_kps = float(np.mean([r.detected_left[0] for r in results]))
    mean_match = float(np.mean([r.stereo_matches[0] for r in results]))
    mean_track = float(np.mean([r.tracked_feats_from_last_frame for r in results]))
    ts_mean = {k: round(float(np.mean([r.track_stats[i] for r in results])), 1) for i, k in enumerate(TS_NAMES)}
    dist_info = dist_audit(allrec, batch.rec, world, rank, local_rank, dev, own_ms_per_step=1e3 * dt_own / args.steps, own_enqueue_ms_per_step=1e3 * t_enqueue / args.steps)
    assert allrec is not None and allrec.shape[0] == world * B
    if args.dump_records:      # tests/test_gpu_parity.py: every rank's own records and what the all-gather handed it
        np.savez(args.dump_records + ".rank%d.npz" % rank, local=batch.rec.cpu().numpy(), gathered=allrec.cpu().numpy(), rank=rank, world=world,
                 own=np.frombuffer(b"".join(bytes(r) for r in results), np.uint8).reshape(B, -1))

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
        # HBM traffic of that kernel: NOT measured in this run -- read from the committed PMC passes (tools/pmc_passes.py:
        # L2 memory-side read / write requests counted by size in separate rocprofv3 --pmc runs of this same command),
        # scaled from the profiled lane count to this run's; None when the profile does not cover this workload.
        # `traffic_source` names the file so that a reader can tell a carried-over constant from a live counter.
        traffic, traffic_source, valu_frac, valu_table = None, None, None, None
        kk = "k_" + ("hamming" if dom.startswith("hamming") else dom)
        prof = PMC_PROFILE.get(args.workload)
        if prof is None:
            traffic_source = "none: no committed PMC profile for workload %s" % args.workload
        else:
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", prof)))
                e = pm["kernels"].get(kk) or pm["kernels"].get(kk + "_f4")
                if pm.get("workload") != args.workload or e is None:
                    raise KeyError("profile covers workload %s and kernels %s" % (pm.get("workload"), sorted(pm["kernels"])))
                traffic = int((e.get("read_bytes", 0) + e.get("write_bytes", 0)) * Bc / pm["lanes"])
                valu_frac = e.get("valu_issue_frac")
                valu_table = {k: v["valu_issue_frac"] for k, v in pm["kernels"].items() if "valu_issue_frac" in v}
                traffic_source = "profiles/%s (separate rocprofv3 --pmc passes of this command, not this run)" % prof
            except Exception as e_:
                traffic_source = "MISSING: profiles/%s does not cover kernel %s of this run (%s: %s) -- traffic is null, not a number carried over from an older profile" % (prof, kk, type(e_).__name__, e_)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": int(abytes), "avg_launch_ms": round(per_kernel[dom]["ms_per_launch"], 4),
                    "streams_per_launch": Bc,
                    "valu_issue_frac": valu_frac, "valu_issue_frac_by_kernel": valu_table,
                    "valu_issue_note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of a launch with one context alone, from the same committed PMC passes as `traffic`: the share of the VALU issue slots the kernel fills -- the bound that binds here when it is near 1 and `frac` is not",
                    "note": ("dominant kernel of the detect stream; stage 3-5 kernels run on the overlap stream and their spans are time-shared, not exclusive" if pipelined else "single stream: every span is an exclusive duration")}
        # end-to-end algorithmic traffic of the whole path (SURVEY.md 8d formula), for the DESIGN.md table
        P = sum(a * b for a, b in lv) / float(W * H)
        pair_bytes = 2 * (3 * P - 1) * W * H + 2 * mean_kps * 60 + 16 * mean_match + 40 * mean_track * 12
        cpu_baseline, pose_rmse, parity_probe, host_fed = None, None, None, None
        legs_s = {"render": round(t_render, 1), "warmup_and_timed": round(time.perf_counter() - t_render_end, 1)}     # wall clock of every leg of this run
        t_leg = time.perf_counter()
        def leg_done(name):
            nonlocal t_leg
            legs_s[name] = round(time.perf_counter() - t_leg, 1); t_leg = time.perf_counter()
        # the state the TIMED region left behind, for the parity probe's final check: taken now, before any other leg touches the contexts
        timed_final = None
        if world == 1 and args.cpu_frames >= total_steps:
            from oracle import probe as PR          # checker only
            rec_cpu = allrec.cpu().numpy()
            timed_final = {}
            for g in sorted(set(k * Bc + l for k in range(NC) for l in (0, Bc // 2 - 1 if Bc > 1 else 0, Bc - 1))):
                ctx_, l_ = batch.lane(g)
                timed_final[g] = PR.digest_of(ctx_, l_, Result.from_buffer_copy(rec_cpu[g].tobytes()))
        # The long leg runs HERE, straight after the --steps region, on a GPU that is warm (round 5).  Rounds 3-4 ran it after the CPU legs and
        # it came out 3-4 % low; the A/B of round 5 (profiles/r05j_*): the same 100 steps 69.3 k pairs/s after ~20 s of GPU idle (an untimed
        # pass of 40 steps first notwithstanding), 71.5 k straight after the warm-up, 71.5 k over 1500 steps at a steady 2.39 GHz / 1.2 kW
        # (`tools/prof.sh clocks`) -- the deficit was the order of the legs, not the length of the measurement.
        long_run = None
        if long_steps > 0:
            long_run = run_long_leg("straight after the --steps region (GPU warm)")
            leg_done("long_run")
        if world == 1 and args.cpu_frames > 0:
            cpu_baseline, pose_rmse, parity_probe = cpu_baseline_and_probe(args, batch, frame_of, ptrs_by_step, worlds, T, p, cam, allrec, timed_final)
            leg_done("cpu_baseline_and_parity_probe")
        if world == 1 and args.host_fed_steps > 0:
            host_fed = host_fed_leg(args, batch, frame_of, dev)
            leg_done("host_fed")
        other_scene = None
        if world == 1 and args.relief_lanes > 0 and args.workload == "config2":
            batch.synchronize()
            try:
                other_scene = other_scene_leg(args, p, dev, local_rank, focal, baseline)
            except Exception as e:
                other_scene = {"error": str(e)}
            leg_done("other_scene")
        scene_cuts = None
        if world == 1 and args.cut_steps > 0:
            batch.synchronize()
            try:
                scene_cuts = scene_cut_leg(args, batch, frames, T, F, n_octaves if detect_fast_orb else 8, p, cam)
            except Exception as e:
                scene_cuts = {"error": str(e)}
            leg_done("scene_cuts")
        other_workloads = None
        if world == 1 and args.other_workloads and args.workload == "config2":
            batch.synchronize()
            other_workloads = other_workloads_leg(args)
            leg_done("other_workloads")
        single_stream = None
        if world == 1 and args.single_stream and args.workload == "config2":
            batch.synchronize()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                import single_stream_bench as SSB
                single_stream = SSB.measure(W, H, args.orb_nfeats, n=120, device=local_rank)
            except Exception as e:
                single_stream = {"error": str(e)}
            leg_done("single_stream")
        # the roofline kernel once more with the GPU to itself (one context, nothing on the overlap stream): in the pipelined
        # schedule its spans are time-shared with the stage 3-5 kernels of the other contexts
        try:
            if not args.exclusive:
                raise RuntimeError("skipped (--exclusive 0)")
            batch.reset()
            c0 = batch.ctxs[0]
            c0.kernel_times_select(dom)
            for i in range(3):
                c0.process_device(ptrs_by_step[i][:Bc], W, H, W)
            c0.wait(); c0.kernel_times_reset()
            for i in range(3, 9):
                c0.process_device(ptrs_by_step[i][:Bc], W, H, W)
            c0.wait()
            tot, calls = c0.kernel_times()[dom]
            ex_ms = tot / max(1, calls) / (7 if dom == "resize" else 1)
            roofline["exclusive"] = {"avg_launch_ms": round(ex_ms, 4), "achieved": round(abytes / (ex_ms * 1e-3) / 1e9, 2), "frac": round(abytes / (ex_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                     "note": "same kernel, same launch shape, %d launches with one context alone on the GPU (HIP events, after the timed region)" % calls}
        except Exception as e:
            roofline["exclusive"] = {"error": str(e)}
        line = {
            "metric": "stereo pairs/sec @%d×%d" % (W, H), "value": round(value, 2), "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f64", "data": "synthetic",
            "config": {"workload": "%s: %dx%d synthetic stereo streams, %s, orb_nfeats %d (~%d kps/image in octave 0 after NMS), BF match, BF track, robust GN; %d independent streams per GPU in %d contexts on separate HIP streams, one frame per stream per step"
                                   % (args.workload, W, H, "FAST+ORB on %d x1/2 octaves" % n_octaves if detect_fast_orb else "ORB x 8 levels", args.orb_nfeats, int(mean_kps), B, NC),
                       "lanes_per_gpu": B, "contexts_per_gpu": NC, "lanes_per_context": Bc, "schedule": args.schedule if NC > 1 else "single stream", "detect_ahead": bool(args.detect_ahead),
                       "sequence": "%d trajectories x %d frames per rank (scene '%s'), stream s = trajectory s %% %d from frame %d * (s // %d) on: %d consecutive DISTINCT frames per stream over warm-up + timed steps, none seen twice"
                                   % (T, F, args.scene, T, PHASE_STRIDE, T, total_steps),
                       "frames_per_trajectory": F, "distinct_frames_per_stream": min(total_steps, F), "render_s": round(t_render, 1),
                       "parallelism": "streams sharded across %d GPU(s), result all-gather per step" % world},
            "timed_region_s": round(dt, 4),
            "host_enqueue_ms_per_step": round(1e3 * t_enqueue / args.steps, 4),
            "fast_redo_rate": round(redo_pairs / float(max(1, args.steps * 2 * B * (n_octaves if detect_fast_orb else 8))), 6),
            "fast_redo_note": "(image, level) pairs of the timed steps whose speculative FAST threshold found too few corners and ran k_fast again at the caller's threshold (%d of %d); every frame of the run is new to its stream" % (redo_pairs, args.steps * 2 * B * (n_octaves if detect_fast_orb else 8)),
            "long_run": long_run,
            "scene_cuts": scene_cuts,
            "other_workloads": other_workloads,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "pose_rmse_vs_cpu": pose_rmse,
            "parity_probe": parity_probe,
            "host_fed": host_fed,
            "single_stream": single_stream,
            "path_hbm_frac": round(pair_bytes * value / 1e9 / HBM_PEAK_GBS, 5),
            "algorithmic_bytes_per_pair": int(pair_bytes),
            "valid_last_step": "%d/%d" % (n_valid, B),
            "mean_kps": round(mean_kps, 1), "mean_matches": round(mean_match, 1), "mean_tracked": round(mean_track, 1),
            "scene": args.scene,
            "track_funnel_mean": ts_mean,
            "track_funnel_note": "svo_result.track_stats of the last timed step, mean over this rank's streams: previous-frame pairings that pass the descriptor threshold on both sides -> survive the joint collision filter (S4:145-160) -> are inliers of the left / right F-matrix RANSAC (1.0 px, S4:202, 237; hyp_* = hypotheses visited before the 0.99-confidence stop) -> of both -> pass the L/R consistency check (S4:282) = tracked",
            "other_scene": other_scene,
            "dist": dist_info,
            "env": {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "SVO_TIMELINE", "SVO_DEBUG_MODE", "SVO_REST_PRIO", "HIP_FORCE_DEV_KERNARG") if os.environ.get(k) is not None},
            "timeline": timeline,
            "kernels_ms_per_context_step": {k: round(v["ms_per_step"], 4) for k, v in per_kernel_warm.items()},
            "legs_s": legs_s,
            "kernels_ms_note": "all kernels: HIP-event spans of the %d warm-up steps; roofline kernel: spans of the timed region" % args.warmup,
        }
        print(json.dumps(line))
    batch.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without torch.distributed.run around it: re-run this same command line as N ranks, one per GPU,
    under torch.distributed.run on 127.0.0.1 (what the driver's own N > 1 invocation does), and hand back its exit code.
    Refuses when the node shows fewer than N devices -- a one-GPU line labelled N GPUs is worse than no line.
    BENCH_FORCE_DEVICE (test hook: every rank on that device) lifts the device-count check."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not os.environ.get("BENCH_FORCE_DEVICE"):
        print("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to run (no silent fallback to fewer GPUs)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def dist_audit(allrec, local_rec, world, rank, local_rank, dev, own_ms_per_step=None, own_enqueue_ms_per_step=None):
    """What lets the first real N > 1 run be audited from its JSON line: the backend, how many distinct
    (host, device) pairs the ranks sit on, the RCCL version, whether every rank's gathered record table is rank 0's --
    with rank r's own records at slot r -- and every rank's OWN clock around the timed steps beside the MAX-reduced one the line's
    `value` uses (a slow rank, a missing rank or a host that cannot feed eight ranks' launches shows here without a re-run).
    None at N = 1."""
    if world == 1:
        return None
    import hashlib, socket
    import torch.distributed as dist
    own = local_rec.cpu().numpy().tobytes()
    table = allrec.cpu().numpy()
    per = table.shape[0] // world
    me = {"rank": rank, "host": socket.gethostname(), "device": int(local_rank), "device_name": torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu",
          "table": hashlib.blake2b(table.tobytes(), digest_size=16).hexdigest(),
          "own_slot_ok": table[rank * per:(rank + 1) * per].tobytes() == own,
          "ms_per_step": own_ms_per_step, "enqueue_ms_per_step": own_enqueue_ms_per_step}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return {"backend": dist.get_backend(), "world_size": world, "ranks_seen": len({(e["host"], e["device"]) for e in everyone}),
            "hosts": sorted({e["host"] for e in everyone}), "rccl_version": ver,
            "gathered_tables_equal": all(e["table"] == everyone[0]["table"] for e in everyone),
            "own_records_at_own_slot": all(e["own_slot_ok"] for e in everyone),
            "rank_ms_per_step": {"min": round(min(e["ms_per_step"] for e in everyone), 4), "max": round(max(e["ms_per_step"] for e in everyone), 4),
                                 "by_rank": [round(e["ms_per_step"], 4) for e in sorted(everyone, key=lambda e: e["rank"])]} if own_ms_per_step is not None else None,
            "rank_host_enqueue_ms_per_step": {"min": round(min(e["enqueue_ms_per_step"] for e in everyone), 4), "max": round(max(e["enqueue_ms_per_step"] for e in everyone), 4)} if own_enqueue_ms_per_step is not None else None,
            "devices": sorted({"%s:%d %s" % (e["host"], e["device"], e["device_name"]) for e in everyone}),
            "host_cores": os.cpu_count(),
            "note": "ranks_seen = distinct (hostname, device) pairs over the ranks (== world_size on a real multi-GPU run; 1 when a test puts every rank on one GPU)"}


def other_scene_leg(args, p, dev, device_index, focal, baseline):
    """The scene type the timed region did NOT use, on a small batch: where stage 4's candidates go and how far the poses are
    from the renderer's ground truth, beside the timed scene's figures.  Untimed; HIP poses against ground truth directly."""
    from stereo_vo_amd.synth import pose6_to_matrix, pose_error
    scene = "relief" if args.scene != "relief" else "planes"
    W, H, F, B = args.width, args.height, 6, args.relief_lanes
    worlds = [SyntheticStereoWorld(W, H, focal, baseline, seed=1000 + s, n_frames=F, device=dev, scene_seed=s % 4, scene=scene) for s in range(B)]
    frames = [[w.render(t) for t in range(F)] for w in worlds]
    torch.cuda.synchronize()
    ctx = hip.Context(n_lanes=B, max_w=W, max_h=H, max_kps=4096, device=device_index)
    ctx.set_params(p); ctx.set_camera(worlds[0].camera())
    acc = {k: [] for k in TS_NAMES}
    et, er, n_valid, n_frames = [], [], 0, 0
    for t in range(F):
        ctx.process_device([(frames[l][t][0].data_ptr(), frames[l][t][1].data_ptr()) for l in range(B)], W, H, W)
        res = ctx.results()
        if t == 0:
            continue
        for l, r in enumerate(res):
            n_frames += 1
            for i, k in enumerate(TS_NAMES):
                acc[k].append(r.track_stats[i])
            if r.valid:
                n_valid += 1
                e_r, e_t = pose_error(pose6_to_matrix(np.array(r.outPose)), worlds[l].gt_delta(t))
                et.append(e_t); er.append(e_r)
    ctx.close()
    return {"scene": scene, "streams": B, "frames_per_stream": F - 1, "valid": "%d/%d" % (n_valid, n_frames),
            "track_funnel_mean": {k: round(float(np.mean(v)), 1) for k, v in acc.items()},
            "pose_vs_ground_truth": {"translation_rmse_m": float(np.sqrt(np.mean(np.square(et)))) if et else None,
                                     "rotation_rmse_rad": float(np.sqrt(np.mean(np.square(er)))) if er else None, "frames": len(et)}}


def host_fed_leg(args, batch, frame_of, dev):
    """The reference's own contract is host images per call (process_new_image_pair.cpp:100-120).  Same batch, same
    schedule, but every step hands PAGE-LOCKED HOST frames to svo_process (SVO_FLAG_PINNED_IMAGES): each context uploads
    on its own copy stream into a two-slot device ring, so the upload of a step overlaps the kernels of the one before.
    PCIe-bound by construction (2 * W * H bytes per pair); never part of `value`.  The host ring holds the first six frames of
    every stream, played forwards and backwards (what is uploaded does not change what an upload costs)."""
    F, B, W, H = 6, batch.B, args.width, args.height
    try:
        host = torch.empty((F, B, 2, H, W), dtype=torch.uint8, pin_memory=True)
    except Exception as e:                                   # not enough lockable memory on this host
        return {"error": "pinned allocation failed: %s" % e}
    for l in range(B):
        for t in range(F):
            L_, R_ = frame_of(l, t)
            host[t, l, 0].copy_(L_); host[t, l, 1].copy_(R_)
    torch.cuda.synchronize()
    hptr = [[(host[t, l, 0].data_ptr(), host[t, l, 1].data_ptr()) for l in range(B)] for t in range(F)]

    def pingpong(i):
        return frame_schedule(i, F)
    for c_ in batch.ctxs:
        c_.kernel_times_select("fast")
    batch.reset()
    n_warm, K = 3, args.host_fed_steps
    for i in range(n_warm):
        batch.step(hptr[pingpong(i)], pinned_host=True)
    batch.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        batch.step(hptr[pingpong(n_warm + i)], pinned_host=True)
    batch.synchronize()
    dt = time.perf_counter() - t0
    res = batch.results()
    rate = B * K / dt
    return {"pairs_per_s": round(rate, 1), "ms_per_step": round(1e3 * dt / K, 3), "steps": K, "pcie_GBps": round(rate * 2 * W * H / 1e9, 2),
            "bytes_per_pair": 2 * W * H, "valid_last_step": "%d/%d" % (sum(1 for r in res if r.valid), B),
            "note": "page-locked host frames, one contiguous upload of %d pairs per context per step on the context's copy stream, two-slot device ring; the resident figure `value` excludes this copy" % batch.Bc}


def scene_cut_leg(args, batch, frames, T, F, levels, p=None, cam=None):
    """What the speculative FAST threshold costs when it is wrong.  Same batch, --cut-steps steps; every 20th step EVERY stream
    jumps to another trajectory (another scene: other textures, other corner statistics) at an arbitrary frame, and the estimators
    are reset there, as an application that detects the cut would (the reference itself keeps the pre-cut frame as `previous`
    for ever after voecBadTracking, P:86-95).  Timed like the main run, resets included."""
    B, K = batch.B, args.cut_steps

    def frame_at(l, step):
        seg = step // 20
        j = (l + seg * 3) % T                                  # a new trajectory after every cut (T >= 2; with one trajectory only the frame jumps)
        return frames[j][frame_schedule(PHASE_STRIDE * (l // T) + 53 * seg + step, F)]

    def ptrs(step):
        out = []
        for l in range(B):
            L_, R_ = frame_at(l, step)
            out.append((L_.data_ptr(), R_.data_ptr()))
        return out
    sched = [ptrs(i) for i in range(K)]
    batch.reset()
    for c_ in batch.ctxs:
        c_.kernel_times_select("fast")
        c_.redo_count(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cuts = 0
    for i in range(K):
        if i and i % 20 == 0:
            batch.reset(); cuts += 1
        batch.step(sched[i])
    batch.synchronize()
    dt = time.perf_counter() - t0
    redo = sum(c_.redo_count(reset=True) for c_ in batch.ctxs)
    res = batch.results()
    parity = None
    if p is not None and args.cpu_frames > 0:
        try:
            parity = final_state_parity(batch, p, cam, res, lambda g: [frame_at(g, i) for i in range(K)], K, cut_every=20)
        except Exception as e:
            parity = {"error": str(e)}
    return {"parity": parity, "pairs_per_s": round(B * K / dt, 1), "ms_per_step": round(1e3 * dt / K, 4), "steps": K, "cuts": cuts, "cut_every": 20,
            "fast_redo_pairs": redo, "fast_redo_rate": round(redo / float(K * 2 * B * levels), 6),
            "valid_last_step": "%d/%d" % (sum(1 for r in res if r.valid), B),
            "note": "every stream jumps to another trajectory / scene every 20 frames, estimators reset at the cut (the resets and their synchronisation are inside the timed span); redo rate over ALL (image, level) pairs of the leg, the first frames (no speculation yet) included"}


def final_state_parity(batch, p, cam, results, frames_for, n_steps, cut_every=0, max_lanes=28):
    """The state a timed leg LEFT BEHIND against the oracle: every stream the HIP path reports invalid at the leg's last
    step, plus three probe lanes of every context, is replayed from its first frame on one host thread each (a fresh estimator after
    every cut, as the leg reset its own), and the last frame's lists, flags and pose are compared.  frames_for(lane) -> the n_steps
    (left, right) device tensors that lane saw.  Bounded: at most max_lanes streams (the rest are counted, not replayed)."""
    import threading
    from oracle import oracle as O
    from oracle import probe as PR
    from stereo_vo_amd.abi import Result
    B, Bc, NC = batch.B, batch.Bc, batch.NC
    probes = sorted(set(k * Bc + l for k in range(NC) for l in (0, Bc // 2 - 1 if Bc > 1 else 0, Bc - 1)))
    invalid = [g for g in range(B) if not results[g].valid]
    want = list(dict.fromkeys(probes + invalid))[:max_lanes]
    rec_cpu = batch.rec.cpu().numpy()
    gpu = {}
    for g in want:
        ctx, l = batch.lane(g)
        gpu[g] = PR.digest_of(ctx, l, Result.from_buffer_copy(rec_cpu[g].tobytes()))
    ref, lock, todo = {}, threading.Lock(), list(want)

    def work():
        while True:
            with lock:
                if not todo:
                    return
                g = todo.pop()
            fr = [tuple(x.cpu().numpy() for x in f) for f in frames_for(g)]
            orc, d = O.Oracle(p), None
            for i in range(n_steps):
                if cut_every and i and i % cut_every == 0:
                    orc.close(); orc = O.Oracle(p)
                r = orc.process(fr[i][0], fr[i][1], cam)
                if i == n_steps - 1:
                    d = PR.digest_of(orc, 0, r)
            orc.close()
            with lock:
                ref[g] = d
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work) for _ in range(max(1, min(len(want), os.cpu_count() or 1)))]
    for t_ in ts: t_.start()
    for t_ in ts: t_.join()
    bad, max_t, max_r = [], 0.0, 0.0
    for g in want:
        lists, flags, et, er = PR.compare(gpu[g], ref[g])
        max_t, max_r = max(max_t, et), max(max_r, er)
        if not (lists and flags and et < 1e-3 and er < 1e-4):
            bad.append({"lane": g, "lists": bool(lists), "flags": bool(flags), "gpu_valid": gpu[g].valid, "cpu_valid": ref[g].valid, "gpu_counts": list(gpu[g].n), "cpu_counts": list(ref[g].n)})
    return {"lanes_replayed": len(want), "of_which_invalid_on_gpu": len([g for g in want if g in invalid]), "invalid_on_gpu_total": len(invalid), "frames_per_lane": n_steps,
            "all_invalid_streams_checked": len(invalid) <= len([g for g in want if g in invalid]),
            "final_lists_bit_exact_and_flags_equal": len(bad) == 0, "pose_max_err_m": max_t, "pose_max_err_rad": max_r, "mismatches": bad[:6],
            "host_seconds": round(time.perf_counter() - t0, 1),
            "note": "oracle replay of every stream the HIP path reports invalid at the leg's last step + 3 probe lanes per context, whole history of the leg; the final frame's keypoints / pairings / tracks / inliers, valid flag, error code and pose compared"}


def other_workloads_leg(args):
    """BASELINE.json configs[2] (KITTI shape) and configs[4] (2048x1536, FAST+ORB on 3 octaves) on this GPU, as short runs of this
    same script in sub-processes (their own contexts and memory), each with its own parity probe; reported inside the default line
    so that the driver's record holds a pairs/s figure for every single-GPU configuration."""
    import subprocess
    out = {}
    for wl, extra in (("config3", ["--lanes", "192", "--contexts", "2"]), ("config5", ["--lanes", "64", "--contexts", "2"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "40", "--warmup", "6", "--cpu-frames", "12", "--long-steps", "0", "--host-fed-steps", "0",
               "--single-stream", "0", "--relief-lanes", "0", "--cut-steps", "0", "--other-workloads", "0", "--exclusive", "1"] + extra
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
            out[wl] = {k: line.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "valid_last_step", "mean_kps", "mean_matches", "mean_tracked", "fast_redo_rate", "path_hbm_frac")}
            out[wl]["workload"] = line["config"]["workload"]
            out[wl]["roofline"] = {k: line["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms", "exclusive")}
            pp = line.get("parity_probe") or {}
            out[wl]["parity_probe"] = {k: pp.get(k) for k in ("lanes", "frames", "lists_bit_exact", "flags_equal", "pose_max_err_m", "pose_max_err_rad")}
        except Exception as e:
            out[wl] = {"error": str(e)}
    return out


def cpu_baseline_and_probe(args, batch, frame_of, ptrs_by_step, worlds, T, p, cam, allrec, timed_final=None):
    """N=1 only, after the timed region.  The oracle (CHECKER / BASELINE, never the thing measured) replays the first
    n frames of the run's own frame schedule for a few probe streams of every context:
      * leg (i): stream 0 alone on one host thread, timed          -> cpu_baseline.value (mirrors the single-threaded reference)
      * leg (ii): the other probe streams on min(S, cores) threads -> cpu_baseline.multi_thread
    Then the SAME batch object (same contexts, lane count, image size, streams and event schedule the timed region used)
    is reset and driven through those frames again, untimed, with a synchronisation after every step, and every probe
    lane's keypoints / descriptors / pairings / tracked pairs / inlier list are compared with the oracle's bit for bit,
    its pose within 1e-3 m / 1e-4 rad -> parity_probe, pose_rmse_vs_cpu."""
    from oracle import oracle as O      # checker / baseline only; never on the product path
    from oracle import probe as PR
    from stereo_vo_amd.abi import Result
    B, Bc, NC = batch.B, batch.Bc, batch.NC
    total = args.warmup + args.steps
    n = min(total, args.cpu_frames)
    order = list(range(n))              # stream g sees frame_of(g, i) at step i
    want = sorted(set(k * Bc + l for k in range(NC) for l in (0, Bc // 2 - 1 if Bc > 1 else 0, Bc - 1)))

    def host_frames(g, count):
        return [tuple(x.cpu().numpy() for x in frame_of(g, i)) for i in range(count)]
    host = {g: host_frames(g, n) for g in want}
    # the timing legs use a -march=native build of the same oracle source made on this host, if gcc is here; it must agree
    # with the portable checker build bit for bit before its digests are trusted
    native = False
    try:
        a, _ = PR.replay(p, cam, host[0], order[:2])
        O.lib(native=True)
        orc_n = O.Oracle(p, native=True)
        b = []
        for t in order[:2]:
            r = orc_n.process(host[0][t][0], host[0][t][1], cam); b.append(PR.digest_of(orc_n, 0, r))
        orc_n.close()
        native = all(x == y and np.array_equal(x.pose, y.pose) for x, y in zip(a, b))
    except Exception:
        native = False

    def replay_one(frs):
        orc = O.Oracle(p, native=native)
        out = []
        c0 = time.perf_counter()
        for t in order:
            r = orc.process(frs[t][0], frs[t][1], cam); out.append(PR.digest_of(orc, 0, r))
        d = time.perf_counter() - c0
        orc.close()
        return out, d
    ref = {}
    ref[0], dt1 = replay_one(host[0])
    others = [g for g in want if g != 0]
    threads = max(1, min(len(others), os.cpu_count() or 1))
    import threading
    lock, todo = threading.Lock(), list(others)
    def work():
        while True:
            with lock:
                if not todo: return
                g = todo.pop()
            d, _ = replay_one(host[g])
            with lock:
                ref[g] = d
    c0 = time.perf_counter()
    ts = [threading.Thread(target=work) for _ in range(threads)]
    for t_ in ts: t_.start()
    for t_ in ts: t_.join()
    dtm = time.perf_counter() - c0
    # leg (ii) proper: S independent streams on min(S, cores) threads, S capped so that the host copies of the
    # frames stay below ~1.5 GB and the leg within seconds: one oracle instance per stream, the first n_mt frames of its schedule
    cores = os.cpu_count() or 1
    mt_streams = list(range(min(B, cores, 96)))
    n_mt = min(n, 12)
    for g in mt_streams:
        if g not in host:
            host[g] = host_frames(g, n_mt)
    todo2, done2 = list(mt_streams), []
    def work2():
        while True:
            with lock:
                if not todo2: return
                g = todo2.pop()
            orc = O.Oracle(p, native=native)
            for t in order[:n_mt]:
                orc.process(host[g][t][0], host[g][t][1], cam)
            orc.close()
            with lock:
                done2.append(g)
    c0 = time.perf_counter()
    ts = [threading.Thread(target=work2) for _ in range(len(mt_streams))]
    for t_ in ts: t_.start()
    for t_ in ts: t_.join()
    dtm2 = time.perf_counter() - c0
    cpu_baseline = {"value": round(n / dt1, 3), "unit": "stereo pairs/s", "cores": 1, "kind": "port", "oracle_version": O.version(),
                    "sample": "first %d frames of stream 0's schedule (%dx%d, orb_nfeats=%d) on ONE thread of the C oracle (%s build); host has %d cores"
                              % (n, args.width, args.height, args.orb_nfeats, "-O3 -march=native, made on this host" if native else "-O3 -msse4.2 portable", cores),
                    "multi_thread": {"value": round(len(done2) * n_mt / dtm2, 3), "unit": "stereo pairs/s", "cores": len(mt_streams), "streams": len(done2),
                                     "sample": "%d independent oracle instances (one per stream, the first %d frames of each stream's schedule) on %d host threads of %d cores"
                                               % (len(done2), n_mt, len(mt_streams), cores),
                                     "probe_replay": {"value": round(len(others) * n / dtm, 3) if others else None, "streams": len(others), "threads": threads,
                                                      "note": "the parity probe's own reference replay (all %d frames of the other probe streams)" % n}}}
    # final state of the timed run itself against the oracle's state after the same history (only when the whole history was replayed)
    final_checked, final_ok = False, None
    if n == total and timed_final is not None:          # (digests taken by main() right after the timed region: other legs have run since)
        final_checked = True
        final_ok = True
        for g in want:
            lists, flags, et, er = PR.compare(timed_final[g], ref[g][-1])
            final_ok = final_ok and lists and flags and et < 1e-3 and er < 1e-4
    batch.reset()
    for c_ in batch.ctxs:
        c_.kernel_times_select(None)
    bad, max_t, max_r, se_t, se_r, n_pose, n_flag = [], 0.0, 0.0, 0.0, 0.0, 0, 0
    for i in range(n):
        batch.step(ptrs_by_step[i])
        batch.synchronize()
        rec_cpu = batch.rec.cpu().numpy()
        for g in want:
            ctx, l = batch.lane(g)
            res = Result.from_buffer_copy(rec_cpu[g].tobytes())
            dg = PR.digest_of(ctx, l, res)
            lists, flags, et, er = PR.compare(dg, ref[g][i])
            if not lists and len(bad) < 8:
                bad.append({"lane": g, "frame": i, "gpu_counts": list(dg.n), "cpu_counts": list(ref[g][i].n)})
            n_flag += int(not flags)
            if dg.valid and ref[g][i].valid:
                dpz = dg.pose - ref[g][i].pose
                se_t += float(np.sum(dpz[:3] ** 2)); se_r += float(np.sum(dpz[3:] ** 2)); n_pose += 1
                max_t, max_r = max(max_t, et), max(max_r, er)
    # accuracy against the synthetic ground truth (stream 0): translation / rotation error of the estimated frame-to-frame pose
    from stereo_vo_amd.synth import pose6_to_matrix, pose_error
    gt_t, gt_r = [], []
    for i in range(1, n):
        d = ref[0][i]
        if not d.valid:
            continue
        G = worlds[0].gt_delta(i)                     # stream 0 = trajectory 0 from its first frame
        er_, et_ = pose_error(pose6_to_matrix(d.pose), G)
        gt_t.append(et_); gt_r.append(er_)
    pose_rmse = {"translation_m": float(np.sqrt(se_t / n_pose)) if n_pose else None, "rotation_rad": float(np.sqrt(se_r / n_pose)) if n_pose else None,
                 "frames": n_pose, "valid_flag_mismatches": n_flag, "tolerance": "1e-3 m / 1e-4 rad per frame (tests/test_gpu_parity.py)",
                 "vs_ground_truth_stream0": {"translation_rmse_m": float(np.sqrt(np.mean(np.square(gt_t)))) if gt_t else None,
                                             "rotation_rmse_rad": float(np.sqrt(np.mean(np.square(gt_r)))) if gt_r else None, "frames": len(gt_t),
                                             "note": "oracle poses of stream 0 (identical to the HIP poses within the error above) against the renderer's ground-truth motion"}}
    parity_probe = {"lanes": want, "frames": n, "shape": "%d contexts x %d lanes x %dx%d, %s schedule -- the timed configuration, same contexts, reset and replayed untimed with a sync per step"
                                                         % (NC, Bc, args.width, args.height, args.schedule if NC > 1 else "single-stream"),
                    "lists_bit_exact": len(bad) == 0, "flags_equal": n_flag == 0, "pose_max_err_m": max_t, "pose_max_err_rad": max_r,
                    "first_mismatches": bad, "final_state_of_timed_run_checked": final_checked, "final_state_of_timed_run_ok": final_ok}
    return cpu_baseline, pose_rmse, parity_probe


if __name__ == "__main__":
    main()
