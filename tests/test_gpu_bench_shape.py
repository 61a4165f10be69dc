"""The configuration bench.py times, checked against the oracle AT ITS OWN SHAPE, and bench.py's N > 1 path run for real.

test_benchmark_shape_*: 3 contexts x 64 lanes x 1280x960 through stereo_vo_amd.pipeline.StreamBatch -- the object
bench.py's step() drives -- on the pipelined two-stream schedule; probe lanes {0, 31, 63} of every context against
independent oracle instances, every list bit for bit.  The XCD-chunked tile orders, fastdiv helpers and the matcher's
train splits all depend on lane count x image size; no smaller test exercises these values.

test_bench_two_ranks_*: `bench.py --gpus 2` under torch.distributed.run with both ranks on device 0 (this box has one
GPU): sharding, event ordering around the all-gather, MAX-time reduction and the gather itself on REAL result records.
RCCL refuses two ranks on one device, so the collective backend is gloo there; the 8-GPU driver run uses nccl.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from stereo_vo_amd import hip
from stereo_vo_amd.abi import north_star_params, Result
from stereo_vo_amd.synth import SyntheticStereoWorld

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("schedule", ["pipelined"])
def test_benchmark_shape_three_contexts_of_64_lanes_match_oracle(schedule):
    import torch
    from oracle import probe as PR
    from stereo_vo_amd.pipeline import StreamBatch
    import bench
    W, H, B, NC, F, STEPS = 1280, 960, 192, 3, 3, 4
    dev = torch.device("cuda", 0)
    worlds = [SyntheticStereoWorld(W, H, 800.0, 0.12, seed=s, n_frames=F, device=dev, scene_seed=s % 4) for s in bench.lane_seeds(0, 1, B)]
    frames = [[w.render(t) for t in range(F)] for w in worlds]
    cam = worlds[0].camera()
    torch.cuda.synchronize()
    p = north_star_params(hip.default_params(), orb_nfeats=2000)
    batch = StreamBatch(p, cam, W, H, B, NC, schedule=schedule)
    Bc = batch.Bc
    probe = sorted(k * Bc + l for k in range(NC) for l in (0, 31, 63))
    order = [bench.frame_schedule(i, F) for i in range(STEPS)]           # 0 1 2 1: the ping-pong bench.py plays
    host = {g: [(frames[g][t][0].cpu().numpy(), frames[g][t][1].cpu().numpy()) for t in range(F)] for g in probe}
    ref, _ = PR.replay_many(p, cam, host, order, threads=min(8, os.cpu_count() or 1))
    ptrs_at = [[(frames[l][t][0].data_ptr(), frames[l][t][1].data_ptr()) for l in range(B)] for t in range(F)]
    # enqueue the steps back to back first (the schedule as timed: no host synchronisation between steps) ...
    for i in range(STEPS):
        batch.step(ptrs_at[order[i]])
    batch.synchronize()
    rec = batch.rec.cpu().numpy()
    for g in probe:
        ctx, l = batch.lane(g)
        res = Result.from_buffer_copy(rec[g].tobytes())
        lists, flags, et, er = PR.compare(PR.digest_of(ctx, l, res), ref[g][-1])
        assert lists and flags and et < 1e-3 and er < 1e-4, ("back-to-back", g, et, er)
        assert ctx.status_word(l) == 0
    # ... then frame by frame after a reset, every probe lane at every step
    batch.reset()
    n_valid = 0
    for i in range(STEPS):
        batch.step(ptrs_at[order[i]])
        batch.synchronize()
        rec = batch.rec.cpu().numpy()
        for g in probe:
            ctx, l = batch.lane(g)
            res = Result.from_buffer_copy(rec[g].tobytes())
            dg = PR.digest_of(ctx, l, res)
            lists, flags, et, er = PR.compare(dg, ref[g][i])
            assert lists, "lists differ: stream %d step %d gpu %s cpu %s" % (g, i, list(dg.n), list(ref[g][i].n))
            assert flags and et < 1e-3 and er < 1e-4, (g, i, et, er)
            if i:
                assert np.allclose(dg.residual[dg.residual < 1e300], ref[g][i].residual[ref[g][i].residual < 1e300], rtol=1e-6, atol=1e-9)
            n_valid += int(dg.valid)
            if i:
                assert dg.n[0] > 1500 and dg.n[3] > 100, dg.n
    assert n_valid >= (STEPS - 1) * len(probe) - 1
    assert sum(1 for r in batch.results() if r.valid) >= B - 4
    batch.close()


def _run_bench_2ranks(backend, tmp, extra_env=None):
    env = dict(os.environ)
    env.update({"BENCH_FORCE_DEVICE": "0", "BENCH_DIST_BACKEND": backend, "MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env or {})
    dump = os.path.join(tmp, "rec_" + backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--lanes", "8", "--contexts", "2", "--frames", "3",
           "--width", "640", "--height", "480", "--orb-nfeats", "500", "--cpu-frames", "0", "--dump-records", dump]
    try:
        pr = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        return None, dump, "timeout"
    return pr, dump, pr.stdout[-2000:] + pr.stderr[-2000:]


def test_bench_two_ranks_on_one_gpu_gather_real_records(tmp_path):
    import json
    pr, dump, log = _run_bench_2ranks("gloo", str(tmp_path))
    assert pr is not None and pr.returncode == 0, log
    line = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["lanes_per_gpu"] == 8
    # the audit block of an N > 1 run: both ranks were forced onto the box's one GPU, so ONE (host, device) pair is seen --
    # on a real multi-GPU node ranks_seen must equal world_size
    d = line["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["ranks_seen"] == 1 and d["gathered_tables_equal"] and d["own_records_at_own_slot"]
    assert set(line["track_funnel_mean"]) == {"threshold", "collision", "inliers_left", "inliers_right", "hyp_left", "hyp_right", "both_masks", "tracked"}
    r0, r1 = np.load(dump + ".rank0.npz"), np.load(dump + ".rank1.npz")
    # every rank holds the same gathered array = rank 0's records followed by rank 1's (stream s -> rank s // lanes)
    assert r0["gathered"].tobytes() == r1["gathered"].tobytes()
    assert r0["gathered"].shape[0] == 16
    assert r0["gathered"][:8].tobytes() == r0["local"].tobytes() and r0["gathered"][8:].tobytes() == r1["local"].tobytes()
    # ... and those records are what svo_get_results returns on the rank that computed them
    assert r0["own"].tobytes() == r0["local"].tobytes() and r1["own"].tobytes() == r1["local"].tobytes()
    # different streams (seeds) on the two ranks: their poses differ, and they are real (valid) results
    recs = [Result.from_buffer_copy(r0["gathered"][i].tobytes()) for i in range(16)]
    assert sum(r.valid for r in recs) >= 14
    assert list(recs[0].outPose) != list(recs[8].outPose)


def test_bench_launches_its_own_ranks_when_started_plainly(tmp_path):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run around it: the process re-runs itself as two
    ranks and the line says n_gpus 2 -- it used to pass an assert and print a one-GPU line."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"BENCH_FORCE_DEVICE": "0", "BENCH_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--lanes", "8", "--contexts", "2", "--frames", "3",
           "--width", "640", "--height", "480", "--orb-nfeats", "500", "--cpu-frames", "0"]
    pr = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-2000:]
    line = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2 and line["dist"]["gathered_tables_equal"]
    # the fields that make the first real N > 1 record self-explaining: ranks seen, RCCL version, and each rank's own
    # ms_per_step beside the MAX-reduced one of the line
    d = line["dist"]
    assert d["ranks_seen"] == 1 and "rccl_version" in d and len(d["rank_ms_per_step"]["by_rank"]) == 2
    assert 0 < d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"] <= line["ms_per_step"] * 1.001 + 1e-3
    assert d["rank_host_enqueue_ms_per_step"]["max"] > 0 and d["host_cores"] >= 1
    # without the test hook a one-GPU box must REFUSE --gpus 2 (exit code 2, nothing that looks like a result line)
    env.pop("BENCH_FORCE_DEVICE")
    pr = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.device_count() < 2:
        assert pr.returncode == 2 and "refusing" in pr.stderr and not [l for l in pr.stdout.splitlines() if l.startswith("{")]


def test_host_fed_path_64_lanes_matches_oracle():
    """Frames handed over as HOST images (the reference's own contract, P:100-120) at the batched shape: page-locked
    contiguous frames (one upload per step), page-locked non-contiguous frames (per-image 2-D uploads) and pageable
    frames (staged through the context's pinned slot) all give the oracle's lists; five steps enqueued back to back
    without waiting walk the two-slot device ring more than twice."""
    import torch
    from oracle import probe as PR
    from stereo_vo_amd.pipeline import StreamBatch
    import bench
    W, H, B, F = 640, 480, 64, 3
    dev = torch.device("cuda", 0)
    worlds = [SyntheticStereoWorld(W, H, 400.0, 0.12, seed=300 + s, n_frames=F, device=dev, scene_seed=s % 4) for s in range(B)]
    cam = worlds[0].camera()
    p = north_star_params(hip.default_params(), orb_nfeats=500)
    host = torch.empty((F, B, 2, H, W), dtype=torch.uint8, pin_memory=True)
    padded = torch.empty((F, B, 2, H, W + 64), dtype=torch.uint8, pin_memory=True)          # row stride != width: 2-D copies
    for l, w in enumerate(worlds):
        for t in range(F):
            a, b = w.render(t)
            host[t, l, 0].copy_(a); host[t, l, 1].copy_(b)
    torch.cuda.synchronize()
    padded[..., :W] = host
    order = [bench.frame_schedule(i, F) for i in range(5)]
    probe = [0, 21, 42, 63]
    ref, _ = PR.replay_many(p, cam, {g: [(host[t, g, 0].numpy(), host[t, g, 1].numpy()) for t in range(F)] for g in probe}, order, threads=4)
    for mode in ("pinned", "pinned-2d", "pageable"):
        batch = StreamBatch(p, cam, W, H, B, 1, max_kps=1024, max_cand=1 << 15)
        ctx = batch.ctxs[0]
        for i, t in enumerate(order):
            if mode == "pinned":
                batch.step([(host[t, l, 0].data_ptr(), host[t, l, 1].data_ptr()) for l in range(B)], pinned_host=True)
            elif mode == "pinned-2d":
                batch.step([(padded[t, l, 0].data_ptr(), padded[t, l, 1].data_ptr()) for l in range(B)], stride=W + 64, pinned_host=True)
            else:
                ctx.process_host([(host[t, l, 0].numpy(), host[t, l, 1].numpy()) for l in range(B)])
            if i in (1, 4):                       # steps 0-1 and 2-4 run back to back; check after each burst
                batch.synchronize()
                res = ctx.results()
                for g in probe:
                    lists, flags, et, er = PR.compare(PR.digest_of(ctx, g, res[g]), ref[g][i])
                    assert lists and flags and et < 1e-3 and er < 1e-4, (mode, i, g)
        ctx.wait_upload()
        batch.close()


def test_frame_parallel_within_one_stream_equals_the_sequential_run(golden_dir):
    """SURVEY.md 8e "Within ONE stream": frames dealt round-robin to two (and three) contexts, hand-over of the previous
    frame's lists + inherited estimator members between them.  Every frame's lists, inlier lists and POSES equal the
    sequential context's and the oracle's -- including across a failed frame (blank images -> voecBadTracking), after
    which the recovery rule (P:86-89) makes the OLDER frame the previous one on whichever context comes next -- with
    match-ID bookkeeping on."""
    import torch
    from stereo_vo_amd.pipeline import FrameParallelStream
    from oracle import probe as PR
    W, H = 640, 480
    w = SyntheticStereoWorld(W, H, 400.0, 0.12, seed=41, n_frames=7, device=torch.device("cuda"))
    cam = w.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=500)
    p.vo_use_matches_ids = 1
    frames = [w.render(t) for t in range(7)]
    blank = (torch.full((H, W), 128, dtype=torch.uint8, device="cuda"), torch.full((H, W), 128, dtype=torch.uint8, device="cuda"))
    seq_frames = [frames[0], frames[1], frames[2], blank, frames[3], frames[4], blank, blank, frames[5], frames[6]]
    torch.cuda.synchronize()
    orc = O_().Oracle(p)
    seq = hip.Context(n_lanes=1, max_w=W, max_h=H, max_kps=1024, max_cand=1 << 15)
    seq.set_params(p); seq.set_camera(cam)
    ref = []
    for (L, R) in seq_frames:
        seq.process_device([(L.data_ptr(), R.data_ptr())], W, H, W)
        r = seq.result(0)
        ro = orc.process(L.cpu().numpy(), R.cpu().numpy(), cam)
        d = PR.digest_of(seq, 0, r)
        assert d == PR.digest_of(orc, 0, ro)
        ref.append((d, list(r.outPose), seq.match_ids(0).tobytes(), r.tracked_feats_from_last_KF))
    assert [d.error_code for d, _, _, _ in ref].count(5) >= 3 and sum(d.valid for d, _, _, _ in ref) >= 5
    for G in (2, 3):
        fp = FrameParallelStream(p, cam, W, H, lanes=1, contexts=G, max_kps=1024, max_cand=1 << 15)
        # (a) pushed back to back, checked at the end ...
        for (L, R) in seq_frames:
            c = fp.push([(L.data_ptr(), R.data_ptr())])
        fp.synchronize()
        r = c.result(0)
        assert PR.digest_of(c, 0, r) == ref[-1][0] and list(r.outPose) == ref[-1][1]
        fp.close()
        # (b) ... and frame by frame
        fp = FrameParallelStream(p, cam, W, H, lanes=1, contexts=G, max_kps=1024, max_cand=1 << 15)
        for i, (L, R) in enumerate(seq_frames):
            c = fp.push([(L.data_ptr(), R.data_ptr())])
            fp.synchronize()
            r = c.result(0)
            d = PR.digest_of(c, 0, r)
            assert d == ref[i][0], (G, i, d.n, ref[i][0].n, d.error_code, ref[i][0].error_code)
            assert list(r.outPose) == ref[i][1], (G, i)                     # the warm start travels with the record: identical poses
            assert c.match_ids(0).tobytes() == ref[i][2] and r.tracked_feats_from_last_KF == ref[i][3], (G, i)
        fp.close()
    seq.close()


def O_():
    from oracle import oracle
    return oracle


def test_frame_parallel_across_two_ranks():
    """The same hand-over between RANKS (torch.distributed send / recv of the record): two processes on this box's one
    GPU, gloo (the nccl path needs one GPU per rank); rank 0 checks every frame against a sequential context."""
    import json
    env = dict(os.environ)
    env.update({"FP_FORCE_DEVICE": "0", "FP_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "tools", "frame_parallel_ranks.py"), "--frames", "9"]
    pr = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]
    line = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
    assert line["identical_to_sequential"] is True and line["frame_parallel_ranks"] == 2 and line["valid_frames"] >= 7


def test_every_schedule_of_the_batch_gives_the_same_records():
    """svo_batch's knobs only move launches between streams: where the NMS / describe block runs (post_mode 0..3, incl. the split at
    the per-level selection), how many stage 3-5 and detect streams there are, which side has the priority, pipelined or free --
    the result records of every stream must be the same bytes under all of them (and valid)."""
    import torch
    from stereo_vo_amd.pipeline import StreamBatch
    W, H, B, NC, F, STEPS = 640, 480, 6, 3, 3, 5
    dev = torch.device("cuda", 0)
    worlds = [SyntheticStereoWorld(W, H, 400.0, 0.12, seed=500 + s, n_frames=F, device=dev) for s in range(B)]
    frames = [[w.render(t) for t in range(F)] for w in worlds]
    cam = worlds[0].camera()
    torch.cuda.synchronize()
    p = north_star_params(hip.default_params(), orb_nfeats=600)
    ptrs_at = [[(frames[l][t][0].data_ptr(), frames[l][t][1].data_ptr()) for l in range(B)] for t in range(F)]
    variants = [dict(),                                                                   # the default: post on one stream per context, detect high
                dict(post_on_rest=False, rest_streams=1, det_priority="low"),             # rounds 1-2
                dict(post_on_rest="own"), dict(post_on_rest="select"),
                dict(rest_streams=2, det_streams=2), dict(schedule="free"),
                dict(detect_ahead=False), dict(post_on_rest="select", detect_ahead=False)]      # rounds 1-3: a context's detector waits for its whole previous frame
    ref = None
    for kw in variants:
        batch = StreamBatch(p, cam, W, H, B, NC, max_kps=1024, max_cand=1 << 15, **kw)
        for i in range(STEPS):
            batch.step(ptrs_at[[0, 1, 2, 1, 0][i]])
        batch.synchronize()
        rec = batch.rec.cpu().numpy().copy()
        res = [Result.from_buffer_copy(rec[g].tobytes()) for g in range(B)]
        assert all(r.valid and r.status == 0 for r in res), kw
        if ref is None:
            ref = rec
        else:
            assert rec.tobytes() == ref.tobytes(), kw
        batch.close()


def test_capacity_bits_of_a_detector_running_ahead_reach_the_right_record(monkeypatch):
    """SVO_FLAG_DETECT_AHEAD: the detector of frame t + 1 runs while stages 3-5 of frame t are still writing frame t's record, so the
    capacity bits it raises are staged and folded in by its own post call.  A candidate list far too small for the image overflows
    on every frame (debug mode 12: no speculative FAST threshold, which would otherwise learn to fit the list after a few frames):
    bit 1 must be in every stream's record -- and in the status word -- with and without the look-ahead (which corners an
    overflowing list keeps is a race, so only the bits are compared)."""
    import torch
    monkeypatch.setenv("SVO_DEBUG_MODE", "12")
    from stereo_vo_amd.pipeline import StreamBatch
    W, H, B, NC, F, STEPS = 640, 480, 4, 2, 3, 4
    dev = torch.device("cuda", 0)
    worlds = [SyntheticStereoWorld(W, H, 400.0, 0.12, seed=700 + s, n_frames=F, device=dev) for s in range(B)]
    frames = [[w.render(t) for t in range(F)] for w in worlds]
    cam = worlds[0].camera()
    torch.cuda.synchronize()
    p = north_star_params(hip.default_params(), orb_nfeats=600)
    ptrs_at = [[(frames[l][t][0].data_ptr(), frames[l][t][1].data_ptr()) for l in range(B)] for t in range(F)]
    for ahead in (True, False):
        batch = StreamBatch(p, cam, W, H, B, NC, max_kps=1024, max_cand=1 << 6, detect_ahead=ahead)
        for i in range(STEPS):
            batch.step(ptrs_at[[0, 1, 2, 1][i]])
            batch.synchronize()
            rec = batch.rec.cpu().numpy().copy()
            res = [Result.from_buffer_copy(rec[g].tobytes()) for g in range(B)]
            assert all(r.status & 1 for r in res), (ahead, i, [r.status for r in res])
            assert [r.status for r in res] == [r2.status for r2 in batch.results()]
            assert all(batch.lane(g)[0].status_word(batch.lane(g)[1]) & 1 for g in range(B))
        batch.close()
    # and a list that fits raises nothing
    monkeypatch.delenv("SVO_DEBUG_MODE")
    batch = StreamBatch(p, cam, W, H, B, NC, max_kps=1024, max_cand=1 << 15)
    for i in range(2):
        batch.step(ptrs_at[i])
    batch.synchronize()
    assert all(r.status == 0 for r in batch.results())
    batch.close()


def test_a_step_that_fails_half_way_leaves_the_batch_usable():
    """a svo_batch_step whose SECOND context refuses its frames (one stride differs) has already enqueued the first context:
    some events of the step are recorded, others are not.  The step reports the error, the batch drains the device and restarts its event
    chain, the held events are dropped -- and the following steps give, lane by lane, what an oracle fed the frames that lane really
    processed gives (the lanes of context 0 saw the failed step's frame, those of context 1 did not)."""
    import torch
    from oracle import oracle as OR
    from oracle import probe as PR
    from stereo_vo_amd.pipeline import StreamBatch
    from stereo_vo_amd.synth import SyntheticStereoWorld
    W, H, B, NC = 640, 480, 4, 2
    dev = torch.device("cuda", 0)
    world = SyntheticStereoWorld(W, H, 400.0, 0.12, seed=5, n_frames=6, device=dev)
    frames = [world.render(t) for t in range(6)]
    cam = world.camera()
    p = north_star_params(hip.default_params(), orb_nfeats=500)
    batch = StreamBatch(p, cam, W, H, B, NC, device=0, max_kps=1024)
    ptrs = lambda t: [(frames[t][0].data_ptr(), frames[t][1].data_ptr())] * B     # noqa: E731  (every lane plays the same sequence)
    seen = [[] for _ in range(B)]
    for t in (0, 1):
        batch.step(ptrs(t))
        for l in range(B): seen[l].append(t)
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
    batch.hold_for(ev)                                   # a held event that the failing step must not leak into the next one
    bad = batch.prepare(ptrs(2))
    bad[B - 1].right.stride = W + 64                     # last lane of context 1: svo_process refuses mixed strides (SVO_ERR_ARG)
    with pytest.raises(hip.SvoError):
        batch.step_prepared(bad)
    for l in range(B // NC): seen[l].append(2)           # context 0 went through
    for t in (3, 4, 5):
        batch.step(ptrs(t))
        for l in range(B): seen[l].append(t)
    batch.synchronize()
    rec = batch.rec.cpu().numpy()
    host = [tuple(x.cpu().numpy() for x in f) for f in frames]
    for l in range(B):
        orc = OR.Oracle(p)
        for t in seen[l]:
            r = orc.process(host[t][0], host[t][1], cam)
        ctx, ll = batch.lane(l)
        res = Result.from_buffer_copy(rec[l].tobytes())
        lists, flags, et, er = PR.compare(PR.digest_of(ctx, ll, res), PR.digest_of(orc, 0, r))
        assert lists and flags and et < 1e-3 and er < 1e-4, (l, seen[l])
        orc.close()
    batch.close()
