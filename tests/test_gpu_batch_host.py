"""The benchmarked shape -- 3 contexts x 64 lanes of 1280x960 streams in the pipelined two-stream schedule -- driven from
the C++ HOST binary tools/batch_streams (include/svo_batch.h: the scheduler lives in libsvo_hip.so, behind the C-ABI), and
checked against the CPU oracle on the probe streams: every step's result record (flags, counts, stage-4 pass-through counters,
pose) and the final frame's keypoints, descriptors, pairings and tracked pairs.  The reference's caller is such a C++ loop
(demo-stereo-odometry/demo-main.cpp:210-220)."""
import ctypes as C
import json
import os
import struct
import subprocess
import sys
import threading

import numpy as np
import pytest

from stereo_vo_amd import hip
from stereo_vo_amd.abi import Result, StereoCamera, north_star_params, keypoint_dtype, dmatch_dtype, index_pair_dtype

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ping_pong(step, F):
    period = 2 * (F - 1); k = step % period
    return k if k < F else period - k


def read_dump(path):
    b = open(path, "rb").read()
    assert b[:8] == b"SVOBDMP1"
    n_probe, n_steps, B, rsz, n_files, F, W, H = struct.unpack_from("<8i", b, 8)
    assert rsz == C.sizeof(Result)
    off = 40
    probe = list(struct.unpack_from("<%di" % n_probe, b, off)); off += 4 * n_probe
    recs = [[None] * n_probe for _ in range(n_steps)]
    for i in range(n_steps):
        for q in range(n_probe):
            recs[i][q] = Result.from_buffer_copy(b[off:off + rsz]); off += rsz
    lists = []
    for q in range(n_probe):
        L = {}
        for side in ("l", "r"):
            (n,) = struct.unpack_from("<i", b, off); off += 4
            L["k" + side] = np.frombuffer(b, keypoint_dtype, n, off).copy(); off += n * keypoint_dtype.itemsize
            L["d" + side] = np.frombuffer(b, np.uint8, n * 32, off).reshape(n, 32).copy(); off += n * 32
        (n,) = struct.unpack_from("<i", b, off); off += 4
        L["m"] = np.frombuffer(b, dmatch_dtype, n, off).copy(); off += n * dmatch_dtype.itemsize
        (n,) = struct.unpack_from("<i", b, off); off += 4
        L["t"] = np.frombuffer(b, index_pair_dtype, n, off).copy(); off += n * index_pair_dtype.itemsize
        lists.append(L)
    assert off == len(b)
    return dict(probe=probe, n_steps=n_steps, B=B, n_files=n_files, F=F, W=W, H=H, recs=recs, lists=lists)


def test_cpp_host_drives_the_benchmarked_shape_and_matches_the_oracle(tmp_path):
    import torch
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_sequence import write_sequence
    exe = os.path.join(ROOT, "tools", "batch_streams")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    W, H, F, NF = 1280, 960, 6, 4
    files, worlds = [], []
    for s in range(NF):
        p = str(tmp_path / ("s%d.svoseq" % s))
        worlds.append(write_sequence(p, W, H, 800.0, 0.12, seed=s, n_frames=F, device="cuda"))
        files.append(p)
    torch.cuda.synchronize()
    dump = str(tmp_path / "probe")
    steps, warm = 6, 2
    out = subprocess.check_output([exe, "--contexts", "2", "--lanes", "96", "--steps", str(steps), "--warmup", str(warm), "--nfeats", "2000", "--dump", dump] + files, timeout=900)
    line = json.loads(out.decode().strip().splitlines()[-1])
    assert line["streams"] == 192 and line["contexts_per_gpu"] == 2 and line["lanes_per_context"] == 96 and line["pairs_per_s"] > 1000
    assert line["valid_last_step"] == "192/192"
    d = read_dump(dump + ".bin")
    assert d["probe"] == [0, 47, 95, 96, 143, 191] and d["n_steps"] == steps + warm and d["B"] == 192
    frames = [[tuple(x.cpu().numpy() for x in w.render(t)) for t in range(F)] for w in worlds]
    cam = worlds[0].camera()
    p = north_star_params(O.default_params(), orb_nfeats=2000)
    ref = {}

    def replay(q, g):
        orc = O.Oracle(p)
        rs = []
        for i in range(d["n_steps"]):
            L, R = frames[g % NF][ping_pong(i + g // NF, F)]
            rs.append(orc.process(L, R, cam))
        ref[q] = (rs, orc.keypoints(0, 0), orc.keypoints(0, 1), orc.matches(0), orc.tracked())
        orc.close()

    ts = [threading.Thread(target=replay, args=(q, g)) for q, g in enumerate(d["probe"])]
    for t in ts: t.start()
    for t in ts: t.join()
    n_valid = 0
    for q, g in enumerate(d["probe"]):
        rs, kl, kr, m, tr = ref[q]
        for i in range(d["n_steps"]):
            a, b = d["recs"][i][q], rs[i]
            tag = (g, i)
            assert (a.valid, a.error_code) == (b.valid, b.error_code), tag
            assert (a.detected_left[0], a.detected_right[0], a.stereo_matches[0], a.tracked_feats_from_last_frame, a.n_outliers, a.n_residual) == \
                   (b.detected_left[0], b.detected_right[0], b.stereo_matches[0], b.tracked_feats_from_last_frame, b.n_outliers, b.n_residual), tag
            assert list(a.track_stats) == list(b.track_stats), tag
            if b.valid:
                n_valid += 1
                dp = np.abs(np.array(a.outPose) - np.array(b.outPose))
                assert dp[:3].max() < 1e-3 and dp[3:].max() < 1e-4, (tag, dp)         # 1e-3 m / 1e-4 rad (BASELINE.json north_star)
        L = d["lists"][q]
        assert L["kl"].tobytes() == kl[0].tobytes() and (L["dl"] == kl[1]).all(), g
        assert L["kr"].tobytes() == kr[0].tobytes() and (L["dr"] == kr[1]).all(), g
        assert L["m"].tobytes() == m.tobytes() and L["t"].tobytes() == tr.tobytes(), g
    assert n_valid >= len(d["probe"]) * (d["n_steps"] - 2)


def test_cpp_host_one_rank_rccl_gather_of_a_batch(tmp_path):
    """--gather rccl with one rank: the batch's records go through the in-place ncclAllGather on the side stream with the
    event ordering of a real multi-GPU run (svo_batch_wait_on_stream / svo_batch_hold_for_event); ncclCommCount is reported."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_sequence import write_sequence
    exe = os.path.join(ROOT, "tools", "batch_streams")
    p = str(tmp_path / "a.svoseq")
    write_sequence(p, 640, 480, 400.0, 0.12, seed=5, n_frames=4, device="cuda")
    out = subprocess.check_output([exe, "--contexts", "2", "--lanes", "4", "--steps", "5", "--warmup", "2", "--nfeats", "500", "--gather", "rccl", p], timeout=600)
    line = json.loads(out.decode().strip().splitlines()[-1])
    assert line["gather"] == "rccl" and line["rccl_comm_count"] == 1 and line["gathered_tables_equal"] is True and line["valid_last_step"] == "8/8"
