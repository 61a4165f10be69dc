"""The exchange steps of the multi-GPU path through the RCCL companion library (include/svo_rccl.h) and the C++ host
tools/multi_gpu_streams (BASELINE.json configs[3]: one estimator per GPU on its own host thread, one all-gather of the
result records per frame).  The box these tests run on has ONE GPU, and RCCL refuses two ranks on one device, so what
runs here is: RCCL with a one-rank group (the real ncclAllGather / communicator set-up, degenerate exchange), and the
two-thread host of the C++ harness with both ranks on GPU 0 exchanging the same records through host memory."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from stereo_vo_amd import hip
from stereo_vo_amd.abi import Result, north_star_params
from stereo_vo_amd.synth import SyntheticStereoWorld

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sequences(tmp_path, seeds):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_sequence import write_sequence
    paths = []
    for s in seeds:
        p = str(tmp_path / ("seq%d.svoseq" % s))
        write_sequence(p, 640, 480, 400.0, 0.12, seed=s, n_frames=8)
        paths.append(p)
    return paths


def _demo_trace(tmp_path, seq, tag):
    out = str(tmp_path / ("demo_%s.txt" % tag))
    subprocess.check_call([os.path.join(ROOT, "tools", "demo_stereo_odometry"), seq, out, "500"], stdout=subprocess.DEVNULL)
    return open(out).read()


def test_cpp_host_one_rank_rccl_gather_equals_the_demo(tmp_path):
    """one rank: the estimator's record goes through ncclAllGather and back; the trajectory written from the GATHERED
    records is the demo's, character for character"""
    exe = os.path.join(ROOT, "tools", "multi_gpu_streams")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    (seq,) = _sequences(tmp_path, [11])
    want = _demo_trace(tmp_path, seq, "a")
    out = subprocess.check_output([exe, "--gpus", "1", "--gather", "rccl", "--nfeats", "500", "--out", str(tmp_path / "mg"), seq], timeout=600)
    line = json.loads(out.decode().strip().splitlines()[-1])
    assert line["ranks"] == 1 and line["gather"] == "rccl" and line["frames"] == 8 and line["rccl_comm_count"] == 1 and line["tables_equal"]
    assert open(str(tmp_path / "mg_stream0.txt")).read() == want


def test_cpp_host_two_threads_two_estimators_equal_two_demos(tmp_path):
    """two host threads, one estimator each (both on this box's one GPU), records exchanged every frame: each stream's
    trajectory equals its own single-estimator run -- estimators share no state (libstereo-odometry.h:732-831)"""
    exe = os.path.join(ROOT, "tools", "multi_gpu_streams")
    a, b = _sequences(tmp_path, [11, 12])
    want = [_demo_trace(tmp_path, a, "a"), _demo_trace(tmp_path, b, "b")]
    assert want[0] != want[1]
    subprocess.check_output([exe, "--gpus", "2", "--same-device", "--gather", "host", "--nfeats", "500", "--out", str(tmp_path / "mg"), a, b], timeout=600)
    for k in range(2):
        assert open(str(tmp_path / ("mg_stream%d.txt" % k))).read() == want[k]


@pytest.mark.parametrize("kind", ["local", "per_process"])
def test_group_allgather_results_one_rank(kind):
    """svo_group_allgather_results on a 4-lane context: the table on the device equals svo_get_results"""
    from stereo_vo_amd import rccl
    lanes = 4
    worlds = [SyntheticStereoWorld(640, 480, 400.0, 0.12, seed=40 + i, n_frames=3) for i in range(lanes)]
    ctx = hip.Context(n_lanes=lanes, max_w=640, max_h=480, max_kps=1024, max_cand=1 << 15)
    ctx.set_params(north_star_params(hip.default_params(), orb_nfeats=400))
    for i, w in enumerate(worlds):
        ctx.set_camera(w.camera(), lane=i)
    grp = rccl.Group.local([0]) if kind == "local" else rccl.Group.rank(rccl.unique_id(), 1, 0, 0)
    assert grp.size == 1
    nbytes = lanes * C.sizeof(Result)
    table = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
    side = torch.cuda.Stream()
    for t in range(3):
        ctx.process_host([tuple(x.numpy() for x in w.render(t)) for w in worlds])
        # once on the context's own stream, once carried by another stream (ordered after the frame by an event)
        grp.allgather_results(ctx, table.data_ptr(), nbytes, rank=0, stream=None if t % 2 == 0 else side.cuda_stream)
        ctx.wait(); side.synchronize(); torch.cuda.synchronize()
        got = rccl.records_from_bytes(table.cpu().numpy().tobytes())
        for i in range(lanes):
            assert bytes(got[i]) == bytes(ctx.result(i)), (t, i)
    with pytest.raises(hip.SvoError):
        grp.allgather_results(ctx, table.data_ptr(), nbytes - 8, rank=0)         # not a whole number of records
    with pytest.raises(hip.SvoError):
        grp.send_frame(0, table.data_ptr(), 16, None, rank=0)                    # a rank does not send to itself
    grp.close(); ctx.close()


def test_contexts_run_on_their_own_device_whatever_is_current():
    """every entry point selects the context's GPU itself (one host thread may own estimators on several GPUs)"""
    import threading
    world = SyntheticStereoWorld(640, 480, 400.0, 0.12, seed=5, n_frames=2)
    p = north_star_params(hip.default_params(), orb_nfeats=300)
    res = {}

    def run(tag):
        ctx = hip.Context(n_lanes=1, max_w=640, max_h=480, max_kps=1024, max_cand=1 << 15)
        ctx.set_params(p); ctx.set_camera(world.camera())
        for t in range(2):
            ctx.process_host([tuple(x.numpy() for x in world.render(t))])
        res[tag] = (bytes(ctx.result(0)), ctx.keypoints(0, 0, 0)[0].tobytes())
        ctx.close()
    run("main")
    th = threading.Thread(target=run, args=("thread",)); th.start(); th.join()
    assert res["main"] == res["thread"]
