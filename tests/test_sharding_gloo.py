"""The N>1 path of bench.py on CPU: two gloo ranks shard the streams, all-gather their result records and reduce
the step time with MAX, exactly as the RCCL path does on GPUs (streams are independent, the only
collective is the gather of the fixed-size result records)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, lanes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from stereo_vo_amd.abi import Result
    seeds = bench.lane_seeds(rank, world, lanes)
    rec = torch.zeros((lanes, C.sizeof(Result)), dtype=torch.uint8)
    for i, s in enumerate(seeds):                       # fake result records: pose x = stream id
        r = Result(); r.outPose[0] = float(s); r.valid = 1; r.tracked_feats_from_last_frame = 100 + s
        rec[i] = torch.frombuffer(bytearray(bytes(r)), dtype=torch.uint8)
    allrec = bench.gather_records(rec, world)
    tmax = bench.reduce_max(0.5 + rank, torch.device("cpu"), world)
    try:
        audit = bench.dist_audit(allrec, rec, world, rank, rank, torch.device("cpu"), own_ms_per_step=2.5 + rank, own_enqueue_ms_per_step=1.0 + 0.5 * rank)   # pretend rank r sits on device r
    except Exception as e:                                                                 # (device name lookup needs a GPU)
        audit = {"error": repr(e)}
    q.put((rank, seeds, allrec.numpy().tobytes(), tmax, audit))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, lanes = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lanes, q)) for r in range(world)]
    for p in procs: p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    from stereo_vo_amd.abi import Result
    outs.sort()
    assert outs[0][1] == [0, 1, 2] and outs[1][1] == [3, 4, 5]          # disjoint, contiguous stream blocks
    assert outs[0][2] == outs[1][2]                                      # every rank sees the same gathered records
    sz = C.sizeof(Result)
    recs = [Result.from_buffer_copy(outs[0][2][i * sz:(i + 1) * sz]) for i in range(world * lanes)]
    assert [r.outPose[0] for r in recs] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]  # ordered by stream id = rank-major
    assert [r.tracked_feats_from_last_frame for r in recs] == [100, 101, 102, 103, 104, 105]
    assert outs[0][3] == outs[1][3] == 1.5                               # MAX over ranks of the step time
    # the audit block bench.py prints for N > 1: both ranks agree, two distinct (host, device) pairs, tables equal, own slot right
    a0, a1 = outs[0][4], outs[1][4]
    assert "error" not in a0, a0
    assert a0 == a1 and a0["backend"] == "gloo" and a0["world_size"] == 2 and a0["ranks_seen"] == 2
    assert a0["gathered_tables_equal"] and a0["own_records_at_own_slot"] and len(a0["hosts"]) == 1
    # every rank's own clock beside the MAX-reduced one: a slow or missing rank shows in the first N > 1 record
    assert a0["rank_ms_per_step"] == {"min": 2.5, "max": 3.5, "by_rank": [2.5, 3.5]} and a0["rank_host_enqueue_ms_per_step"] == {"min": 1.0, "max": 1.5}
    assert len(a0["devices"]) == 2 and a0["host_cores"] >= 1 and "rccl_version" in a0


def test_frame_schedule_is_ping_pong():
    import bench
    assert [bench.frame_schedule(i, 4) for i in range(9)] == [0, 1, 2, 3, 2, 1, 0, 1, 2]
    assert all(abs(bench.frame_schedule(i + 1, 6) - bench.frame_schedule(i, 6)) == 1 for i in range(40))
    assert bench.frame_schedule(5, 1) == 0


def test_plain_multi_gpu_invocation_launches_ranks_or_refuses(monkeypatch, capsys):
    """`python bench.py --gpus N` with no WORLD_SIZE becomes the launcher of N ranks: with fewer than N
    devices visible it refuses (exit code 2, no result line); otherwise the command it starts is torch.distributed.run with
    --nproc-per-node N on 127.0.0.1 followed by this script and the caller's own arguments."""
    import subprocess
    import sys
    import bench
    import torch
    monkeypatch.delenv("BENCH_FORCE_DEVICE", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.self_launch(2) == 2
    assert "refusing" in capsys.readouterr().err
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20"])
    assert bench.self_launch(8) == 0
    c = seen["cmd"]
    assert c[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and c[c.index("--nproc-per-node") + 1] == "8"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-4:] == ["--gpus", "8", "--steps", "20"] and c[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a rank count that disagrees with --gpus is an error, not a one-GPU line
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    try:
        bench.main()
        assert False, "bench.main() accepted WORLD_SIZE 1 with --gpus 2"
    except SystemExit as e:
        assert "WORLD_SIZE" in str(e)
