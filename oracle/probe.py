"""Replay of stereo streams on the CPU oracle, frame by frame, with a digest of every list the path produces.

TEST INFRASTRUCTURE (checker): used by tests/ and by bench.py's cpu_baseline / parity_probe leg to check probe lanes of
the batched HIP run against independent oracle instances, and to time the oracle on the box's host cores.  Nothing on
the product path imports this.
"""
import hashlib
import threading
import time

import numpy as np

from . import oracle as O


def _h(*arrays):
    m = hashlib.blake2b(digest_size=16)
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


class FrameDigest:
    """What one frame of one stream left behind: list digests, counts, result record fields."""
    __slots__ = ("kps", "matches", "tracked", "outliers", "valid", "error_code", "pose", "n", "residual")

    def __eq__(self, o):
        return all(getattr(self, k) == getattr(o, k) for k in ("kps", "matches", "tracked", "outliers", "valid", "error_code", "n"))


def digest_of(src, lane, res):
    """src: an Oracle (lane ignored) or a hip.Context; res: the Result of the frame."""
    is_orc = isinstance(src, O.Oracle)
    d = FrameDigest()
    kl = src.keypoints(0, 0) if is_orc else src.keypoints(lane, 0, 0)
    kr = src.keypoints(0, 1) if is_orc else src.keypoints(lane, 0, 1)
    d.kps = _h(kl[0], kl[1], kr[0], kr[1])
    d.matches = _h(src.matches(0) if is_orc else src.matches(lane))
    d.tracked = _h(src.tracked() if is_orc else src.tracked(lane))
    d.valid, d.error_code = int(res.valid), int(res.error_code)
    d.outliers = _h(src.outliers() if is_orc else src.outliers(lane)) if res.valid else ""
    d.pose = np.array(res.outPose, np.float64)
    d.n = (len(kl[0]), len(kr[0]), int(res.stereo_matches[0]), int(res.tracked_feats_from_last_frame), int(res.n_outliers), int(res.n_residual)) + tuple(res.track_stats)
    d.residual = (src.residuals() if is_orc else src.residuals(lane)) if res.valid else None
    return d


def replay(params, cam, frames, order):
    """One oracle instance through frames[order[0]], frames[order[1]], ...; frames[t] = (left, right) uint8 arrays.
    Returns (digests, seconds)."""
    orc = O.Oracle(params)
    out = []
    t0 = time.perf_counter()
    for t in order:
        r = orc.process(frames[t][0], frames[t][1], cam)
        out.append(digest_of(orc, 0, r))
    dt = time.perf_counter() - t0
    orc.close()
    return out, dt


def replay_many(params, cam, frames_by_stream, order, threads):
    """Independent oracle instances, one per stream, on `threads` host threads (ctypes releases the GIL; the C oracle
    has no global mutable state).  Returns ({stream: digests}, wall seconds)."""
    keys = list(frames_by_stream)
    out, lock, nxt = {}, threading.Lock(), [0]

    def work():
        while True:
            with lock:
                if nxt[0] >= len(keys):
                    return
                k = keys[nxt[0]]; nxt[0] += 1
            d, _ = replay(params, cam, frames_by_stream[k], order)
            with lock:
                out[k] = d

    t0 = time.perf_counter()
    ts = [threading.Thread(target=work) for _ in range(max(1, min(threads, len(keys))))]
    for t in ts: t.start()
    for t in ts: t.join()
    return out, time.perf_counter() - t0


def compare(gpu, cpu, tol_m=1e-3, tol_rad=1e-4):
    """(lists_bit_exact, flags_equal, translation err, rotation err) of one frame."""
    lists = gpu == cpu
    flags = (gpu.valid, gpu.error_code) == (cpu.valid, cpu.error_code)
    et = er = 0.0
    if gpu.valid and cpu.valid:
        dp = np.abs(gpu.pose - cpu.pose)
        et, er = float(dp[:3].max()), float(dp[3:].max())
    return lists, flags, et, er
