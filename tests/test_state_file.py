"""CPU tests of the state-file format (saveStateToFile / loadStateFromFile, libstereo-odometry/src/common.cpp:88-350,
475-543): byte layout against a hand-assembled known answer, and write -> read round trips including empty lists."""
import struct

import numpy as np

from stereo_vo_amd.abi import keypoint_dtype, dmatch_dtype
from stereo_vo_amd.state_file import read_state, write_state


def _lists(rng, n, m):
    k = np.zeros(n, keypoint_dtype)
    k["x"] = rng.uniform(0, 640, n).astype(np.float32); k["y"] = rng.uniform(0, 480, n).astype(np.float32)
    k["size"] = 31; k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["response"] = rng.normal(size=n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n); k["class_id"] = -1
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    mm = np.zeros(m, dmatch_dtype)
    mm["queryIdx"] = rng.integers(0, max(1, n), m); mm["trainIdx"] = rng.integers(0, max(1, n), m); mm["distance"] = rng.integers(0, 60, m)
    return k, d, mm


def test_state_file_byte_layout(tmp_path):
    k = np.zeros(1, keypoint_dtype)
    k[0] = (1.5, 2.5, 31.0, 90.0, 0.25, 3, -1)              # x y size angle response octave class_id (cv::KeyPoint order)
    d = np.arange(32, dtype=np.uint8)[None]
    m = np.zeros(1, dmatch_dtype); m[0] = (0, 0, 0, 7.0)    # queryIdx trainIdx imgIdx distance (cv::DMatch order)
    empty = {"left": (k[:0], d[:0]), "right": (k[:0], d[:0]), "matches": m[:0], "ids": []}
    cur = {"left": (k, d), "right": (k[:0], d[:0]), "matches": m, "ids": [42]}
    p = tmp_path / "state.bin"
    write_state(p, empty, cur, reset=True, num_tracked_last_kf=5, num_tracked_last_frame=6, last_match_id=43, kf_max_match_id=41)
    kp_empty = struct.pack("<Q", 0) + struct.pack("<iii", 0, 0, 0)
    want = struct.pack("<Q", 1)
    want += kp_empty + kp_empty + struct.pack("<QQ", 0, 0)                                            # PRE
    want += struct.pack("<Q", 1) + struct.pack("<fffffii", 1.5, 2.5, 0.25, 31.0, 90.0, 3, -1) + struct.pack("<iii", 1, 32, 0) + bytes(range(32))
    want += kp_empty + struct.pack("<QQ", 1, 1) + struct.pack("<Q", 42) + struct.pack("<iifi", 0, 0, 7.0, 0)   # CUR right, pairings
    want += struct.pack("<B", 1) + struct.pack("<QQQQQ", 0, 5, 6, 43, 41)
    assert p.read_bytes() == want


def test_state_file_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    kl, dl, mm = _lists(rng, 57, 31)
    kr, dr, _ = _lists(rng, 49, 0)
    pre = {"left": (kl, dl), "right": (kr, dr), "matches": mm, "ids": np.arange(100, 131)}
    cur = {"left": (kr, dr), "right": (kl, dl), "matches": mm[:7], "ids": []}              # id count != pairing count: no ids stored
    p = tmp_path / "s.bin"
    write_state(p, pre, cur, last_match_id=131, kf_max_match_id=99, num_tracked_last_kf=3, num_tracked_last_frame=20)
    s = read_state(p)
    for name, d in (("pre", pre), ("cur", cur)):
        for side in ("left", "right"):
            assert s[name][side][0].tobytes() == np.asarray(d[side][0]).tobytes() and (s[name][side][1] == d[side][1]).all()
        assert s[name]["matches"].tobytes() == np.asarray(d["matches"]).tobytes()
        assert list(s[name]["ids"]) == list(d["ids"])
    assert (s["reset"], s["last_match_id"], s["kf_max_match_id"], s["num_tracked_last_kf"], s["num_tracked_last_frame"]) == (False, 131, 99, 3, 20)
