"""Reader / writer of the estimator state file, the on-disk format either side of the hot path.

Layout = what CStereoOdometryEstimator::saveStateToFile writes (libstereo-odometry/src/common.cpp:475-543 with the
helpers m_dump_keypoints_to_stream :88-133 and m_dump_matches_to_stream :138-163), little-endian, size_t = 8 bytes:

    npyr                                                         u64
    PRE left, PRE right, PRE pairings, CUR left, CUR right, CUR pairings, where
      keypoints = count u64, then per keypoint  x y response size angle (f32)  octave class_id (i32),
                  then rows cols type (i32) and rows*cols descriptor bytes
      pairings  = count u64, id_count u64, then per pairing  [id u64 if count == id_count]
                  queryIdx trainIdx (i32) distance (f32) imgIdx (i32)
    m_reset u8, m_lastID, m_num_tracked_pairs_from_last_kf, m_num_tracked_pairs_from_last_frame,
    m_last_match_ID, m_kf_max_match_ID                            u64 each

The reference's loadStateFromFile (common.cpp:261-350) expects one more u64 ("v_s", :342-343) before m_last_match_ID
that its own saver never writes; this module, like svo_load_state, reads what the saver writes.
"""
import struct

import numpy as np

from .abi import keypoint_dtype, dmatch_dtype

_KP_FILE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def _dump_keypoints(kps, desc):
    kps = np.asarray(kps, keypoint_dtype)
    n = len(kps)
    rec = np.zeros(n, _KP_FILE)
    for f in _KP_FILE.names:
        rec[f] = kps[f]
    desc = np.ascontiguousarray(desc, np.uint8).reshape(n, 32) if n else np.zeros((0, 32), np.uint8)
    return struct.pack("<Q", n) + rec.tobytes() + struct.pack("<iii", n, 32 if n else 0, 0) + desc.tobytes()


def _load_keypoints(buf, off):
    (n,) = struct.unpack_from("<Q", buf, off); off += 8
    rec = np.frombuffer(buf, _KP_FILE, n, off); off += n * _KP_FILE.itemsize
    rows, cols, _typ = struct.unpack_from("<iii", buf, off); off += 12
    kps = np.zeros(n, keypoint_dtype)
    for f in _KP_FILE.names:
        kps[f] = rec[f]
    desc = np.frombuffer(buf, np.uint8, rows * cols, off).reshape(rows, cols).copy(); off += rows * cols
    return kps, desc, off


def _dump_matches(m, ids):
    m = np.asarray(m, dmatch_dtype)
    ids = np.asarray(ids, np.int64)
    out = [struct.pack("<QQ", len(m), len(ids))]
    for i in range(len(m)):
        if len(m) == len(ids):
            out.append(struct.pack("<Q", int(ids[i])))
        out.append(struct.pack("<iifi", int(m["queryIdx"][i]), int(m["trainIdx"][i]), float(m["distance"][i]), int(m["imgIdx"][i])))
    return b"".join(out)


def _load_matches(buf, off):
    n, ni = struct.unpack_from("<QQ", buf, off); off += 16
    m = np.zeros(n, dmatch_dtype); ids = np.zeros(ni, np.int64)
    for i in range(n):
        if n == ni:
            (ids[i],) = struct.unpack_from("<Q", buf, off); off += 8
        q, t, d, im = struct.unpack_from("<iifi", buf, off); off += 16
        m[i] = (q, t, im, d)
    return m, ids, off


def write_state(path, pre, cur, reset=False, num_tracked_last_kf=0, num_tracked_last_frame=0, last_match_id=0, kf_max_match_id=0, npyr=1):
    """pre / cur: dicts with left=(kps, desc), right=(kps, desc), matches, ids."""
    blob = [struct.pack("<Q", npyr)]
    for d in (pre, cur):
        blob += [_dump_keypoints(*d["left"]), _dump_keypoints(*d["right"]), _dump_matches(d["matches"], d["ids"])]
    blob.append(struct.pack("<BQQQQQ", 1 if reset else 0, 0, num_tracked_last_kf, num_tracked_last_frame, last_match_id, kf_max_match_id))
    with open(path, "wb") as f:
        f.write(b"".join(blob))


def read_state(path):
    buf = open(path, "rb").read()
    (npyr,) = struct.unpack_from("<Q", buf, 0); off = 8
    out = {"npyr": npyr}
    for name in ("pre", "cur"):
        lk, ld, off = _load_keypoints(buf, off)
        rk, rd, off = _load_keypoints(buf, off)
        m, ids, off = _load_matches(buf, off)
        out[name] = {"left": (lk, ld), "right": (rk, rd), "matches": m, "ids": ids}
    r, last_id, nkf, nfr, lm, kfm = struct.unpack_from("<BQQQQQ", buf, off); off += 41
    assert off == len(buf), "trailing bytes in the state file"
    out.update(reset=bool(r), last_id=last_id, num_tracked_last_kf=nkf, num_tracked_last_frame=nfr, last_match_id=lm, kf_max_match_id=kfm)
    return out
