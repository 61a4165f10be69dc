"""ctypes binding of the CPU oracle (oracle/libsvo_oracle.so).

TEST INFRASTRUCTURE: import only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from stereo_vo_amd.abi import (Params, Result, StereoCamera, keypoint_dtype, dmatch_dtype, index_pair_dtype)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = {}

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def _ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def build(native=False):
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["native"] if native else []))


def lib(native=False):
    """native=True: the same source built -march=native ON THIS HOST (oracle/_native/, never shipped): the timing leg of
    bench.py's cpu_baseline.  The portable build is the checker everywhere else; -ffp-contract=off and no fast-math in
    both, so they agree bit for bit (bench.py checks that before it uses the native one)."""
    if native not in _LIB:
        path = os.path.join(_HERE, "_native", "libsvo_oracle.so") if native else os.path.join(_HERE, "libsvo_oracle.so")
        if not native and os.environ.get("SVO_ORACLE_SO"):      # tools/oracle_sanitize.sh: the sanitizer build of the same source
            path = os.environ["SVO_ORACLE_SO"]
        elif native or not os.path.exists(path):
            build(native)
        L = C.CDLL(path)
        L.svo_oracle_create.restype = C.c_void_p
        L.svo_oracle_sad8.restype = C.c_uint32
        for name in ("svo_oracle_destroy", "svo_oracle_set_params", "svo_oracle_get_params", "svo_oracle_set_fast_threshold",
                     "svo_oracle_set_orb_threshold", "svo_oracle_reset_ids", "svo_oracle_set_this_frame_as_kf"):
            getattr(L, name).restype = None
        _LIB[native] = L
    return _LIB[native]


def version() -> int:
    """SVO_ORACLE_VERSION: the version of the frozen definitions (oracle/svo_oracle.h); stored in the golden files and the bench line"""
    return int(lib().svo_oracle_version())


def default_params() -> Params:
    p = Params()
    lib().svo_oracle_params_defaults(C.byref(p))
    return p


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


class Oracle:
    """One rso::CStereoOdometryEstimator worth of state, CPU side."""

    def __init__(self, params: Params = None, native=False):
        self.L = lib(native)
        self.h = C.c_void_p(self.L.svo_oracle_create())
        if params is not None:
            self.set_params(params)

    def close(self):
        if self.h:
            self.L.svo_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, p: Params):
        self.L.svo_oracle_set_params(self.h, C.byref(p))

    def set_fast_threshold(self, v):
        self.L.svo_oracle_set_fast_threshold(self.h, int(v))

    def set_orb_threshold(self, v):
        self.L.svo_oracle_set_orb_threshold(self.h, int(v))

    def fast_threshold(self):
        return self.L.svo_oracle_get_fast_threshold(self.h)

    def orb_threshold(self):
        return self.L.svo_oracle_get_orb_threshold(self.h)

    def process(self, left, right, cam: StereoCamera, repeat=False) -> Result:
        left, right = _img(left), _img(right)
        assert left.shape == right.shape
        h, w = left.shape
        res = Result()
        rc = self.L.svo_oracle_process(self.h, _ptr(left, u8p), _ptr(right, u8p), w, h, w, C.byref(cam), int(repeat), C.byref(res))
        if rc != 0:
            raise RuntimeError("svo_oracle_process failed: %d" % rc)
        return res

    def keypoints(self, which=0, side=0, octave=0):
        n = self.L.svo_oracle_get_keypoints(self.h, which, side, octave, None, None, 0)
        k = np.zeros(n, keypoint_dtype)
        d = np.zeros((n, 32), np.uint8)
        if n:
            self.L.svo_oracle_get_keypoints(self.h, which, side, octave, k.ctypes.data_as(C.c_void_p), _ptr(d, u8p), n)
        return k, d

    def row_index(self, which=0, side=0, octave=0):
        n = self.L.svo_oracle_get_row_index(self.h, which, side, octave, None, 0)
        a = np.zeros(n, np.int64)
        if n:
            self.L.svo_oracle_get_row_index(self.h, which, side, octave, _ptr(a, i64p), n)
        return a

    def matches(self, which=0, octave=0):
        n = self.L.svo_oracle_get_matches(self.h, which, octave, None, 0)
        m = np.zeros(n, dmatch_dtype)
        if n:
            self.L.svo_oracle_get_matches(self.h, which, octave, m.ctypes.data_as(C.c_void_p), n)
        return m

    def matches_row_index(self, which=0, octave=0):
        n = self.L.svo_oracle_get_matches_row_index(self.h, which, octave, None, 0)
        a = np.zeros(n, np.int64)
        if n:
            self.L.svo_oracle_get_matches_row_index(self.h, which, octave, _ptr(a, i64p), n)
        return a

    def match_ids(self, which=0, octave=0):
        n = self.L.svo_oracle_get_match_ids(self.h, which, octave, None, 0)
        a = np.zeros(n, np.int64)
        if n:
            self.L.svo_oracle_get_match_ids(self.h, which, octave, _ptr(a, i64p), n)
        return a

    def tracked(self, octave=0):
        n = self.L.svo_oracle_get_tracked(self.h, octave, None, 0)
        t = np.zeros(n, index_pair_dtype)
        if n:
            self.L.svo_oracle_get_tracked(self.h, octave, t.ctypes.data_as(C.c_void_p), n)
        return t

    def residuals(self):
        n = self.L.svo_oracle_get_residuals(self.h, None, 0)
        r = np.zeros(n, np.float64)
        if n:
            self.L.svo_oracle_get_residuals(self.h, _ptr(r, f64p), n)
        return r

    def outliers(self):
        n = self.L.svo_oracle_get_outliers(self.h, None, 0)
        r = np.zeros(n, np.int32)
        if n:
            self.L.svo_oracle_get_outliers(self.h, _ptr(r, i32p), n)
        return r

    def change_in_pose(self, tracked, pre_m, cur_m, pre_l, pre_r, cur_l, cur_r, cam, init6=None):
        n = len(tracked)
        res = Result()
        residual = np.zeros(max(n, 1), np.float64)
        outl = np.zeros(max(n, 1), np.int32)
        init = None if init6 is None else np.ascontiguousarray(init6, np.float64)
        vp = C.c_void_p
        valid = self.L.svo_oracle_change_in_pose(
            self.h, tracked.ctypes.data_as(vp), n, pre_m.ctypes.data_as(vp), cur_m.ctypes.data_as(vp),
            pre_l.ctypes.data_as(vp), pre_r.ctypes.data_as(vp), cur_l.ctypes.data_as(vp), cur_r.ctypes.data_as(vp),
            C.byref(cam), _ptr(init, f64p), C.byref(res), _ptr(residual, f64p), _ptr(outl, i32p))
        return bool(valid), res, residual[:res.n_residual], outl[:res.n_outliers]


# ---- stage-level functions -------------------------------------------------------------------------

def orb_detect(img, nfeatures, nlevels=8, fast_th=20):
    img = _img(img)
    h, w = img.shape
    cap = nfeatures + 64
    k = np.zeros(cap, keypoint_dtype)
    d = np.zeros((cap, 32), np.uint8)
    n = lib().svo_oracle_orb_detect(_ptr(img, u8p), w, h, w, nfeatures, nlevels, fast_th, k.ctypes.data_as(C.c_void_p), _ptr(d, u8p), cap)
    return k[:n].copy(), d[:n].copy()


def fast_orb_detect(img, fast_th=20):
    img = _img(img)
    h, w = img.shape
    cap = w * h // 9 + 16
    k = np.zeros(cap, keypoint_dtype)
    d = np.zeros((cap, 32), np.uint8)
    n = lib().svo_oracle_fast_orb_detect(_ptr(img, u8p), w, h, w, fast_th, k.ctypes.data_as(C.c_void_p), _ptr(d, u8p), cap)
    return k[:n].copy(), d[:n].copy()


def fast_score_map(img, th):
    img = _img(img)
    h, w = img.shape
    s = np.zeros((h, w), np.uint8)
    lib().svo_oracle_fast_score_map(_ptr(img, u8p), w, h, w, th, _ptr(s, u8p))
    return s


def orb_angle(img, x, y):
    """orientation in degrees of the position (x, y) (intensity centroid over the radius-15 disc, the oracle's atan2)"""
    img = _img(img)
    L = lib()
    L.svo_oracle_orb_angle.restype = C.c_float
    return float(L.svo_oracle_orb_angle(_ptr(img, u8p), img.shape[1], int(x), int(y), None))


def steered_brief(blurred, x, y, angle_deg):
    """the 256 steered tests alone (cv::ORB's computeOrbDescriptor) on an ALREADY blurred image; angle in degrees (float32)"""
    img = _img(blurred)
    d = np.zeros(32, np.uint8)
    f = lib().svo_oracle_steered_brief
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    f(img.ctypes.data, img.shape[1], int(x), int(y), float(np.float32(angle_deg)), d.ctypes.data)
    return d


def sincosf(x):
    """the frozen single-precision (sin, cos) of the steering"""
    f = lib().svo_oracle_sincosf
    f.restype = None
    f.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    s, c = C.c_float(), C.c_float()
    f(float(np.float32(x)), C.byref(s), C.byref(c))
    return float(s.value), float(c.value)


def harris(img, x, y):
    """cv::ORB's HarrisResponses at one position (7 x 7 block, k = 0.04)"""
    img = _img(img)
    f = lib().svo_oracle_harris
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    return float(f(img.ctypes.data, img.shape[1], int(x), int(y)))


def level_quota(nfeatures, nlevels):
    q = (C.c_int * nlevels)()
    lib().svo_oracle_level_quota(int(nfeatures), int(nlevels), q)
    return list(q)


def pyramid_sizes(w, h, nlevels):
    lw = (C.c_int * nlevels)()
    lh = (C.c_int * nlevels)()
    sc = (C.c_float * nlevels)()
    lib().svo_oracle_pyramid_sizes(w, h, nlevels, lw, lh, sc)
    return list(lw), list(lh), [float(np.float32(x)) for x in sc]


def resize(img, dw, dh):
    img = _img(img)
    h, w = img.shape
    out = np.zeros((dh, dw), np.uint8)
    lib().svo_oracle_resize(_ptr(img, u8p), w, h, w, _ptr(out, u8p), dw, dh)
    return out


def resize_table(src, dst):
    """cv::resize's INTER_LINEAR tables of one axis (oracle v7): (first tap index, weight pair a0 | a1 << 16) per destination position"""
    idx = np.zeros(dst, np.int32); w01 = np.zeros(dst, np.int32)
    i32p = C.POINTER(C.c_int32)
    lib().svo_oracle_resize_table(int(src), int(dst), _ptr(idx, i32p), _ptr(w01, i32p))
    return idx, w01


def half_smooth(img):
    img = _img(img)
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), np.uint8)
    lib().svo_oracle_half_smooth(_ptr(img, u8p), w, h, w, _ptr(out, u8p))
    return out


def nms_copy(kps, min_distance, img_w, img_h, num_out):
    n = len(kps)
    order = np.zeros(max(n, 1), np.int32)
    c = lib().svo_oracle_nms_copy(kps.ctypes.data_as(C.c_void_p), n, min_distance, img_w, img_h, num_out, _ptr(order, i32p))
    return order[:c].copy()


def anms_copy(kps, num_out, min_radius_th=0.0):
    """m_adaptive_non_max_sup (S2:141-215): indices of the kept keypoints in radius-descending order."""
    kps = np.ascontiguousarray(kps, dtype=keypoint_dtype)
    order = np.zeros(max(1, len(kps)), np.int32)
    f = lib().svo_oracle_anms_copy
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
    n = f(kps.ctypes.data, len(kps), int(num_out), float(min_radius_th), order.ctypes.data)
    return order[:n].copy()


def nms_mask(kps, min_distance, img_w, img_h, num_out):
    n = len(kps)
    m = np.zeros(max(n, 1), np.uint8)
    lib().svo_oracle_nms_mask(kps.ctypes.data_as(C.c_void_p), n, min_distance, img_w, img_h, num_out, _ptr(m, u8p))
    return m[:n].copy()


def row_sort_index(kps, img_h):
    n = len(kps)
    order = np.zeros(max(n, 1), np.int32)
    idx = np.zeros(img_h, np.int64)
    lib().svo_oracle_row_sort_index(kps.ctypes.data_as(C.c_void_p), n, img_h, _ptr(order, i32p), _ptr(idx, i64p))
    return order[:n].copy(), idx


def hamming_bf(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros(max(len(q), 1), np.int32)
    dist = np.zeros(max(len(q), 1), np.int32)
    lib().svo_oracle_hamming_bf(_ptr(q, u8p), len(q), _ptr(t, u8p), len(t), _ptr(idx, i32p), _ptr(dist, i32p))
    return idx[:len(q)].copy(), dist[:len(q)].copy()


def match_lr(params, orb_th, kl, dl, idxl, kr, dr, idxr, img_w, img_h):
    cap = max(len(kl), 1)
    out = np.zeros(cap, dmatch_dtype)
    ri = np.zeros(img_h + 1, np.int64)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    vp = C.c_void_p
    m = lib().svo_oracle_match_lr(C.byref(params), int(orb_th), kl.ctypes.data_as(vp), _ptr(dl, u8p), len(kl), _ptr(idxl, i64p),
                                  kr.ctypes.data_as(vp), _ptr(dr, u8p), len(kr), _ptr(idxr, i64p), img_w, img_h,
                                  out.ctypes.data_as(vp), cap, _ptr(ri, i64p))
    if m < 0:
        raise RuntimeError("match_lr: %d" % m)
    return out[:m].copy(), ri


def ransac_fundamental(p1, p2):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = len(p1)
    mask = np.zeros(max(n, 1), np.uint8)
    F = np.zeros(9, np.float64)
    bh = C.c_int(0)
    nu = C.c_int(0)
    cnt = lib().svo_oracle_ransac_fundamental(_ptr(p1, f32p), _ptr(p2, f32p), n, _ptr(mask, u8p), _ptr(F, f64p), C.byref(bh), C.byref(nu))
    return cnt, mask[:n].copy(), F.reshape(3, 3), bh.value, nu.value


def cv_rng_raw(count):
    """cv::RNG's first `count` raw 32-bit outputs from the seed findFundamentalMat's RANSAC uses, (uint64)-1"""
    out = np.zeros(count, np.uint32)
    f = lib().svo_oracle_cv_rng_raw
    f.restype = None
    f(_ptr(out, C.POINTER(C.c_uint32)), int(count))
    return out


def ransac_samples(p1, p2, count):
    """the first `count` minimal samples (rows of seven indices) cv::findFundamentalMat's RANSAC draws for these point pairs"""
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    idx = np.zeros((count, 7), np.int32)
    k = lib().svo_oracle_ransac_samples(_ptr(p1, f32p), _ptr(p2, f32p), len(p1), int(count), _ptr(idx, C.POINTER(C.c_int32)))
    return idx[:k].copy()


def seven_point(p1, p2):
    """the models (1 or 3, 3x3 each) of the 7-point algorithm for the first seven correspondences"""
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    assert len(p1) >= 7 and len(p2) >= 7
    F = np.zeros(27, np.float64)
    n = lib().svo_oracle_seven_point(_ptr(p1, f32p), _ptr(p2, f32p), _ptr(F, f64p))
    return F.reshape(3, 3, 3)[:n].copy()


def track(params, orb_th, pkl, pdl, pkr, pdr, pm, pri, ckl, cdl, ckr, cdr, cm, cri, img_w, img_h, stats=False):
    """stage 4 on caller-supplied lists; stats=True: (tracked pairs, the call's eight SVO_TS_* counters)"""
    cap = max(len(pm), len(cm), 1)
    out = np.zeros(cap, index_pair_dtype)
    vp = C.c_void_p
    arrs = [np.ascontiguousarray(a, np.uint8) for a in (pdl, pdr, cdl, cdr)]
    ts = np.zeros(8, np.int32)
    t = lib().svo_oracle_track_stats(C.byref(params), int(orb_th),
                                     pkl.ctypes.data_as(vp), _ptr(arrs[0], u8p), pkr.ctypes.data_as(vp), _ptr(arrs[1], u8p), pm.ctypes.data_as(vp), len(pm), _ptr(pri, i64p),
                                     ckl.ctypes.data_as(vp), _ptr(arrs[2], u8p), ckr.ctypes.data_as(vp), _ptr(arrs[3], u8p), cm.ctypes.data_as(vp), len(cm), _ptr(cri, i64p),
                                     img_w, img_h, out.ctypes.data_as(vp), cap, ts.ctypes.data_as(vp))
    if t < 0:
        raise RuntimeError("track: %d" % t)
    return (out[:t].copy(), ts) if stats else out[:t].copy()


def project(lmks, cam, delta):
    lmks = np.ascontiguousarray(lmks, np.float64).reshape(-1, 3)
    delta = np.ascontiguousarray(delta, np.float64)
    n = len(lmks)
    pix = np.zeros((n, 4), np.float32)
    jac = np.zeros((n, 4, 6), np.float64)
    lib().svo_oracle_project(_ptr(lmks, f64p), n, C.byref(cam), _ptr(delta, f64p), _ptr(pix, f32p), _ptr(jac, f64p))
    return pix, jac


def delta_to_pose(delta):
    delta = np.ascontiguousarray(delta, np.float64)
    out = np.zeros(6, np.float64)
    lib().svo_oracle_delta_to_pose(_ptr(delta, f64p), _ptr(out, f64p))
    return out


def prepare(src, map_x=None, map_y=None):
    """Stage 1 (grey conversion + rectification): src is HxW (grey) or HxWx3 (BGR) uint8; maps are HxW float32 or None."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape[:2]
    ch = 1 if src.ndim == 2 else src.shape[2]
    assert ch in (1, 3)
    dst = np.zeros((h, w), dtype=np.uint8)
    mx = my = None
    if map_x is not None:
        mx = np.ascontiguousarray(map_x, dtype=np.float32); my = np.ascontiguousarray(map_y, dtype=np.float32)
        assert mx.shape == (h, w) and my.shape == (h, w)
    f = lib().svo_oracle_prepare
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    f(src.ctypes.data, w, h, src.strides[0], ch, mx.ctypes.data if mx is not None else None, my.ctypes.data if my is not None else None,
      dst.ctypes.data, dst.strides[0])
    return dst


def pose_to_delta(pose):
    p = np.ascontiguousarray(pose, np.float64); d = np.zeros(6)
    f = lib().svo_oracle_pose_to_delta
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p]
    f(p.ctypes.data, d.ctypes.data)
    return d


def projected_coords(pre_matches, pre_left, pre_right, tracked_first, cam, change_pose):
    """getProjectedCoords (C:415-466): (B, 4) float32 pixels uL vL uR vR of the pairings whose tracked_first is -1."""
    m = np.ascontiguousarray(pre_matches, dmatch_dtype); kl = np.ascontiguousarray(pre_left, keypoint_dtype); kr = np.ascontiguousarray(pre_right, keypoint_dtype)
    tf = np.ascontiguousarray(tracked_first, np.int32); pose = np.ascontiguousarray(change_pose, np.float64)
    pix = np.zeros((max(1, len(m)), 4), np.float32)
    f = lib().svo_oracle_projected_coords
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = f(m.ctypes.data, len(m), kl.ctypes.data, kr.ctypes.data, tf.ctypes.data, C.addressof(cam), pose.ctypes.data, pix.ctypes.data)
    return pix[:n].copy()


def sad8(l, r, lx, ly, rx, ry):
    l, r = _img(l), _img(r)
    return int(lib().svo_oracle_sad8(_ptr(l, u8p), _ptr(r, u8p), C.c_size_t(l.shape[1]), lx, ly, rx, ry))


def ransac_niters(cnt, n, max_iters=1000):
    """the RANSAC's stop rule (cv::RANSACUpdateNumIters(0.99, (n - cnt) / n, 7, max_iters)) as the oracle and the kernels evaluate it"""
    return int(lib().svo_oracle_ransac_niters(int(cnt), int(n), int(max_iters)))
