"""ctypes binding of the HIP C-ABI (include/svo_hip.h -> stereo_vo_amd/libsvo_hip.so).

This is the product path.  There is no CPU fallback: `lib()` raises if the shared library is missing and
`Context()` raises if no HIP device is present.
"""
import ctypes as C
import os
import numpy as np

from .abi import Params, Result, StereoCamera, keypoint_dtype, dmatch_dtype, index_pair_dtype

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVO_HIP_LIB") or os.path.join(_HERE, "libsvo_hip.so")      # SVO_HIP_LIB: a library built from another commit, for A/B runs (tools/prof.sh abbench)
_LIB = None

MAX_LANES = 128
RUN_DETECT, RUN_MATCH, RUN_TRACK, RUN_OPTIMIZE, RUN_ALL = 1, 2, 4, 8, 15
FLAG_REPEAT, FLAG_NO_SHIFT, FLAG_DEVICE_IMAGES, FLAG_BGR_IMAGES, FLAG_DETECT_NO_POST, RUN_DETECT_POST, FLAG_PINNED_IMAGES = 16, 32, 64, 128, 256, 512, 1024

# every entry point include/svo_hip.h declares (tests check that the library exports all of them)
EXPORTS = [
    "svo_config_defaults", "svo_params_defaults", "svo_create", "svo_destroy", "svo_strerror", "svo_last_error",
    "svo_set_params", "svo_get_params", "svo_params_load_ini", "svo_set_fast_threshold", "svo_set_orb_threshold", "svo_get_fast_threshold",
    "svo_get_orb_threshold", "svo_set_stream", "svo_get_stream", "svo_get_device", "svo_set_camera", "svo_set_rectify_map", "svo_reset", "svo_process", "svo_wait", "svo_get_result", "svo_get_results", "svo_copy_results_async",
    "svo_get_keypoints", "svo_get_matches", "svo_get_tracked", "svo_get_residuals", "svo_get_outliers",
    "svo_get_keypoints_oct", "svo_get_matches_oct", "svo_get_tracked_oct", "svo_get_row_index", "svo_get_matches_row_index", "svo_get_match_ids", "svo_reset_ids", "svo_set_this_frame_as_kf",
    "svo_put_features", "svo_put_matches", "svo_put_tracked", "svo_put_match_ids", "svo_save_state", "svo_load_state", "svo_change_in_pose", "svo_projected_coords", "svo_hamming_match",
    "svo_debug_get_level", "svo_debug_get_raw_keypoints", "svo_debug_get_status_word", "svo_debug_get_redo_count", "svo_debug_timeline", "svo_profiler_sections_enabled",
    "svo_kernel_times", "svo_kernel_times_reset", "svo_kernel_times_select", "svo_abi_sizes",
    "svo_get_values", "svo_put_features_oct", "svo_put_matches_oct", "svo_put_match_ids_oct",
    "svo_handover_bytes", "svo_export_frame", "svo_import_frame",
    "svo_use_graphs", "svo_record_after_post", "svo_wait_upload", "svo_host_alloc", "svo_host_free", "svo_host_register", "svo_host_unregister",
]


BATCH_EXPORTS = [
    "svo_batch_abi_sizes", "svo_batch_config_defaults", "svo_batch_create", "svo_batch_create_sized", "svo_batch_destroy", "svo_batch_last_error", "svo_batch_lanes", "svo_batch_contexts",
    "svo_batch_context", "svo_batch_set_params", "svo_batch_set_camera", "svo_batch_set_results_buffer", "svo_batch_switch_results_buffer", "svo_batch_step",
    "svo_batch_wait_on_stream", "svo_batch_hold_for_event", "svo_batch_synchronize", "svo_batch_results", "svo_batch_reset",
    "svo_fpstream_create", "svo_fpstream_destroy", "svo_fpstream_last_error", "svo_fpstream_contexts", "svo_fpstream_context",
    "svo_fpstream_last_owner", "svo_fpstream_set_params", "svo_fpstream_set_camera", "svo_fpstream_push", "svo_fpstream_synchronize",
]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_lanes", C.c_int32), ("max_w", C.c_int32), ("max_h", C.c_int32),
                ("max_kps", C.c_int32), ("max_cand", C.c_int32), ("kernel_times", C.c_int32), ("max_octaves", C.c_int32),
                ("stream", C.c_void_p)]


class BatchConfig(C.Structure):
    """svo_batch_config (include/svo_batch.h)"""
    _fields_ = [("ctx", Config), ("n_contexts", C.c_int32), ("schedule", C.c_int32), ("det_priority_high", C.c_int32),
                ("post_mode", C.c_int32), ("det_streams", C.c_int32), ("rest_streams", C.c_int32), ("no_detect_ahead", C.c_int32)]


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("stride", C.c_int64)]


class Frame(C.Structure):
    _fields_ = [("left", Image), ("right", Image)]


class Values(C.Structure):
    _fields_ = [("left_kps", C.c_void_p), ("left_desc", C.c_void_p), ("right_kps", C.c_void_p), ("right_desc", C.c_void_p),
                ("matches", C.c_void_p), ("match_ids", C.c_void_p), ("cap_kps", C.c_int32), ("cap_matches", C.c_int32),
                ("n_left", C.c_int32), ("n_right", C.c_int32), ("n_matches", C.c_int32), ("n_ids", C.c_int32)]


class SvoError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SvoError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.svo_strerror.restype = C.c_char_p
        L.svo_last_error.restype = C.c_char_p
        L.svo_last_error.argtypes = [C.c_void_p]
        for n in ("svo_destroy", "svo_config_defaults", "svo_params_defaults", "svo_abi_sizes"):
            getattr(L, n).restype = None
        L.svo_destroy.argtypes = [C.c_void_p]
        for n in ("svo_batch_last_error", "svo_fpstream_last_error"):
            getattr(L, n).restype = C.c_char_p; getattr(L, n).argtypes = [C.c_void_p]
        for n in ("svo_batch_context", "svo_fpstream_context", "svo_fpstream_last_owner"):
            getattr(L, n).restype = C.c_void_p
        for n in ("svo_batch_destroy", "svo_fpstream_destroy", "svo_batch_config_defaults", "svo_batch_abi_sizes"):
            getattr(L, n).restype = None
        L.svo_batch_destroy.argtypes = [C.c_void_p]; L.svo_fpstream_destroy.argtypes = [C.c_void_p]
        L.svo_batch_create_sized.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _LIB = L
    return _LIB


def default_params() -> Params:
    p = Params()
    lib().svo_params_defaults(C.byref(p))
    return p


def load_params_ini(path, sections, p: Params = None) -> Params:
    """loadParamsFromConfigFileName (H:665-672): the reference's INI keys from the seven sections
    [rectify, detect, match, if-match, least_squares, gui, general] ('' skips a group) over `p` (defaults when None)."""
    if len(sections) != 7:
        raise ValueError("seven section names expected (H:556)")
    if p is None:
        p = default_params()
    arr = (C.c_char_p * 7)(*[s.encode() if s else None for s in sections])
    rc = lib().svo_params_load_ini(str(path).encode(), arr, C.byref(p))
    if rc != 0:
        raise SvoError("svo_params_load_ini(%s): %s" % (path, lib().svo_strerror(rc).decode()))
    return p


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """n_lanes independent rso::CStereoOdometryEstimator streams driven through the HIP kernels together."""

    def __init__(self, n_lanes=1, max_w=1280, max_h=960, max_kps=4096, max_cand=1 << 17, device=0, kernel_times=False, stream=None, max_octaves=1):
        self.L = lib()
        cfg = Config()
        self.L.svo_config_defaults(C.byref(cfg))
        cfg.device, cfg.n_lanes, cfg.max_w, cfg.max_h, cfg.max_kps, cfg.max_cand = device, n_lanes, max_w, max_h, max_kps, max_cand
        cfg.kernel_times = int(kernel_times)
        cfg.max_octaves = int(max_octaves)
        cfg.stream = stream
        self.n_lanes, self.max_kps = n_lanes, max_kps
        h = C.c_void_p()
        rc = self.L.svo_create(C.byref(cfg), C.byref(h))
        self.h = h
        if rc != 0:
            msg = self._err(rc)
            if h:
                self.L.svo_destroy(h)
            self.h = None
            raise SvoError("svo_create failed: " + msg)
        self._keep = None
        self._owned = True

    @classmethod
    def from_handle(cls, handle, n_lanes, max_kps):
        """A view of a context that something else owns (svo_batch_context / svo_fpstream_context): close() leaves it alone."""
        self = cls.__new__(cls)
        self.L = lib()
        self.h = C.c_void_p(handle)
        self.n_lanes, self.max_kps = n_lanes, max_kps
        self._keep = None
        self._owned = False
        return self

    def _err(self, rc):
        s = self.L.svo_strerror(rc).decode()
        if self.h:
            le = self.L.svo_last_error(self.h)
            if le:
                s += " [" + le.decode() + "]"
        return s

    def _ck(self, rc, what):
        if rc < 0:
            raise SvoError("%s: %s" % (what, self._err(rc)))
        return rc

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_owned", True):
                self.L.svo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters ------------------------------------------------------------------------------------
    def set_params(self, p: Params):
        self._ck(self.L.svo_set_params(self.h, C.byref(p)), "svo_set_params")

    def set_camera(self, cam: StereoCamera, lane=-1):
        self._ck(self.L.svo_set_camera(self.h, lane, C.byref(cam)), "svo_set_camera")

    def set_fast_threshold(self, v):
        self._ck(self.L.svo_set_fast_threshold(self.h, int(v)), "svo_set_fast_threshold")

    def set_orb_threshold(self, v):
        self._ck(self.L.svo_set_orb_threshold(self.h, int(v)), "svo_set_orb_threshold")

    def fast_threshold(self):
        return self.L.svo_get_fast_threshold(self.h)

    def orb_threshold(self):
        return self.L.svo_get_orb_threshold(self.h)

    def reset(self, lane=-1):
        self._ck(self.L.svo_reset(self.h, lane), "svo_reset")

    # -- frames ----------------------------------------------------------------------------------------
    def process_host(self, pairs, flags=RUN_ALL):
        """pairs: list of (left, right) uint8 numpy arrays [h, w], one per lane."""
        assert len(pairs) == self.n_lanes
        fr = (Frame * self.n_lanes)()
        keep = []
        for i, (l, r) in enumerate(pairs):
            l = np.ascontiguousarray(l, np.uint8)
            r = np.ascontiguousarray(r, np.uint8)
            keep += [l, r]
            h, w = l.shape[:2]
            if l.ndim == 3:
                assert l.shape[2] == 3 and r.shape == l.shape
                flags |= FLAG_BGR_IMAGES
            fr[i].left = Image(l.ctypes.data, w, h, l.strides[0])
            fr[i].right = Image(r.ctypes.data, w, h, r.strides[0])
        self._keep = keep
        self._ck(self.L.svo_process(self.h, fr, C.c_uint32(flags & ~FLAG_DEVICE_IMAGES)), "svo_process")

    def process_device(self, ptr_pairs, w, h, stride, flags=RUN_ALL):
        """ptr_pairs: list of (left_ptr, right_ptr) device addresses (e.g. torch tensor.data_ptr()), one per lane."""
        assert len(ptr_pairs) == self.n_lanes
        fr = (Frame * self.n_lanes)()
        for i, (l, r) in enumerate(ptr_pairs):
            fr[i].left = Image(l, w, h, stride)
            fr[i].right = Image(r, w, h, stride)
        self._ck(self.L.svo_process(self.h, fr, C.c_uint32(flags | FLAG_DEVICE_IMAGES)), "svo_process")

    def process_pinned(self, ptr_pairs, w, h, stride, flags=RUN_ALL):
        """ptr_pairs: (left_ptr, right_ptr) PAGE-LOCKED host addresses per lane (torch pin_memory() tensors, svo_host_alloc):
        the upload is enqueued on the context's copy stream; the buffers must stay untouched until wait_upload()."""
        assert len(ptr_pairs) == self.n_lanes
        fr = (Frame * self.n_lanes)()
        for i, (l, r) in enumerate(ptr_pairs):
            fr[i].left = Image(l, w, h, stride)
            fr[i].right = Image(r, w, h, stride)
        self._ck(self.L.svo_process(self.h, fr, C.c_uint32((flags | FLAG_PINNED_IMAGES) & ~FLAG_DEVICE_IMAGES)), "svo_process")

    def handover_bytes(self):
        self.L.svo_handover_bytes.restype = C.c_size_t
        return int(self.L.svo_handover_bytes(self.h))

    def export_frame(self, dev_ptr, nbytes):
        """Enqueue the hand-over record ("what the next call finds as its previous frame", all lanes) into device memory."""
        self._ck(self.L.svo_export_frame(self.h, C.c_void_p(dev_ptr), C.c_size_t(nbytes)), "svo_export_frame")

    def import_frame(self, dev_ptr, nbytes):
        self._ck(self.L.svo_import_frame(self.h, C.c_void_p(dev_ptr), C.c_size_t(nbytes)), "svo_import_frame")

    def use_graphs(self, enable=True):
        """Capture each distinct svo_process call into a hipGraph and replay it (one launch per frame)."""
        self._ck(self.L.svo_use_graphs(self.h, int(enable)), "svo_use_graphs")

    def wait_upload(self):
        self._ck(self.L.svo_wait_upload(self.h), "svo_wait_upload")

    def set_stream(self, stream):
        """Later process / copy calls enqueue on this raw HIP stream (None: the stream the context was created with)."""
        self._ck(self.L.svo_set_stream(self.h, C.c_void_p(stream) if stream else None), "svo_set_stream")

    def get_stream(self):
        """the raw hipStream_t (int) calls enqueue on right now"""
        st = C.c_void_p()
        self._ck(self.L.svo_get_stream(self.h, C.byref(st)), "svo_get_stream")
        return st.value or 0

    def set_rectify_map(self, lane, side, map_x, map_y):
        """Stage-1 rectification map of one camera (HxW float32 source coordinates); None, None clears it."""
        if map_x is None:
            self._ck(self.L.svo_set_rectify_map(self.h, lane, side, None, None, 0, 0), "svo_set_rectify_map")
            return
        mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
        assert mx.ndim == 2 and mx.shape == my.shape
        self._ck(self.L.svo_set_rectify_map(self.h, lane, side, C.c_void_p(mx.ctypes.data), C.c_void_p(my.ctypes.data), mx.shape[1], mx.shape[0]), "svo_set_rectify_map")

    def projected_coords(self, pre_matches, pre_left, pre_right, tracked_first, cam, change_pose):
        """getProjectedCoords (common.cpp:415-466): (B, 4) float32 pixels of the pairings whose tracked_first is -1."""
        m = np.ascontiguousarray(pre_matches, dmatch_dtype); kl = np.ascontiguousarray(pre_left, keypoint_dtype); kr = np.ascontiguousarray(pre_right, keypoint_dtype)
        tf = np.ascontiguousarray(tracked_first, np.int32); pose = np.ascontiguousarray(change_pose, np.float64)
        pix = np.zeros((max(1, len(m)), 4), np.float32)
        n = self._ck(self.L.svo_projected_coords(self.h, _vp(m), len(m), _vp(kl), len(kl), _vp(kr), len(kr), _vp(tf), C.byref(cam), _vp(pose), _vp(pix), len(pix)), "svo_projected_coords")
        return pix[:n].copy()

    def save_state(self, lane, path):
        """saveStateToFile (common.cpp:475-543) of one lane."""
        self._ck(self.L.svo_save_state(self.h, lane, os.fsencode(path)), "svo_save_state")

    def load_state(self, lane, path):
        """loadStateFromFile (common.cpp:261-350) into one lane."""
        self._ck(self.L.svo_load_state(self.h, lane, os.fsencode(path)), "svo_load_state")

    def run_stages(self, flags):
        """Run stages on data already in the context (svo_put_* / previous svo_process), no prev/cur shift."""
        self._ck(self.L.svo_process(self.h, None, C.c_uint32((flags | FLAG_NO_SHIFT) & ~RUN_DETECT)), "svo_process")

    def wait(self):
        self._ck(self.L.svo_wait(self.h), "svo_wait")

    def result(self, lane=0) -> Result:
        r = Result()
        self._ck(self.L.svo_get_result(self.h, lane, C.byref(r)), "svo_get_result")
        return r

    def results(self):
        arr = (Result * self.n_lanes)()
        self._ck(self.L.svo_get_results(self.h, arr), "svo_get_results")
        return list(arr)

    def copy_results_async(self, dst_ptr, nbytes):
        self._ck(self.L.svo_copy_results_async(self.h, C.c_void_p(dst_ptr), C.c_size_t(nbytes)), "svo_copy_results_async")

    # -- lists -----------------------------------------------------------------------------------------
    def keypoints(self, lane=0, which=0, side=0, octave=0):
        n = self._ck(self.L.svo_get_keypoints_oct(self.h, lane, which, side, octave, None, None, 0), "svo_get_keypoints_oct")
        k = np.zeros(n, keypoint_dtype)
        d = np.zeros((n, 32), np.uint8)
        if n:
            self._ck(self.L.svo_get_keypoints_oct(self.h, lane, which, side, octave, _vp(k), _vp(d), n), "svo_get_keypoints_oct")
        return k, d

    def row_index(self, lane=0, which=0, side=0, octave=0):
        n = self._ck(self.L.svo_get_row_index(self.h, lane, which, side, octave, None, 0), "svo_get_row_index")
        a = np.zeros(n, np.int32)
        if n:
            self._ck(self.L.svo_get_row_index(self.h, lane, which, side, octave, _vp(a), n), "svo_get_row_index")
        return a

    def matches_row_index(self, lane=0, which=0, octave=0):
        n = self._ck(self.L.svo_get_matches_row_index(self.h, lane, which, octave, None, 0), "svo_get_matches_row_index")
        a = np.zeros(n, np.int32)
        if n:
            self._ck(self.L.svo_get_matches_row_index(self.h, lane, which, octave, _vp(a), n), "svo_get_matches_row_index")
        return a

    def raw_keypoints(self, lane=0, side=0):
        n = self._ck(self.L.svo_debug_get_raw_keypoints(self.h, lane, side, None, None, 0), "svo_debug_get_raw_keypoints")
        k = np.zeros(n, keypoint_dtype)
        d = np.zeros((n, 32), np.uint8)
        if n:
            self._ck(self.L.svo_debug_get_raw_keypoints(self.h, lane, side, _vp(k), _vp(d), n), "svo_debug_get_raw_keypoints")
        return k, d

    def level(self, lane, side, level):
        w, h = C.c_int(0), C.c_int(0)
        n = self._ck(self.L.svo_debug_get_level(self.h, lane, side, level, None, 0, C.byref(w), C.byref(h)), "svo_debug_get_level")
        out = np.zeros((h.value, w.value), np.uint8)
        self._ck(self.L.svo_debug_get_level(self.h, lane, side, level, _vp(out), n, C.byref(w), C.byref(h)), "svo_debug_get_level")
        return out

    def redo_count(self, reset=False):
        """(image, level) pairs whose speculative FAST threshold failed (a second FAST pass at the caller's threshold) since creation / the last reset"""
        v = C.c_uint32(0)
        self._ck(self.L.svo_debug_get_redo_count(self.h, C.byref(v), int(bool(reset))), "svo_debug_get_redo_count")
        return int(v.value)

    def timeline(self, reset=False):
        """SVO_TIMELINE=1: (frame counter, array [16 frames][32 kinds][8 aux][2] of wall-clock ticks (10 ns), t0 = 2**64 - 1 where nothing ran)"""
        a = np.zeros((16, 32, 8, 2), np.uint64)
        n = self._ck(self.L.svo_debug_timeline(self.h, _vp(a), 16 * 256, int(bool(reset))), "svo_debug_timeline")
        return n, a

    def status_word(self, lane=0):
        w = C.c_uint32(0)
        self._ck(self.L.svo_debug_get_status_word(self.h, lane, C.byref(w)), "svo_debug_get_status_word")
        return w.value

    def matches(self, lane=0, which=0, octave=0):
        n = self._ck(self.L.svo_get_matches_oct(self.h, lane, which, octave, None, 0), "svo_get_matches_oct")
        m = np.zeros(n, dmatch_dtype)
        if n:
            self._ck(self.L.svo_get_matches_oct(self.h, lane, which, octave, _vp(m), n), "svo_get_matches_oct")
        return m

    def match_ids(self, lane=0, which=0, octave=0):
        n = self._ck(self.L.svo_get_match_ids(self.h, lane, which, octave, None, 0), "svo_get_match_ids")
        a = np.zeros(n, np.int32)
        if n:
            self._ck(self.L.svo_get_match_ids(self.h, lane, which, octave, _vp(a), n), "svo_get_match_ids")
        return a

    def reset_ids(self, lane=-1):
        self._ck(self.L.svo_reset_ids(self.h, lane), "svo_reset_ids")

    def set_this_frame_as_kf(self, lane=0):
        self._ck(self.L.svo_set_this_frame_as_kf(self.h, lane), "svo_set_this_frame_as_kf")

    def tracked(self, lane=0, octave=0):
        n = self._ck(self.L.svo_get_tracked_oct(self.h, lane, octave, None, 0), "svo_get_tracked_oct")
        t = np.zeros(n, index_pair_dtype)
        if n:
            self._ck(self.L.svo_get_tracked_oct(self.h, lane, octave, _vp(t), n), "svo_get_tracked_oct")
        return t

    def residuals(self, lane=0):
        n = self._ck(self.L.svo_get_residuals(self.h, lane, None, 0), "svo_get_residuals")
        r = np.zeros(n, np.float64)
        if n:
            self._ck(self.L.svo_get_residuals(self.h, lane, _vp(r), n), "svo_get_residuals")
        return r

    def outliers(self, lane=0):
        n = self._ck(self.L.svo_get_outliers(self.h, lane, None, 0), "svo_get_outliers")
        r = np.zeros(n, np.int32)
        if n:
            self._ck(self.L.svo_get_outliers(self.h, lane, _vp(r), n), "svo_get_outliers")
        return r

    def values(self, lane=0, which=0, octave=0):
        """getValues (H:704-724) in one device synchronisation: (left_kps, left_desc, right_kps, right_desc, matches, ids)."""
        mk = self.max_kps
        kl, kr = np.zeros(mk, keypoint_dtype), np.zeros(mk, keypoint_dtype)
        dl, dr = np.zeros((mk, 32), np.uint8), np.zeros((mk, 32), np.uint8)
        m, ids = np.zeros(mk, dmatch_dtype), np.zeros(mk, np.int32)
        v = Values(kl.ctypes.data, dl.ctypes.data, kr.ctypes.data, dr.ctypes.data, m.ctypes.data, ids.ctypes.data, mk, mk, 0, 0, 0, 0)
        self._ck(self.L.svo_get_values(self.h, lane, which, octave, C.byref(v)), "svo_get_values")
        return kl[:v.n_left].copy(), dl[:v.n_left].copy(), kr[:v.n_right].copy(), dr[:v.n_right].copy(), m[:v.n_matches].copy(), ids[:v.n_ids].copy()

    # -- precomputed-data bypass -------------------------------------------------------------------------
    def put_features(self, lane, which, side, kps, desc, img_w, img_h, octave=0):
        kps = np.ascontiguousarray(kps)
        desc = None if desc is None else np.ascontiguousarray(desc, np.uint8)
        self._ck(self.L.svo_put_features_oct(self.h, lane, which, side, octave, _vp(kps), _vp(desc), len(kps), img_w, img_h), "svo_put_features_oct")

    def put_matches(self, lane, which, m, octave=0):
        m = np.ascontiguousarray(m)
        self._ck(self.L.svo_put_matches_oct(self.h, lane, which, octave, _vp(m), len(m)), "svo_put_matches_oct")

    def put_match_ids(self, lane, which, ids, octave=0):
        ids = np.ascontiguousarray(ids, np.int32)
        self._ck(self.L.svo_put_match_ids_oct(self.h, lane, which, octave, _vp(ids), len(ids)), "svo_put_match_ids_oct")

    def put_tracked(self, lane, t):
        t = np.ascontiguousarray(t)
        self._ck(self.L.svo_put_tracked(self.h, lane, _vp(t), len(t)), "svo_put_tracked")

    def change_in_pose(self, tracked, pre_m, cur_m, pre_l, pre_r, cur_l, cur_r, cam, init6=None):
        n = len(tracked)
        res = Result()
        residual = np.zeros(max(n, 1), np.float64)
        outl = np.zeros(max(n, 1), np.int32)
        init = None if init6 is None else np.ascontiguousarray(init6, np.float64)
        arrs = [np.ascontiguousarray(a) for a in (tracked, pre_m, cur_m, pre_l, pre_r, cur_l, cur_r)]
        rc = self.L.svo_change_in_pose(self.h, _vp(arrs[0]), n, _vp(arrs[1]), len(pre_m), _vp(arrs[2]), len(cur_m),
                                       _vp(arrs[3]), len(pre_l), _vp(arrs[4]), len(pre_r), _vp(arrs[5]), len(cur_l), _vp(arrs[6]), len(cur_r),
                                       C.byref(cam), _vp(init), C.byref(res), _vp(residual), _vp(outl))
        self._ck(rc, "svo_change_in_pose")
        return bool(rc), res, residual[:res.n_residual], outl[:res.n_outliers]

    def hamming_match(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.zeros(max(len(q), 1), np.int32)
        dist = np.zeros(max(len(q), 1), np.int32)
        self._ck(self.L.svo_hamming_match(self.h, _vp(q), len(q), _vp(t), len(t), _vp(idx), _vp(dist)), "svo_hamming_match")
        return idx[:len(q)], dist[:len(q)]

    # -- timing ----------------------------------------------------------------------------------------
    def kernel_times(self):
        names = (C.c_char_p * 32)()
        tot = (C.c_double * 32)()
        calls = (C.c_int64 * 32)()
        n = self._ck(self.L.svo_kernel_times(self.h, names, tot, calls, 32), "svo_kernel_times")
        return {names[i].decode(): (tot[i], calls[i]) for i in range(n)}

    def kernel_times_select(self, name=None):
        self._ck(self.L.svo_kernel_times_select(self.h, name.encode() if name else None), "svo_kernel_times_select")

    def kernel_times_reset(self):
        self._ck(self.L.svo_kernel_times_reset(self.h), "svo_kernel_times_reset")
