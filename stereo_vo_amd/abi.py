"""ctypes / numpy mirrors of include/svo_types.h (the POD records that cross the C-ABI).

Pure declarations: importing this module loads no native library.
"""
import ctypes as C
import numpy as np

MAX_OCTAVES = 4
DESC_BYTES = 32

# == cv::KeyPoint / cv::DMatch layouts (libstereo-odometry.h:108-109)
keypoint_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
dmatch_dtype = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
index_pair_dtype = np.dtype([("first", "<i4"), ("second", "<i4")])
assert keypoint_dtype.itemsize == 28 and dmatch_dtype.itemsize == 16 and index_pair_dtype.itemsize == 8

VOEC_NONE, VOEC_BAD_COND_NUMBER, VOEC_INCR_FUNC_COST_STG1, VOEC_INCR_FUNC_COST_STG2, VOEC_FIRST_ITERATION, VOEC_BAD_TRACKING = range(6)
DM_ORB, DM_FAST_ORB, DM_FASTER, DM_KLT = range(4)
SM_DESC_BF, SM_DESC_RBR, SM_SAD = range(3)
IFM_DESC_BF, IFM_DESC_WIN, IFM_SAD, IFM_OPTICAL_FLOW = range(4)


class StereoCamera(C.Structure):
    _fields_ = [("l_fx", C.c_double), ("l_fy", C.c_double), ("l_cx", C.c_double), ("l_cy", C.c_double),
                ("r_fx", C.c_double), ("r_fy", C.c_double), ("r_cx", C.c_double), ("r_cy", C.c_double),
                ("baseline", C.c_double), ("ncols", C.c_int32), ("nrows", C.c_int32)]

    @classmethod
    def simple(cls, f, cx, cy, baseline, ncols, nrows):
        return cls(f, f, cx, cy, f, f, cx, cy, baseline, ncols, nrows)


class Params(C.Structure):
    _fields_ = [
        ("nOctaves", C.c_int32),
        ("detect_method", C.c_int32), ("non_maximal_suppression", C.c_int32), ("nmsMethod", C.c_int32),
        ("min_distance", C.c_int32), ("orb_nfeats", C.c_int32), ("orb_nlevels", C.c_int32),
        ("fast_min_th", C.c_int32), ("fast_max_th", C.c_int32), ("initial_FAST_threshold", C.c_int32),
        ("minimum_ORB_response", C.c_double),
        ("match_method", C.c_int32), ("enable_robust_1to1_match", C.c_int32),
        ("orb_min_th", C.c_int32), ("orb_max_th", C.c_int32),
        ("max_y_diff", C.c_double), ("orb_max_distance", C.c_double),
        ("ifm_method", C.c_int32), ("ifm_win_w", C.c_int32), ("ifm_win_h", C.c_int32), ("filter_fund_matrix", C.c_int32),
        ("use_robust_kernel", C.c_int32), ("max_iters", C.c_int32), ("initial_max_iters", C.c_int32),
        ("max_incr_cost", C.c_int32), ("bad_tracking_th", C.c_int32), ("use_previous_pose_as_initial", C.c_int32),
        ("use_custom_initial_pose", C.c_int32), ("_pad0", C.c_int32),
        ("kernel_param", C.c_double), ("min_mod_out_vector", C.c_double), ("residual_threshold", C.c_double),
        ("vo_use_matches_ids", C.c_int32), ("_pad1", C.c_int32),
    ]

    def copy(self):
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


class Result(C.Structure):
    _fields_ = [
        ("outPose", C.c_double * 6), ("delta", C.c_double * 6),
        ("num_it", C.c_int32), ("num_it_final", C.c_int32), ("valid", C.c_int32), ("error_code", C.c_int32),
        ("tracked_feats_from_last_KF", C.c_int32), ("tracked_feats_from_last_frame", C.c_int32),
        ("detected_left", C.c_int32 * 4), ("detected_right", C.c_int32 * 4), ("stereo_matches", C.c_int32 * 4),
        ("n_octaves", C.c_int32), ("n_outliers", C.c_int32), ("n_residual", C.c_int32), ("status", C.c_int32),
        ("track_stats", C.c_int32 * 8),
    ]

# indices of Result.track_stats (include/svo_types.h SVO_TS_*)
TS_NAMES = ("threshold", "collision", "inliers_left", "inliers_right", "hyp_left", "hyp_right", "both_masks", "tracked")


def north_star_params(base: Params, orb_nfeats=1350) -> Params:
    """SURVEY.md 8d common parameters: ORB + BF (1-to-1, max_y_diff 1, orb_max_distance 60) + BF tracking."""
    p = base.copy()
    p.detect_method = DM_ORB
    p.orb_nfeats = orb_nfeats
    p.orb_nlevels = 8
    p.match_method = SM_DESC_BF
    p.max_y_diff = 1.0
    p.enable_robust_1to1_match = 1
    p.orb_max_distance = 60.0
    p.ifm_method = IFM_DESC_BF
    return p
