/* svo_types.h -- plain-old-data records that cross the C-ABI boundary of the MI355X stereo-VO hot path.
 *
 * They replace, field for field, the OpenCV / MRPT types that appear in the reference's public interface
 * (libstereo-odometry/include/libstereo-odometry.h, "H" below), which cannot be used here because OpenCV,
 * MRPT and Eigen are absent on both the build container and the GPU box.
 */
#ifndef SVO_TYPES_H
#define SVO_TYPES_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == cv::KeyPoint field order (pt.x, pt.y, size, angle, response, octave, class_id), 28 bytes. H:108 */
typedef struct svo_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} svo_keypoint;

/* == cv::DMatch (queryIdx, trainIdx, imgIdx, distance), 16 bytes. H:109 */
typedef struct svo_dmatch {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} svo_dmatch;

/* one (previous-match-index, current-match-index) entry of rso::vector_index_pairs_t. H:139 */
typedef struct svo_index_pair {
    int32_t first, second;
} svo_index_pair;

/* the fields of mrpt::utils::TStereoCamera that the path reads (S5:185-193, 510-515; C:402-403) */
typedef struct svo_stereo_camera {
    double l_fx, l_fy, l_cx, l_cy;
    double r_fx, r_fy, r_cx, r_cy;
    double baseline;        /* rightCameraPose[0] */
    int32_t ncols, nrows;   /* leftCamera.ncols / nrows (used by getChangeInPose for the NMS grid) */
} svo_stereo_camera;

/* rso::VOErrorCode, same numeric values. H:142 */
enum {
    SVO_VOEC_NONE = 0, SVO_VOEC_BAD_COND_NUMBER = 1, SVO_VOEC_INCR_FUNC_COST_STG1 = 2,
    SVO_VOEC_INCR_FUNC_COST_STG2 = 3, SVO_VOEC_FIRST_ITERATION = 4, SVO_VOEC_BAD_TRACKING = 5
};

/* TDetectParams::TDMethod H:388, NMSMethod H:387, TSMMethod H:454, TIFMMethod H:291 */
enum { SVO_DM_ORB = 0, SVO_DM_FAST_ORB = 1, SVO_DM_FASTER = 2, SVO_DM_KLT = 3 };
enum { SVO_NMS_STANDARD = 0, SVO_NMS_ADAPTIVE = 1 };
enum { SVO_SM_DESC_BF = 0, SVO_SM_DESC_RBR = 1, SVO_SM_SAD = 2 };
enum { SVO_IFM_DESC_BF = 0, SVO_IFM_DESC_WIN = 1, SVO_IFM_SAD = 2, SVO_IFM_OPTICAL_FLOW = 3 };

/* The INI keys / struct fields that parameterise the path (H:266-508, H:554-663), one flat record.
 * Field names are the reference's. Defaults: svo_params_defaults() (S2:44-58, S3:46-57, C:69-82, S1:27-30),
 * except that the selectors default to the north-star configuration (ORB + BF + BF), because the
 * reference's own defaults select variants that are out of scope. */
typedef struct svo_params {
    /* RECTIFY */
    int32_t nOctaves;
    /* DETECT */
    int32_t detect_method;
    int32_t non_maximal_suppression;
    int32_t nmsMethod;
    int32_t min_distance;
    int32_t orb_nfeats;
    int32_t orb_nlevels;
    int32_t fast_min_th, fast_max_th, initial_FAST_threshold;
    double  minimum_ORB_response;
    /* MATCH */
    int32_t match_method;
    int32_t enable_robust_1to1_match;
    int32_t orb_min_th, orb_max_th;
    double  max_y_diff;
    double  orb_max_distance;
    /* IF-MATCH */
    int32_t ifm_method;
    int32_t ifm_win_w, ifm_win_h;
    int32_t filter_fund_matrix;   /* read by the reference's INI loader, unused by it (H:304) */
    /* LEAST_SQUARES */
    int32_t use_robust_kernel;
    int32_t max_iters, initial_max_iters;
    int32_t max_incr_cost;
    int32_t bad_tracking_th;
    int32_t use_previous_pose_as_initial;
    int32_t use_custom_initial_pose;
    int32_t _pad0;
    double  kernel_param;
    double  min_mod_out_vector;
    double  residual_threshold;
    /* GENERAL */
    int32_t vo_use_matches_ids;
    int32_t _pad1;
} svo_params;

/* TStereoOdometryResult (H:235-264) without the variable-length members, which have their own getters. */
typedef struct svo_result {
    double  outPose[6];      /* x y z yaw pitch roll  (CPose3D of the inverse of delta, S5:717-718) */
    double  delta[6];        /* w1 w2 w3 t1 t2 t3 as optimised (S5:45-52) */
    int32_t num_it, num_it_final;
    int32_t valid;
    int32_t error_code;
    int32_t tracked_feats_from_last_KF, tracked_feats_from_last_frame;
    int32_t detected_left[4], detected_right[4];   /* detected_feats[octave].first/.second */
    int32_t stereo_matches[4];
    int32_t n_octaves;
    int32_t n_outliers;      /* size of result.outliers (which holds INLIER cur-match indices, S5:603-610) */
    int32_t n_residual;      /* size of result.out_residual */
    int32_t status;          /* 0, or capacity bits: 1 = a level's FAST candidate list overflowed svo_config.max_cand, 2 = a keypoint /
                                track list was cut at svo_config.max_kps (the reference has no such limits: stage2_detect.cpp:461-464),
                                4 = svo_import_frame was handed a hand-over record of another layout */
    int32_t track_stats[8];  /* where stage 4's candidates go, summed over the octaves (SVO_TS_*): no reference counterpart -- the
                                reference prints some of them at verbose level 2 (stage4_match_consecutive.cpp:205, 240) */
} svo_result;

/* indices of svo_result.track_stats: previous-frame pairings that ...                                      (ifmDescBF | ifmDescWin) */
enum {
    SVO_TS_THRESHOLD = 0,    /* pass the descriptor-distance threshold on both sides (S4:149)               | = SVO_TS_COLLISION */
    SVO_TS_COLLISION = 1,    /* survive the joint collision filter (S4:145-160): the RANSAC's input         | survivors of S4:640-679 */
    SVO_TS_INLIERS_L = 2,    /* inliers of the left-left fundamental matrix (S4:202-205); 0 = no model */
    SVO_TS_INLIERS_R = 3,    /* inliers of the right-right one (S4:237-240) */
    SVO_TS_HYP_L = 4,        /* hypotheses the left RANSAC visited before its confidence stop */
    SVO_TS_HYP_R = 5,        /* the same, right */
    SVO_TS_BOTH_MASKS = 6,   /* are inliers of both models (S4:243-255; = SVO_TS_COLLISION when a model was not found) */
    SVO_TS_TRACKED = 7       /* also pass the left/right consistency check (S4:282): tracked_pairs */
};

#define SVO_MAX_OCTAVES 4
#define SVO_DESC_BYTES 32

#ifdef __cplusplus
}
#endif
#endif
