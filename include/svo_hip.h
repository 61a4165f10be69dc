/* svo_hip.h -- C-ABI of the MI355X (gfx950) stereo-VO hot path.  Shared library: stereo_vo_amd/libsvo_hip.so
 *
 * Drop-in boundary for stages 2-5 of rso::CStereoOdometryEstimator::processNewImagePair()
 * (libstereo-odometry/src/process_new_image_pair.cpp:41-385, "P"; declarations in
 * libstereo-odometry/include/libstereo-odometry.h, "H").  Plain pointers and sizes only; every call returns an
 * int status (0 = ok, <0 = error, see svo_strerror) and never throws.  Out-buffers are caller-owned with their
 * capacity passed in; device buffers are context-owned and persist across frames (the previous frame's
 * keypoints / descriptors / pairings stay in HBM).
 *
 * One context drives `n_lanes` INDEPENDENT estimator streams ("lanes") through every kernel launch together:
 * a lane is what one rso::CStereoOdometryEstimator instance is in the reference (all of its state is
 * per-instance, H:732-831), and batching lanes per launch is how a 256-CU device is filled by a workload whose
 * single-stream form is launch-latency bound.  A context with n_lanes = 1 is exactly one
 * estimator.  There is no CPU fallback: without a HIP device svo_create fails.
 */
#ifndef SVO_HIP_H
#define SVO_HIP_H
#include "svo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_MAX_LANES 128
#define SVO_MAX_LEVELS 8

/* status codes */
enum {
    SVO_OK = 0, SVO_ERR_HIP = -1, SVO_ERR_ARG = -2, SVO_ERR_UNSUPPORTED = -3, SVO_ERR_NO_DEVICE = -4,
    SVO_ERR_CAPACITY = -5, SVO_ERR_STATE = -6
};

/* svo_process flags */
enum {
    SVO_RUN_DETECT = 1,     /* stage 2  (P:166-167 -> stage2_detect_features, H:1017)  */
    SVO_RUN_MATCH = 2,      /* stage 3  (P:269 -> stage3_match_left_right, H:1025)     */
    SVO_RUN_TRACK = 4,      /* stage 4  (P:314 -> stage4_track, H:1033)                */
    SVO_RUN_OPTIMIZE = 8,   /* stage 5  (P:338 -> stage5_optimize, H:1041)             */
    SVO_RUN_ALL = 15,
    SVO_FLAG_REPEAT = 16,   /* request_data.repeat (H:221, P:86-92)                    */
    SVO_FLAG_NO_SHIFT = 32, /* do not run the prev/cur shift of P:86-100 (staged tests and the
                               precomputed-data bypass of P:131-162 / P:219-251 after svo_put_*) */
    SVO_FLAG_DEVICE_IMAGES = 64, /* svo_image.data are device pointers (already resident in HBM) */
    SVO_FLAG_DETECT_NO_POST = 256, /* with SVO_RUN_DETECT: stop before the reference's own post-processing of the detector output
                                    (m_non_max_sup + m_update_indexes, S2:583-618: one latency-bound block per image) and the
                                    description of its survivors; both are left to a later call with SVO_RUN_DETECT_POST,
                                    e.g. on another stream */
    SVO_RUN_DETECT_POST = 512,   /* that post-processing alone, ahead of the other stages of the call */
    SVO_FLAG_BGR_IMAGES = 128,   /* svo_image.data are 8-bit 3-channel BGR (stride in bytes): stage 1's grey conversion runs
                                    on the device (stage1_rectify.cpp:50-51) */
    SVO_FLAG_DETECT_SPLIT_AT_SELECT = 2048, /* moves the split point of SVO_FLAG_DETECT_NO_POST / SVO_RUN_DETECT_POST forward: the detect call stops
                                    after the FAST kernel (pyramid + corner candidates: the throughput half), and the post call
                                    starts with the per-level selection (top-K, Harris, sort) -- pass it to BOTH calls; ORB mode only */
    SVO_FLAG_DETECT_AHEAD = 4096, /* a detect call that may run AHEAD of the stages 3-5 of the frame before it (svo_batch's pipelined
                                    schedule, round 4).  With SVO_RUN_DETECT | SVO_FLAG_DETECT_NO_POST: the call writes the detector's
                                    per-image scratch ONLY (level-0 pointer table, pyramid, candidates, per-level winners) -- no lane
                                    state, no result record, no status word; capacity bits it raises are staged.  The matching
                                    SVO_RUN_DETECT_POST call carries the flag too and is issued WITHOUT SVO_FLAG_NO_SHIFT: it runs the
                                    prev/cur shift of P:86-100, clears the record and folds the staged bits in.  Everything the ahead
                                    call overwrites is last read by the post-processing of the previous frame: svo_record_after_post
                                    hands the caller an event for exactly that point */
    SVO_FLAG_PINNED_IMAGES = 1024 /* svo_image.data are PAGE-LOCKED host pointers (svo_host_alloc / svo_host_register): the upload is
                                    enqueued on the context's copy stream and svo_process returns without waiting for it; the
                                    images must stay untouched until svo_wait_upload (or svo_wait) returns.  Without this flag host
                                    images are copied into the context's own page-locked staging first, so they are only borrowed
                                    for the duration of the call, as in the reference (P:111-120) */
};

typedef struct svo_ctx svo_ctx;

typedef struct svo_config {
    int32_t device;         /* HIP device ordinal */
    int32_t n_lanes;        /* 1..SVO_MAX_LANES independent estimator streams per launch */
    int32_t max_w, max_h;   /* largest image the context will see */
    int32_t max_kps;        /* capacity of every keypoint / pairing list per octave (power of two, <= 8192; above 4096 the NMS and
                               Gauss-Newton kernels keep their sort / hash arrays in global memory instead of LDS: slower per lane) */
    int32_t max_cand;       /* capacity of the per-level FAST candidate list at level 0 (scaled by area above) */
    int32_t kernel_times;   /* 1: bracket every kernel with HIP events (svo_kernel_times) */
    int32_t max_octaves;    /* 1..4: octave lists per lane (params_rectify.nOctaves of the FAST+ORB mode); 1 suffices for ORB */
    void*   stream;         /* hipStream_t to run on; NULL = the context creates its own */
} svo_config;

/* 8-bit gray, row-major (what stage 1 hands to stage 2: S1:47-85 is outside this library) */
typedef struct svo_image {
    const uint8_t* data;
    int32_t w, h;
    int64_t stride;         /* bytes between rows */
} svo_image;

typedef struct svo_frame {  /* one lane's rectified stereo pair (request_data.stereo_imgs, H:208) */
    svo_image left, right;
} svo_frame;

void svo_config_defaults(svo_config* c);
void svo_params_defaults(svo_params* p);          /* reference defaults, north-star selectors (svo_types.h) */

int  svo_create(const svo_config* cfg, svo_ctx** out);
void svo_destroy(svo_ctx* ctx);
const char* svo_strerror(int status);
const char* svo_last_error(const svo_ctx* ctx);   /* text of the last HIP failure */

/* loadParamsFromConfigFile (H:554-663): stores the record, then resetFASTThreshold / resetORBThreshold.  The reference has no
 * keypoint cap (stage2_detect.cpp:461-464); a request that this context's lists cannot hold (orb_nfeats against
 * svo_config.max_kps, nOctaves against max_octaves) is refused here with SVO_ERR_CAPACITY -- the numbers are in svo_last_error --
 * and the parameters in force stay as they were. */
int svo_set_params(svo_ctx* ctx, const svo_params* p);
int svo_get_params(const svo_ctx* ctx, svo_params* p);
/* The file half of loadParamsFromConfigFile / loadParamsFromConfigFileName (H:551-672): reads the reference's INI keys from the
 * seven sections {RECTIFY, DETECT, MATCH, IF-MATCH, LEAST_SQUARES, GUI, GENERAL} (an empty or NULL name skips the group) into
 * *p, keeping the current value of every key that is absent (`if_match_method` falls back to 0 as at H:611).  Host only: needs
 * no context and no GPU; follow it with svo_set_params.  SVO_ERR_ARG when the file cannot be opened (H:669 asserts it exists). */
int svo_params_load_ini(const char* path, const char* const sections[7], svo_params* p);
int svo_set_fast_threshold(svo_ctx* ctx, int v);  /* setFASTThreshold, clamped (H:531) */
int svo_set_orb_threshold(svo_ctx* ctx, int v);   /* setORBThreshold, clamped (H:538) */
int svo_get_fast_threshold(const svo_ctx* ctx);
int svo_get_orb_threshold(const svo_ctx* ctx);
/* Switch the HIP stream that later svo_process / svo_copy_results_async calls enqueue on (NULL: the context's own).
 * Ordering between work already enqueued on the old stream and work on the new one is the caller's business (events):
 * this is what lets a caller run stage 2 of one context on a normal-priority stream and stages 3-5 of another on a
 * high-priority one (bench.py).  Stream lifetime: a stream must stay alive while calls that enqueue on it are being made; once
 * the caller has switched away it may synchronise and destroy the stream -- later svo_wait / svo_get_* / svo_destroy wait on
 * context-owned events recorded behind the work, never on the stream handle. */
int svo_set_stream(svo_ctx* ctx, void* stream);
int svo_get_device(const svo_ctx* ctx);           /* HIP device ordinal of the context (svo_config.device); < 0 on error */
/* Arms ONE hipEvent_t: the next svo_process call that runs the detector's post-processing (SVO_RUN_DETECT_POST, or SVO_RUN_DETECT
 * without SVO_FLAG_DETECT_NO_POST) records it on its stream right behind the last kernel that reads the detector's per-image
 * scratch (NMS / row sort + description), i.e. BEFORE stages 3-5 of the same call.  A detect call with SVO_FLAG_DETECT_AHEAD for
 * the next frame need wait for nothing later.  (Inside a hipGraph replay the event is recorded behind the whole graph.) */
int svo_record_after_post(svo_ctx* ctx, void* event);
/* the hipStream_t later calls enqueue on (so that a caller can order its own work -- an RCCL call, an event -- after a frame) */
int svo_get_stream(svo_ctx* ctx, void** stream);
/* request_data.stereo_cam (H:211), per lane; lane = -1 sets every lane */
int svo_set_camera(svo_ctx* ctx, int lane, const svo_stereo_camera* cam);
/* Stage 1 on the device (stage1_rectify.cpp:47-85).  Rectification maps of one camera of one lane (lane = -1: every
 * lane): HOST float arrays [h][w] of source coordinates per output pixel, as cv::initUndistortRectifyMap(CV_32FC1)
 * -- what mrpt::vision::CStereoRectifyMap::setFromCamParams (S1:66-68) precomputes -- yields.  They are converted
 * once to the 1/32-pixel fixed point of cv::remap and kept on the device; from then on svo_process rectifies that
 * camera's images (bilinear, constant-0 border, S1:70-72) before detection.  map_x = map_y = NULL clears the map
 * (areImagesRectified(), S1:61-65). */
int svo_set_rectify_map(svo_ctx* ctx, int lane, int side, const float* map_x, const float* map_y, int w, int h);
/* forget both frames, the warm start and the match-ID counters of one lane (a freshly constructed estimator, C:28-50); -1 = all.
 * The FAST / ORB thresholds belong to the context (all its lanes share svo_params) and stay as they are: svo_set_*_threshold. */
int svo_reset(svo_ctx* ctx, int lane);

/* processNewImagePair for every lane (P:41-385): ENQUEUES the whole frame on the context's stream and
 * returns; frames[lane] for lane in [0,n_lanes).  Nothing is copied back until svo_wait/svo_get_*. */
int svo_process(svo_ctx* ctx, const svo_frame* frames, uint32_t flags);
/* block until every enqueued frame has finished */
int svo_wait(svo_ctx* ctx);
/* Host-fed frames (the reference's contract: P:100-120 takes host images per call).  Uploads go through a ring of two
 * device buffers on a dedicated copy stream, so the upload of frame t+1 overlaps the kernels of frame t; stage 2 of a
 * frame waits for its own upload only.  svo_wait_upload blocks until every enqueued upload has left the host buffers.
 * svo_host_alloc / svo_host_free hand out page-locked memory to callers that have no HIP headers; svo_host_register /
 * svo_host_unregister page-lock memory the caller already owns (e.g. a camera driver's frame buffers). */
int svo_wait_upload(svo_ctx* ctx);
/* One stream alone is bound by launch latency, not by its kernels (~30 launches per frame).  With graphs enabled, the
 * kernel sequence of a svo_process call is captured once into a hipGraph -- per combination of stage flags, image ring slot
 * and dynamic thresholds -- and replayed by one graph launch afterwards.  Images (host or device) then always pass
 * through the context's two-slot ring, which gives the captured kernels fixed addresses.  Calls that run stage 1 on the
 * device (rectification maps / BGR input) or have kernel timing on fall back to plain launches.  Same results. */
int svo_use_graphs(svo_ctx* ctx, int enable);
int svo_host_alloc(size_t bytes, void** out);
int svo_host_free(void* p);
int svo_host_register(void* p, size_t bytes);
int svo_host_unregister(void* p);
/* TStereoOdometryResult of the last frame, per lane (H:235-264); implies svo_wait */
int svo_get_result(svo_ctx* ctx, int lane, svo_result* res);
int svo_get_results(svo_ctx* ctx, svo_result* res /* n_lanes entries */);
/* enqueue a device-to-device copy of the n_lanes result records into caller-owned device memory (e.g. the send
 * buffer of an RCCL all-gather of poses) on the context's stream; no host synchronisation */
int svo_copy_results_async(svo_ctx* ctx, void* dst_device, size_t bytes);

/* getValues (H:704-724) and friends.  which: 0 = current frame, 1 = previous frame; side: 0 left, 1 right.
 * Each returns the list length (possibly > cap; only min(len,cap) entries are written) or <0. */
int svo_get_keypoints(svo_ctx* ctx, int lane, int which, int side, svo_keypoint* kps, uint8_t* desc, int cap);
int svo_get_matches(svo_ctx* ctx, int lane, int which, svo_dmatch* m, int cap);
int svo_get_tracked(svo_ctx* ctx, int lane, svo_index_pair* t, int cap);          /* tracked_pairs[0] (H:823) */
int svo_get_residuals(svo_ctx* ctx, int lane, double* r, int cap);                /* result.out_residual */
int svo_get_outliers(svo_ctx* ctx, int lane, int32_t* idx, int cap);              /* result.outliers */
/* the same lists of one octave (the FAST+ORB mode keeps one list per x1/2 octave, H:760-764, H:799), the row table
 * of m_update_indexes (pyr_feats_index, H:763; img_h entries) and matches_lr_row_index (H:789; img_h + 1 entries) */
int svo_get_keypoints_oct(svo_ctx* ctx, int lane, int which, int side, int octave, svo_keypoint* kps, uint8_t* desc, int cap);
int svo_get_matches_oct(svo_ctx* ctx, int lane, int which, int octave, svo_dmatch* m, int cap);
int svo_get_tracked_oct(svo_ctx* ctx, int lane, int octave, svo_index_pair* t, int cap);
int svo_get_row_index(svo_ctx* ctx, int lane, int which, int side, int octave, int32_t* idx, int cap);
int svo_get_matches_row_index(svo_ctx* ctx, int lane, int which, int octave, int32_t* idx, int cap);
/* match-ID bookkeeping of params_general.vo_use_matches_ids: matches_IDs (H:794, getRefCurrentIDs H:701),
 * resetIds (H:684) and setThisFrameAsKF (H:675-683); result.tracked_feats_from_last_KF counts against the key frame */
int svo_get_match_ids(svo_ctx* ctx, int lane, int which, int octave, int32_t* ids, int cap);
/* getValues (H:704-724) in ONE device synchronisation: every list of one octave of one frame of one lane.  The caller
 * fills the array pointers (any may be NULL) and the two capacities; the call fills the four counts (the full list
 * lengths, possibly above the capacities; min(length, capacity) entries are written).  The lists are packed on the
 * device, copied to the context's page-locked staging in one transfer, and unpacked on the host: one stream
 * synchronisation per call where the individual getters above pay one or two each. */
typedef struct svo_values {
    svo_keypoint* left_kps;  uint8_t* left_desc;      /* cap_kps entries / cap_kps * 32 bytes */
    svo_keypoint* right_kps; uint8_t* right_desc;
    svo_dmatch* matches;     int32_t* match_ids;      /* cap_matches entries each */
    int32_t cap_kps, cap_matches;
    int32_t n_left, n_right, n_matches, n_ids;        /* out */
} svo_values;
int svo_get_values(svo_ctx* ctx, int lane, int which, int octave, svo_values* v);
int svo_reset_ids(svo_ctx* ctx, int lane);
int svo_set_this_frame_as_kf(svo_ctx* ctx, int lane);   /* H:675-683; SVO_ERR_STATE while the lane has no current frame with pairings */

/* the precomputed-data bypass (request_data.use_precomputed_data, H:214-218, P:131-162, P:219-251):
 * load caller-supplied features / pairings into a lane's current (which=0) or previous (which=1) frame. */
int svo_put_features(svo_ctx* ctx, int lane, int which, int side, const svo_keypoint* kps, const uint8_t* desc, int n,
                     int img_w, int img_h);
/* (svo_put_matches also builds the list's matches_lr_row_index (S3:425-445) from the LEFT keypoints put before it: the reference builds that
 * index in stage 3 only, which this path skips (P:219-251), and its windowed tracker (S4:517-530) would read an index nobody built.) */
int svo_put_matches(svo_ctx* ctx, int lane, int which, const svo_dmatch* m, int n);
int svo_put_tracked(svo_ctx* ctx, int lane, const svo_index_pair* t, int n);
/* request_data.precomputed_matches_ID (H:218, P:233-244): the IDs of the octave-0 pairings put before; m_last_match_ID
 * becomes their maximum (P:236-243).  Honoured by the pipeline only when vo_use_matches_ids is set (P:246-250). */
int svo_put_match_ids(svo_ctx* ctx, int lane, int which, const int32_t* ids, int n);
/* the same per OCTAVE (P:141-161 and P:219-244 loop over params_rectify.nOctaves lists): octave < svo_config.max_octaves and,
 * in the FAST+ORB mode, < nOctaves.  img_w / img_h stay the size of the octave-0 image.  svo_put_match_ids_oct: octave 0
 * restarts m_last_match_ID at its maximum ID, higher octaves raise it (call in ascending octave order, as P:236-243 does). */
int svo_put_features_oct(svo_ctx* ctx, int lane, int which, int side, int octave, const svo_keypoint* kps, const uint8_t* desc, int n,
                         int img_w, int img_h);
int svo_put_matches_oct(svo_ctx* ctx, int lane, int which, int octave, const svo_dmatch* m, int n);
int svo_put_match_ids_oct(svo_ctx* ctx, int lane, int which, int octave, const int32_t* ids, int n);

/* getProjectedCoords (H:175-182, C:415-466): pixel coordinates (uL vL uR vR, 4 floats each) that the previous
 * pairings NOT marked as tracked (tracked_first[m] == -1, C:430-431) take after the change in pose: triangulation as
 * in stage 5, then m_pinhole_stereo_projection with the inverse of change_pose (x y z yaw pitch roll).  Returns the
 * number B of such pairings; nothing is written when pix is NULL or cap < B (size query). */
int svo_projected_coords(svo_ctx* ctx, const svo_dmatch* pre_matches, int n_pre, const svo_keypoint* pre_left, int n_left,
                         const svo_keypoint* pre_right, int n_right, const int32_t* tracked_first,
                         const svo_stereo_camera* cam, const double* change_pose6, float* pix, int cap);

/* Frame-parallelism WITHIN one stream: consecutive frames of a stream dealt round-robin to several
 * contexts (one GPU or several).  Stages 2-3 of a frame need only its images; stages 4-5 need the previous frame's lists
 * and the members a call inherits from the one before (m_error for the recovery rule P:86-95, m_last_computed_pose
 * S5:506-507, the match-ID counters H:735-742).  The owner of frame t-1 exports them once its stages 4-5 are enqueued,
 * the owner of frame t imports them after its own stages 2-3 and before its stages 4-5:
 *     owner(t):   svo_process(frames, SVO_RUN_DETECT | SVO_RUN_MATCH);            // overlaps owner(t-1)'s stages 4-5
 *                 <wait for owner(t-1)'s export>  svo_import_frame(blob);
 *                 svo_process(NULL, SVO_RUN_TRACK | SVO_RUN_OPTIMIZE | SVO_FLAG_NO_SHIFT);  svo_export_frame(blob');
 * The record is one flat DEVICE buffer of svo_handover_bytes() bytes (all lanes): hand it over device-to-device, or with
 * ncclSend / ncclRecv between GPUs.  Both calls only enqueue on the context's stream; ordering between the two
 * contexts' streams is the caller's (an event, or the send/recv pair).  The run equals the sequential one list for
 * list and pose for pose: nothing is dropped, not even the warm start. */
size_t svo_handover_bytes(const svo_ctx* ctx);
int svo_export_frame(svo_ctx* ctx, void* dev_blob, size_t bytes);
int svo_import_frame(svo_ctx* ctx, const void* dev_blob, size_t bytes);

/* saveStateToFile / loadStateFromFile (H:184-185, C:475-543, C:261-350): the state of one lane in the reference's
 * binary layout -- npyr; then for PRE and CUR: left keypoints, right keypoints (count, then x y response size angle as
 * float and octave class_id as int per keypoint, then rows cols type and the descriptor bytes), pairings (count,
 * id count, then [id] queryIdx trainIdx distance imgIdx each); then m_reset (1 byte) and m_lastID,
 * m_num_tracked_pairs_from_last_kf, m_num_tracked_pairs_from_last_frame, m_last_match_ID, m_kf_max_match_ID as
 * 8-byte integers.  Octave-0 lists (single-octave contexts only).  svo_load_state reads what svo_save_state (and
 * the reference's saveStateToFile) writes; the reference's own loader expects one more 8-byte word before
 * m_last_match_ID that its saver never writes (C:342-343 vs C:533-539) -- deviation, see SURVEY.md appendix A #18.
 * After a load both frames are present, m_error is cleared and the warm start is the identity. */
int svo_save_state(svo_ctx* ctx, int lane, const char* path);
int svo_load_state(svo_ctx* ctx, int lane, const char* path);

/* getChangeInPose (H:162-172, C:355-413): stage 5 alone on caller arrays, lane 0 of the context.
 * residual: n_tracked doubles; outliers: n_tracked int32 (holds INLIER cur-match indices, S5:603-610).
 * init6 may be NULL.  Returns result.valid (0/1) or <0. */
int svo_change_in_pose(svo_ctx* ctx, const svo_index_pair* tracked, int n_tracked,
                       const svo_dmatch* pre_matches, int n_pre, const svo_dmatch* cur_matches, int n_cur,
                       const svo_keypoint* pre_left, int n_pl, const svo_keypoint* pre_right, int n_pr,
                       const svo_keypoint* cur_left, int n_cl, const svo_keypoint* cur_right, int n_cr,
                       const svo_stereo_camera* cam, const double* init6,
                       svo_result* res, double* residual, int32_t* outliers);

/* cv::BFMatcher(NORM_HAMMING,false).match stand-in (S3:88-94, S4:141-142) on caller arrays (host pointers):
 * for each of nq 32-byte query rows the FIRST minimum-distance train row.  idx = -1 when nt == 0. */
int svo_hamming_match(svo_ctx* ctx, const uint8_t* query, int nq, const uint8_t* train, int nt,
                      int32_t* idx, int32_t* dist);

/* debugging / parity probes */
int svo_debug_get_level(svo_ctx* ctx, int lane, int side, int level, uint8_t* out, int cap, int* w, int* h);
int svo_debug_get_raw_keypoints(svo_ctx* ctx, int lane, int side, svo_keypoint* kps, uint8_t* desc, int cap);
int svo_debug_get_status_word(svo_ctx* ctx, int lane, uint32_t* w);   /* capacity-overflow bits */
/* 1 when the reference's profiler sections (CTimeLogger names: "processNewImagePair", "_stg1" .. "_stg5", "stg3.find_pairings",
 * "stg4.track") are being emitted as roctx ranges around the enqueue of each stage: SVO_ROCTX=1 in the environment, or a rocprofv3
 * session (it sets ROCP_TOOL_LIBRARIES); the roctx library is looked up with dlopen, never linked. */
int svo_profiler_sections_enabled(void);
/* How often the speculative FAST threshold of k_fast failed since svo_create (or the last reset = 1 call): (image, level) pairs
 * that a frame had to run again at the caller's threshold (k_fast_redo + the second k_select pass).  Waits for the enqueued work. */
int svo_debug_get_redo_count(svo_ctx* ctx, uint32_t* pairs, int reset);
/* What was really in flight (SVO_TIMELINE=1 in the environment of svo_create; a measuring knob, off in production): every kernel stamps
 * the hull [first wave in, last wave out] of its launch on the device-wide 100 MHz wall clock.  out receives 4096 records of two uint64
 * (t0, t1; t0 = ~0: no such launch) for the last 16 frames: index = (frame % 16) * 256 + kind * 8 + aux, kinds in bench.py's
 * TIMELINE_KINDS order.  Returns the context's frame counter (0: the knob is off); reset = 1 clears the table.  Waits for the work. */
int svo_debug_timeline(svo_ctx* ctx, uint64_t* out, int cap_records, int reset);

/* per-kernel HIP-event timing (svo_config.kernel_times = 1): names[i] points to static storage.
 * total_ms[i] / calls[i] accumulate since the last svo_kernel_times_reset. Returns number of kernels. */
int svo_kernel_times(svo_ctx* ctx, const char** names, double* total_ms, int64_t* calls, int cap);
int svo_kernel_times_reset(svo_ctx* ctx);
/* time only the kernel of that name from now on (NULL or "": all again): two events per kernel cost a few percent of
 * a step, a benchmark that needs the live duration of one kernel need not pay for the others */
int svo_kernel_times_select(svo_ctx* ctx, const char* name);

/* sizeof() of the ABI records as this library was compiled: out[0..5] = keypoint, dmatch, stereo_camera,
 * params, result, config.  Lets a binding verify its mirror of svo_types.h. */
void svo_abi_sizes(int32_t* out6);

#ifdef __cplusplus
}
#endif
#endif
