// svo_kernels.h -- host-side launchers of the HIP kernels (defined in k_detect.hip, k_match.hip, k_gn.hip)
#pragma once
#include "svo_device.h"

struct GNParams {
    int use_robust_kernel, max_iters, initial_max_iters, max_incr_cost;
    int use_previous_pose_as_initial, use_custom_initial_pose;
    int min_distance, img_w, img_h;      // grid of the stage-5 NMS mask (S5:465-468)
    int pmax;                            // power of two >= max_kps (LDS carve)
    int standalone;                      // 1: getChangeInPose entry (no prev/bad-tracking gate)
    int pad;
    double kernel_param, min_mod_out_vector, residual_threshold;
    double init[6];
};

// Which kernel forms are the PRODUCT and which are kept as A/B anchors only.  Every form below is compiled and run
// against the oracle by tests/test_gpu_parity.py (test_every_ransac_kernel_form_gives_the_oracle_models,
// test_kernel_launch_knobs_do_not_change_results); the launchers pick the product form unless a debug knob says otherwise.
//   product, many lanes (> 8 lane-octaves):  k_hamming_f4, k_ransac_hyp_thread, k_ransac_count_mfma16
//   product, few lanes (one stream alone):   k_hamming_f4, k_ransac_hyp (16 lanes per sample), k_ransac_count<4> (VALU, for latency)
//   product, other entry points:             k_hamming_plain (svo_hamming_match), k_track_gate (multi-octave contexts), k_project_points
//   A/B only (never launched by default):    k_hamming + k_gather_mdesc (the int8 matcher, SVO_HAM_FP4=0), k_ransac_count_mfma (4 x 4 tiles,
//                                            SVO_DEBUG_MODE=52), k_ransac_count<16> (SVO_DEBUG_MODE=14).  Since round 6 these are compiled only
//                                            under -DSVO_AB_KERNELS: the product libsvo_hip.so does not contain them (svo_create refuses the
//                                            knobs there), libsvo_hip_ab.so does -- tests and A/B runs load it through SVO_HIP_LIB.
bool svo_ab_kernels_built();                   // k_match.hip: compiled with -DSVO_AB_KERNELS (libsvo_hip_ab.so)
bool svo_ab_form_requested(int debug_mode);    // SVO_HAM_FP4=0 or SVO_DEBUG_MODE 14 / 52 in the environment
hipError_t svo_upload_tables();
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, DEVICE) and is shared by every context of the process on that
// device: keyed by both, only ever raised -- a later, smaller context must not lower the limit under an earlier context's launches,
// and a host with one thread per GPU must set it on every device it uses (the calling thread's current device is the context's).
hipError_t svo_raise_dyn_smem(const void* kernel, size_t bytes);
// stage 1 (k_prepare.hip)
struct PrepArgs {
    const uint8_t* src[2 * SVO_MAX_LANES];   // source image pointers (device memory), by value in the kernel arguments
    const uint2* const* maps;         // device table [n_img] of fixed-point maps (nullptr entries: no rectification), or nullptr
    uint8_t* dst; long long dst_img_stride; int dst_pitch;
    long long src_stride; int channels, w, h;
};
void launch_prepare(const PrepArgs& a, int n_img, hipStream_t st);
// frame hand-over between contexts (k_handover.hip)
size_t handover_record_bytes(const DevCtx& c);
void launch_export_frame(const DevCtx& c, uint8_t* blob, hipStream_t st);
void launch_import_frame(const DevCtx& c, const uint8_t* blob, hipStream_t st);
void launch_pack_values(const DevCtx& c, int lane, int which, int octave, uint8_t* dst, hipStream_t st);
void launch_begin_frame(const DevCtx& c, const uint8_t* const* ptrs, unsigned flags, hipStream_t st);
void launch_resize(const DevCtx& c, int level, hipStream_t st);
void launch_fast(const DevCtx& c, hipStream_t st);
void launch_select(const DevCtx& c, hipStream_t st);
void launch_describe(const DevCtx& c, int pre, hipStream_t st);
hipError_t configure_nms_rowsort(const DevCtx& c);
size_t nms_rowsort_scratch_bytes(const DevCtx& c);
void launch_nms_rowsort(const DevCtx& c, int do_nms, int min_distance, int pre, hipStream_t st);
void launch_half(const DevCtx& c, int level, hipStream_t st);
void launch_fastorb_anms(const DevCtx& c, uint32_t* scratch3, hipStream_t st);     // scratch3: 3 x n_img x cand_total words
void launch_fastorb_nms(const DevCtx& c, int do_nms, int min_distance, hipStream_t st);
void launch_hamming(const DevCtx& c, int mode, int nsplit, hipStream_t st);
void launch_match_lr_filter(const DevCtx& c, int one_to_one, double max_y_diff, hipStream_t st);
void launch_track_filter(const DevCtx& c, hipStream_t st);
void launch_ransac_hyp(const DevCtx& c, int chunk, hipStream_t st);
void launch_ransac_count(const DevCtx& c, int chunk, hipStream_t st);
void launch_track_finalize(const DevCtx& c, int bad_tracking_th, int win_mode, hipStream_t st);
void launch_match_lr_rbr(const DevCtx& c, int one_to_one, double max_y_diff, double minimum_response, int max_distance, hipStream_t st);
void launch_track_win(const DevCtx& c, int win_w, int win_h, hipStream_t st);
void launch_match_ids(const DevCtx& c, unsigned flags, hipStream_t st);
void launch_hamming_plain(const uint8_t* q, int nq, const uint8_t* t, int nt, unsigned* out, int nsplit, hipStream_t st);
hipError_t configure_gauss_newton(int pmax);
size_t gn_scratch_bytes_per_lane(int pmax);
hipError_t configure_match(int max_kps);
void launch_project_points(const float* uvu, int n, const svo_stereo_camera& cam, const double* delta6, float* pix, hipStream_t st);
void launch_gauss_newton(const DevCtx& c, const GNParams& P, hipStream_t st);
