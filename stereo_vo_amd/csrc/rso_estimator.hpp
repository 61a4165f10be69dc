// rso_estimator.hpp -- host-side mirror of the reference's public class over the HIP C-ABI.
//
// Same namespace, class, method and field names as libstereo-odometry/include/libstereo-odometry.h (H:147-1047),
// with the MRPT / OpenCV / Eigen types (absent in this environment) swapped for the POD records of
// include/svo_types.h.  Header-only; link against stereo_vo_amd/libsvo_hip.so.  One instance == one lane context
// (n_lanes = 1): exactly the reference's one-estimator-one-thread model (H:732-831).  Hard errors throw
// std::runtime_error where the reference uses THROW_EXCEPTION / ASSERT_ (P:54-81); soft errors come back in
// result.error_code with the reference's enum values (H:142).
#pragma once
#include <cstdint>
#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
extern "C" {
#include "../../include/svo_hip.h"
}

namespace rso {

typedef svo_keypoint KeyPoint;                    // == cv::KeyPoint layout (H:108)
typedef svo_dmatch DMatch;                        // == cv::DMatch layout   (H:109)
typedef std::vector<KeyPoint> TKeyPointList;      // H:108
typedef std::vector<DMatch> TDMatchList;          // H:109
typedef std::vector<std::pair<size_t, size_t> > vector_index_pairs_t;   // H:139

enum VOErrorCode { voecNone, voecBadCondNumber, voecIncrFuncCostStg1, voecIncrFuncCostStg2, voecFirstIteration, voecBadTracking };   // H:142

/** 8-bit gray, row-major, already rectified: what stage 1 (S1:47-85) hands to stage 2 */
struct TGrayImage { const uint8_t* data; int w, h; size_t stride; };

/** the fields of mrpt::utils::TStereoCamera the path reads */
struct TStereoCamera {
    struct TCamera { double m_fx, m_fy, m_cx, m_cy; unsigned ncols, nrows;
        double fx() const { return m_fx; } double fy() const { return m_fy; } double cx() const { return m_cx; } double cy() const { return m_cy; } };
    TCamera leftCamera, rightCamera;
    double rightCameraPose[7];                    // [0] = baseline (S5:185, 515)
    TStereoCamera() : leftCamera(), rightCamera() { for (double& v : rightCameraPose) v = 0; }
};

/** mrpt::utils::TPixelCoordf stand-in */
struct TPixelCoordf { float x, y; TPixelCoordf() : x(0), y(0) {} TPixelCoordf(float x_, float y_) : x(x_), y(y_) {} };

/** mrpt::poses::CPose3D stand-in: x y z yaw pitch roll, R = Rz(yaw) Ry(pitch) Rx(roll) */
struct CPose3D {
    double m_coords[3]; double m_yaw, m_pitch, m_roll;
    CPose3D() : m_yaw(0), m_pitch(0), m_roll(0) { m_coords[0] = m_coords[1] = m_coords[2] = 0; }
    CPose3D(double x, double y, double z, double yaw, double pitch, double roll) : m_yaw(yaw), m_pitch(pitch), m_roll(roll) { m_coords[0] = x; m_coords[1] = y; m_coords[2] = z; }
    double x() const { return m_coords[0]; } double y() const { return m_coords[1]; } double z() const { return m_coords[2]; }
    double yaw() const { return m_yaw; } double pitch() const { return m_pitch; } double roll() const { return m_roll; }
    void getHomogeneousMatrix(double M[16]) const {
        const double cy = std::cos(m_yaw), sy = std::sin(m_yaw), cp = std::cos(m_pitch), sp = std::sin(m_pitch), cr = std::cos(m_roll), sr = std::sin(m_roll);
        const double R[9] = { cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr };
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[4 * r + c] = R[3 * r + c]; M[4 * r + 3] = m_coords[r]; }
        M[12] = M[13] = M[14] = 0; M[15] = 1;
    }
    static CPose3D fromHomogeneousMatrix(const double M[16]) {
        const double pitch = std::atan2(-M[8], std::hypot(M[0], M[4]));
        double yaw, roll;
        if (std::fabs(M[9]) + std::fabs(M[10]) < 10 * 2.220446049250313e-16) { roll = 0; yaw = pitch > 0 ? std::atan2(M[6], M[2]) : std::atan2(-M[6], -M[2]); }
        else { roll = std::atan2(M[9], M[10]); yaw = std::atan2(M[4], M[0]); }
        return CPose3D(M[3], M[7], M[11], yaw, pitch, roll);
    }
};

class CStereoOdometryEstimator {
public:
    struct TStereoOdometryRequest {               // H:205-233
        TGrayImage imageLeft, imageRight;         // stereo_imgs (borrowed for the duration of the call)
        TStereoCamera stereo_cam;
        bool use_precomputed_data;
        std::vector<TKeyPointList>* precomputed_left_feats, *precomputed_right_feats;
        std::vector<std::vector<uint8_t> >* precomputed_left_desc, *precomputed_right_desc;   // N x 32 bytes per octave
        std::vector<TDMatchList>* precomputed_matches;
        std::vector<std::vector<size_t> >* precomputed_matches_ID;                         // H:218, honoured when vo_use_matches_ids (P:233-250)
        bool repeat;
        TStereoOdometryRequest() : imageLeft(), imageRight(), stereo_cam(), use_precomputed_data(false), precomputed_left_feats(NULL),
            precomputed_right_feats(NULL), precomputed_left_desc(NULL), precomputed_right_desc(NULL), precomputed_matches(NULL),
            precomputed_matches_ID(NULL), repeat(false) {}
    };
    struct TStereoOdometryResult {                // H:235-264
        CPose3D outPose;
        std::vector<size_t> outliers;             // holds INLIER cur-match indices (S5:603-610)
        std::vector<double> out_residual;
        int num_it, num_it_final;
        bool valid;
        VOErrorCode error_code;
        size_t tracked_feats_from_last_KF, tracked_feats_from_last_frame;
        std::vector<std::pair<size_t, size_t> > detected_feats;
        std::vector<size_t> stereo_matches;
        TStereoOdometryResult() : num_it(0), num_it_final(0), valid(false), error_code(voecNone), tracked_feats_from_last_KF(0), tracked_feats_from_last_frame(0) {}
    };
    // the seven parameter groups (H:266-508) collapse into the flat record whose fields carry the reference's names
    svo_params params;

    explicit CStereoOdometryEstimator(int max_w = 1280, int max_h = 960, int device = 0, int max_octaves = 4) : m_ctx(NULL), m_verbose_level(1) {
        svo_params_defaults(&params);
        svo_config cfg; svo_config_defaults(&cfg);
        cfg.device = device; cfg.n_lanes = 1; cfg.max_w = max_w; cfg.max_h = max_h; cfg.max_octaves = max_octaves;
        const int rc = svo_create(&cfg, &m_ctx);
        if (rc != SVO_OK) { std::string msg = std::string("svo_create: ") + svo_strerror(rc) + " " + (m_ctx ? svo_last_error(m_ctx) : ""); if (m_ctx) svo_destroy(m_ctx); m_ctx = NULL; throw std::runtime_error(msg); }
    }
    ~CStereoOdometryEstimator() { if (m_ctx) svo_destroy(m_ctx); }
    CStereoOdometryEstimator(const CStereoOdometryEstimator&) = delete;
    CStereoOdometryEstimator& operator=(const CStereoOdometryEstimator&) = delete;

    /** loadParamsFromConfigFile's effect (H:554-663): push `params`, then reset both dynamic thresholds */
    void applyParams() { check(svo_set_params(m_ctx, &params), "svo_set_params"); }
    /** Loads configuration from an INI file from its name (H:665-672); sections in the order RECTIFY, DETECT, MATCH, IF-MATCH,
      * LEAST_SQUARES, GUI, GENERAL (H:551-553), an empty name skips the group */
    void loadParamsFromConfigFileName(const std::string& fileName, const std::vector<std::string>& sections) {
        if (sections.size() != 7) throw std::runtime_error("loadParamsFromConfigFile: seven section names expected");    // H:556
        const char* names[7];
        for (int i = 0; i < 7; i++) names[i] = sections[i].c_str();
        check(svo_params_load_ini(fileName.c_str(), names, &params), "svo_params_load_ini");
        applyParams();                                                                   // resetFASTThreshold / resetORBThreshold, H:662-663
    }
    void setVerbosityLevel(int level) { m_verbose_level = level; }                       // H:527
    int getFASTThreshold() { return svo_get_fast_threshold(m_ctx); }                     // H:530
    void setFASTThreshold(int v) { check(svo_set_fast_threshold(m_ctx, v), "svo_set_fast_threshold"); }   // H:531
    void resetFASTThreshold() { setFASTThreshold(params.initial_FAST_threshold); }       // H:532
    int getORBThreshold() { return svo_get_orb_threshold(m_ctx); }                       // H:537
    void setORBThreshold(int v) { check(svo_set_orb_threshold(m_ctx, v), "svo_set_orb_threshold"); }      // H:538
    void resetORBThreshold() { setORBThreshold((int)params.orb_max_distance); }          // H:539

    /** The main entry point (H:157-159, P:41-385) */
    void processNewImagePair(TStereoOdometryRequest& request_data, TStereoOdometryResult& result) {
        set_camera(request_data.stereo_cam);
        uint32_t flags = SVO_RUN_ALL | (request_data.repeat ? SVO_FLAG_REPEAT : 0);
        svo_frame f;
        f.left.data = request_data.imageLeft.data; f.left.w = request_data.imageLeft.w; f.left.h = request_data.imageLeft.h; f.left.stride = (int64_t)request_data.imageLeft.stride;
        f.right.data = request_data.imageRight.data; f.right.w = request_data.imageRight.w; f.right.h = request_data.imageRight.h; f.right.stride = (int64_t)request_data.imageRight.stride;
        if (request_data.use_precomputed_data) {                                         // P:131-162, 219-251
            if (!request_data.precomputed_left_feats || !request_data.precomputed_right_feats || !request_data.precomputed_left_desc ||
                !request_data.precomputed_right_desc || !request_data.precomputed_matches) throw std::runtime_error("precomputed data pointers must be set");   // P:137, 157, 221
            // shift prev/cur first (P:86-100; request_data.repeat and the recovery rule apply here as in the full path), then
            // load the caller's lists, octave by octave (P:141-161, 219-244), into the new current frame
            check(svo_process(m_ctx, NULL, request_data.repeat ? SVO_FLAG_REPEAT : 0), "svo_process(shift)");
            const int w = request_data.imageLeft.w, h = request_data.imageLeft.h;
            const size_t nOct = request_data.precomputed_left_feats->size();
            if (request_data.precomputed_right_feats->size() != nOct || request_data.precomputed_left_desc->size() != nOct ||
                request_data.precomputed_right_desc->size() != nOct || request_data.precomputed_matches->size() != nOct)
                throw std::runtime_error("Number of octaves in precomputed data does not match");                          // P:138-139
            if (params.vo_use_matches_ids && (!request_data.precomputed_matches_ID || request_data.precomputed_matches_ID->size() != nOct))
                throw std::runtime_error("precomputed_matches_ID must be set when vo_use_matches_ids is on");                // P:235
            for (size_t o = 0; o < nOct; o++) {
                const TKeyPointList& kl = (*request_data.precomputed_left_feats)[o], &kr = (*request_data.precomputed_right_feats)[o];
                check(svo_put_features_oct(m_ctx, 0, 0, 0, (int)o, kl.data(), (*request_data.precomputed_left_desc)[o].data(), (int)kl.size(), w, h), "svo_put_features_oct");
                check(svo_put_features_oct(m_ctx, 0, 0, 1, (int)o, kr.data(), (*request_data.precomputed_right_desc)[o].data(), (int)kr.size(), w, h), "svo_put_features_oct");
                const TDMatchList& m = (*request_data.precomputed_matches)[o];
                check(svo_put_matches_oct(m_ctx, 0, 0, (int)o, m.data(), (int)m.size()), "svo_put_matches_oct");
                if (params.vo_use_matches_ids) {
                    const std::vector<size_t>& idv = (*request_data.precomputed_matches_ID)[o];
                    std::vector<int32_t> ids(idv.begin(), idv.end());
                    check(svo_put_match_ids_oct(m_ctx, 0, 0, (int)o, ids.data(), (int)ids.size()), "svo_put_match_ids_oct");
                }
            }
            check(svo_process(m_ctx, NULL, SVO_RUN_TRACK | SVO_RUN_OPTIMIZE | SVO_FLAG_NO_SHIFT), "svo_process");
        } else {
            if (!f.left.data || !f.right.data) throw std::runtime_error("Pointer 'request_data.stereo_imgs' must be set to stereo observation data!");   // P:81
            check(svo_process(m_ctx, &f, flags), "svo_process");
        }
        fill_result(result);
    }

    /** getChangeInPose (H:162-172, C:355-413) */
    bool getChangeInPose(const vector_index_pairs_t& tracked_pairs, const TDMatchList& pre_matches, const TDMatchList& cur_matches,
                         const TKeyPointList& pre_left_feats, const TKeyPointList& pre_right_feats,
                         const TKeyPointList& cur_left_feats, const TKeyPointList& cur_right_feats,
                         const TStereoCamera& stereo_camera, TStereoOdometryResult& result,
                         const std::vector<double>& ini_estimation = std::vector<double>(6, 0)) {
        std::vector<svo_index_pair> t(tracked_pairs.size());
        for (size_t i = 0; i < t.size(); i++) { t[i].first = (int32_t)tracked_pairs[i].first; t[i].second = (int32_t)tracked_pairs[i].second; }
        const svo_stereo_camera cam = to_cam(stereo_camera);
        svo_result r; std::vector<double> res(t.size() + 1); std::vector<int32_t> outl(t.size() + 1);
        const int rc = svo_change_in_pose(m_ctx, t.data(), (int)t.size(), pre_matches.data(), (int)pre_matches.size(), cur_matches.data(), (int)cur_matches.size(),
                                          pre_left_feats.data(), (int)pre_left_feats.size(), pre_right_feats.data(), (int)pre_right_feats.size(),
                                          cur_left_feats.data(), (int)cur_left_feats.size(), cur_right_feats.data(), (int)cur_right_feats.size(),
                                          &cam, ini_estimation.size() == 6 ? ini_estimation.data() : NULL, &r, res.data(), outl.data());
        check(rc, "svo_change_in_pose");
        from_record(r, result);
        result.out_residual.assign(res.begin(), res.begin() + r.n_residual);
        result.outliers.assign(outl.begin(), outl.begin() + r.n_outliers);
        return result.valid;
    }

    /** getProjectedCoords (H:175-182, C:415-466): coordinates of the previous pairings that were NOT tracked elsewhere
     *  (other_matches_tracked[m].first == -1) after the change in pose; pro_pre_feats gets (uL, vL), (uR, vR) pairs */
    void getProjectedCoords(const TDMatchList& pre_matches, const TKeyPointList& pre_left_feats, const TKeyPointList& pre_right_feats,
                            const std::vector<std::pair<int, float> >& other_matches_tracked, const TStereoCamera& stereo_camera,
                            const CPose3D& change_pose, std::vector<std::pair<TPixelCoordf, TPixelCoordf> >& pro_pre_feats) {
        std::vector<int32_t> first(other_matches_tracked.size());
        for (size_t i = 0; i < first.size(); i++) first[i] = other_matches_tracked[i].first;
        const svo_stereo_camera cam = to_cam(stereo_camera);
        const double pose[6] = { change_pose.x(), change_pose.y(), change_pose.z(), change_pose.yaw(), change_pose.pitch(), change_pose.roll() };
        std::vector<float> pix(4 * pre_matches.size() + 4);
        const int n = check(svo_projected_coords(m_ctx, pre_matches.data(), (int)pre_matches.size(), pre_left_feats.data(), (int)pre_left_feats.size(),
                                                 pre_right_feats.data(), (int)pre_right_feats.size(), first.data(), &cam, pose, pix.data(), (int)pre_matches.size() + 1), "svo_projected_coords");
        pro_pre_feats.resize((size_t)n);
        for (int i = 0; i < n; i++) { pro_pre_feats[i].first = TPixelCoordf(pix[4 * i], pix[4 * i + 1]); pro_pre_feats[i].second = TPixelCoordf(pix[4 * i + 2], pix[4 * i + 3]); }
    }

    /** getValues (H:704-724): copies of the current frame's octave-0 lists */
    void getValues(TKeyPointList& leftKP, TKeyPointList& rightKP, std::vector<uint8_t>& leftDesc, std::vector<uint8_t>& rightDesc, TDMatchList& matches,
                   std::vector<size_t>& matches_id) {
        // one device synchronisation for all six lists (svo_get_values); a second call only if a list outgrew the guess
        int cap_k = 4096, cap_m = 4096;
        for (int pass = 0; pass < 2; pass++) {
            leftKP.resize(cap_k); rightKP.resize(cap_k); leftDesc.resize((size_t)cap_k * 32); rightDesc.resize((size_t)cap_k * 32); matches.resize(cap_m);
            std::vector<int32_t> ids((size_t)cap_m);
            svo_values v; v.left_kps = leftKP.data(); v.left_desc = leftDesc.data(); v.right_kps = rightKP.data(); v.right_desc = rightDesc.data();
            v.matches = matches.data(); v.match_ids = ids.data(); v.cap_kps = cap_k; v.cap_matches = cap_m;
            check(svo_get_values(m_ctx, 0, 0, 0, &v), "svo_get_values");
            if (v.n_left > cap_k || v.n_right > cap_k || v.n_matches > cap_m || v.n_ids > cap_m) { cap_k = v.n_left > v.n_right ? v.n_left : v.n_right; cap_m = v.n_matches > v.n_ids ? v.n_matches : v.n_ids; continue; }
            leftKP.resize(v.n_left); leftDesc.resize((size_t)v.n_left * 32); rightKP.resize(v.n_right); rightDesc.resize((size_t)v.n_right * 32);
            matches.resize(v.n_matches); matches_id.assign(ids.begin(), ids.begin() + v.n_ids);
            break;
        }
    }
    /** Stage 1 (S1:47-85): what m_stereo_rectifier.setFromCamParams() precomputes, handed over as float maps [h][w] of
     *  source coordinates (cv::initUndistortRectifyMap, CV_32FC1); empty vectors = areImagesRectified() (S1:61-65). */
    void setRectifyMaps(int side, const std::vector<float>& map_x, const std::vector<float>& map_y, int w, int h) {
        const bool clear = map_x.empty();
        check(svo_set_rectify_map(m_ctx, 0, side, clear ? NULL : map_x.data(), clear ? NULL : map_y.data(), w, h), "svo_set_rectify_map");
    }
    /** saveStateToFile / loadStateFromFile (H:184-185, C:475-543, C:261-350): false on error, like the reference */
    bool saveStateToFile(const std::string& filename) { return svo_save_state(m_ctx, 0, filename.c_str()) == SVO_OK; }
    bool loadStateFromFile(const std::string& filename) { return svo_load_state(m_ctx, 0, filename.c_str()) == SVO_OK; }
    void resetIds() { check(svo_reset_ids(m_ctx, 0), "svo_reset_ids"); }                         // H:684
    void setThisFrameAsKF() { check(svo_set_this_frame_as_kf(m_ctx, 0), "svo_set_this_frame_as_kf"); }   // H:675-683

    svo_ctx* handle() { return m_ctx; }

private:
    svo_ctx* m_ctx;
    int m_verbose_level;

    int check(int rc, const char* what) {
        if (rc < 0) throw std::runtime_error(std::string(what) + ": " + svo_strerror(rc) + " " + svo_last_error(m_ctx));
        return rc;
    }
    static svo_stereo_camera to_cam(const TStereoCamera& c) {
        svo_stereo_camera o;
        o.l_fx = c.leftCamera.fx(); o.l_fy = c.leftCamera.fy(); o.l_cx = c.leftCamera.cx(); o.l_cy = c.leftCamera.cy();
        o.r_fx = c.rightCamera.fx(); o.r_fy = c.rightCamera.fy(); o.r_cx = c.rightCamera.cx(); o.r_cy = c.rightCamera.cy();
        o.baseline = c.rightCameraPose[0]; o.ncols = (int32_t)c.leftCamera.ncols; o.nrows = (int32_t)c.leftCamera.nrows;
        return o;
    }
    void set_camera(const TStereoCamera& c) { const svo_stereo_camera o = to_cam(c); check(svo_set_camera(m_ctx, 0, &o), "svo_set_camera"); }
    static void from_record(const svo_result& r, TStereoOdometryResult& result) {
        result.outPose = CPose3D(r.outPose[0], r.outPose[1], r.outPose[2], r.outPose[3], r.outPose[4], r.outPose[5]);
        result.num_it = r.num_it; result.num_it_final = r.num_it_final; result.valid = r.valid != 0; result.error_code = (VOErrorCode)r.error_code;
        result.tracked_feats_from_last_KF = (size_t)r.tracked_feats_from_last_KF; result.tracked_feats_from_last_frame = (size_t)r.tracked_feats_from_last_frame;
        const int no = r.n_octaves > 0 ? (r.n_octaves < 4 ? r.n_octaves : 4) : 1;
        result.detected_feats.clear(); result.stereo_matches.clear();
        for (int o = 0; o < no; o++) { result.detected_feats.push_back(std::make_pair((size_t)r.detected_left[o], (size_t)r.detected_right[o])); result.stereo_matches.push_back((size_t)r.stereo_matches[o]); }
    }
    void fill_result(TStereoOdometryResult& result) {
        svo_result r; check(svo_get_result(m_ctx, 0, &r), "svo_get_result");
        from_record(r, result);
        result.out_residual.resize((size_t)r.n_residual);
        if (r.n_residual) check(svo_get_residuals(m_ctx, 0, result.out_residual.data(), r.n_residual), "svo_get_residuals");
        std::vector<int32_t> o((size_t)r.n_outliers);
        if (r.n_outliers) check(svo_get_outliers(m_ctx, 0, o.data(), r.n_outliers), "svo_get_outliers");
        result.outliers.assign(o.begin(), o.end());
    }
    void fetch_kps(int side, TKeyPointList& k, std::vector<uint8_t>& d) {
        const int n = check(svo_get_keypoints(m_ctx, 0, 0, side, NULL, NULL, 0), "svo_get_keypoints");
        k.resize(n); d.resize((size_t)n * 32);
        if (n) check(svo_get_keypoints(m_ctx, 0, 0, side, k.data(), d.data(), n), "svo_get_keypoints");
    }
};

}  // namespace rso
