// demo_stereo_odometry.cpp -- the analogue of the reference's demo-stereo-odometry/demo-main.cpp on a synthetic
// sequence file: frame loop (D:210-220), pose chaining pose <- pose * (K * outPose * K^-1) (D:235-242) and the
// camera_pose.txt trace "x y z yaw pitch roll" with %.3f (D:251-253).
//
// Sequence file ("SVOSEQ1\0", written by tools/make_sequence.py): int32 W, H, F; float64 fx, cx, cy, baseline;
// then F x (left W*H bytes, right W*H bytes).
// usage: demo_stereo_odometry <sequence.svoseq> <camera_pose.txt> [orb_nfeats | --opt config.ini]
//        --opt: the reference's application INI (demo-main.cpp:46-48, 154-167), read by loadParamsFromConfigFileName
#include "../stereo_vo_amd/csrc/rso_estimator.hpp"
#include <cstdio>
#include <cstring>
#include <cstdlib>

static void mat_mul(const double* A, const double* B, double* C) {
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { double s = 0; for (int k = 0; k < 4; k++) s += A[4 * r + k] * B[4 * k + c]; C[4 * r + c] = s; }
}
static void rigid_inverse(const double* M, double* I) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) I[4 * r + c] = M[4 * c + r];
    for (int r = 0; r < 3; r++) I[4 * r + 3] = -(I[4 * r] * M[3] + I[4 * r + 1] * M[7] + I[4 * r + 2] * M[11]);
    I[12] = I[13] = I[14] = 0; I[15] = 1;
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <sequence.svoseq> <camera_pose.txt> [orb_nfeats | --opt config.ini]\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    char magic[8]; int32_t W, H, F; double fx, cx, cy, baseline;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "SVOSEQ1", 8) != 0 || std::fread(&W, 4, 1, f) != 1 || std::fread(&H, 4, 1, f) != 1 ||
        std::fread(&F, 4, 1, f) != 1 || std::fread(&fx, 8, 1, f) != 1 || std::fread(&cx, 8, 1, f) != 1 || std::fread(&cy, 8, 1, f) != 1 || std::fread(&baseline, 8, 1, f) != 1) {
        std::fprintf(stderr, "bad sequence header\n"); return 2;
    }
    try {
        rso::CStereoOdometryEstimator stereo_odom_engine(W, H);
        stereo_odom_engine.params.detect_method = SVO_DM_ORB;             // the north-star configuration
        const bool has_ini = argc > 4 && std::strcmp(argv[3], "--opt") == 0;
        stereo_odom_engine.params.orb_nfeats = (argc > 3 && !has_ini) ? std::atoi(argv[3]) : 500;
        stereo_odom_engine.params.match_method = SVO_SM_DESC_BF; stereo_odom_engine.params.max_y_diff = 1.0;
        stereo_odom_engine.params.enable_robust_1to1_match = 1; stereo_odom_engine.params.orb_max_distance = 60.0;
        stereo_odom_engine.params.ifm_method = SVO_IFM_DESC_BF;
        stereo_odom_engine.applyParams();
        if (has_ini) {                                                    // demo-main.cpp:154-167
            std::vector<std::string> paramSections;
            paramSections.push_back("RECTIFY"); paramSections.push_back("DETECT"); paramSections.push_back("MATCH"); paramSections.push_back("IF-MATCH");
            paramSections.push_back("LEAST_SQUARES"); paramSections.push_back("GUI"); paramSections.push_back("GENERAL");
            stereo_odom_engine.loadParamsFromConfigFileName(argv[4], paramSections);
        }
        rso::CStereoOdometryEstimator::TStereoOdometryRequest odom_request;
        rso::TStereoCamera& cam = odom_request.stereo_cam;
        cam.leftCamera.m_fx = cam.leftCamera.m_fy = cam.rightCamera.m_fx = cam.rightCamera.m_fy = fx;
        cam.leftCamera.m_cx = cam.rightCamera.m_cx = cx; cam.leftCamera.m_cy = cam.rightCamera.m_cy = cy;
        cam.leftCamera.ncols = cam.rightCamera.ncols = (unsigned)W; cam.leftCamera.nrows = cam.rightCamera.nrows = (unsigned)H;
        cam.rightCameraPose[0] = baseline;
        std::vector<uint8_t> L((size_t)W * H), R((size_t)W * H);
        FILE* out = std::fopen(argv[2], "wt");
        if (!out) { std::perror(argv[2]); return 2; }
        double pose[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        const double K[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };      // camera_pose_on_robot = identity (D:168-172)
        double Kinv[16]; rigid_inverse(K, Kinv);
        for (int t = 0; t < F; t++) {
            if (std::fread(L.data(), 1, L.size(), f) != L.size() || std::fread(R.data(), 1, R.size(), f) != R.size()) { std::fprintf(stderr, "short read\n"); return 2; }
            odom_request.imageLeft = rso::TGrayImage{ L.data(), W, H, (size_t)W };
            odom_request.imageRight = rso::TGrayImage{ R.data(), W, H, (size_t)W };
            rso::CStereoOdometryEstimator::TStereoOdometryResult odom_result;
            stereo_odom_engine.processNewImagePair(odom_request, odom_result);          // D:220
            if (odom_result.valid) {                                                     // D:222-244
                double D[16], T1[16], T2[16], P[16];
                odom_result.outPose.getHomogeneousMatrix(D);
                mat_mul(K, D, T1); mat_mul(T1, Kinv, T2); mat_mul(pose, T2, P);
                std::memcpy(pose, P, sizeof(P));
            }
            const rso::CPose3D p = rso::CPose3D::fromHomogeneousMatrix(pose);
            std::fprintf(out, "%.3f %.3f %.3f %.3f %.3f %.3f\n", p.x(), p.y(), p.z(), p.yaw(), p.pitch(), p.roll());   // D:251-253
            std::printf("Frame: %d valid=%d err=%d tracked=%zu it=%d/%d\n", t + 1, (int)odom_result.valid, (int)odom_result.error_code,
                        odom_result.tracked_feats_from_last_frame, odom_result.num_it, odom_result.num_it_final);
        }
        std::fclose(out);
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    std::fclose(f);
    return 0;
}
