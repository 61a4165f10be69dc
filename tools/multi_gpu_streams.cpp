// multi_gpu_streams.cpp -- BASELINE.json configs[3] as a C++ host: G independent stereo streams, one
// rso::CStereoOdometryEstimator per GPU on one host thread each (the reference's threading model: an estimator is not
// thread-safe and shares nothing with another, libstereo-odometry.h:732-831), and ONE all-gather of the fixed-size result
// records per frame so that every rank sees every stream's pose (include/svo_rccl.h, RCCL over xGMI).
// Every rank chains all G trajectories from what it RECEIVED (pose <- pose * outPose, demo-main.cpp:235-242) and rank 0
// writes them as <prefix>_stream<k>.txt in the demo's format (D:251-253): equal to G separate runs of
// tools/demo_stereo_odometry if and only if the gather carried every record.
//
// usage: multi_gpu_streams [--gpus G] [--same-device] [--gather rccl|host] [--nfeats N] [--out PREFIX] a.svoseq [b.svoseq ...]
//   stream k plays sequence file k % n_files.  --same-device puts every rank on GPU 0 (a one-GPU box; RCCL refuses two
//   ranks on one device, so that needs --gather host: the same records exchanged through host memory and a barrier).
#include "../stereo_vo_amd/csrc/rso_estimator.hpp"
#include "../include/svo_rccl.h"
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct Sequence { int32_t W = 0, H = 0, F = 0; double fx = 0, cx = 0, cy = 0, baseline = 0; std::vector<uint8_t> px; };

static bool load_sequence(const char* path, Sequence& s)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); return false; }
    char magic[8];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "SVOSEQ1", 8) == 0 && std::fread(&s.W, 4, 1, f) == 1 && std::fread(&s.H, 4, 1, f) == 1 &&
              std::fread(&s.F, 4, 1, f) == 1 && std::fread(&s.fx, 8, 1, f) == 1 && std::fread(&s.cx, 8, 1, f) == 1 && std::fread(&s.cy, 8, 1, f) == 1 && std::fread(&s.baseline, 8, 1, f) == 1;
    if (ok) { s.px.resize((size_t)2 * s.W * s.H * s.F); ok = std::fread(s.px.data(), 1, s.px.size(), f) == s.px.size(); }
    std::fclose(f);
    if (!ok) std::fprintf(stderr, "%s: bad sequence file\n", path);
    return ok;
}

static void mat_mul(const double* A, const double* B, double* C) {
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { double s = 0; for (int k = 0; k < 4; k++) s += A[4 * r + k] * B[4 * k + c]; C[4 * r + c] = s; }
}

struct Shared {
    int G = 1, nfeats = 500, frames = 0; bool same_device = false, use_rccl = true;
    std::vector<Sequence> seqs; std::string prefix;
    svo_group* group = nullptr;
    std::vector<svo_result> host_exchange;             // --gather host: one record per rank
    pthread_barrier_t barrier;
    std::vector<int> rc; std::vector<std::string> err;
    std::vector<std::vector<svo_result>> seen;          // per rank: frames x G records as received
};

static void rank_main(Shared* sh, int rank)
{
    const int G = sh->G, dev = sh->same_device ? 0 : rank;
    const Sequence& seq = sh->seqs[(size_t)rank % sh->seqs.size()];
    try {
        if (hipSetDevice(dev) != hipSuccess) throw std::runtime_error("hipSetDevice");
        rso::CStereoOdometryEstimator est(seq.W, seq.H, dev);
        est.params.detect_method = SVO_DM_ORB; est.params.orb_nfeats = sh->nfeats;
        est.params.match_method = SVO_SM_DESC_BF; est.params.max_y_diff = 1.0; est.params.enable_robust_1to1_match = 1; est.params.orb_max_distance = 60.0;
        est.params.ifm_method = SVO_IFM_DESC_BF;
        est.applyParams();
        est.setVerbosityLevel(0);
        rso::CStereoOdometryEstimator::TStereoOdometryRequest req;
        rso::TStereoCamera& cam = req.stereo_cam;
        cam.leftCamera.m_fx = cam.leftCamera.m_fy = cam.rightCamera.m_fx = cam.rightCamera.m_fy = seq.fx;
        cam.leftCamera.m_cx = cam.rightCamera.m_cx = seq.cx; cam.leftCamera.m_cy = cam.rightCamera.m_cy = seq.cy;
        cam.leftCamera.ncols = cam.rightCamera.ncols = (unsigned)seq.W; cam.leftCamera.nrows = cam.rightCamera.nrows = (unsigned)seq.H;
        cam.rightCameraPose[0] = seq.baseline;
        void* d_records = nullptr; svo_result* h_records = nullptr; void* stream = nullptr;
        const size_t bytes = (size_t)G * sizeof(svo_result);
        if (hipMalloc(&d_records, bytes) != hipSuccess || hipHostMalloc((void**)&h_records, bytes, hipHostMallocDefault) != hipSuccess) throw std::runtime_error("allocating the gather buffers");
        if (svo_get_stream(est.handle(), &stream) != SVO_OK) throw std::runtime_error("svo_get_stream");
        const size_t img = (size_t)seq.W * seq.H;
        for (int t = 0; t < sh->frames; t++) {
            req.imageLeft = rso::TGrayImage{ seq.px.data() + (size_t)2 * t * img, seq.W, seq.H, (size_t)seq.W };
            req.imageRight = rso::TGrayImage{ seq.px.data() + (size_t)(2 * t + 1) * img, seq.W, seq.H, (size_t)seq.W };
            rso::CStereoOdometryEstimator::TStereoOdometryResult res;
            est.processNewImagePair(req, res);
            if (sh->use_rccl) {
                const int rc = svo_group_allgather_results(sh->group, rank, est.handle(), nullptr, d_records, bytes);
                if (rc != SVO_OK) throw std::runtime_error(std::string("svo_group_allgather_results: ") + svo_group_last_error(sh->group));
                if (hipMemcpyAsync(h_records, d_records, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                    throw std::runtime_error("reading the gathered records");
            } else {
                if (svo_get_result(est.handle(), 0, &sh->host_exchange[(size_t)rank]) != SVO_OK) throw std::runtime_error("svo_get_result");
                pthread_barrier_wait(&sh->barrier);
                std::memcpy(h_records, sh->host_exchange.data(), bytes);
                pthread_barrier_wait(&sh->barrier);                           // nobody overwrites its slot before all have read
            }
            // what came back for OUR slot must be what the estimator handed to its caller
            const svo_result& mine = h_records[rank];
            if ((mine.valid != 0) != res.valid || mine.error_code != (int)res.error_code || mine.outPose[0] != res.outPose.x() || mine.outPose[3] != res.outPose.yaw())
                throw std::runtime_error("gathered record differs from the local result");
            for (int k = 0; k < G; k++) sh->seen[(size_t)rank].push_back(h_records[k]);
        }
        (void)hipFree(d_records); (void)hipHostFree(h_records);
        sh->rc[(size_t)rank] = 0;
    } catch (const std::exception& e) {
        sh->rc[(size_t)rank] = 1; sh->err[(size_t)rank] = e.what();
        std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
        std::exit(1);                                                         // the others would wait for this rank forever
    }
}

int main(int argc, char** argv)
{
    // HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); streams that share one never overlap.
    // Before the first HIP call, and only if the caller has not chosen:
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    Shared sh; sh.prefix = "camera_pose";
    std::vector<const char*> files;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--gpus" && i + 1 < argc) sh.G = std::atoi(argv[++i]);
        else if (a == "--same-device") sh.same_device = true;
        else if (a == "--gather" && i + 1 < argc) sh.use_rccl = std::string(argv[++i]) == "rccl";
        else if (a == "--nfeats" && i + 1 < argc) sh.nfeats = std::atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) sh.prefix = argv[++i];
        else files.push_back(argv[i]);
    }
    if (files.empty() || sh.G < 1) { std::fprintf(stderr, "usage: %s [--gpus G] [--same-device] [--gather rccl|host] [--nfeats N] [--out PREFIX] a.svoseq [b.svoseq ...]\n", argv[0]); return 2; }
    if (sh.same_device && sh.use_rccl && sh.G > 1) { std::fprintf(stderr, "--same-device with more than one rank needs --gather host (RCCL refuses two ranks on one device)\n"); return 2; }
    sh.seqs.resize(files.size());
    for (size_t i = 0; i < files.size(); i++) if (!load_sequence(files[i], sh.seqs[i])) return 2;
    sh.frames = sh.seqs[0].F;
    for (const Sequence& s : sh.seqs) sh.frames = s.F < sh.frames ? s.F : sh.frames;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || (!sh.same_device && ndev < sh.G)) { std::fprintf(stderr, "%d GPU(s) visible, %d wanted\n", ndev, sh.G); return 3; }
    if (sh.use_rccl) {
        std::vector<int> devs((size_t)sh.G);
        for (int r = 0; r < sh.G; r++) devs[(size_t)r] = r;
        const int rc = svo_group_create_local(devs.data(), sh.G, &sh.group);
        if (rc != SVO_OK) { std::fprintf(stderr, "svo_group_create_local: %s\n", sh.group ? svo_group_last_error(sh.group) : "bad arguments"); return 3; }
    }
    const int comm_count = sh.group ? svo_group_comm_count(sh.group, 0) : 0;          // what RCCL itself says it connected
    sh.host_exchange.resize((size_t)sh.G); sh.rc.assign((size_t)sh.G, -1); sh.err.resize((size_t)sh.G); sh.seen.resize((size_t)sh.G);
    pthread_barrier_init(&sh.barrier, nullptr, (unsigned)sh.G);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int r = 0; r < sh.G; r++) th.emplace_back(rank_main, &sh, r);
    for (std::thread& t : th) t.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (sh.group) svo_group_destroy(sh.group);
    // every rank must have seen the same table
    for (int r = 1; r < sh.G; r++)
        if (sh.seen[(size_t)r].size() != sh.seen[0].size() || std::memcmp(sh.seen[(size_t)r].data(), sh.seen[0].data(), sh.seen[0].size() * sizeof(svo_result)) != 0) { std::fprintf(stderr, "rank %d saw a different table than rank 0\n", r); return 1; }
    for (int k = 0; k < sh.G; k++) {
        const std::string path = sh.prefix + "_stream" + std::to_string(k) + ".txt";
        FILE* out = std::fopen(path.c_str(), "wt");
        if (!out) { std::perror(path.c_str()); return 2; }
        double pose[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        for (int t = 0; t < sh.frames; t++) {
            const svo_result& r = sh.seen[0][(size_t)t * sh.G + k];
            if (r.valid) {
                double D[16], P[16];
                rso::CPose3D(r.outPose[0], r.outPose[1], r.outPose[2], r.outPose[3], r.outPose[4], r.outPose[5]).getHomogeneousMatrix(D);
                mat_mul(pose, D, P); std::memcpy(pose, P, sizeof(P));
            }
            const rso::CPose3D p = rso::CPose3D::fromHomogeneousMatrix(pose);
            std::fprintf(out, "%.3f %.3f %.3f %.3f %.3f %.3f\n", p.x(), p.y(), p.z(), p.yaw(), p.pitch(), p.roll());
        }
        std::fclose(out);
    }
    std::printf("{\"ranks\": %d, \"gather\": \"%s\", \"rccl_comm_count\": %d, \"tables_equal\": true, \"frames\": %d, \"seconds\": %.4f, \"pairs_per_s\": %.2f}\n", sh.G,
                sh.use_rccl ? "rccl" : "host", comm_count, sh.frames, secs, (double)sh.frames * sh.G / secs);
    return 0;
}
