// batch_streams.cpp -- the BENCHMARKED schedule driven from a C++ host: contexts x lanes independent stereo streams per GPU in
// the pipelined two-stream schedule of include/svo_batch.h, frames resident in HBM, one svo_batch_step per frame of every
// stream.  The reference's caller is a C++ loop around one estimator (demo-stereo-odometry/demo-main.cpp:210-220); this is
// that loop for as many estimators as an MI355X takes (all estimator state is per instance, libstereo-odometry.h:732-831).
//
// usage: batch_streams [--contexts C] [--lanes L] [--steps K] [--warmup W] [--nfeats N] [--gpus G] [--device D] [--gather rccl|none]
//                      [--dump PREFIX] a.svoseq [b.svoseq ...]
//   Stream s plays sequence file s % n_files ping-pong, starting s / n_files frames in, so that streams sharing a file still
//   differ.  --gpus G: one host thread and one batch per GPU; with --gather rccl the result records of all G x C x L streams
//   are all-gathered every step over RCCL (BASELINE.json configs[3] at batch size), and rank 0 reports what ncclCommCount /
//   the gathered table say.  Prints one JSON line: pairs/s over the K timed steps (host clock between device synchronisations).
//   --dump PREFIX (one GPU): afterwards the batch is reset and the same schedule replayed with a synchronisation per step; the
//   probe streams' result records of every step and their final lists go to PREFIX.bin for a checker (tests/test_gpu_batch_host.py
//   compares them with the CPU oracle).
#include "../include/svo_batch.h"
#ifdef SVO_WITH_RCCL
#include "../include/svo_rccl.h"
#else
// built on a host without RCCL (stereo_vo_amd/csrc/Makefile): the one-GPU batched schedule is all there is; --gather rccl is refused
struct svo_group;
static int svo_group_create_local(const int*, int, svo_group** out) { if (out) *out = nullptr; return SVO_ERR_UNSUPPORTED; }
static const char* svo_group_last_error(const svo_group*) { return "this binary was built without RCCL"; }
static int svo_group_comm_count(const svo_group*, int) { return 0; }
static void svo_group_destroy(svo_group*) {}
static int svo_group_allgather_inplace(svo_group*, int, void*, size_t, void*) { return SVO_ERR_UNSUPPORTED; }
#endif
#include <hip/hip_runtime.h>
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

static int ping_pong(int step, int F) { if (F <= 1) return 0; const int period = 2 * (F - 1), k = step % period; return k < F ? k : period - k; }

struct Options {
    int contexts = 2, lanes = 96, steps = 20, warmup = 4, nfeats = 2000, gpus = 1, device = 0; bool rccl = false; std::string dump;
    std::vector<const char*> files;
};

struct Rank {
    int rank = 0, device = 0; const Options* opt = nullptr; const std::vector<Sequence>* seqs = nullptr; svo_group* group = nullptr;
    int rc = -1; std::string err; double seconds = 0; int valid_last = 0; std::vector<svo_result> table;      // gathered records of the last step
};

#define CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { r->err = std::string(#expr) + ": " + hipGetErrorString(_e); return; } } while (0)
#define CHECK_B(expr) do { int _rc = (expr); if (_rc < 0) { r->err = std::string(#expr) + ": " + svo_strerror(_rc) + " [" + svo_batch_last_error(b) + "]"; return; } } while (0)

static void set_north_star(svo_params& p, int nfeats)
{
    svo_params_defaults(&p);
    p.detect_method = SVO_DM_ORB; p.orb_nfeats = nfeats; p.orb_nlevels = 8;
    p.match_method = SVO_SM_DESC_BF; p.max_y_diff = 1.0; p.enable_robust_1to1_match = 1; p.orb_max_distance = 60.0;
    p.ifm_method = SVO_IFM_DESC_BF;
}

static void rank_main(Rank* r)
{
    const Options& o = *r->opt; const std::vector<Sequence>& seqs = *r->seqs;
    const int W = seqs[0].W, H = seqs[0].H, F = seqs[0].F, B = o.contexts * o.lanes, G = o.gpus;
    const size_t img = (size_t)W * H;
    CHECK_HIP(hipSetDevice(r->device));
    // every sequence file resident in HBM once
    std::vector<uint8_t*> d_seq(seqs.size(), nullptr);
    for (size_t i = 0; i < seqs.size(); i++) { CHECK_HIP(hipMalloc((void**)&d_seq[i], seqs[i].px.size())); CHECK_HIP(hipMemcpy(d_seq[i], seqs[i].px.data(), seqs[i].px.size(), hipMemcpyHostToDevice)); }
    svo_batch_config cfg; svo_batch_config_defaults(&cfg);
    cfg.ctx.device = r->device; cfg.ctx.n_lanes = o.lanes; cfg.ctx.max_w = W; cfg.ctx.max_h = H; cfg.ctx.max_kps = 4096; cfg.ctx.max_cand = 1 << 17;
    cfg.n_contexts = o.contexts;
    svo_batch* b = nullptr;
    { const int rc = svo_batch_create(&cfg, &b); if (rc < 0) { r->err = std::string("svo_batch_create: ") + svo_strerror(rc) + " [" + (b ? svo_batch_last_error(b) : "") + "]"; if (b) svo_batch_destroy(b); return; } }
    svo_params p; set_north_star(p, o.nfeats);
    CHECK_B(svo_batch_set_params(b, &p));
    svo_stereo_camera cam; std::memset(&cam, 0, sizeof(cam));
    cam.l_fx = cam.l_fy = cam.r_fx = cam.r_fy = seqs[0].fx; cam.l_cx = cam.r_cx = seqs[0].cx; cam.l_cy = cam.r_cy = seqs[0].cy; cam.baseline = seqs[0].baseline; cam.ncols = W; cam.nrows = H;
    CHECK_B(svo_batch_set_camera(b, -1, &cam));
    // records of ALL ranks' streams: this rank's batch writes its slot, the all-gather fills the others
    // (two tables, used alternately: step t + 1 writes the other one while the all-gather of step t -- which needs every rank to have
    // finished step t -- still reads the first, so the ranks do not fall into lock step; only step t + 2 waits for that gather)
    uint8_t* d_tables[2] = { nullptr, nullptr }; hipStream_t s_gather = nullptr; hipEvent_t gathered[2] = { nullptr, nullptr };
    const size_t chunk = (size_t)B * sizeof(svo_result);
    for (int k = 0; k < 2; k++) { CHECK_HIP(hipMalloc((void**)&d_tables[k], chunk * G)); CHECK_HIP(hipMemset(d_tables[k], 0, chunk * G)); CHECK_HIP(hipEventCreateWithFlags(&gathered[k], hipEventDisableTiming)); }
    CHECK_HIP(hipStreamCreateWithFlags(&s_gather, hipStreamNonBlocking));
    CHECK_B(svo_batch_set_results_buffer(b, d_tables[0] + chunk * r->rank, chunk));
    uint8_t* d_table = d_tables[0];                                   // the table the last step wrote
    std::vector<svo_frame> frames((size_t)B);
    auto fill = [&](int step) {
        for (int l = 0; l < B; l++) {
            const int s = r->rank * B + l, file = s % (int)seqs.size(), t = ping_pong(step + s / (int)seqs.size(), F);
            const uint8_t* base = d_seq[(size_t)file] + (size_t)2 * t * img;
            frames[(size_t)l].left = svo_image{ base, W, H, (int64_t)W };
            frames[(size_t)l].right = svo_image{ base + img, W, H, (int64_t)W };
        }
    };
    auto step = [&](int i) -> bool {
        fill(i);
        const int k = i & 1;
        if (o.rccl) {
            d_table = d_tables[k];
            if (svo_batch_switch_results_buffer(b, d_table + chunk * r->rank, chunk) < 0) { r->err = "svo_batch_switch_results_buffer"; return false; }
            if (i >= 2 && svo_batch_hold_for_event(b, gathered[k]) < 0) { r->err = "ordering the step behind the gather that read its table"; return false; }
        }
        if (svo_batch_step(b, frames.data(), SVO_FLAG_DEVICE_IMAGES) < 0) { r->err = std::string("svo_batch_step: ") + svo_batch_last_error(b); return false; }
        if (o.rccl) {
            if (svo_batch_wait_on_stream(b, s_gather) < 0) { r->err = "svo_batch_wait_on_stream"; return false; }
            if (svo_group_allgather_inplace(r->group, r->rank, d_table, chunk * G, s_gather) != SVO_OK) { r->err = std::string("svo_group_allgather_inplace: ") + svo_group_last_error(r->group); return false; }
            if (hipEventRecord(gathered[k], s_gather) != hipSuccess) { r->err = "recording the gather"; return false; }
        }
        return true;
    };
    for (int i = 0; i < o.warmup; i++) if (!step(i)) return;
    CHECK_B(svo_batch_synchronize(b)); CHECK_HIP(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < o.steps; i++) if (!step(o.warmup + i)) return;
    CHECK_B(svo_batch_synchronize(b)); CHECK_HIP(hipDeviceSynchronize());
    r->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    r->table.resize((size_t)B * G);
    CHECK_HIP(hipMemcpy(r->table.data(), d_table, chunk * G, hipMemcpyDeviceToHost));
    for (int l = 0; l < B; l++) r->valid_last += r->table[(size_t)r->rank * B + l].valid ? 1 : 0;
    if (!o.dump.empty() && G == 1) {
        // the checker's pass: same schedule from a fresh state, a synchronisation per step, probe streams recorded
        std::vector<int> probe;
        for (int k = 0; k < o.contexts; k++) for (int q : { 0, o.lanes / 2 - 1 > 0 ? o.lanes / 2 - 1 : 0, o.lanes - 1 }) { const int g = k * o.lanes + q; bool have = false; for (int x : probe) have |= x == g; if (!have) probe.push_back(g); }
        CHECK_B(svo_batch_reset(b));
        const int n_steps = o.warmup + o.steps;
        FILE* f = std::fopen((o.dump + ".bin").c_str(), "wb");
        if (!f) { r->err = "cannot open the dump file"; return; }
        const int32_t hdr[8] = { (int32_t)probe.size(), n_steps, B, (int32_t)sizeof(svo_result), (int32_t)seqs.size(), F, W, H };
        std::fwrite("SVOBDMP1", 1, 8, f); std::fwrite(hdr, 4, 8, f);
        for (int g : probe) { const int32_t v = g; std::fwrite(&v, 4, 1, f); }
        std::vector<svo_result> res((size_t)B);
        for (int i = 0; i < n_steps; i++) {
            fill(i);
            CHECK_B(svo_batch_step(b, frames.data(), SVO_FLAG_DEVICE_IMAGES));
            CHECK_B(svo_batch_results(b, res.data()));
            for (int g : probe) std::fwrite(&res[(size_t)g], sizeof(svo_result), 1, f);
        }
        std::vector<svo_keypoint> kps(4096); std::vector<uint8_t> desc((size_t)4096 * 32); std::vector<svo_dmatch> mm(4096); std::vector<svo_index_pair> tr(4096);
        for (int g : probe) {
            svo_ctx* c = svo_batch_context(b, g / o.lanes); const int l = g % o.lanes;
            for (int side = 0; side < 2; side++) {
                const int32_t n = svo_get_keypoints(c, l, 0, side, kps.data(), desc.data(), 4096);
                std::fwrite(&n, 4, 1, f); std::fwrite(kps.data(), sizeof(svo_keypoint), (size_t)(n > 0 ? n : 0), f); std::fwrite(desc.data(), 32, (size_t)(n > 0 ? n : 0), f);
            }
            int32_t n = svo_get_matches(c, l, 0, mm.data(), 4096);
            std::fwrite(&n, 4, 1, f); std::fwrite(mm.data(), sizeof(svo_dmatch), (size_t)(n > 0 ? n : 0), f);
            n = svo_get_tracked(c, l, tr.data(), 4096);
            std::fwrite(&n, 4, 1, f); std::fwrite(tr.data(), sizeof(svo_index_pair), (size_t)(n > 0 ? n : 0), f);
        }
        std::fclose(f);
    }
    svo_batch_destroy(b);
    for (uint8_t* d : d_seq) (void)hipFree(d);
    for (int k = 0; k < 2; k++) { (void)hipFree(d_tables[k]); (void)hipEventDestroy(gathered[k]); } (void)hipStreamDestroy(s_gather);
    r->rc = 0;
}

int main(int argc, char** argv)
{
    // HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); streams that share one never overlap.
    // Before the first HIP call, and only if the caller has not chosen:
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    Options o;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? std::atoi(argv[++i]) : 0; };
        if (a == "--contexts") o.contexts = next(); else if (a == "--lanes") o.lanes = next(); else if (a == "--steps") o.steps = next();
        else if (a == "--warmup") o.warmup = next(); else if (a == "--nfeats") o.nfeats = next(); else if (a == "--gpus") o.gpus = next();
        else if (a == "--device") o.device = next();
        else if (a == "--gather" && i + 1 < argc) o.rccl = std::string(argv[++i]) == "rccl";
        else if (a == "--dump" && i + 1 < argc) o.dump = argv[++i];
        else o.files.push_back(argv[i]);
    }
    if (o.files.empty() || o.contexts < 1 || o.lanes < 1 || o.lanes > SVO_MAX_LANES || o.gpus < 1 || o.steps < 1) {
        std::fprintf(stderr, "usage: %s [--contexts C] [--lanes L] [--steps K] [--warmup W] [--nfeats N] [--gpus G] [--device D] [--gather rccl|none] [--dump PREFIX] a.svoseq [b.svoseq ...]\n", argv[0]);
        return 2;
    }
    std::vector<Sequence> seqs(o.files.size());
    for (size_t i = 0; i < o.files.size(); i++) if (!load_sequence(o.files[i], seqs[i])) return 2;
    for (const Sequence& s : seqs) if (s.W != seqs[0].W || s.H != seqs[0].H || s.F != seqs[0].F) { std::fprintf(stderr, "sequence files must share size and length\n"); return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < o.device + o.gpus) { std::fprintf(stderr, "%d GPU(s) visible, devices %d..%d wanted\n", ndev, o.device, o.device + o.gpus - 1); return 3; }
    svo_group* group = nullptr; int comm_count = 0;
    if (o.rccl) {
        std::vector<int> devs((size_t)o.gpus);
        for (int r = 0; r < o.gpus; r++) devs[(size_t)r] = o.device + r;
        const int rc = svo_group_create_local(devs.data(), o.gpus, &group);
        if (rc != SVO_OK) { std::fprintf(stderr, "svo_group_create_local: %s\n", group ? svo_group_last_error(group) : "bad arguments"); return 3; }
        comm_count = svo_group_comm_count(group, 0);
    }
    std::vector<Rank> ranks((size_t)o.gpus);
    std::vector<std::thread> th;
    for (int r = 0; r < o.gpus; r++) { ranks[(size_t)r].rank = r; ranks[(size_t)r].device = o.device + r; ranks[(size_t)r].opt = &o; ranks[(size_t)r].seqs = &seqs; ranks[(size_t)r].group = group; }
    for (int r = 0; r < o.gpus; r++) th.emplace_back(rank_main, &ranks[(size_t)r]);
    for (std::thread& t : th) t.join();
    if (group) svo_group_destroy(group);
    double secs = 0; int valid = 0; bool tables_equal = true;
    for (const Rank& r : ranks) {
        if (r.rc != 0) { std::fprintf(stderr, "rank %d: %s\n", r.rank, r.err.c_str()); return 1; }
        secs = r.seconds > secs ? r.seconds : secs; valid += r.valid_last;
        if (o.rccl && std::memcmp(r.table.data(), ranks[0].table.data(), r.table.size() * sizeof(svo_result)) != 0) tables_equal = false;
    }
    const int B = o.contexts * o.lanes;
    std::printf("{\"host\": \"c++ (tools/batch_streams.cpp over include/svo_batch.h)\", \"gpus\": %d, \"contexts_per_gpu\": %d, \"lanes_per_context\": %d, \"streams\": %d, "
                "\"steps\": %d, \"warmup\": %d, \"seconds\": %.5f, \"ms_per_step\": %.4f, \"pairs_per_s\": %.1f, \"valid_last_step\": \"%d/%d\", "
                "\"gather\": \"%s\", \"rccl_comm_count\": %d, \"gathered_tables_equal\": %s}\n",
                o.gpus, o.contexts, o.lanes, B * o.gpus, o.steps, o.warmup, secs, 1e3 * secs / o.steps, (double)B * o.gpus * o.steps / secs, valid, B * o.gpus,
                o.rccl ? "rccl" : "none", comm_count, o.rccl ? (tables_equal ? "true" : "false") : "null");
    return (o.rccl && !tables_equal) ? 1 : 0;
}
