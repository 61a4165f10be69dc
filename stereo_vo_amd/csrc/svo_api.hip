// svo_api.hip -- the C-ABI of include/svo_hip.h: context, geometry, frame pipeline, getters.
//
// svo_process enqueues one frame of every lane on the context's stream as a fixed sequence of ~20 kernel
// launches; every data-dependent size (keypoint, pairing, track counts) stays in device memory, so there is no
// host synchronisation inside a frame and frames can be enqueued back to back.
#include "svo_device.h"
#include "svo_kernels.h"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <mutex>
#include <map>
#include <memory>
#include <condition_variable>

#define HIPCHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e); return SVO_ERR_HIP; } } while (0)

enum { KT_BEGIN, KT_RESIZE, KT_FAST, KT_SELECT, KT_DESCRIBE, KT_NMS, KT_HAM_LR, KT_LR_FILTER, KT_HAM_TRK, KT_TRK_FILTER,
       KT_RANSAC_HYP, KT_RANSAC_CNT, KT_TRK_FINAL, KT_GN, KT_RANSAC_HYP1, KT_RANSAC_CNT1, KT_RANSAC_HYP2, KT_RANSAC_CNT2, KT_COUNT };
static const char* kt_names[KT_COUNT] = { "begin_frame", "resize", "fast", "select", "describe", "nms_rowsort", "hamming_lr",
    "match_lr_filter", "hamming_track", "track_filter", "ransac_hyp", "ransac_count", "track_finalize", "gauss_newton",
    "ransac_hyp_1", "ransac_count_1", "ransac_hyp_2", "ransac_count_2" };

struct TimedSpan { int id; hipEvent_t a, b; };

// ---- profiler sections ---------------------------------------------------------------------------------------------------
// The reference brackets its stages with mrpt's CTimeLogger: m_profiler.enter / leave("processNewImagePair", "_stg1" ... "_stg5",
// "stg3.find_pairings", "stg4.track", ...: process_new_image_pair.cpp:79, 377; stage1_rectify.cpp:41-86; stage2_detect.cpp:392, 670;
// stage3_match_left_right.cpp:64-473; stage4_match_consecutive.cpp:76-800; stage5_optimization.cpp:398, 729).  Here the same names
// are roctx ranges around the ENQUEUE of each stage (the kernels run asynchronously: rocprofv3 --marker-trace --kernel-trace ties a
// kernel to the range it was launched in).  The roctx library is looked up at run time and only when asked for (SVO_ROCTX=1, or a
// rocprofv3 session: ROCP_TOOL_LIBRARIES is set), so libsvo_hip.so has no link-time dependency on the profiler SDK.
#include <dlfcn.h>
namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi()
    {
        const char* e = getenv("SVO_ROCTX");
        if (e && e[0] == '0') return;
        if (!(e && e[0] == '1') && !getenv("ROCP_TOOL_LIBRARIES")) return;
        for (const char* name : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" }) {
            void* h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
RoctxApi& roctx_api() { static RoctxApi a; return a; }
struct Section {                                       // m_profiler.enter(name) ... leave(name)
    bool on;
    explicit Section(const char* name) : on(roctx_api().push != nullptr) { if (on) roctx_api().push(name); }
    ~Section() { if (on) roctx_api().pop(); }
    Section(const Section&) = delete; Section& operator=(const Section&) = delete;
};
}
extern "C" int svo_profiler_sections_enabled(void) { return roctx_api().push != nullptr; }

struct svo_ctx {
    svo_config cfg;
    svo_params params;
    int fast_th, orb_th;
    hipStream_t stream, stream0; bool own_stream;      // stream0: the stream the context was created with / owns
    DevCtx dc;
    bool geom_ready; int geom_w, geom_h, geom_nfe, geom_nlevels, geom_method, geom_noct;
    int raw_cap_alloc;
    uint8_t* d_img0; int img0_pitch_internal;
    // host-fed frames (process_new_image_pair.cpp:100-120 hands over host images): a ring of two device level-0 buffers
    // filled by a dedicated copy stream, so that the upload of frame t+1 overlaps the kernels of frame t
    uint8_t* d_img0_ring[2]; uint8_t* h_stage[2]; size_t slot_bytes;
    hipStream_t s_copy; hipEvent_t ev_h2d[2], ev_det[2]; bool ev_det_valid[2], ev_h2d_valid[2], up_ready; int up_slot, det_slot;
    long long pyr_bytes_alloc; int cand_total_alloc, rtab_alloc, tile_tab_alloc;
    std::vector<void*> allocs;
    std::string last_error;
    std::vector<TimedSpan> spans; std::vector<hipEvent_t> free_events;
    double kt_total[KT_COUNT]; long long kt_calls[KT_COUNT]; unsigned kt_mask;
    unsigned* d_ham_out; uint8_t* d_ham_q, *d_ham_t; int ham_cap_q, ham_cap_t;
    // stage 1 (svo_set_rectify_map, SVO_FLAG_BGR_IMAGES): all allocated on first use
    uint8_t* d_src; int src_pitch;                    // staging of host source images (grey or BGR)
    uint2** d_map_ptrs; std::vector<uint2*> map_ptrs;  // per image: fixed-point map on the device or nullptr
    int map_w, map_h, n_maps;
    std::vector<uint2*> map_bufs;                      // per image: the allocation behind map_ptrs (kept across clear / set cycles)
    // getChangeInPose works on temporaries (common.cpp:362-400): a one-lane scratch view of the device context
    DevCtx cip; bool cip_ready;
    // svo_get_values: device packing buffer and its page-locked host mirror
    uint8_t* d_vals; uint8_t* h_vals; size_t vals_bytes;
    uint32_t* d_anms;                                  // scratch of k_fastorb_anms (3 x n_img x cand_total), allocated on first use
    bool imported_pending;                             // svo_import_frame ran since the last svo_process
    int sampler_nmax;                                  // > 0: holds a reference on the shared sampler table of (device, sampler_nmax)
    hipEvent_t post_event;                             // svo_record_after_post: armed for the next call that runs the detector's post-processing
    // svo_use_graphs: the kernel sequence of a frame captured once per (flags, ring slot, thresholds) and replayed
    struct GraphEntry { uint32_t flags; int slot, fast_th, orb_th; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs; bool use_graphs;
    // every stream that has had work of this context enqueued since the last full synchronisation (svo_set_stream), each with
    // a context-owned event recorded behind the last work enqueued on it.  Synchronising waits on the EVENTS, never on the
    // foreign stream handles: a caller may synchronise and destroy its own stream and switch back with svo_set_stream(NULL);
    // an event recorded on a stream that has since been destroyed is simply complete.
    struct UsedStream { hipStream_t s; hipEvent_t ev; bool dirty; };
    std::vector<UsedStream> used_streams;
};

static void drop_graphs(svo_ctx* ctx);
static int sampler_table_acquire(svo_ctx* ctx, int nmax);
static void sampler_table_release(int device, int nmax);
static void note_stream(svo_ctx* ctx)
{
    for (auto& u : ctx->used_streams) if (u.s == ctx->stream) { u.dirty = true; return; }
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;      // mark_stream then synchronises the stream itself
    ctx->used_streams.push_back({ ctx->stream, ev, true });
}
// record "everything enqueued so far" on the current stream (behind the last launch of an entry point, on its error exits too;
// never while capturing)
static void mark_stream(svo_ctx* ctx)
{
    if (ctx->stream == ctx->stream0 && ctx->own_stream) return;  // the context's own stream is synchronised by handle
    for (auto& u : ctx->used_streams) if (u.s == ctx->stream) {
        if (u.ev) (void)hipEventRecord(u.ev, ctx->stream);
        else { (void)hipStreamSynchronize(ctx->stream); u.dirty = false; }   // no event to leave behind: wait now, while the handle is certainly alive
        return;
    }
}
// wait for what this context left on streams other than the current one
static hipError_t sync_foreign(svo_ctx* ctx)
{
    hipError_t first = hipSuccess;
    for (auto& u : ctx->used_streams) {
        if (!u.dirty || u.s == ctx->stream) continue;
        hipError_t e = hipSuccess;
        if (u.s == ctx->stream0 && ctx->own_stream) e = hipStreamSynchronize(u.s);
        else if (u.ev) e = hipEventSynchronize(u.ev);                    // (an entry without an event was synchronised when it was marked)
        if (first == hipSuccess) first = e;
        u.dirty = false;
    }
    return first;
}
// wait for everything this context has enqueued, on whichever stream (a caller that switched streams with
// svo_set_stream may have left work on the earlier ones)
static hipError_t sync_all(svo_ctx* ctx)
{
    hipError_t first = sync_foreign(ctx);
    if (ctx->up_ready) { const hipError_t e2 = hipStreamSynchronize(ctx->s_copy); if (first == hipSuccess) first = e2; }
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (first == hipSuccess) first = e;
    // everything is complete: forget the foreign streams the caller has switched away from (a host that makes a stream per frame
    // would otherwise leave one entry and one event behind per frame)
    size_t keep = 0;
    for (auto& u : ctx->used_streams) {
        u.dirty = false;
        if (u.s == ctx->stream || (u.s == ctx->stream0 && ctx->own_stream)) ctx->used_streams[keep++] = u;
        else if (u.ev) (void)hipEventDestroy(u.ev);
    }
    ctx->used_streams.resize(keep);
    return first;
}

static int align_up(int v, int a) { return (v + a - 1) / a * a; }

extern "C" void svo_config_defaults(svo_config* c)
{
    memset(c, 0, sizeof(*c));
    c->device = 0; c->n_lanes = 1; c->max_w = 1280; c->max_h = 960; c->max_kps = 4096; c->max_cand = 1 << 17; c->kernel_times = 0; c->max_octaves = 1; c->stream = nullptr;
}

extern "C" void svo_params_defaults(svo_params* p)
{
    memset(p, 0, sizeof(*p));
    p->nOctaves = 3;                                   // stage1_rectify.cpp:27-30
    p->detect_method = SVO_DM_ORB;                     // north-star selector (reference default dmFASTER, stage2_detect.cpp:45)
    p->non_maximal_suppression = 1; p->nmsMethod = SVO_NMS_STANDARD; p->min_distance = 3;   // stage2_detect.cpp:49-51
    p->orb_nfeats = 500; p->orb_nlevels = 8; p->minimum_ORB_response = 0.0;                 // stage2_detect.cpp:52-54
    p->fast_min_th = 5; p->fast_max_th = 30; p->initial_FAST_threshold = 20;                // stage2_detect.cpp:55-56
    p->match_method = SVO_SM_DESC_BF;                  // north-star selector (reference default smSAD, stage3:47)
    p->orb_max_distance = 40; p->orb_min_th = 30; p->orb_max_th = 100;                      // stage3_match_left_right.cpp:50-51
    p->enable_robust_1to1_match = 0; p->max_y_diff = 0;                                     // stage3_match_left_right.cpp:52-54
    p->ifm_method = SVO_IFM_DESC_BF; p->ifm_win_w = 16; p->ifm_win_h = 16; p->filter_fund_matrix = 0;   // H:610; no reference defaults (common.cpp:84)
    p->use_robust_kernel = 1; p->kernel_param = 3.0; p->max_iters = 100; p->initial_max_iters = 10;     // common.cpp:69-82
    p->min_mod_out_vector = 1e-3; p->max_incr_cost = 3; p->residual_threshold = 10.0; p->bad_tracking_th = 5;
    p->use_previous_pose_as_initial = 1; p->use_custom_initial_pose = 0;
    p->vo_use_matches_ids = 0;                         // process_new_image_pair.cpp:35
}

extern "C" const char* svo_strerror(int s)
{
    switch (s) {
        case SVO_OK: return "ok";
        case SVO_ERR_HIP: return "HIP runtime error (see svo_last_error)";
        case SVO_ERR_ARG: return "invalid argument";
        case SVO_ERR_UNSUPPORTED: return "configuration outside the supported hot path";
        case SVO_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case SVO_ERR_CAPACITY: return "context capacity exceeded";
        case SVO_ERR_STATE: return "call not valid in the current state";
        default: return "unknown status";
    }
}
// Every entry point runs on the context's GPU whatever device the calling thread had current (a host may own one
// estimator per GPU and call them from one thread, or from one thread each: SURVEY.md 8b "Threading")
static inline void use_device(const svo_ctx* ctx)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->cfg.device) (void)hipSetDevice(ctx->cfg.device);
}
extern "C" const char* svo_last_error(const svo_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" void svo_abi_sizes(int32_t* out)
{
    out[0] = sizeof(svo_keypoint); out[1] = sizeof(svo_dmatch); out[2] = sizeof(svo_stereo_camera);
    out[3] = sizeof(svo_params); out[4] = sizeof(svo_result); out[5] = sizeof(svo_config);
}

template <typename T>
static hipError_t dev_alloc(svo_ctx* ctx, T** p, size_t n)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
    if (e != hipSuccess) return e;
    e = hipMemset(q, 0, n * sizeof(T) + 256);
    if (e != hipSuccess) { hipFree(q); return e; }
    ctx->allocs.push_back(q);
    *p = (T*)q;
    return hipSuccess;
}

static void level_sizes(int w, int h, int nlevels, int* lw, int* lh, float* sc)
{
    for (int l = 0; l < nlevels; l++) {                      // same expression as oracle svo_oracle_pyramid_sizes
        const float sf = (float)pow(1.2, (double)l);
        sc[l] = sf; lw[l] = (int)lrintf((float)w / sf); lh[l] = (int)lrintf((float)h / sf);
    }
}

static long long pyramid_bytes(int w, int h, int nlevels)
{
    int lw[SVO_MAX_LEVELS], lh[SVO_MAX_LEVELS]; float sc[SVO_MAX_LEVELS];
    level_sizes(w, h, nlevels, lw, lh, sc);
    long long b = 0;
    for (int l = 1; l < nlevels; l++) b += (long long)align_up(lw[l], 64) * lh[l];
    return align_up((int)((b + 255) & ~255LL), 256);
}

// ---- the sampler's attempt table ---------------------------------------------------------------------------------------------
// cv::RNG from the seed RANSACPointSetRegistrator::run uses, (uint64)-1 -- state = (uint32)state * 4164903690 + (state >> 32), output
// (uint32)state --, rng.uniform(0, n) = next() % n, a draw that repeats an index of the same attempt is drawn again (getSubset); the
// collinearity rejection depends on the data and stays on the device.  The attempts depend on the point count n alone: tabulated once per
// process and nmax on the host (~0.1 s, 18 KB per n: 84 MB at max_kps 4096) and uploaded ONCE per (device, nmax) -- three contexts of a
// batch used to hold three copies -- together with the generator's state at the end of every row, from which the device
// continues the stream when a lane needs more attempts than its row holds (rs_schedule_block).  The host mutex covers the maps only,
// not the upload of another device's copy.
namespace {
struct SamplerHost { std::vector<uint16_t> att; std::vector<uint64_t> state; };
struct SamplerDev { uint16_t* att = nullptr; uint64_t* state = nullptr; int refs = 0; bool ready = false; hipError_t err = hipSuccess; };
std::mutex g_sampler_mu;
std::map<int, std::shared_ptr<SamplerHost>> g_sampler_host;                 // by nmax
std::map<std::pair<int, int>, std::shared_ptr<SamplerDev>> g_sampler_dev;   // by (device, nmax)
std::condition_variable g_sampler_cv;
size_t sampler_row_off(int n) { return n < SVO_RS_SMALL_N ? (size_t)(n - 8) * SVO_RS_ATT_SMALL : (size_t)(SVO_RS_SMALL_N - 8) * SVO_RS_ATT_SMALL + (size_t)(n - SVO_RS_SMALL_N) * SVO_RS_ATT; }
std::shared_ptr<SamplerHost> sampler_build(int nmax)
{
    auto h = std::make_shared<SamplerHost>();
    h->att.assign(sampler_row_off(nmax + 1) * 8, 0);
    h->state.assign((size_t)nmax + 1, 0xFFFFFFFFFFFFFFFFULL);
    for (int n = 8; n <= nmax; n++) {
        uint64_t stt = 0xFFFFFFFFFFFFFFFFULL;
        uint16_t* row = h->att.data() + sampler_row_off(n) * 8;
        const int n_att = n < SVO_RS_SMALL_N ? SVO_RS_ATT_SMALL : SVO_RS_ATT;
        for (int a = 0; a < n_att; a++) {
            uint16_t* s7 = row + (size_t)a * 8;
            for (int i = 0; i < 7; i++) {
                for (;;) {
                    stt = (uint64_t)(uint32_t)stt * 4164903690ULL + (uint32_t)(stt >> 32);
                    const uint16_t v = (uint16_t)((uint32_t)stt % (uint32_t)n);
                    bool dup = false;
                    for (int k = 0; k < i; k++) dup = dup || s7[k] == v;
                    if (!dup) { s7[i] = v; break; }
                }
            }
        }
        h->state[(size_t)n] = stt;
    }
    return h;
}
}
static int sampler_table_acquire(svo_ctx* ctx, int nmax)
{
    const std::pair<int, int> key(ctx->cfg.device, nmax);
    std::shared_ptr<SamplerDev> dv; std::shared_ptr<SamplerHost> host; bool mine = false;
    {
        std::unique_lock<std::mutex> lock(g_sampler_mu);
        auto& slot = g_sampler_dev[key];
        if (!slot) { slot = std::make_shared<SamplerDev>(); mine = true; }
        dv = slot; dv->refs++;
        if (!mine) g_sampler_cv.wait(lock, [&] { return dv->ready; });      // another thread is uploading this very copy: wait for THAT one only
        else {
            auto& hs = g_sampler_host[nmax];
            if (!hs) hs = sampler_build(nmax);                               // (built under the lock: once per process and nmax)
            host = hs;
        }
    }
    if (mine) {                                                              // the upload runs WITHOUT the lock: contexts on other devices / of other sizes go on
        hipError_t e = hipMalloc((void**)&dv->att, host->att.size() * sizeof(uint16_t) + 256);
        if (e == hipSuccess) e = hipMalloc((void**)&dv->state, host->state.size() * sizeof(uint64_t) + 256);
        if (e == hipSuccess) e = hipMemcpy(dv->att, host->att.data(), host->att.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dv->state, host->state.data(), host->state.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
        std::lock_guard<std::mutex> lock(g_sampler_mu);
        dv->err = e; dv->ready = true;
        g_sampler_cv.notify_all();
    }
    if (dv->err != hipSuccess) { ctx->last_error = std::string("sampler table: ") + hipGetErrorString(dv->err); sampler_table_release(ctx->cfg.device, nmax); return SVO_ERR_HIP; }
    ctx->dc.rs_att = dv->att; ctx->dc.rs_att_state = (const unsigned long long*)dv->state; ctx->dc.rs_att_nmax = nmax;
    ctx->sampler_nmax = nmax;
    return SVO_OK;
}
static void sampler_table_release(int device, int nmax)
{
    std::lock_guard<std::mutex> lock(g_sampler_mu);
    auto it = g_sampler_dev.find(std::make_pair(device, nmax));
    if (it == g_sampler_dev.end()) return;
    if (--it->second->refs > 0) return;
    if (it->second->att) hipFree(it->second->att);
    if (it->second->state) hipFree(it->second->state);
    g_sampler_dev.erase(it);
}

extern "C" int svo_create(const svo_config* cfg, svo_ctx** out)
{
    if (!cfg || !out) return SVO_ERR_ARG;
    *out = nullptr;
    if (cfg->n_lanes < 1 || cfg->n_lanes > SVO_MAX_LANES || cfg->max_w < 64 || cfg->max_h < 64) return SVO_ERR_ARG;
    if (cfg->max_kps < 64 || cfg->max_kps > 8192 || (cfg->max_kps & (cfg->max_kps - 1))) return SVO_ERR_ARG;   // above 4096 the NMS / GN kernels keep their sort and hash arrays in global scratch
    if ((long long)cfg->max_w * cfg->max_h >= (1 << 24)) return SVO_ERR_UNSUPPORTED;      // position packs into 24 bits
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) return SVO_ERR_NO_DEVICE;
    svo_ctx* ctx = new svo_ctx();
    ctx->cfg = *cfg;
    svo_params_defaults(&ctx->params);
    ctx->fast_th = 20; ctx->orb_th = 60;                  // common.cpp:35-36
    ctx->sampler_nmax = 0;
    ctx->geom_ready = false;
    ctx->d_ham_out = nullptr; ctx->d_ham_q = ctx->d_ham_t = nullptr; ctx->ham_cap_q = ctx->ham_cap_t = 0;
    ctx->d_src = nullptr; ctx->src_pitch = 0; ctx->d_map_ptrs = nullptr; ctx->map_w = ctx->map_h = ctx->n_maps = 0;
    ctx->imported_pending = false; ctx->use_graphs = false; ctx->d_anms = nullptr; ctx->post_event = nullptr;
    ctx->cip_ready = false; ctx->d_vals = nullptr; ctx->h_vals = nullptr; ctx->vals_bytes = 0;
    ctx->up_ready = false; ctx->up_slot = 0; ctx->det_slot = -1; ctx->s_copy = nullptr; ctx->slot_bytes = 0;
    for (int i = 0; i < 2; i++) { ctx->d_img0_ring[i] = nullptr; ctx->h_stage[i] = nullptr; ctx->ev_det_valid[i] = ctx->ev_h2d_valid[i] = false; }
    for (int i = 0; i < KT_COUNT; i++) { ctx->kt_total[i] = 0; ctx->kt_calls[i] = 0; }
    ctx->kt_mask = 0xFFFFFFFFu;
    *out = ctx;                                           // so that the caller can read last_error and destroy
    HIPCHECK(hipSetDevice(cfg->device));
    if (cfg->stream) { ctx->stream = (hipStream_t)cfg->stream; ctx->own_stream = false; }
    else { HIPCHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    ctx->stream0 = ctx->stream;
    HIPCHECK(svo_upload_tables());
    DevCtx& d = ctx->dc;
    memset(&d, 0, sizeof(d));
    const int L = cfg->n_lanes, NI = 2 * L, MK = cfg->max_kps;
    const int OC = cfg->max_octaves < 1 ? 1 : (cfg->max_octaves > SVO_MAX_OCTAVES ? SVO_MAX_OCTAVES : cfg->max_octaves);
    const int NV = L * OC;                                  // lane-octaves
    d.n_lanes = L; d.n_img = NI; d.max_kps = MK; d.n_levels = SVO_MAX_LEVELS; d.oct_cap = OC; d.n_oct = 1; d.max_h = cfg->max_h;
    ctx->raw_cap_alloc = 2 * MK;
    ctx->pyr_bytes_alloc = pyramid_bytes(cfg->max_w, cfg->max_h, SVO_MAX_LEVELS);
    ctx->cand_total_alloc = (int)((long long)cfg->max_cand * 33 / 10) + 8 * 1024;
    ctx->rtab_alloc = 2 * (cfg->max_w + cfg->max_h) * SVO_MAX_LEVELS;
    ctx->img0_pitch_internal = align_up(cfg->max_w, 64);
    HIPCHECK(dev_alloc(ctx, &ctx->d_img0, (size_t)NI * ctx->img0_pitch_internal * cfg->max_h));
    HIPCHECK(dev_alloc(ctx, (uint8_t***)&d.img0, (size_t)NI));
    HIPCHECK(dev_alloc(ctx, &d.pyr, (size_t)NI * ctx->pyr_bytes_alloc));
    HIPCHECK(dev_alloc(ctx, &d.rtab, (size_t)ctx->rtab_alloc));
    // FAST tile table: the level tilings of an image of the largest size, all levels (scale 1.2: 3.3 x level 0; x1/2 octaves: 1.34 x)
    ctx->tile_tab_alloc = 4 * ((cfg->max_w / SVO_FT_W + 2) * (cfg->max_h / SVO_FT_H + 2)) + 64;
    HIPCHECK(dev_alloc(ctx, (uint4**)&d.fast_tiles, (size_t)ctx->tile_tab_alloc));
    HIPCHECK(dev_alloc(ctx, &d.cand_keys, (size_t)NI * ctx->cand_total_alloc));
    HIPCHECK(dev_alloc(ctx, &d.cand_cnt, (size_t)NI * SVO_MAX_LEVELS * SVO_CNT_STRIDE));
    HIPCHECK(dev_alloc(ctx, &d.fast_th_dyn, (size_t)NI * SVO_MAX_LEVELS * 4 + 32));
    d.fast_th_used = d.fast_th_dyn + (size_t)NI * SVO_MAX_LEVELS; d.redo_flag = d.fast_th_used + (size_t)NI * SVO_MAX_LEVELS;
    d.redo_list = d.redo_flag + (size_t)NI * SVO_MAX_LEVELS; d.redo_n = d.redo_list + (size_t)NI * SVO_MAX_LEVELS;
    HIPCHECK(dev_alloc(ctx, &d.lvl_pos, (size_t)NI * ctx->raw_cap_alloc));
    HIPCHECK(dev_alloc(ctx, &d.lvl_resp, (size_t)NI * ctx->raw_cap_alloc));
    d.sel_max = MK > 4096 ? 2 * SVO_SEL_MAX : SVO_SEL_MAX;
    HIPCHECK(dev_alloc(ctx, &d.sel_keys, (size_t)NI * SVO_MAX_LEVELS * d.sel_max));
    HIPCHECK(dev_alloc(ctx, &d.sel_resp, (size_t)NI * SVO_MAX_LEVELS * d.sel_max));
    d.big_scratch = nullptr; d.gn_scratch = nullptr;
    if (MK > 4096) HIPCHECK(dev_alloc(ctx, &d.big_scratch, (size_t)NI * OC * (size_t)MK * 28));
    HIPCHECK(dev_alloc(ctx, &d.gn_scratch, (size_t)L * gn_scratch_bytes_per_lane(MK)));      // lanes that track more than GN_LCAP pairs (k_gn.hip)
    HIPCHECK(dev_alloc(ctx, &d.sel_n, (size_t)NI * SVO_MAX_LEVELS));
    HIPCHECK(dev_alloc(ctx, &d.lvl_n, (size_t)NI * SVO_MAX_LEVELS));
    HIPCHECK(dev_alloc(ctx, &d.raw_kps, (size_t)NI * ctx->raw_cap_alloc));
    HIPCHECK(dev_alloc(ctx, &d.raw_desc, (size_t)NI * ctx->raw_cap_alloc * 32));
    HIPCHECK(dev_alloc(ctx, &d.raw_n, (size_t)NI));
    HIPCHECK(dev_alloc(ctx, &d.desc_work, (size_t)NI * ctx->raw_cap_alloc + 8));
    HIPCHECK(dev_alloc(ctx, &d.desc_n, (size_t)NI));
    HIPCHECK(dev_alloc(ctx, &d.kps, (size_t)NV * 4 * MK));
    HIPCHECK(dev_alloc(ctx, &d.desc, (size_t)NV * 4 * MK * 32));
    HIPCHECK(dev_alloc(ctx, &d.mdesc, (size_t)NV * 4 * MK * 32));
    HIPCHECK(dev_alloc(ctx, &d.final_slot, (size_t)NV * 4 * MK));
    HIPCHECK(dev_alloc(ctx, &d.n_kps, (size_t)NV * 4));
    HIPCHECK(dev_alloc(ctx, &d.matches, (size_t)NV * 2 * MK));
    HIPCHECK(dev_alloc(ctx, &d.n_matches, (size_t)NV * 2));
    HIPCHECK(dev_alloc(ctx, &d.row_index, (size_t)NV * 4 * cfg->max_h));
    HIPCHECK(dev_alloc(ctx, &d.mrow_index, (size_t)NV * 2 * (cfg->max_h + 1)));
    HIPCHECK(dev_alloc(ctx, &d.ids, (size_t)NV * 2 * MK));
    HIPCHECK(dev_alloc(ctx, &d.n_ids, (size_t)NV * 2));
    HIPCHECK(dev_alloc(ctx, &d.bf_idx, (size_t)NV * 3 * MK));
    HIPCHECK(dev_alloc(ctx, &d.trk_kq, (size_t)NV * MK));
    HIPCHECK(dev_alloc(ctx, &d.trk_nk, (size_t)NV));
    HIPCHECK(dev_alloc(ctx, &d.trk_pts, (size_t)NV * 2 * MK * 4));
    HIPCHECK(dev_alloc(ctx, &d.rs_F, (size_t)NV * 2 * SVO_RANSAC_SLOTS * 9));
    HIPCHECK(dev_alloc(ctx, &d.rs_guard, (size_t)NV * 2 * SVO_RANSAC_SLOTS * 2));
    HIPCHECK(dev_alloc(ctx, &d.rs_cnt, (size_t)NV * 2 * SVO_RANSAC_SLOTS));
    HIPCHECK(dev_alloc(ctx, &d.rs_k, (size_t)NV * 2 * SVO_RANSAC_SLOTS));
    HIPCHECK(dev_alloc(ctx, &d.rs_nvalid, (size_t)NV * 2 * (SVO_RANSAC_PAD / SVO_RANSAC_REG)));
    HIPCHECK(dev_alloc(ctx, &d.rs_bound, (size_t)NV * 2));
    HIPCHECK(dev_alloc(ctx, &d.rs_gen, (size_t)NV * 2));
    HIPCHECK(dev_alloc(ctx, &d.rs_floor, (size_t)NV * 4));
    HIPCHECK(dev_alloc(ctx, &d.rs_ticket, (size_t)NV * 2 * (SVO_RANSAC_SLOTS / 16)));
    HIPCHECK(dev_alloc(ctx, &d.rs_smp, (size_t)NV * 2 * SVO_RANSAC_PAD * 8));
    HIPCHECK(dev_alloc(ctx, &d.rs_sched, (size_t)NV * SVO_RS_ST));
    {   // The attempts of cv::findFundamentalMat's sampler for every point count a context can meet (k_match.hip, k_ransac_schedule):
        // ONE device copy per (device, max_kps), shared by every context of the process and freed with the last of them (sampler_table_*).
        int rc_t = sampler_table_acquire(ctx, MK); if (rc_t) return rc_t;
    }
    HIPCHECK(dev_alloc(ctx, &d.tracked, (size_t)NV * MK));
    HIPCHECK(dev_alloc(ctx, &d.n_tracked, (size_t)NV));
    HIPCHECK(dev_alloc(ctx, &d.gn_lmk, (size_t)L * MK * 3));
    HIPCHECK(dev_alloc(ctx, &d.gn_obs, (size_t)L * MK * 8));
    HIPCHECK(dev_alloc(ctx, &d.residual, (size_t)L * MK));
    HIPCHECK(dev_alloc(ctx, &d.outliers, (size_t)L * MK));
    HIPCHECK(dev_alloc(ctx, &d.cams, (size_t)L));
    HIPCHECK(dev_alloc(ctx, &d.lane, (size_t)L));
    HIPCHECK(dev_alloc(ctx, &d.results, (size_t)L));
    HIPCHECK(dev_alloc(ctx, &d.status, (size_t)L));
    HIPCHECK(dev_alloc(ctx, &d.det_status, (size_t)L)); d.det_ahead = 0;
    d.bf_dist = nullptr;
    { const char* dm = getenv("SVO_DEBUG_MODE"); d.debug_mode = dm ? atoi(dm) : 0; }
    if (svo_ab_form_requested(d.debug_mode) && !svo_ab_kernels_built()) {
        ctx->last_error = "SVO_HAM_FP4=0 / SVO_DEBUG_MODE=14 / 52 ask for an A/B kernel form that this library was built without: load libsvo_hip_ab.so (SVO_HIP_LIB) or build with -DSVO_AB_KERNELS";
        return SVO_ERR_UNSUPPORTED;
    }
    d.rs_c0 = SVO_RANSAC_CHUNK0; d.rs_c1 = SVO_RANSAC_CHUNK1;      // (set per call where the RANSAC is launched)
    { const char* rp = getenv("SVO_REST_PRIO"); d.rest_prio = rp ? (atoi(rp) & 3) : 0; }
    // SVO_TIMELINE=1: every kernel stamps the hull of its launch on the device's wall clock (svo_device.h, TlScope; svo_debug_timeline)
    d.tl = nullptr; d.tl_step = 0;
    { const char* tl = getenv("SVO_TIMELINE"); if (tl && tl[0] == '1') { HIPCHECK(dev_alloc(ctx, &d.tl, (size_t)SVO_TL_STEPS * 256 * SVO_TL_SUB)); HIPCHECK(svo_debug_timeline(ctx, nullptr, 0, 1) < 0 ? hipErrorUnknown : hipSuccess); } }
    HIPCHECK(configure_gauss_newton(MK));
    HIPCHECK(configure_match(MK));
    return SVO_OK;
}

extern "C" void svo_destroy(svo_ctx* ctx)
{
    if (ctx) use_device(ctx);
    if (!ctx) return;
    sync_all(ctx);                                        // work may still be running on a stream the caller switched away from
    for (void* p : ctx->allocs) hipFree(p);
    if (ctx->sampler_nmax > 0) sampler_table_release(ctx->cfg.device, ctx->sampler_nmax);
    for (auto& g : ctx->graphs) hipGraphExecDestroy(g.exec);
    if (ctx->h_vals) hipHostFree(ctx->h_vals);
    if (ctx->d_ham_out) hipFree(ctx->d_ham_out);
    if (ctx->d_ham_q) hipFree(ctx->d_ham_q);
    if (ctx->d_ham_t) hipFree(ctx->d_ham_t);
    if (ctx->up_ready) {
        hipStreamSynchronize(ctx->s_copy);
        for (int i = 0; i < 2; i++) { hipEventDestroy(ctx->ev_h2d[i]); hipEventDestroy(ctx->ev_det[i]); if (ctx->h_stage[i]) hipHostFree(ctx->h_stage[i]); }
        hipStreamDestroy(ctx->s_copy);
    }
    for (auto& u : ctx->used_streams) if (u.ev) hipEventDestroy(u.ev);
    for (auto& s : ctx->spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    for (auto e : ctx->free_events) hipEventDestroy(e);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream0);
    delete ctx;
}

void level_quota(int nfeatures, int nlevels, int* q);

// The reference has no keypoint cap (stage2_detect.cpp:461-464: orb_nfeats is free); a context has one (svo_config.max_kps).
// A request that cannot fit whatever the image size is refused HERE, when the parameters are loaded, with SVO_ERR_CAPACITY and
// the numbers in svo_last_error -- not at the first frame, and never by cutting a list silently.  (What depends on the image
// size -- pyramid and candidate buffers -- is still checked by the first svo_process.)
static int params_fit(svo_ctx* ctx, const svo_params& p)
{
    const bool fast_orb = p.detect_method == SVO_DM_FAST_ORB;
    if (p.detect_method != SVO_DM_ORB && !fast_orb) return SVO_OK;          // refused as unsupported by svo_process
    char msg[256];
    const int MK = ctx->dc.max_kps;
    if (p.orb_nfeats <= 0) { snprintf(msg, sizeof(msg), "orb_nfeats %d: the detector needs a positive feature count", p.orb_nfeats); ctx->last_error = msg; return SVO_ERR_ARG; }
    if (fast_orb) {
        const int noct = p.nOctaves < 1 ? 1 : p.nOctaves;
        if (noct > ctx->dc.oct_cap) { snprintf(msg, sizeof(msg), "nOctaves %d exceeds svo_config.max_octaves %d", noct, ctx->dc.oct_cap); ctx->last_error = msg; return SVO_ERR_CAPACITY; }
        if (p.non_maximal_suppression) {
            const size_t k0 = (size_t)((double)(size_t)p.orb_nfeats * (double)(2 * noct) / (pow(2, noct) - 1));            // S2:404-407
            if ((long long)k0 > MK) { snprintf(msg, sizeof(msg), "orb_nfeats %d keeps up to %zu keypoints in octave 0, svo_config.max_kps is %d", p.orb_nfeats, k0, MK); ctx->last_error = msg; return SVO_ERR_CAPACITY; }
        }
        return SVO_OK;
    }
    const int nfe = p.non_maximal_suppression ? (int)(size_t)(1.5 * (double)(size_t)p.orb_nfeats) : p.orb_nfeats;       // S2:461-464
    int nlev = p.orb_nlevels < 1 ? 1 : p.orb_nlevels;
    if (nlev > SVO_MAX_LEVELS) return SVO_OK;                                  // refused as unsupported by the first frame
    if (nfe > MK) { snprintf(msg, sizeof(msg), "orb_nfeats %d asks the detector for %d keypoints, svo_config.max_kps is %d (largest supported: 8192)", p.orb_nfeats, nfe, MK); ctx->last_error = msg; return SVO_ERR_CAPACITY; }
    int quota[SVO_MAX_LEVELS];
    level_quota(nfe, nlev, quota);
    for (int l = 0; l < nlev; l++)
        if (2 * quota[l] > ctx->dc.sel_max) { snprintf(msg, sizeof(msg), "level %d ranks 2 x %d corners, the selection list of this context holds %d (max_kps > 4096 doubles it)", l, quota[l], ctx->dc.sel_max); ctx->last_error = msg; return SVO_ERR_CAPACITY; }
    return SVO_OK;
}

extern "C" int svo_set_params(svo_ctx* ctx, const svo_params* p)
{
    if (ctx) use_device(ctx);
    if (!ctx || !p) return SVO_ERR_ARG;
    { const int rc = params_fit(ctx, *p); if (rc) return rc; }
    ctx->params = *p;
    ctx->fast_th = p->initial_FAST_threshold;            // resetFASTThreshold (H:532, 661)
    ctx->orb_th = (int)p->orb_max_distance;              // resetORBThreshold (H:539, 662)
    ctx->geom_ready = false;
    drop_graphs(ctx);
    return SVO_OK;
}
extern "C" int svo_get_params(const svo_ctx* ctx, svo_params* p) { if (!ctx || !p) return SVO_ERR_ARG; *p = ctx->params; return SVO_OK; }
extern "C" int svo_set_fast_threshold(svo_ctx* ctx, int v)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    const int lo = ctx->params.fast_min_th, hi = ctx->params.fast_max_th, m = v > lo ? v : lo;
    ctx->fast_th = hi < m ? hi : m;
    return SVO_OK;
}
extern "C" int svo_set_orb_threshold(svo_ctx* ctx, int v)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    const int lo = ctx->params.orb_min_th, hi = ctx->params.orb_max_th, m = v > lo ? v : lo;
    ctx->orb_th = hi < m ? hi : m;
    return SVO_OK;
}
extern "C" int svo_get_fast_threshold(const svo_ctx* ctx) { return ctx ? ctx->fast_th : SVO_ERR_ARG; }
extern "C" int svo_get_orb_threshold(const svo_ctx* ctx) { return ctx ? ctx->orb_th : SVO_ERR_ARG; }

extern "C" int svo_set_stream(svo_ctx* ctx, void* stream)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    if (stream) ctx->stream = (hipStream_t)stream;
    else ctx->stream = ctx->stream0;
    return SVO_OK;
}

extern "C" int svo_get_device(const svo_ctx* ctx) { return ctx ? ctx->cfg.device : SVO_ERR_ARG; }
extern "C" int svo_get_stream(svo_ctx* ctx, void** stream) { if (!ctx || !stream) return SVO_ERR_ARG; *stream = (void*)ctx->stream; return SVO_OK; }

extern "C" int svo_set_camera(svo_ctx* ctx, int lane, const svo_stereo_camera* cam)
{
    if (ctx) use_device(ctx);
    if (!ctx || !cam || lane < -1 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    HIPCHECK(sync_all(ctx));
    for (int l = 0; l < ctx->cfg.n_lanes; l++)
        if (lane < 0 || lane == l) HIPCHECK(hipMemcpy(ctx->dc.cams + l, cam, sizeof(*cam), hipMemcpyHostToDevice));
    return SVO_OK;
}

extern "C" int svo_set_rectify_map(svo_ctx* ctx, int lane, int side, const float* map_x, const float* map_y, int w, int h)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < -1 || lane >= ctx->cfg.n_lanes || side < 0 || side > 1 || ((map_x == nullptr) != (map_y == nullptr))) return SVO_ERR_ARG;
    const int NI = 2 * ctx->cfg.n_lanes;
    if (map_x && (w <= 0 || h <= 0 || w > ctx->cfg.max_w || h > ctx->cfg.max_h || (ctx->n_maps > 0 && (w != ctx->map_w || h != ctx->map_h)))) return SVO_ERR_ARG;
    HIPCHECK(sync_all(ctx));
    if (ctx->map_ptrs.empty()) {
        ctx->map_ptrs.assign(NI, nullptr); ctx->map_bufs.assign(NI, nullptr);
        HIPCHECK(dev_alloc(ctx, (uint2***)&ctx->d_map_ptrs, (size_t)NI));
        HIPCHECK(hipMemset(ctx->d_map_ptrs, 0, sizeof(uint2*) * NI));
    }
    std::vector<uint2> fixed;
    if (map_x) {
        // cv::remap's fixed point (oracle: svo_oracle_map_fixed): cvRound(32 x) -> integer part + 5-bit fraction
        fixed.resize((size_t)w * h);
        for (size_t i = 0; i < fixed.size(); i++) {
            const float mx = map_x[i], my = map_y[i];
            uint2 e; e.x = 0xFFFFFFFFu; e.y = 0;
            if (mx > -4.0f && mx < (float)(w + 4) && my > -4.0f && my < (float)(h + 4)) {
                const long ix = lrintf(mx * 32.0f), iy = lrintf(my * 32.0f);
                const int x = (int)(ix >> 5), y = (int)(iy >> 5);
                if (x >= -1 && x < w && y >= -1 && y < h) { e.x = (uint32_t)(x + 1) | ((uint32_t)(y + 1) << 16); e.y = (uint32_t)(ix & 31) | ((uint32_t)(iy & 31) << 8); }
            }
            fixed[i] = e;
        }
    }
    for (int l = 0; l < ctx->cfg.n_lanes; l++) {
        if (lane >= 0 && lane != l) continue;
        uint2*& mp = ctx->map_ptrs[2 * l + side];
        if (map_x) {
            if (!mp) {                                       // a cleared camera's buffer is reused, not allocated again
                uint2*& buf = ctx->map_bufs[2 * l + side];
                if (!buf) HIPCHECK(dev_alloc(ctx, &buf, (size_t)ctx->cfg.max_w * ctx->cfg.max_h));
                mp = buf; ctx->n_maps++;
            }
            HIPCHECK(hipMemcpy(mp, fixed.data(), fixed.size() * sizeof(uint2), hipMemcpyHostToDevice));
            ctx->map_w = w; ctx->map_h = h;
        } else if (mp) { mp = nullptr; ctx->n_maps--; }     // the buffer stays in map_bufs for the next set
    }
    HIPCHECK(hipMemcpy(ctx->d_map_ptrs, ctx->map_ptrs.data(), sizeof(uint2*) * NI, hipMemcpyHostToDevice));
    return SVO_OK;
}

extern "C" int svo_reset(svo_ctx* ctx, int lane)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < -1 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    HIPCHECK(sync_all(ctx));
    for (int l = 0; l < ctx->cfg.n_lanes; l++)
        if (lane < 0 || lane == l) {
            HIPCHECK(hipMemset(ctx->dc.lane + l, 0, sizeof(LaneState)));
            const int OC = ctx->dc.oct_cap;
            HIPCHECK(hipMemset(ctx->dc.n_kps + (size_t)l * OC * 4, 0, (size_t)OC * 4 * sizeof(int)));
            HIPCHECK(hipMemset(ctx->dc.n_matches + (size_t)l * OC * 2, 0, (size_t)OC * 2 * sizeof(int)));
            HIPCHECK(hipMemset(ctx->dc.n_tracked + (size_t)l * OC, 0, (size_t)OC * sizeof(int)));
            HIPCHECK(hipMemset(ctx->dc.n_ids + (size_t)l * OC * 2, 0, (size_t)OC * 2 * sizeof(int)));
            HIPCHECK(hipMemset(ctx->dc.results + l, 0, sizeof(svo_result)));
            HIPCHECK(hipMemset(ctx->dc.fast_th_dyn + (size_t)2 * l * SVO_MAX_LEVELS, 0, (size_t)2 * SVO_MAX_LEVELS * sizeof(uint32_t)));   // a fresh estimator speculates nothing
        }
    return SVO_OK;
}

// cv::ORB per-level feature budget (oracle: orb_level_quota)
void level_quota(int nfeatures, int nlevels, int* q)
{
    const float factor = (float)(1.0 / 1.2);
    float nd = (float)nfeatures * (1.0f - factor) / (1.0f - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { q[l] = (int)lrintf(nd); sum += q[l]; nd *= factor; }
    q[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

// cv::resize's INTER_LINEAR tables for one axis of an 8-bit image (oracle v7: svo_oracle_resize_table, same float operations in the same
// order; this file is compiled with -ffp-contract=off): idx = first tap, w01 = weight of that tap | weight of the next << 16
static void resize_table(int src, int dst, int* idx, int* w01)
{
    const double inv_scale = (double)dst / src, scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; d++) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)floor((double)fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= src - 1) { fx = 0.f; sx = src - 1; }
        long a0 = lrintf((1.f - fx) * 2048.f), a1 = lrintf(fx * 2048.f);       // cvRound: half to even
        if (a0 > 32767) a0 = 32767;
        if (a1 > 32767) a1 = 32767;
        idx[d] = sx; w01[d] = (int)a0 | ((int)a1 << 16);
    }
}

static void drop_graphs(svo_ctx* ctx)
{
    if (ctx->graphs.empty()) return;
    sync_all(ctx);
    for (auto& g : ctx->graphs) hipGraphExecDestroy(g.exec);
    ctx->graphs.clear();
}

static int ensure_geometry(svo_ctx* ctx, int w, int h)
{
    const svo_params& p = ctx->params;
    const bool fast_orb = p.detect_method == SVO_DM_FAST_ORB;
    const int nfe = p.non_maximal_suppression ? (int)(size_t)(1.5 * (double)(size_t)p.orb_nfeats) : p.orb_nfeats;   // stage2_detect.cpp:461-464
    // stage1_rectify.cpp:80: one octave for dmORB (cv::ORB builds its own x1/1.2 pyramid), params_rectify.nOctaves otherwise
    const int noct = fast_orb ? (p.nOctaves < 1 ? 1 : p.nOctaves) : 1;
    int nlev = fast_orb ? noct : p.orb_nlevels; if (nlev < 1) nlev = 1;
    const int nfe_key = fast_orb ? (p.non_maximal_suppression ? p.orb_nfeats : -1) : nfe;      // FAST + ORB without NMS keeps every corner: its own slot layout
    if (ctx->geom_ready && ctx->geom_w == w && ctx->geom_h == h && ctx->geom_nfe == nfe_key && ctx->geom_nlevels == nlev &&
        ctx->geom_method == p.detect_method && ctx->geom_noct == noct) return SVO_OK;
    if (w > ctx->cfg.max_w || h > ctx->cfg.max_h || w < 64 || h < 64) return SVO_ERR_CAPACITY;
    if (nlev > SVO_MAX_LEVELS) return SVO_ERR_UNSUPPORTED;
    if (noct > ctx->dc.oct_cap) return SVO_ERR_CAPACITY;       // svo_config.max_octaves
    HIPCHECK(sync_all(ctx));
    drop_graphs(ctx);
    DevCtx& d = ctx->dc;
    int lw[SVO_MAX_LEVELS], lh[SVO_MAX_LEVELS], quota[SVO_MAX_LEVELS]; float sc[SVO_MAX_LEVELS];
    d.fast_orb = fast_orb ? 1 : 0; d.n_oct = noct;
    // number of keypoints the reference's NMS may keep per octave (stage2_detect.cpp:404-407)
    {
        size_t k0 = (size_t)((double)(size_t)p.orb_nfeats * (double)(2 * noct) / (pow(2, noct) - 1));
        for (int o = 0; o < SVO_MAX_OCTAVES; o++) d.kps_to_detect[o] = o == 0 ? (int)k0 : (o < noct ? (int)(size_t)round((double)k0 / pow(2, o)) : 0);
    }
    if (fast_orb) {
        // mrpt x1/2 octaves (S1:82-83).  Slots per octave: what the NMS may keep (kps_to_detect); without NMS the reference keeps EVERY
        // FAST corner (S2:613-614: no cap at all) -- here as many as the context's lists hold, max_kps for octave 0 and half of the one
        // before for the others (a quarter of the pixels each), beyond which SVO_ST_KPS_OVERFLOW is raised
        for (int l = 0; l < nlev; l++) {
            lw[l] = l ? lw[l - 1] / 2 : w; lh[l] = l ? lh[l - 1] / 2 : h; sc[l] = 1.0f;
            quota[l] = p.non_maximal_suppression ? d.kps_to_detect[l] : std::max(64, d.max_kps >> l);
        }
    } else {
        level_sizes(w, h, nlev, lw, lh, sc);
        level_quota(nfe, nlev, quota);
    }
    for (int o = 0; o < SVO_MAX_OCTAVES; o++) { d.ow[o] = fast_orb ? (o < noct ? lw[o] : 0) : (o == 0 ? w : 0); d.oh[o] = fast_orb ? (o < noct ? lh[o] : 0) : (o == 0 ? h : 0); }
    d.W = w; d.H = h; d.n_levels = nlev;
    long long off = 0; int tile_off = 0, slot_off = 0, cand_off = 0, rt_off = 0;
    std::vector<int> rtab;
    for (int l = 0; l < nlev; l++) {
        LevelGeom& g = d.lv[l];
        g.w = lw[l]; g.h = lh[l]; g.pitch = align_up(lw[l], 64); g.scale = sc[l];
        g.offset = off; if (l >= 1) off += (long long)g.pitch * g.h;
        const int iw = g.w - 2 * SVO_EDGE, ih = g.h - 2 * SVO_EDGE;
        const bool live = iw > 0 && ih > 0 && quota[l] > 0;
        g.tiles_x = live ? (iw + SVO_FT_W - 1) / SVO_FT_W : 0; g.tiles_y = live ? (ih + SVO_FT_H - 1) / SVO_FT_H : 0;
        g.tile_off = tile_off; tile_off += g.tiles_x * g.tiles_y;
        g.quota = live ? quota[l] : 0; g.slot_off = slot_off; slot_off += g.quota;
        long long cc = (long long)ctx->cfg.max_cand * ((long long)g.w * g.h) / ((long long)lw[0] * lh[0]);
        if (cc < 1024) cc = 1024;
        g.cand_cap = (int)cc; g.cand_off = cand_off; cand_off += g.cand_cap;
        g.rtab_off = rt_off;
        if (l >= 1 && !fast_orb) {
            std::vector<int> xi(g.w), xf(g.w), yi(g.h), yf(g.h);
            resize_table(lw[l - 1], g.w, xi.data(), xf.data());
            resize_table(lh[l - 1], g.h, yi.data(), yf.data());
            // what k_resize's tile shape relies on (true for every x1/1.2 step; a geometry that broke it would be refused, not mis-sampled):
            // the four adjacent pixels of a thread sample within one 8-byte window, a 128 x 32 tile's taps fit the 176 x 42 LDS window
            for (int x = 0; x + 3 < g.w; x++) if (xi[x + 3] < xi[x] || xi[x + 3] - xi[x] > 5) return SVO_ERR_UNSUPPORTED;
            for (int x = 0; x < g.w; x += 128) if (xi[std::min(x + 127, g.w - 1)] + 1 - (xi[x] & ~15) + 4 > 176) return SVO_ERR_UNSUPPORTED;
            for (int y = 0; y < g.h; y += 32) if (yi[std::min(y + 31, g.h - 1)] + 1 - yi[y] > 41) return SVO_ERR_UNSUPPORTED;
            rtab.insert(rtab.end(), xi.begin(), xi.end()); rtab.insert(rtab.end(), xf.begin(), xf.end());
            rtab.insert(rtab.end(), yi.begin(), yi.end()); rtab.insert(rtab.end(), yf.begin(), yf.end());
            rt_off += 2 * (g.w + g.h);
        }
        if (!fast_orb && 2 * g.quota > d.sel_max) return SVO_ERR_CAPACITY;         // k_select's list capacity (2048, 4096 with max_kps > 4096)
        if (fast_orb && g.quota > d.max_kps) return SVO_ERR_CAPACITY;               // one octave's list must fit max_kps
    }
    for (int l = nlev; l < SVO_MAX_LEVELS; l++) { memset(&d.lv[l], 0, sizeof(LevelGeom)); d.lv[l].tile_off = tile_off; d.lv[l].slot_off = slot_off; }
    d.pyr_bytes = ctx->pyr_bytes_alloc;
    d.raw_cap = ctx->raw_cap_alloc;
    if (off > ctx->pyr_bytes_alloc || cand_off > ctx->cand_total_alloc || (int)rtab.size() > ctx->rtab_alloc || slot_off > d.raw_cap) return SVO_ERR_CAPACITY;
    if (!fast_orb && slot_off > d.max_kps) return SVO_ERR_CAPACITY;                 // ORB mode: all levels feed one list
    d.n_tiles = tile_off; d.n_slots = slot_off; d.cand_total = ctx->cand_total_alloc;
    d.div_tiles = make_fastdiv((uint32_t)(tile_off > 0 ? tile_off : 1));
    if (!rtab.empty()) HIPCHECK(hipMemcpy(d.rtab, rtab.data(), rtab.size() * sizeof(int), hipMemcpyHostToDevice));
    {   // k_fast's tile table (same for every image and frame of this geometry)
        if (tile_off > ctx->tile_tab_alloc) return SVO_ERR_CAPACITY;
        std::vector<uint4> tt((size_t)tile_off);
        for (int l = 0; l < nlev; l++) {
            const LevelGeom& g = d.lv[l];
            for (int t = 0; t < g.tiles_x * g.tiles_y; t++) {
                const int by = t / g.tiles_x, bx = t - by * g.tiles_x;
                tt[(size_t)g.tile_off + t] = make_uint4((uint32_t)(SVO_EDGE + bx * SVO_FT_W) | ((uint32_t)(SVO_EDGE + by * SVO_FT_H) << 16), (uint32_t)g.w | ((uint32_t)g.h << 16),
                                                        (uint32_t)l | ((uint32_t)g.pitch << 8), (uint32_t)g.offset);
            }
        }
        if (!tt.empty()) HIPCHECK(hipMemcpy((void*)d.fast_tiles, tt.data(), tt.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    HIPCHECK(configure_nms_rowsort(d));
    ctx->geom_ready = true; ctx->geom_w = w; ctx->geom_h = h; ctx->geom_nfe = nfe_key; ctx->geom_nlevels = nlev;
    ctx->geom_method = p.detect_method; ctx->geom_noct = noct;
    return SVO_OK;
}

// ---- kernel timing ------------------------------------------------------------------------------------------
static hipEvent_t get_event(svo_ctx* ctx)
{
    if (!ctx->free_events.empty()) { hipEvent_t e = ctx->free_events.back(); ctx->free_events.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
struct Span {
    svo_ctx* ctx; int id; hipEvent_t a, b; bool on; hipStream_t st;
    Span(svo_ctx* c, int i, hipStream_t on_stream = nullptr) : ctx(c), id(i), on(c->cfg.kernel_times != 0 && ((c->kt_mask >> i) & 1u)), st(on_stream ? on_stream : c->stream) { if (on) { a = get_event(ctx); b = get_event(ctx); hipEventRecord(a, st); } }
    ~Span() { if (on) { hipEventRecord(b, st); ctx->spans.push_back({ id, a, b }); } }
};
static void collect_spans(svo_ctx* ctx)
{
    for (auto& s : ctx->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { ctx->kt_total[s.id] += ms; ctx->kt_calls[s.id] += 1; }
        ctx->free_events.push_back(s.a); ctx->free_events.push_back(s.b);
    }
    ctx->spans.clear();
}

extern "C" int svo_wait(svo_ctx* ctx)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    HIPCHECK(sync_all(ctx));
    collect_spans(ctx);
    return SVO_OK;
}

extern "C" int svo_kernel_times(svo_ctx* ctx, const char** names, double* total_ms, int64_t* calls, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    for (int i = 0; i < KT_COUNT && i < cap; i++) { if (names) names[i] = kt_names[i]; if (total_ms) total_ms[i] = ctx->kt_total[i]; if (calls) calls[i] = ctx->kt_calls[i]; }
    return KT_COUNT;
}
extern "C" int svo_kernel_times_select(svo_ctx* ctx, const char* name)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    if (!name || !*name) { ctx->kt_mask = 0xFFFFFFFFu; return SVO_OK; }
    for (int i = 0; i < KT_COUNT; i++) if (!strcmp(name, kt_names[i])) { ctx->kt_mask = 1u << i; return SVO_OK; }
    return SVO_ERR_ARG;
}

extern "C" int svo_kernel_times_reset(svo_ctx* ctx)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    for (int i = 0; i < KT_COUNT; i++) { ctx->kt_total[i] = 0; ctx->kt_calls[i] = 0; }
    return SVO_OK;
}

// train-side splits of the brute-force matcher: enough workgroups for ~8 waves per SIMD (a ~2000 x 2000 problem is
// 8 query blocks x 8 train tiles; with one split per lane a 16-lane launch keeps ONE wave per SIMD busy)
static int hamming_splits(const svo_ctx* ctx)
{
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("SVO_HAM_SPLITS"); forced = e ? atoi(e) : 0; }
    if (forced > 0) return forced;
    int s = 1024 / (8 * ctx->cfg.n_lanes); return s < 1 ? 1 : (s > 8 ? 8 : s);     // train splits: enough workgroups to fill the GPU, few enough to amortise the query expansion (64 lanes: 2 -- 32.1 us per launch against 34.8 with 4, 38.2 with 6: profiles/r04o)
}

// ---- host-fed frames ------------------------------------------------------------------------------------------
static int ensure_upload(svo_ctx* ctx)
{
    if (ctx->up_ready) return SVO_OK;
    {   // the upload stream at the highest priority: when a caller's compute stream is a high-priority one (svo_batch's detect stream),
        // an upload queued at normal priority was scheduled behind it (host-fed batch: 21.3 k -> 16.1 k pairs/s, 52 -> 40 GB/s)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const char* e = getenv("SVO_COPY_PRIO");
        HIPCHECK(hipStreamCreateWithPriority(&ctx->s_copy, hipStreamNonBlocking, (e && atoi(e) == 0) ? 0 : greatest));
    }
    for (int i = 0; i < 2; i++) { HIPCHECK(hipEventCreateWithFlags(&ctx->ev_h2d[i], hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&ctx->ev_det[i], hipEventDisableTiming)); }
    ctx->slot_bytes = (size_t)2 * ctx->cfg.n_lanes * ctx->img0_pitch_internal * ctx->cfg.max_h;
    ctx->d_img0_ring[0] = ctx->d_img0;
    HIPCHECK(dev_alloc(ctx, &ctx->d_img0_ring[1], ctx->slot_bytes));
    ctx->up_ready = true;
    return SVO_OK;
}

// Enqueue the upload of one frame of every lane (8-bit grey, host memory) into the next slot of the device ring, on the
// copy stream; the compute stream `st` is made to wait for it.  No host synchronisation when the images are
// page-locked (SVO_FLAG_PINNED_IMAGES): the call returns while the copy is in flight and the caller keeps the images
// valid until svo_wait_upload / svo_wait.  Pageable images are first copied into the context's own page-locked
// staging slot (that is the cost of "borrowed for the duration of the call", P:111-120), then uploaded the same way.
static int upload_frames(svo_ctx* ctx, const svo_frame* frames, int w, int h, bool pinned, hipStream_t st, const uint8_t** ptrs)
{
    int rc = ensure_upload(ctx); if (rc) return rc;
    const int L = ctx->cfg.n_lanes, ipitch = ctx->img0_pitch_internal, slot = ctx->up_slot;
    const size_t img_bytes = (size_t)ipitch * ctx->cfg.max_h;
    uint8_t* dbase = ctx->d_img0_ring[slot];
    // the slot's previous frame must be through stage 2 (the only stage that reads level 0) before it is overwritten
    if (ctx->ev_det_valid[slot]) HIPCHECK(hipStreamWaitEvent(ctx->s_copy, ctx->ev_det[slot], 0));
    const uint8_t* src0 = frames[0].left.data;
    bool contiguous = (w == ipitch) && (h == ctx->cfg.max_h);
    for (int l = 0; l < L && contiguous; l++)
        for (int sd = 0; sd < 2; sd++) {
            const svo_image& im = sd ? frames[l].right : frames[l].left;
            if (im.stride != (int64_t)w || im.data != src0 + (size_t)(2 * l + sd) * img_bytes) contiguous = false;
        }
    if (pinned) {
        if (contiguous) HIPCHECK(hipMemcpyAsync(dbase, src0, img_bytes * 2 * L, hipMemcpyHostToDevice, ctx->s_copy));       // [lane][side][h][w] in one piece
        else for (int l = 0; l < L; l++)
            for (int sd = 0; sd < 2; sd++) {
                const svo_image& im = sd ? frames[l].right : frames[l].left;
                HIPCHECK(hipMemcpy2DAsync(dbase + (size_t)(2 * l + sd) * img_bytes, ipitch, im.data, (size_t)im.stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, ctx->s_copy));
            }
    } else {
        if (!ctx->h_stage[slot]) HIPCHECK(hipHostMalloc((void**)&ctx->h_stage[slot], ctx->slot_bytes, hipHostMallocDefault));
        if (ctx->ev_h2d_valid[slot]) HIPCHECK(hipEventSynchronize(ctx->ev_h2d[slot]));          // the staging slot's last upload has left it
        for (int l = 0; l < L; l++)
            for (int sd = 0; sd < 2; sd++) {
                const svo_image& im = sd ? frames[l].right : frames[l].left;
                uint8_t* hs = ctx->h_stage[slot] + (size_t)(2 * l + sd) * img_bytes;
                if (im.stride == (int64_t)ipitch && w == ipitch) memcpy(hs, im.data, (size_t)ipitch * h);
                else for (int y = 0; y < h; y++) memcpy(hs + (size_t)y * ipitch, im.data + (size_t)y * im.stride, (size_t)w);
            }
        if (h == ctx->cfg.max_h) HIPCHECK(hipMemcpyAsync(dbase, ctx->h_stage[slot], img_bytes * 2 * L, hipMemcpyHostToDevice, ctx->s_copy));
        else for (int i = 0; i < 2 * L; i++) HIPCHECK(hipMemcpyAsync(dbase + (size_t)i * img_bytes, ctx->h_stage[slot] + (size_t)i * img_bytes, (size_t)ipitch * h, hipMemcpyHostToDevice, ctx->s_copy));
    }
    HIPCHECK(hipEventRecord(ctx->ev_h2d[slot], ctx->s_copy)); ctx->ev_h2d_valid[slot] = true;
    HIPCHECK(hipStreamWaitEvent(st, ctx->ev_h2d[slot], 0));
    for (int i = 0; i < 2 * L; i++) ptrs[i] = dbase + (size_t)i * img_bytes;
    ctx->det_slot = slot; ctx->up_slot = slot ^ 1;
    return SVO_OK;
}

extern "C" int svo_use_graphs(svo_ctx* ctx, int enable)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    if (!enable) drop_graphs(ctx);
    ctx->use_graphs = enable != 0;
    return SVO_OK;
}

extern "C" int svo_wait_upload(svo_ctx* ctx)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    if (ctx->up_ready) HIPCHECK(hipStreamSynchronize(ctx->s_copy));
    return SVO_OK;
}
extern "C" int svo_host_alloc(size_t bytes, void** out)
{
    if (!out || !bytes) return SVO_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? SVO_OK : SVO_ERR_HIP;
}
extern "C" int svo_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? SVO_OK : SVO_ERR_HIP; }
extern "C" int svo_host_register(void* p, size_t bytes) { return (p && bytes && hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) ? SVO_OK : SVO_ERR_HIP; }
extern "C" int svo_host_unregister(void* p) { return (p && hipHostUnregister(p) == hipSuccess) ? SVO_OK : SVO_ERR_HIP; }

// ---- processNewImagePair ---------------------------------------------------------------------------------
extern "C" int svo_process(svo_ctx* ctx, const svo_frame* frames, uint32_t flags)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    // svo_record_after_post arms ONE call: the next one that runs the detector's post-processing.  Whatever way that call leaves --
    // an argument check, SVO_ERR_STATE, a failed launch -- the event is recorded behind what was enqueued (possibly nothing) and
    // disarmed, so that a waiter is released and a later, unrelated call cannot record it at the wrong point.
    struct PostDisarm { svo_ctx* c; bool consumes; ~PostDisarm() { if (consumes && c->post_event) { hipStreamCaptureStatus cs = hipStreamCaptureStatusNone; hipStreamIsCapturing(c->stream, &cs);
                                                                    if (cs == hipStreamCaptureStatusNone) hipEventRecord(c->post_event, c->stream); c->post_event = nullptr; } } }
        post_disarm{ ctx, (flags & SVO_RUN_DETECT_POST) || ((flags & SVO_RUN_DETECT) && !(flags & SVO_FLAG_DETECT_NO_POST)) };
    const svo_params& p = ctx->params;
    // P:54-76: invalid selectors are hard errors; the variants outside the hot path are refused explicitly
    if (p.detect_method < 0 || p.detect_method > 3 || p.match_method < 0 || p.match_method > 2 || p.ifm_method < 0 || p.ifm_method > 3) return SVO_ERR_ARG;
    if ((flags & SVO_RUN_DETECT) && p.detect_method != SVO_DM_ORB && p.detect_method != SVO_DM_FAST_ORB) return SVO_ERR_UNSUPPORTED;   // KLT / FASTER: out of scope
    if ((flags & SVO_RUN_MATCH) && p.match_method != SVO_SM_DESC_BF && p.match_method != SVO_SM_DESC_RBR) return SVO_ERR_UNSUPPORTED;   // smSAD: out of scope
    if ((flags & SVO_RUN_TRACK) && p.ifm_method != SVO_IFM_DESC_BF && p.ifm_method != SVO_IFM_DESC_WIN) return SVO_ERR_UNSUPPORTED;      // ifmSAD / optical flow: out of scope
    if (p.non_maximal_suppression && p.nmsMethod != SVO_NMS_STANDARD && p.nmsMethod != SVO_NMS_ADAPTIVE) return SVO_ERR_ARG;          // S2:608
    if (p.min_distance < 2) return SVO_ERR_ARG;            // cell size 0 divides by zero in the reference (S2:331-332)
    // SVO_FLAG_DETECT_AHEAD: a detect call that leaves lane state and records alone (it may overlap stages 3-5 of the frame before),
    // or the post call that completes it (and therefore runs the shift itself)
    const bool ahead = (flags & SVO_FLAG_DETECT_AHEAD) != 0;
    if (ahead) {
        if (flags & SVO_RUN_DETECT) { if (!(flags & SVO_FLAG_DETECT_NO_POST) || (flags & (SVO_RUN_MATCH | SVO_RUN_TRACK | SVO_RUN_OPTIMIZE | SVO_RUN_DETECT_POST))) return SVO_ERR_ARG; }
        else if (!(flags & SVO_RUN_DETECT_POST) || (flags & SVO_FLAG_NO_SHIFT)) return SVO_ERR_ARG;
    }
    Section sec_all("processNewImagePair");                                       // P:79, 377
    DevCtx& d = ctx->dc;
    const hipStream_t st = ctx->stream;
    note_stream(ctx);
    bool capturing = false;
    // whatever was enqueued before an error exit is still covered by the stream's event (declared before the capture guard: a
    // capture is closed first)
    struct MarkGuard { svo_ctx* c; ~MarkGuard() { mark_stream(c); } } mark_guard{ ctx };
    const uint8_t* ptrs[2 * SVO_MAX_LANES];
    PrepArgs prep; memset(&prep, 0, sizeof(prep)); bool prepare = false;
    if (flags & SVO_RUN_DETECT) {
        if (!frames) return SVO_ERR_ARG;                    // P:81
        const int w = frames[0].left.w, h = frames[0].left.h;
        for (int l = 0; l < d.n_lanes; l++) {
            const svo_frame& f = frames[l];
            if (!f.left.data || !f.right.data || f.left.w != w || f.left.h != h || f.right.w != w || f.right.h != h) return SVO_ERR_ARG;
        }
        int rc = ensure_geometry(ctx, w, h); if (rc) return rc;
        const int ch = (flags & SVO_FLAG_BGR_IMAGES) ? 3 : 1;
        prepare = ch == 3 || ctx->n_maps > 0;
        if (ctx->n_maps > 0 && (ctx->map_w != w || ctx->map_h != h)) return SVO_ERR_ARG;
        const int ipitch = ctx->img0_pitch_internal;
        if (flags & SVO_FLAG_DEVICE_IMAGES) {
            const long long stride = frames[0].left.stride;
            for (int l = 0; l < d.n_lanes; l++) {
                if (frames[l].left.stride != stride || frames[l].right.stride != stride) return SVO_ERR_ARG;
                ptrs[2 * l] = frames[l].left.data; ptrs[2 * l + 1] = frames[l].right.data;
            }
            d.img0_pitch = (int)stride;
            if (prepare) { for (int i = 0; i < 2 * d.n_lanes; i++) prep.src[i] = ptrs[i]; prep.src_stride = stride; }
            else if (ctx->use_graphs) {
                // a captured frame reads fixed addresses: the images go to the ring slot first (device to device, same stream)
                rc = ensure_upload(ctx); if (rc) return rc;
                const int slot = ctx->up_slot; ctx->up_slot = slot ^ 1; ctx->det_slot = slot;
                const size_t img_bytes = (size_t)ipitch * ctx->cfg.max_h;
                for (int i = 0; i < 2 * d.n_lanes; i++) {
                    uint8_t* dst = ctx->d_img0_ring[slot] + (size_t)i * img_bytes;
                    HIPCHECK(hipMemcpy2DAsync(dst, ipitch, ptrs[i], (size_t)stride, (size_t)w, (size_t)h, hipMemcpyDeviceToDevice, st));
                    ptrs[i] = dst;
                }
                d.img0_pitch = ipitch;
            }
        } else if (!prepare) {
            rc = upload_frames(ctx, frames, w, h, (flags & SVO_FLAG_PINNED_IMAGES) != 0, st, ptrs); if (rc) return rc;
            d.img0_pitch = ipitch;
        } else {
            if (!ctx->d_src) { ctx->src_pitch = align_up(3 * ctx->cfg.max_w, 64); HIPCHECK(dev_alloc(ctx, &ctx->d_src, (size_t)2 * ctx->cfg.n_lanes * ctx->src_pitch * ctx->cfg.max_h)); }
            for (int l = 0; l < d.n_lanes; l++)
                for (int s = 0; s < 2; s++) {
                    const svo_image& im = s ? frames[l].right : frames[l].left;
                    uint8_t* dst = ctx->d_src + (size_t)(2 * l + s) * ctx->src_pitch * ctx->cfg.max_h;
                    HIPCHECK(hipMemcpy2DAsync(dst, ctx->src_pitch, im.data, (size_t)im.stride, (size_t)w * ch, (size_t)h, hipMemcpyHostToDevice, st));
                    prep.src[2 * l + s] = dst;
                }
            prep.src_stride = ctx->src_pitch;
        }
        if (prepare) {       // stage 1 writes the context's own level-0 buffers; detection reads those
            for (int i = 0; i < 2 * d.n_lanes; i++) ptrs[i] = ctx->d_img0 + (size_t)i * ipitch * ctx->cfg.max_h;
            d.img0_pitch = ipitch;
            prep.maps = ctx->n_maps > 0 ? (const uint2* const*)ctx->d_map_ptrs : nullptr;
            prep.dst = ctx->d_img0; prep.dst_img_stride = (long long)ipitch * ctx->cfg.max_h; prep.dst_pitch = ipitch;
            prep.channels = ch; prep.w = w; prep.h = h;
        }
    } else if (!ctx->geom_ready && (flags & (SVO_RUN_MATCH | SVO_RUN_OPTIMIZE))) return SVO_ERR_STATE;
    d.fast_th = ctx->fast_th; d.orb_th = ctx->orb_th;
    if (flags & SVO_RUN_DETECT) d.tl_step++;                 // (a frame of the detect-ahead schedule is a detect call and then a post call)
    // svo_use_graphs: the launches below are captured once into a hipGraph per (flags, ring slot, thresholds) and replayed:
    // ~30 kernel launches of a frame become one graph launch (what bounds ONE stream is launch latency, not the kernels)
    // every lazily allocated buffer a frame may need exists BEFORE a capture begins: hipMalloc / hipMemset are refused on a
    // capturing thread and would invalidate the capture (the adaptive NMS after the FAST+ORB detector has such a buffer)
    if ((flags & SVO_RUN_DETECT) && d.fast_orb && p.non_maximal_suppression && p.nmsMethod == SVO_NMS_ADAPTIVE && !ctx->d_anms)
        HIPCHECK(dev_alloc(ctx, &ctx->d_anms, (size_t)3 * d.n_img * ctx->cand_total_alloc));
    const bool graph_ok = ctx->use_graphs && !prepare && !ctx->cfg.kernel_times && (!(flags & SVO_RUN_DETECT) || ctx->det_slot >= 0);
    const uint32_t gflags = flags & ~(uint32_t)(SVO_FLAG_DEVICE_IMAGES | SVO_FLAG_PINNED_IMAGES);
    const int gslot = (flags & SVO_RUN_DETECT) ? ctx->det_slot : -1;
    if (graph_ok) {
        const bool ids_first = p.vo_use_matches_ids && ctx->imported_pending;       // changes one kernel argument: not replayable
        for (auto& g : ctx->graphs)
            if (!ids_first && g.flags == gflags && g.slot == gslot && g.fast_th == ctx->fast_th && g.orb_th == ctx->orb_th) {
                HIPCHECK(hipGraphLaunch(g.exec, st));
                ctx->imported_pending = false;
                if ((flags & (SVO_RUN_DETECT | SVO_RUN_DETECT_POST)) && ctx->up_ready && ctx->det_slot >= 0) { HIPCHECK(hipEventRecord(ctx->ev_det[ctx->det_slot], st)); ctx->ev_det_valid[ctx->det_slot] = true; }
                if (ctx->post_event && ((flags & SVO_RUN_DETECT_POST) || ((flags & SVO_RUN_DETECT) && !(flags & SVO_FLAG_DETECT_NO_POST)))) { HIPCHECK(hipEventRecord(ctx->post_event, st)); ctx->post_event = nullptr; }
                return SVO_OK;
            }
        if (!ids_first) { HIPCHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); capturing = true; }
    }
    struct CaptureGuard { hipStream_t st; bool* on; ~CaptureGuard() { if (*on) { hipGraph_t g = nullptr; hipStreamEndCapture(st, &g); if (g) hipGraphDestroy(g); *on = false; } } } guard{ st, &capturing };
    d.det_ahead = (ahead && (flags & SVO_RUN_DETECT)) ? 1 : 0;
    { Span s(ctx, KT_BEGIN); launch_begin_frame(d, (flags & SVO_RUN_DETECT) ? ptrs : nullptr, flags, st); if (prepare) { Section sec("_stg1"); launch_prepare(prep, 2 * d.n_lanes, st); } }
    // the detector's per-image scratch has had its last reader: the armed event (svo_record_after_post) goes here
    auto after_post = [&]() -> hipError_t {
        if (!ctx->post_event || capturing) return hipSuccess;
        const hipError_t e = hipEventRecord(ctx->post_event, st);
        ctx->post_event = nullptr;
        return e;
    };
    if (flags & SVO_RUN_DETECT) {
        Section sec("_stg2");                                                        // S2:392, 670
        if (d.fast_orb) {       // stage2_detect.cpp:502-515 on the x1/2 octave pyramid
            { Span s(ctx, KT_RESIZE); for (int l = 1; l < d.n_levels; l++) launch_half(d, l, st); }
            { Span s(ctx, KT_FAST); launch_fast(d, st); }
            if (p.non_maximal_suppression && p.nmsMethod == SVO_NMS_ADAPTIVE) {      // S2:599-606 on the FAST detector's output
                Span s(ctx, KT_SELECT); launch_fastorb_anms(d, ctx->d_anms, st);
            } else { Span s(ctx, KT_SELECT); launch_fastorb_nms(d, p.non_maximal_suppression, p.min_distance, st); }
            { Span s(ctx, KT_DESCRIBE); launch_describe(d, 0, st); }
            if (!(flags & SVO_FLAG_DETECT_NO_POST)) { Span s(ctx, KT_NMS); launch_nms_rowsort(d, p.non_maximal_suppression ? 0 : 3, p.min_distance, 0, st); }   // 0: the NMS ran above; 3: none, raster order
        } else {                // stage2_detect.cpp:458-497
            { Span s(ctx, KT_RESIZE); for (int l = 1; l < d.n_levels; l++) launch_resize(d, l, st); }
            { Span s(ctx, KT_FAST); launch_fast(d, st); }
            const bool split_early = (flags & SVO_FLAG_DETECT_NO_POST) && (flags & SVO_FLAG_DETECT_SPLIT_AT_SELECT);
            if (!split_early) { Span s(ctx, KT_SELECT); launch_select(d, st); }
            // The reference's NMS needs positions and responses only, so it runs BEFORE orientation + description and only
            // its survivors are described (debug mode 9 keeps the detector's order: describe everything, then NMS --
            // that is what svo_debug_get_raw_keypoints shows)
            const int nms_mode = p.non_maximal_suppression ? (p.nmsMethod == SVO_NMS_ADAPTIVE ? 2 : 1) : 0;
            if (d.debug_mode == 9) {
                { Span s(ctx, KT_DESCRIBE); launch_describe(d, 0, st); }
                if (!(flags & SVO_FLAG_DETECT_NO_POST)) { Span s(ctx, KT_NMS); launch_nms_rowsort(d, nms_mode, p.min_distance, 0, st); }
            } else if (!(flags & SVO_FLAG_DETECT_NO_POST)) {
                { Span s(ctx, KT_NMS); launch_nms_rowsort(d, nms_mode, p.min_distance, 1, st); }
                { Span s(ctx, KT_DESCRIBE); launch_describe(d, 1, st); }
            }
        }
    } else if (flags & SVO_RUN_DETECT_POST) {       // the post-processing a SVO_FLAG_DETECT_NO_POST call left out
        if (!ctx->geom_ready) return SVO_ERR_STATE;
        Section sec("_stg2");
        const int nms_mode = p.non_maximal_suppression ? (p.nmsMethod == SVO_NMS_ADAPTIVE ? 2 : 1) : 0;
        if ((flags & SVO_FLAG_DETECT_SPLIT_AT_SELECT) && !d.fast_orb) { Span s(ctx, KT_SELECT); launch_select(d, st); }
        if (d.fast_orb || d.debug_mode == 9) { Span s(ctx, KT_NMS); launch_nms_rowsort(d, d.fast_orb ? (p.non_maximal_suppression ? 0 : 3) : nms_mode, p.min_distance, 0, st); }
        else {
            { Span s(ctx, KT_NMS); launch_nms_rowsort(d, nms_mode, p.min_distance, 1, st); }
            { Span s(ctx, KT_DESCRIBE); launch_describe(d, 1, st); }
        }
    }
    if ((flags & SVO_RUN_DETECT_POST) || ((flags & SVO_RUN_DETECT) && !(flags & SVO_FLAG_DETECT_NO_POST))) HIPCHECK(after_post());
    const int nsplit = hamming_splits(ctx);
    if (flags & SVO_RUN_MATCH) {
        Section sec("_stg3"), sec2("stg3.find_pairings");                           // S3:64, 77
        // (the brute-force result words were set to all ones by k_begin_frame)
        if (p.match_method == SVO_SM_DESC_BF) {
            { Span s(ctx, KT_HAM_LR); launch_hamming(d, 0, nsplit, st); }
            { Span s(ctx, KT_LR_FILTER); launch_match_lr_filter(d, p.enable_robust_1to1_match, p.max_y_diff, st); }
        } else {                                                // smDescRbR (stage3_match_left_right.cpp:185-419)
            const double minresp = p.detect_method == SVO_DM_ORB ? p.minimum_ORB_response : 0.0;   // S3:189-193
            Span s(ctx, KT_LR_FILTER);
            launch_match_lr_rbr(d, p.enable_robust_1to1_match, p.max_y_diff, minresp, (int)(size_t)p.orb_max_distance, st);
        }
    }
    if (flags & SVO_RUN_TRACK) {
        Section sec("_stg4"), sec2("stg4.track");                                    // S4:76, 457
        const int win = p.ifm_method == SVO_IFM_DESC_WIN;
        // The chunks of the RANSAC's sample schedule (set before the tracker kernels: phase 0 of the schedule rides in them).  A handful of
        // lanes (one stream on its own): chunk 0 takes chunk 1's samples and more -- a hypothesis + count launch pair is ~20 us of latency
        // there, more than evaluating the samples an early bound might have saved.  SVO_RS_C0 = 32 | 160 | 320 forces a form (A/B, tests).
        {
            static const int forced = [] { const char* e = getenv("SVO_RS_C0"); const int v = e ? atoi(e) : 0; return (v == SVO_RANSAC_CHUNK0 || v == SVO_RANSAC_CHUNK1 || v == SVO_RANSAC_FEW) ? v : 0; }();
            d.rs_c0 = forced ? forced : (d.n_lanes * d.n_oct > 8 ? SVO_RANSAC_CHUNK0 : SVO_RANSAC_FEW);
            d.rs_c1 = d.rs_c0 > SVO_RANSAC_CHUNK1 ? d.rs_c0 : SVO_RANSAC_CHUNK1;
        }
        if (!win) {
            { Span s(ctx, KT_HAM_TRK); launch_hamming(d, 1, nsplit, st); }
            { Span s(ctx, KT_TRK_FILTER); launch_track_filter(d, st); }
        } else {                                                // ifmDescWin (stage4_match_consecutive.cpp:435-738)
            Span s(ctx, KT_TRK_FILTER); launch_track_win(d, p.ifm_win_w, p.ifm_win_h, st);
        }
        // F-matrix RANSAC: the first SVO_RANSAC_CHUNK0 hypotheses of the fixed schedule, then two more chunks, each only
        // as far as the 0.99-confidence stop of the sequential algorithm can still reach given what has been counted so far
        { Span s(ctx, KT_RANSAC_HYP); launch_ransac_hyp(d, 0, st); }
        { Span s(ctx, KT_RANSAC_CNT); launch_ransac_count(d, 0, st); }
        if (d.rs_c0 < d.rs_c1) {
            { Span s(ctx, KT_RANSAC_HYP1); launch_ransac_hyp(d, 1, st); }
            { Span s(ctx, KT_RANSAC_CNT1); launch_ransac_count(d, 1, st); }
        }
        { Span s(ctx, KT_RANSAC_HYP2); launch_ransac_hyp(d, 2, st); }
        { Span s(ctx, KT_RANSAC_CNT2); launch_ransac_count(d, 2, st); }
        { Span s(ctx, KT_TRK_FINAL); launch_track_finalize(d, p.bad_tracking_th, win, st); }
    }
    // (after svo_import_frame the lane may turn out to be at its stream's FIRST frame: the first-frame ID rule of S3:172-173
    // then applies in this call although stage 3 ran in an earlier one)
    if (p.vo_use_matches_ids && (flags & (SVO_RUN_MATCH | SVO_RUN_TRACK))) { Span s(ctx, KT_TRK_FINAL); launch_match_ids(d, flags | (ctx->imported_pending ? SVO_RUN_MATCH : 0), st); }
    ctx->imported_pending = false;
    if (flags & SVO_RUN_OPTIMIZE) {
        Section sec("_stg5");                                                        // S5:398, 729
        GNParams g; memset(&g, 0, sizeof(g));
        g.use_robust_kernel = p.use_robust_kernel; g.max_iters = p.max_iters; g.initial_max_iters = p.initial_max_iters; g.max_incr_cost = p.max_incr_cost;
        g.use_previous_pose_as_initial = p.use_previous_pose_as_initial; g.use_custom_initial_pose = 0;   // processNewImagePair passes no initial estimate (P:338)
        g.min_distance = p.min_distance; g.img_w = d.W; g.img_h = d.H; g.pmax = d.max_kps; g.standalone = 0;
        g.kernel_param = p.kernel_param; g.min_mod_out_vector = p.min_mod_out_vector; g.residual_threshold = p.residual_threshold;
        if (p.use_custom_initial_pose) g.use_custom_initial_pose = 1;        // deltaPose = initial_estimation = zeros (S5:504-505, default argument H:1045)
        { Span s(ctx, KT_GN); launch_gauss_newton(d, g, st); }
    }
    if (capturing) {
        hipGraph_t graph = nullptr;
        capturing = false;
        HIPCHECK(hipStreamEndCapture(st, &graph));
        hipGraphExec_t exec = nullptr;
        const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        HIPCHECK(e);
        ctx->graphs.push_back({ gflags, gslot, ctx->fast_th, ctx->orb_th, exec });
        HIPCHECK(hipGraphLaunch(exec, st));
        if ((flags & SVO_RUN_DETECT_POST) || ((flags & SVO_RUN_DETECT) && !(flags & SVO_FLAG_DETECT_NO_POST))) HIPCHECK(after_post());
    }
    // stage 2 is the only reader of the level-0 images: once it is through, the ring slot may take the next upload
    if ((flags & (SVO_RUN_DETECT | SVO_RUN_DETECT_POST)) && ctx->up_ready && ctx->det_slot >= 0) {
        HIPCHECK(hipEventRecord(ctx->ev_det[ctx->det_slot], st)); ctx->ev_det_valid[ctx->det_slot] = true;
    }
    HIPCHECK(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_record_after_post(svo_ctx* ctx, void* event)
{
    if (!ctx) return SVO_ERR_ARG;
    ctx->post_event = (hipEvent_t)event;
    return SVO_OK;
}

// ---- getters ------------------------------------------------------------------------------------------------
static int lane_state(svo_ctx* ctx, int lane, LaneState* s)
{
    int rc = svo_wait(ctx); if (rc) return rc;
    HIPCHECK(hipMemcpy(s, ctx->dc.lane + lane, sizeof(LaneState), hipMemcpyDeviceToHost));
    return SVO_OK;
}
static int slot_of(const LaneState& s, int which) { return which ? s.prev_slot : 1 - s.prev_slot; }

extern "C" int svo_get_results(svo_ctx* ctx, svo_result* res)
{
    if (ctx) use_device(ctx);
    if (!ctx || !res) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    HIPCHECK(hipMemcpy(res, ctx->dc.results, sizeof(svo_result) * ctx->cfg.n_lanes, hipMemcpyDeviceToHost));
    return SVO_OK;
}
extern "C" int svo_copy_results_async(svo_ctx* ctx, void* dst, size_t bytes)
{
    if (ctx) use_device(ctx);
    if (!ctx || !dst || bytes < sizeof(svo_result) * (size_t)ctx->cfg.n_lanes) return SVO_ERR_ARG;
    note_stream(ctx);
    HIPCHECK(hipMemcpyAsync(dst, ctx->dc.results, sizeof(svo_result) * ctx->cfg.n_lanes, hipMemcpyDeviceToDevice, ctx->stream));
    mark_stream(ctx);
    return SVO_OK;
}
extern "C" int svo_get_result(svo_ctx* ctx, int lane, svo_result* res)
{
    if (ctx) use_device(ctx);
    if (!ctx || !res || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    HIPCHECK(hipMemcpy(res, ctx->dc.results + lane, sizeof(svo_result), hipMemcpyDeviceToHost));
    return SVO_OK;
}

extern "C" int svo_get_keypoints_oct(svo_ctx* ctx, int lane, int which, int side, int octave, svo_keypoint* kps, uint8_t* desc, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || (side | 1) != 1 || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    if (which ? !s.has_prev : !s.has_cur) return 0;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave;
    int n = 0;
    HIPCHECK(hipMemcpy(&n, ctx->dc.n_kps + (vl * 2 + slot) * 2 + side, sizeof(int), hipMemcpyDeviceToHost));
    const int m = n < cap ? n : cap;
    const long long base = (((long long)vl * 2 + slot) * 2 + side) * ctx->dc.max_kps;
    if (kps && m > 0) HIPCHECK(hipMemcpy(kps, ctx->dc.kps + base, sizeof(svo_keypoint) * m, hipMemcpyDeviceToHost));
    if (desc && m > 0) HIPCHECK(hipMemcpy(desc, ctx->dc.desc + base * 32, (size_t)32 * m, hipMemcpyDeviceToHost));
    return n;
}
extern "C" int svo_get_keypoints(svo_ctx* ctx, int lane, int which, int side, svo_keypoint* kps, uint8_t* desc, int cap)
{
    if (ctx) use_device(ctx);
    return svo_get_keypoints_oct(ctx, lane, which, side, 0, kps, desc, cap);
}

extern "C" int svo_get_row_index(svo_ctx* ctx, int lane, int which, int side, int octave, int32_t* idx, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || !ctx->geom_ready || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || (side | 1) != 1 || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave, H = ctx->dc.oh[octave];
    const int m = H < cap ? H : cap;
    if (idx && m > 0) HIPCHECK(hipMemcpy(idx, ctx->dc.row_index + (long long)((vl * 2 + slot) * 2 + side) * ctx->dc.max_h, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    return H;
}

extern "C" int svo_get_matches_oct(svo_ctx* ctx, int lane, int which, int octave, svo_dmatch* mm, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    if (which ? !s.has_prev : !s.has_cur) return 0;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave;
    int n = 0;
    HIPCHECK(hipMemcpy(&n, ctx->dc.n_matches + vl * 2 + slot, sizeof(int), hipMemcpyDeviceToHost));
    const int m = n < cap ? n : cap;
    if (mm && m > 0) HIPCHECK(hipMemcpy(mm, ctx->dc.matches + ((long long)vl * 2 + slot) * ctx->dc.max_kps, sizeof(svo_dmatch) * m, hipMemcpyDeviceToHost));
    return n;
}
extern "C" int svo_get_matches(svo_ctx* ctx, int lane, int which, svo_dmatch* mm, int cap) { return svo_get_matches_oct(ctx, lane, which, 0, mm, cap); }

// ---- getValues in one synchronisation (H:704-724) ---------------------------------------------------------------
extern "C" int svo_get_values(svo_ctx* ctx, int lane, int which, int octave, svo_values* v)
{
    if (ctx) use_device(ctx);
    if (!ctx || !v || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || octave < 0 || octave >= ctx->dc.oct_cap || v->cap_kps < 0 || v->cap_matches < 0) return SVO_ERR_ARG;
    const int MK = ctx->dc.max_kps;
    if (!ctx->d_vals) {
        ctx->vals_bytes = 64 + (size_t)MK * (2 * (sizeof(svo_keypoint) + 32) + sizeof(svo_dmatch) + sizeof(int32_t));
        HIPCHECK(dev_alloc(ctx, &ctx->d_vals, ctx->vals_bytes));
        HIPCHECK(hipHostMalloc((void**)&ctx->h_vals, ctx->vals_bytes, hipHostMallocDefault));
    }
    // whatever the context still has in flight on other streams must be done; the pack + copy below then run on the
    // current stream and the single synchronisation of this call waits for them
    HIPCHECK(sync_foreign(ctx));
    note_stream(ctx);
    launch_pack_values(ctx->dc, lane, which, octave, ctx->d_vals, ctx->stream);
    HIPCHECK(hipMemcpyAsync(ctx->h_vals, ctx->d_vals, ctx->vals_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));                                      // THE synchronisation
    collect_spans(ctx);
    const int32_t* hdr = (const int32_t*)ctx->h_vals;
    v->n_left = hdr[0]; v->n_right = hdr[1]; v->n_matches = hdr[2]; v->n_ids = hdr[3];
    const uint8_t* b = ctx->h_vals + 64;
    const uint8_t* kl = b, *kr = kl + (size_t)MK * sizeof(svo_keypoint), *dl = kr + (size_t)MK * sizeof(svo_keypoint), *dr = dl + (size_t)MK * 32;
    const uint8_t* mm = dr + (size_t)MK * 32, *ii = mm + (size_t)MK * sizeof(svo_dmatch);
    const int nl = v->n_left < v->cap_kps ? v->n_left : v->cap_kps, nr = v->n_right < v->cap_kps ? v->n_right : v->cap_kps;
    const int nm = v->n_matches < v->cap_matches ? v->n_matches : v->cap_matches, ni = v->n_ids < v->cap_matches ? v->n_ids : v->cap_matches;
    if (v->left_kps && nl > 0) memcpy(v->left_kps, kl, sizeof(svo_keypoint) * nl);
    if (v->right_kps && nr > 0) memcpy(v->right_kps, kr, sizeof(svo_keypoint) * nr);
    if (v->left_desc && nl > 0) memcpy(v->left_desc, dl, (size_t)32 * nl);
    if (v->right_desc && nr > 0) memcpy(v->right_desc, dr, (size_t)32 * nr);
    if (v->matches && nm > 0) memcpy(v->matches, mm, sizeof(svo_dmatch) * nm);
    if (v->match_ids && ni > 0) memcpy(v->match_ids, ii, sizeof(int32_t) * ni);
    return SVO_OK;
}

extern "C" int svo_get_matches_row_index(svo_ctx* ctx, int lane, int which, int octave, int32_t* idx, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || !ctx->geom_ready || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave, H1 = ctx->dc.oh[octave] + 1;
    const int m = H1 < cap ? H1 : cap;
    if (idx && m > 0) HIPCHECK(hipMemcpy(idx, ctx->dc.mrow_index + (long long)(vl * 2 + slot) * (ctx->dc.max_h + 1), sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    return H1;
}

extern "C" int svo_get_tracked_oct(svo_ctx* ctx, int lane, int octave, svo_index_pair* t, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    const int vl = lane * ctx->dc.oct_cap + octave;
    int n = 0;
    HIPCHECK(hipMemcpy(&n, ctx->dc.n_tracked + vl, sizeof(int), hipMemcpyDeviceToHost));
    const int m = n < cap ? n : cap;
    if (t && m > 0) HIPCHECK(hipMemcpy(t, ctx->dc.tracked + (long long)vl * ctx->dc.max_kps, sizeof(svo_index_pair) * m, hipMemcpyDeviceToHost));
    return n;
}
extern "C" int svo_get_tracked(svo_ctx* ctx, int lane, svo_index_pair* t, int cap) { return svo_get_tracked_oct(ctx, lane, 0, t, cap); }

extern "C" int svo_get_match_ids(svo_ctx* ctx, int lane, int which, int octave, int32_t* ids, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    if (which ? !s.has_prev : !s.has_cur) return 0;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave;
    int n = 0;
    HIPCHECK(hipMemcpy(&n, ctx->dc.n_ids + vl * 2 + slot, sizeof(int), hipMemcpyDeviceToHost));
    const int m = n < cap ? n : cap;
    if (ids && m > 0) HIPCHECK(hipMemcpy(ids, ctx->dc.ids + ((long long)vl * 2 + slot) * ctx->dc.max_kps, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    return n;
}

// resetIds (H:684): the next frame renumbers the previous IDs and becomes the key frame (P:254-267)
extern "C" int svo_reset_ids(svo_ctx* ctx, int lane)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < -1 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    for (int l = 0; l < ctx->cfg.n_lanes; l++)
        if (lane < 0 || lane == l) { LaneState s; int rc = lane_state(ctx, l, &s); if (rc) return rc; s.reset_ids = 1; HIPCHECK(hipMemcpy(ctx->dc.lane + l, &s, sizeof(s), hipMemcpyHostToDevice)); }
    return SVO_OK;
}

// setThisFrameAsKF (H:675-683): m_last_kf_max_id = max ID of the current frame's octave-0 pairings
extern "C" int svo_set_this_frame_as_kf(svo_ctx* ctx, int lane)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    if (!s.has_cur) return SVO_ERR_STATE;                 // ASSERTMSG_ "Current frame does not exist" (H:677)
    const int vl = lane * ctx->dc.oct_cap, slot = slot_of(s, 0);
    int n = 0;
    HIPCHECK(hipMemcpy(&n, ctx->dc.n_ids + vl * 2 + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n <= 0) return SVO_ERR_STATE;
    std::vector<int32_t> ids((size_t)n);
    HIPCHECK(hipMemcpy(ids.data(), ctx->dc.ids + ((long long)vl * 2 + slot) * ctx->dc.max_kps, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    int mx = ids[0]; for (int i = 1; i < n; i++) if (ids[i] > mx) mx = ids[i];
    s.last_kf_max_id = mx;
    HIPCHECK(hipMemcpy(ctx->dc.lane + lane, &s, sizeof(s), hipMemcpyHostToDevice));
    return SVO_OK;
}

extern "C" int svo_get_residuals(svo_ctx* ctx, int lane, double* r, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    svo_result res; int rc = svo_get_result(ctx, lane, &res); if (rc) return rc;
    const int n = res.n_residual, m = n < cap ? n : cap;
    if (r && m > 0) HIPCHECK(hipMemcpy(r, ctx->dc.residual + (long long)lane * ctx->dc.max_kps, sizeof(double) * m, hipMemcpyDeviceToHost));
    return n;
}

extern "C" int svo_get_outliers(svo_ctx* ctx, int lane, int32_t* idx, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    svo_result res; int rc = svo_get_result(ctx, lane, &res); if (rc) return rc;
    const int n = res.n_outliers, m = n < cap ? n : cap;
    if (idx && m > 0) HIPCHECK(hipMemcpy(idx, ctx->dc.outliers + (long long)lane * ctx->dc.max_kps, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    return n;
}

// ---- precomputed-data bypass ----------------------------------------------------------------------------------
static int mark_present(svo_ctx* ctx, int lane, int which)
{
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    if (which) s.has_prev = 1; else s.has_cur = 1;
    HIPCHECK(hipMemcpy(ctx->dc.lane + lane, &s, sizeof(s), hipMemcpyHostToDevice));
    return SVO_OK;
}

extern "C" int svo_put_features_oct(svo_ctx* ctx, int lane, int which, int side, int octave, const svo_keypoint* kps, const uint8_t* desc, int n, int img_w, int img_h)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || (side | 1) != 1 || n < 0 || (n > 0 && !kps) || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    if (n > ctx->dc.max_kps) return SVO_ERR_CAPACITY;
    int rc = ensure_geometry(ctx, img_w, img_h); if (rc) return rc;
    if (octave >= ctx->dc.n_oct) return SVO_ERR_ARG;                       // P:139: the octave count of the data must match the system's
    LaneState s; rc = lane_state(ctx, lane, &s); if (rc) return rc;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave;
    const long long base = (((long long)vl * 2 + slot) * 2 + side) * ctx->dc.max_kps;
    if (n > 0) HIPCHECK(hipMemcpy(ctx->dc.kps + base, kps, sizeof(svo_keypoint) * n, hipMemcpyHostToDevice));
    if (n > 0 && desc) HIPCHECK(hipMemcpy(ctx->dc.desc + base * 32, desc, (size_t)32 * n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->dc.n_kps + (vl * 2 + slot) * 2 + side, &n, sizeof(int), hipMemcpyHostToDevice));
    if (which == 0) {                                                      // result.detected_feats[octave] (P:171-176)
        int32_t* cnt = side ? ctx->dc.results[lane].detected_right : ctx->dc.results[lane].detected_left;
        HIPCHECK(hipMemcpy(cnt + octave, &n, sizeof(int), hipMemcpyHostToDevice));
    }
    return mark_present(ctx, lane, which);
}
extern "C" int svo_put_features(svo_ctx* ctx, int lane, int which, int side, const svo_keypoint* kps, const uint8_t* desc, int n, int img_w, int img_h)
{
    if (ctx) use_device(ctx);
    return svo_put_features_oct(ctx, lane, which, side, 0, kps, desc, n, img_w, img_h);
}

extern "C" int svo_put_matches_oct(svo_ctx* ctx, int lane, int which, int octave, const svo_dmatch* m, int n)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || n < 0 || (n > 0 && !m) || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    if (n > ctx->dc.max_kps) return SVO_ERR_CAPACITY;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave;
    if (n > 0) HIPCHECK(hipMemcpy(ctx->dc.matches + ((long long)vl * 2 + slot) * ctx->dc.max_kps, m, sizeof(svo_dmatch) * n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->dc.n_matches + vl * 2 + slot, &n, sizeof(int), hipMemcpyHostToDevice));
    if (which == 0) HIPCHECK(hipMemcpy(ctx->dc.results[lane].stereo_matches + octave, &n, sizeof(int), hipMemcpyHostToDevice));   // P:274-276
    // matches_lr_row_index (S3:425-445) of the list just put.  The reference builds it in stage 3 only, which the precomputed-data path
    // (P:219-251) and the state loader skip: its windowed tracker (S4:517-530) then reads an index nobody built.  Here it is built from the
    // left keypoints put BEFORE the pairings (the oracle's sequential rule, ri[H] = n as there), so that ifmDescWin works on caller data.
    if (ctx->geom_ready && octave < ctx->dc.n_oct) {
        int nl = 0;
        const long long kbase = (((long long)vl * 2 + slot) * 2 + 0) * ctx->dc.max_kps;
        HIPCHECK(hipMemcpy(&nl, ctx->dc.n_kps + (vl * 2 + slot) * 2 + 0, sizeof(int), hipMemcpyDeviceToHost));
        nl = std::max(0, std::min(nl, ctx->dc.max_kps));
        bool ok = true;
        for (int i = 0; i < n && ok; i++) ok = m[i].queryIdx >= 0 && m[i].queryIdx < nl;
        if (ok) {
            std::vector<svo_keypoint> kl((size_t)nl);
            if (nl > 0) HIPCHECK(hipMemcpy(kl.data(), ctx->dc.kps + kbase, sizeof(svo_keypoint) * nl, hipMemcpyDeviceToHost));
            const int H = std::max(0, std::min(ctx->dc.oh[octave], ctx->dc.max_h));
            std::vector<int32_t> ri((size_t)H + 1);
            int idx = 0;
            for (int y = 0; y < H; y++) {
                ri[y] = idx;
                while (idx < n && kl[m[idx].queryIdx].y <= (float)y) idx++;               // S3:441
            }
            ri[H] = n;
            HIPCHECK(hipMemcpy(ctx->dc.mrow_index + (long long)(vl * 2 + slot) * (ctx->dc.max_h + 1), ri.data(), sizeof(int32_t) * (H + 1), hipMemcpyHostToDevice));
        }
    }
    return mark_present(ctx, lane, which);
}
extern "C" int svo_put_matches(svo_ctx* ctx, int lane, int which, const svo_dmatch* m, int n) { return svo_put_matches_oct(ctx, lane, which, 0, m, n); }

extern "C" int svo_put_match_ids_oct(svo_ctx* ctx, int lane, int which, int octave, const int32_t* ids, int n)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || (which | 1) != 1 || n < 0 || (n > 0 && !ids) || octave < 0 || octave >= ctx->dc.oct_cap) return SVO_ERR_ARG;
    if (n > ctx->dc.max_kps) return SVO_ERR_CAPACITY;
    LaneState s; int rc = lane_state(ctx, lane, &s); if (rc) return rc;
    const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap + octave;
    if (n > 0) HIPCHECK(hipMemcpy(ctx->dc.ids + ((long long)vl * 2 + slot) * ctx->dc.max_kps, ids, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->dc.n_ids + vl * 2 + slot, &n, sizeof(int), hipMemcpyHostToDevice));
    int mx = octave == 0 ? 0 : s.last_match_id;                                                    // P:236: m_last_match_ID = 0, then the maximum over the octaves (P:243)
    for (int i = 0; i < n; i++) if (ids[i] > mx) mx = ids[i];
    s.last_match_id = mx;
    HIPCHECK(hipMemcpy(ctx->dc.lane + lane, &s, sizeof(s), hipMemcpyHostToDevice));
    return SVO_OK;
}
extern "C" int svo_put_match_ids(svo_ctx* ctx, int lane, int which, const int32_t* ids, int n) { return svo_put_match_ids_oct(ctx, lane, which, 0, ids, n); }

extern "C" int svo_put_tracked(svo_ctx* ctx, int lane, const svo_index_pair* t, int n)
{
    if (ctx) use_device(ctx);
    if (!ctx || lane < 0 || lane >= ctx->cfg.n_lanes || n < 0 || (n > 0 && !t)) return SVO_ERR_ARG;
    if (n > ctx->dc.max_kps) return SVO_ERR_CAPACITY;
    int rc = svo_wait(ctx); if (rc) return rc;
    const int vl = lane * ctx->dc.oct_cap;                                   // octave 0; the other octaves of the lane are emptied
    HIPCHECK(hipMemset(ctx->dc.n_tracked + vl, 0, sizeof(int) * ctx->dc.oct_cap));
    if (n > 0) HIPCHECK(hipMemcpy(ctx->dc.tracked + (long long)vl * ctx->dc.max_kps, t, sizeof(svo_index_pair) * n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->dc.n_tracked + vl, &n, sizeof(int), hipMemcpyHostToDevice));
    return SVO_OK;
}

// ---- getProjectedCoords (common.cpp:415-466) --------------------------------------------------------------------
// delta = [rotation vector, translation] of the inverse of change_pose (C:456-461); same operations as
// svo_oracle_pose_to_delta (log map through the unit quaternion)
static void pose_to_delta(const double* pose, double* dp)
{
    const double cy = cos(pose[3]), sy = sin(pose[3]), cp = cos(pose[4]), sp = sin(pose[4]), cr = cos(pose[5]), sr = sin(pose[5]);
    const double R[9] = { cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                          sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                          -sp, cp * sr, cp * cr };
    const double Ri[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };
    dp[3] = -(Ri[0] * pose[0] + Ri[1] * pose[1] + Ri[2] * pose[2]);
    dp[4] = -(Ri[3] * pose[0] + Ri[4] * pose[1] + Ri[5] * pose[2]);
    dp[5] = -(Ri[6] * pose[0] + Ri[7] * pose[1] + Ri[8] * pose[2]);
    double q0, q1, q2, q3;
    const double tr = Ri[0] + Ri[4] + Ri[8];
    if (tr > 0.0) { const double s = sqrt(tr + 1.0) * 2.0; q0 = 0.25 * s; q1 = (Ri[7] - Ri[5]) / s; q2 = (Ri[2] - Ri[6]) / s; q3 = (Ri[3] - Ri[1]) / s; }
    else if (Ri[0] > Ri[4] && Ri[0] > Ri[8]) { const double s = sqrt(1.0 + Ri[0] - Ri[4] - Ri[8]) * 2.0; q0 = (Ri[7] - Ri[5]) / s; q1 = 0.25 * s; q2 = (Ri[1] + Ri[3]) / s; q3 = (Ri[2] + Ri[6]) / s; }
    else if (Ri[4] > Ri[8]) { const double s = sqrt(1.0 + Ri[4] - Ri[0] - Ri[8]) * 2.0; q0 = (Ri[2] - Ri[6]) / s; q1 = (Ri[1] + Ri[3]) / s; q2 = 0.25 * s; q3 = (Ri[5] + Ri[7]) / s; }
    else { const double s = sqrt(1.0 + Ri[8] - Ri[0] - Ri[4]) * 2.0; q0 = (Ri[3] - Ri[1]) / s; q1 = (Ri[2] + Ri[6]) / s; q2 = (Ri[5] + Ri[7]) / s; q3 = 0.25 * s; }
    if (q0 < 0.0) { q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3; }
    const double vn = sqrt(q1 * q1 + q2 * q2 + q3 * q3);
    if (vn < 1e-12) { dp[0] = 2.0 * q1; dp[1] = 2.0 * q2; dp[2] = 2.0 * q3; }
    else { const double k = 2.0 * atan2(vn, q0) / vn; dp[0] = k * q1; dp[1] = k * q2; dp[2] = k * q3; }
}

extern "C" int svo_projected_coords(svo_ctx* ctx, const svo_dmatch* pre_matches, int n_pre, const svo_keypoint* pre_left, int n_left,
                                    const svo_keypoint* pre_right, int n_right, const int32_t* tracked_first,
                                    const svo_stereo_camera* cam, const double* change_pose6, float* pix, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || n_pre < 0 || (n_pre > 0 && (!pre_matches || !pre_left || !pre_right || !tracked_first)) || !cam || !change_pose6) return SVO_ERR_ARG;
    std::vector<float> uvu;
    for (int m = 0; m < n_pre; m++) {
        if (tracked_first[m] != -1) continue;                                   // C:430-431
        const int l = pre_matches[m].queryIdx, r = pre_matches[m].trainIdx;
        if (l < 0 || l >= n_left || r < 0 || r >= n_right) return SVO_ERR_ARG;
        uvu.push_back(pre_left[l].x); uvu.push_back(pre_left[l].y); uvu.push_back(pre_right[r].x);
    }
    const int nb = (int)(uvu.size() / 3);
    if (!pix || cap < nb || nb == 0) return nb;                                 // size query / nothing to do
    double dp[6]; pose_to_delta(change_pose6, dp);
    float* d_in = nullptr, *d_out = nullptr;
    HIPCHECK(hipMalloc((void**)&d_in, uvu.size() * sizeof(float)));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)nb * 4 * sizeof(float));
    if (e != hipSuccess) { hipFree(d_in); HIPCHECK(e); }
    const hipStream_t st = ctx->stream;
    e = hipMemcpyAsync(d_in, uvu.data(), uvu.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { launch_project_points(d_in, nb, *cam, dp, d_out, st); e = hipMemcpyAsync(pix, d_out, (size_t)nb * 4 * sizeof(float), hipMemcpyDeviceToHost, st); }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_in); hipFree(d_out);
    HIPCHECK(e);
    return nb;
}

// ---- frame hand-over between contexts (k_handover.hip) ----------------------------------------------------------
extern "C" size_t svo_handover_bytes(const svo_ctx* ctx)
{
    return ctx ? handover_record_bytes(ctx->dc) * (size_t)ctx->dc.n_lanes * (size_t)ctx->dc.oct_cap : 0;
}
extern "C" int svo_export_frame(svo_ctx* ctx, void* dev_blob, size_t bytes)
{
    if (ctx) use_device(ctx);
    if (!ctx || !dev_blob || bytes < svo_handover_bytes(ctx)) return SVO_ERR_ARG;
    if (!ctx->geom_ready) return SVO_ERR_STATE;
    note_stream(ctx);
    launch_export_frame(ctx->dc, (uint8_t*)dev_blob, ctx->stream);
    mark_stream(ctx);
    HIPCHECK(hipGetLastError());
    return SVO_OK;
}
extern "C" int svo_import_frame(svo_ctx* ctx, const void* dev_blob, size_t bytes)
{
    if (ctx) use_device(ctx);
    if (!ctx || !dev_blob || bytes < svo_handover_bytes(ctx)) return SVO_ERR_ARG;
    if (!ctx->geom_ready) return SVO_ERR_STATE;
    note_stream(ctx);
    launch_import_frame(ctx->dc, (const uint8_t*)dev_blob, ctx->stream);
    ctx->imported_pending = true;
    mark_stream(ctx);
    HIPCHECK(hipGetLastError());
    return SVO_OK;
}

// ---- saveStateToFile / loadStateFromFile (common.cpp:475-543, 261-350; helpers :88-255) --------------------------
namespace {
struct StateList { std::vector<svo_keypoint> kps; std::vector<uint8_t> desc; };
bool wr(FILE* f, const void* p, size_t n) { return fwrite(p, 1, n, f) == n; }
bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }
bool dump_keypoints(FILE* f, const StateList& L)              // m_dump_keypoints_to_stream (C:88-133)
{
    const uint64_t n = L.kps.size();
    if (!wr(f, &n, 8)) return false;
    for (const svo_keypoint& k : L.kps) {
        const float v[5] = { k.x, k.y, k.response, k.size, k.angle };
        const int32_t w[2] = { k.octave, k.class_id };
        if (!wr(f, v, sizeof(v)) || !wr(f, w, sizeof(w))) return false;
    }
    const int32_t hdr[3] = { (int32_t)n, n ? 32 : 0, 0 /* CV_8UC1 */ };
    return wr(f, hdr, sizeof(hdr)) && (L.desc.empty() || wr(f, L.desc.data(), L.desc.size()));
}
bool load_keypoints(FILE* f, StateList& L, size_t cap)         // m_load_keypoints_from_stream (C:168-211)
{
    uint64_t n = 0;
    if (!rd(f, &n, 8) || n > cap) return false;
    L.kps.resize((size_t)n);
    for (svo_keypoint& k : L.kps) {
        float v[5]; int32_t w[2];
        if (!rd(f, v, sizeof(v)) || !rd(f, w, sizeof(w))) return false;
        k.x = v[0]; k.y = v[1]; k.response = v[2]; k.size = v[3]; k.angle = v[4]; k.octave = w[0]; k.class_id = w[1];
    }
    int32_t hdr[3];
    if (!rd(f, hdr, sizeof(hdr)) || hdr[0] < 0 || hdr[1] < 0) return false;
    if ((uint64_t)hdr[0] != n || (n && hdr[1] != 32)) return false;             // this path only knows 256-bit descriptors
    L.desc.resize((size_t)n * 32);
    return L.desc.empty() || rd(f, L.desc.data(), L.desc.size());
}
bool dump_matches(FILE* f, const std::vector<svo_dmatch>& m, const std::vector<int32_t>& ids)   // m_dump_matches_to_stream (C:138-163)
{
    const uint64_t n = m.size(), ni = ids.size();
    if (!wr(f, &n, 8) || !wr(f, &ni, 8)) return false;
    for (size_t i = 0; i < m.size(); i++) {
        if (n == ni) { const uint64_t id = (uint64_t)(int64_t)ids[i]; if (!wr(f, &id, 8)) return false; }
        if (!wr(f, &m[i].queryIdx, 4) || !wr(f, &m[i].trainIdx, 4) || !wr(f, &m[i].distance, 4) || !wr(f, &m[i].imgIdx, 4)) return false;
    }
    return true;
}
bool load_matches(FILE* f, std::vector<svo_dmatch>& m, std::vector<int32_t>& ids, size_t cap)    // m_load_matches_from_stream (C:216-255)
{
    uint64_t n = 0, ni = 0;
    if (!rd(f, &n, 8) || !rd(f, &ni, 8) || n > cap || ni > cap) return false;
    m.resize((size_t)n); ids.assign((size_t)ni, 0);
    for (size_t i = 0; i < m.size(); i++) {
        if (n == ni) { uint64_t id; if (!rd(f, &id, 8)) return false; ids[i] = (int32_t)id; }
        if (!rd(f, &m[i].queryIdx, 4) || !rd(f, &m[i].trainIdx, 4) || !rd(f, &m[i].distance, 4) || !rd(f, &m[i].imgIdx, 4)) return false;
    }
    return true;
}
}  // namespace

extern "C" int svo_save_state(svo_ctx* ctx, int lane, const char* path)
{
    if (ctx) use_device(ctx);
    if (!ctx || !path || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    if (ctx->dc.n_oct > 1) return SVO_ERR_UNSUPPORTED;                 // the reference's format holds one list per eye
    int rc = svo_wait(ctx); if (rc) return rc;
    LaneState s; rc = lane_state(ctx, lane, &s); if (rc) return rc;
    StateList L[2][2]; std::vector<svo_dmatch> M[2]; std::vector<int32_t> I[2];           // [which: 0 cur, 1 prev]
    for (int which = 0; which < 2; which++) {
        for (int side = 0; side < 2; side++) {
            const int n = svo_get_keypoints_oct(ctx, lane, which, side, 0, nullptr, nullptr, 0); if (n < 0) return n;
            L[which][side].kps.resize(n); L[which][side].desc.resize((size_t)n * 32);
            if (n) { rc = svo_get_keypoints_oct(ctx, lane, which, side, 0, L[which][side].kps.data(), L[which][side].desc.data(), n); if (rc < 0) return rc; }
        }
        int n = svo_get_matches_oct(ctx, lane, which, 0, nullptr, 0); if (n < 0) return n;
        M[which].resize(n); if (n) { rc = svo_get_matches_oct(ctx, lane, which, 0, M[which].data(), n); if (rc < 0) return rc; }
        n = svo_get_match_ids(ctx, lane, which, 0, nullptr, 0); if (n < 0) return n;
        I[which].resize(n); if (n) { rc = svo_get_match_ids(ctx, lane, which, 0, I[which].data(), n); if (rc < 0) return rc; }
    }
    svo_result res; HIPCHECK(hipMemcpy(&res, ctx->dc.results + lane, sizeof(res), hipMemcpyDeviceToHost));
    FILE* f = fopen(path, "wb");
    if (!f) { ctx->last_error = std::string("cannot open ") + path; return SVO_ERR_ARG; }
    const uint64_t npyr = 1;
    bool ok = wr(f, &npyr, 8);
    for (int which = 1; which >= 0 && ok; which--)                       // PRE first, then CUR (C:491-527)
        ok = dump_keypoints(f, L[which][0]) && dump_keypoints(f, L[which][1]) && dump_matches(f, M[which], I[which]);
    const uint8_t m_reset = s.reset_ids ? 1 : 0;
    const uint64_t tail[5] = { 0 /* m_lastID: legacy, never used */, (uint64_t)s.num_tracked_last_kf, (uint64_t)res.tracked_feats_from_last_frame,
                               (uint64_t)s.last_match_id, (uint64_t)s.last_kf_max_id };
    ok = ok && wr(f, &m_reset, 1) && wr(f, tail, sizeof(tail));
    ok = (fclose(f) == 0) && ok;
    if (!ok) { ctx->last_error = std::string("short write to ") + path; return SVO_ERR_STATE; }
    return SVO_OK;
}

extern "C" int svo_load_state(svo_ctx* ctx, int lane, const char* path)
{
    if (ctx) use_device(ctx);
    if (!ctx || !path || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    if (ctx->dc.oct_cap > 1 && ctx->dc.n_oct > 1) return SVO_ERR_UNSUPPORTED;
    FILE* f = fopen(path, "rb");
    if (!f) { ctx->last_error = std::string("cannot open ") + path; return SVO_ERR_ARG; }
    StateList L[2][2]; std::vector<svo_dmatch> M[2]; std::vector<int32_t> I[2];
    uint64_t npyr = 0, tail[5]; uint8_t m_reset = 0;
    const size_t cap = (size_t)ctx->dc.max_kps;
    bool ok = rd(f, &npyr, 8);
    for (int which = 1; which >= 0 && ok; which--)
        ok = load_keypoints(f, L[which][0], cap) && load_keypoints(f, L[which][1], cap) && load_matches(f, M[which], I[which], cap);
    ok = ok && rd(f, &m_reset, 1) && rd(f, tail, sizeof(tail));
    fclose(f);
    if (!ok) { ctx->last_error = std::string("malformed or truncated state file ") + path; return SVO_ERR_ARG; }
    // the file carries no image size: the geometry of the last frame, or the context's maximum for a fresh context
    const int gw = ctx->geom_ready ? ctx->geom_w : ctx->cfg.max_w, gh = ctx->geom_ready ? ctx->geom_h : ctx->cfg.max_h;
    int rc = svo_reset(ctx, lane); if (rc) return rc;
    for (int which = 1; which >= 0; which--) {
        for (int side = 0; side < 2; side++) {
            rc = svo_put_features(ctx, lane, which, side, L[which][side].kps.data(), L[which][side].desc.data(), (int)L[which][side].kps.size(), gw, gh);
            if (rc) return rc;
        }
        rc = svo_put_matches(ctx, lane, which, M[which].data(), (int)M[which].size()); if (rc) return rc;
        LaneState s; rc = lane_state(ctx, lane, &s); if (rc) return rc;
        const int slot = slot_of(s, which), vl = lane * ctx->dc.oct_cap, ni = (int)I[which].size();
        if (ni > 0) HIPCHECK(hipMemcpy(ctx->dc.ids + ((long long)vl * 2 + slot) * ctx->dc.max_kps, I[which].data(), sizeof(int32_t) * ni, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(ctx->dc.n_ids + vl * 2 + slot, &ni, sizeof(int), hipMemcpyHostToDevice));
    }
    LaneState s; rc = lane_state(ctx, lane, &s); if (rc) return rc;
    s.reset_ids = m_reset; s.num_tracked_last_kf = (int)tail[1]; s.last_match_id = (int)tail[3]; s.last_kf_max_id = (int)tail[4];
    HIPCHECK(hipMemcpy(ctx->dc.lane + lane, &s, sizeof(s), hipMemcpyHostToDevice));
    return SVO_OK;
}

// ---- getChangeInPose (common.cpp:355-413) -------------------------------------------------------------------
// The reference packs the caller's arrays into TEMPORARY TImagePairData / TTrackingData (C:362-400) and runs
// stage5_optimize on them: m_prev_imgpair / m_current_imgpair are not touched.  The only estimator members the call
// shares with the pipeline are the ones stage 5 itself reads and writes: m_last_computed_pose (S5:506-507, 720-721)
// and m_error (S5:380-386).  Same here: the Gauss-Newton kernel runs on a ONE-LANE SCRATCH VIEW of the device context
// (its own keypoint / pairing / track / residual buffers and lane record); lane 0's warm start and m_error are copied
// into it before the launch and back afterwards, nothing else of the live lanes is read or written.
static int ensure_cip(svo_ctx* ctx)
{
    if (ctx->cip_ready) return SVO_OK;
    DevCtx& s = ctx->cip;
    s = ctx->dc;
    const int MK = ctx->dc.max_kps;
    s.n_lanes = 1; s.n_img = 2; s.oct_cap = 1; s.n_oct = 1;
    HIPCHECK(dev_alloc(ctx, &s.kps, (size_t)4 * MK));
    HIPCHECK(dev_alloc(ctx, &s.n_kps, (size_t)4));
    HIPCHECK(dev_alloc(ctx, &s.matches, (size_t)2 * MK));
    HIPCHECK(dev_alloc(ctx, &s.n_matches, (size_t)2));
    HIPCHECK(dev_alloc(ctx, &s.tracked, (size_t)MK));
    HIPCHECK(dev_alloc(ctx, &s.n_tracked, (size_t)1));
    HIPCHECK(dev_alloc(ctx, &s.trk_kq, (size_t)MK));
    HIPCHECK(dev_alloc(ctx, &s.gn_lmk, (size_t)MK * 3));
    HIPCHECK(dev_alloc(ctx, &s.gn_obs, (size_t)MK * 8));
    HIPCHECK(dev_alloc(ctx, &s.residual, (size_t)MK));
    HIPCHECK(dev_alloc(ctx, &s.outliers, (size_t)MK));
    HIPCHECK(dev_alloc(ctx, &s.cams, (size_t)1));
    HIPCHECK(dev_alloc(ctx, &s.lane, (size_t)1));
    HIPCHECK(dev_alloc(ctx, &s.results, (size_t)1));
    HIPCHECK(dev_alloc(ctx, &s.status, (size_t)1));
    ctx->cip_ready = true;
    return SVO_OK;
}

extern "C" int svo_change_in_pose(svo_ctx* ctx, const svo_index_pair* tracked, int n_tracked,
                                  const svo_dmatch* pre_matches, int n_pre, const svo_dmatch* cur_matches, int n_cur,
                                  const svo_keypoint* pre_left, int n_pl, const svo_keypoint* pre_right, int n_pr,
                                  const svo_keypoint* cur_left, int n_cl, const svo_keypoint* cur_right, int n_cr,
                                  const svo_stereo_camera* cam, const double* init6,
                                  svo_result* res, double* residual, int32_t* outliers)
{
    if (ctx) use_device(ctx);
    if (!ctx || !cam || !res || n_tracked < 0 || n_pre < 0 || n_cur < 0 || n_pl < 0 || n_pr < 0 || n_cl < 0 || n_cr < 0) return SVO_ERR_ARG;
    if ((n_tracked > 0 && !tracked) || (n_pre > 0 && !pre_matches) || (n_cur > 0 && !cur_matches) || (n_pl > 0 && !pre_left) ||
        (n_pr > 0 && !pre_right) || (n_cl > 0 && !cur_left) || (n_cr > 0 && !cur_right)) return SVO_ERR_ARG;
    if (cam->ncols < 1 || cam->nrows < 1) return SVO_ERR_ARG;
    const svo_params& p = ctx->params;
    const int MK = ctx->dc.max_kps;
    if (n_tracked > MK || n_pre > MK || n_cur > MK || n_pl > MK || n_pr > MK || n_cl > MK || n_cr > MK) return SVO_ERR_CAPACITY;
    int rc = svo_wait(ctx); if (rc) return rc;
    if ((rc = ensure_cip(ctx))) return rc;
    DevCtx& s = ctx->cip;
    s.debug_mode = ctx->dc.debug_mode;
    s.W = cam->ncols; s.H = cam->nrows; s.ow[0] = cam->ncols; s.oh[0] = cam->nrows;      // common.cpp:402-403
    // lane record of the scratch lane: slot 0 = previous, slot 1 = current, both present; warm start and m_error of lane 0
    LaneState live; HIPCHECK(hipMemcpy(&live, ctx->dc.lane, sizeof(live), hipMemcpyDeviceToHost));
    LaneState ls = live; ls.prev_slot = 0; ls.has_prev = 1; ls.has_cur = 1;
    HIPCHECK(hipMemcpy(s.lane, &ls, sizeof(ls), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(s.cams, cam, sizeof(*cam), hipMemcpyHostToDevice));
    const svo_keypoint* kp[4] = { pre_left, pre_right, cur_left, cur_right };        // (slot 0: prev | slot 1: cur) x (left, right)
    const int nk[4] = { n_pl, n_pr, n_cl, n_cr };
    for (int i = 0; i < 4; i++) if (nk[i] > 0) HIPCHECK(hipMemcpy(s.kps + (size_t)i * MK, kp[i], sizeof(svo_keypoint) * nk[i], hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(s.n_kps, nk, sizeof(nk), hipMemcpyHostToDevice));
    if (n_pre > 0) HIPCHECK(hipMemcpy(s.matches, pre_matches, sizeof(svo_dmatch) * n_pre, hipMemcpyHostToDevice));
    if (n_cur > 0) HIPCHECK(hipMemcpy(s.matches + MK, cur_matches, sizeof(svo_dmatch) * n_cur, hipMemcpyHostToDevice));
    const int nm[2] = { n_pre, n_cur };
    HIPCHECK(hipMemcpy(s.n_matches, nm, sizeof(nm), hipMemcpyHostToDevice));
    if (n_tracked > 0) HIPCHECK(hipMemcpy(s.tracked, tracked, sizeof(svo_index_pair) * n_tracked, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(s.n_tracked, &n_tracked, sizeof(int), hipMemcpyHostToDevice));
    svo_result r0; memset(&r0, 0, sizeof(r0)); r0.n_octaves = 1;                         // what k_begin_frame leaves in a result record
    HIPCHECK(hipMemcpy(s.results, &r0, sizeof(r0), hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(s.status, 0, sizeof(uint32_t)));
    GNParams g; memset(&g, 0, sizeof(g));
    g.use_robust_kernel = p.use_robust_kernel; g.max_iters = p.max_iters; g.initial_max_iters = p.initial_max_iters; g.max_incr_cost = p.max_incr_cost;
    g.use_previous_pose_as_initial = p.use_previous_pose_as_initial; g.use_custom_initial_pose = p.use_custom_initial_pose;
    g.min_distance = p.min_distance; g.img_w = cam->ncols; g.img_h = cam->nrows;      // common.cpp:402-403
    g.pmax = MK; g.standalone = 1;
    g.kernel_param = p.kernel_param; g.min_mod_out_vector = p.min_mod_out_vector; g.residual_threshold = p.residual_threshold;
    for (int k = 0; k < 6; k++) g.init[k] = init6 ? init6[k] : 0.0;
    note_stream(ctx);
    { Span sp(ctx, KT_GN); launch_gauss_newton(s, g, ctx->stream); }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    HIPCHECK(hipMemcpy(res, s.results, sizeof(*res), hipMemcpyDeviceToHost));
    if (residual && res->n_residual > 0) HIPCHECK(hipMemcpy(residual, s.residual, sizeof(double) * res->n_residual, hipMemcpyDeviceToHost));
    if (outliers && res->n_outliers > 0) HIPCHECK(hipMemcpy(outliers, s.outliers, sizeof(int32_t) * res->n_outliers, hipMemcpyDeviceToHost));
    // m_last_computed_pose and m_error are the estimator's own members: hand them back to lane 0
    HIPCHECK(hipMemcpy(&ls, s.lane, sizeof(ls), hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; k++) live.last_pose[k] = ls.last_pose[k];
    live.m_error = ls.m_error;
    HIPCHECK(hipMemcpy(ctx->dc.lane, &live, sizeof(live), hipMemcpyHostToDevice));
    return res->valid;
}

// ---- standalone brute-force matcher -------------------------------------------------------------------------
extern "C" int svo_hamming_match(svo_ctx* ctx, const uint8_t* query, int nq, const uint8_t* train, int nt, int32_t* idx, int32_t* dist)
{
    if (ctx) use_device(ctx);
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!query || !idx || !dist)) || (nt > 0 && !train)) return SVO_ERR_ARG;
    if (nt > 65535) return SVO_ERR_UNSUPPORTED;            // train index packs into 16 bits
    if (nq == 0) return SVO_OK;
    if (nt == 0) { for (int i = 0; i < nq; i++) { idx[i] = -1; dist[i] = 0; } return SVO_OK; }
    if (nq > ctx->ham_cap_q) {
        if (ctx->d_ham_q) hipFree(ctx->d_ham_q); if (ctx->d_ham_out) hipFree(ctx->d_ham_out);
        ctx->d_ham_q = nullptr; ctx->d_ham_out = nullptr; ctx->ham_cap_q = 0;
        HIPCHECK(hipMalloc((void**)&ctx->d_ham_q, (size_t)nq * 32)); HIPCHECK(hipMalloc((void**)&ctx->d_ham_out, (size_t)nq * 4)); ctx->ham_cap_q = nq;
    }
    if (nt > ctx->ham_cap_t) {
        if (ctx->d_ham_t) hipFree(ctx->d_ham_t);
        ctx->d_ham_t = nullptr; ctx->ham_cap_t = 0;
        HIPCHECK(hipMalloc((void**)&ctx->d_ham_t, (size_t)nt * 32)); ctx->ham_cap_t = nt;
    }
    const hipStream_t st = ctx->stream;
    HIPCHECK(hipMemcpyAsync(ctx->d_ham_q, query, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(ctx->d_ham_t, train, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(ctx->d_ham_out, 0xFF, (size_t)nq * 4, st));
    const int tiles = (nt + 255) / 256;
    int nsplit = 1; while (nsplit < 16 && nsplit * 2 <= tiles && ((nq + 255) / 256) * nsplit < 1024) nsplit *= 2;
    launch_hamming_plain(ctx->d_ham_q, nq, ctx->d_ham_t, nt, ctx->d_ham_out, nsplit, st);
    std::vector<unsigned> out((size_t)nq);
    HIPCHECK(hipMemcpyAsync(out.data(), ctx->d_ham_out, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    for (int i = 0; i < nq; i++) { idx[i] = (int)(out[i] & 0xFFFFu); dist[i] = (int)(out[i] >> 16); }
    return SVO_OK;
}

// ---- probes ------------------------------------------------------------------------------------------------
extern "C" int svo_debug_get_level(svo_ctx* ctx, int lane, int side, int level, uint8_t* out, int cap, int* w, int* h)
{
    if (ctx) use_device(ctx);
    if (!ctx || !ctx->geom_ready || lane < 0 || lane >= ctx->cfg.n_lanes || level < 0 || level >= ctx->dc.n_levels) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    const LevelGeom& g = ctx->dc.lv[level];
    if (w) *w = g.w; if (h) *h = g.h;
    if (!out) return g.w * g.h;
    if (cap < g.w * g.h) return SVO_ERR_CAPACITY;
    const int img = lane * 2 + side;
    const uint8_t* src; int pitch;
    if (level == 0) { const uint8_t* p0 = nullptr; HIPCHECK(hipMemcpy(&p0, ctx->dc.img0 + img, sizeof(p0), hipMemcpyDeviceToHost)); src = p0; pitch = ctx->dc.img0_pitch; if (!src) return SVO_ERR_STATE; }
    else { src = ctx->dc.pyr + (long long)img * ctx->dc.pyr_bytes + g.offset; pitch = g.pitch; }
    HIPCHECK(hipMemcpy2D(out, (size_t)g.w, src, (size_t)pitch, (size_t)g.w, (size_t)g.h, hipMemcpyDeviceToHost));
    return g.w * g.h;
}

extern "C" int svo_debug_get_raw_keypoints(svo_ctx* ctx, int lane, int side, svo_keypoint* kps, uint8_t* desc, int cap)
{
    if (ctx) use_device(ctx);
    if (!ctx || !ctx->geom_ready || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    const DevCtx& d = ctx->dc;
    const int img = lane * 2 + side;
    int ln[SVO_MAX_LEVELS];
    HIPCHECK(hipMemcpy(ln, d.lvl_n + img * SVO_MAX_LEVELS, sizeof(ln), hipMemcpyDeviceToHost));
    int n = 0;
    for (int l = 0; l < d.n_levels; l++) {
        for (int i = 0; i < ln[l]; i++, n++) {
            if (n >= cap) continue;
            const long long o = (long long)img * d.raw_cap + d.lv[l].slot_off + i;
            if (kps) HIPCHECK(hipMemcpy(kps + n, d.raw_kps + o, sizeof(svo_keypoint), hipMemcpyDeviceToHost));
            if (desc) HIPCHECK(hipMemcpy(desc + (size_t)n * 32, d.raw_desc + o * 32, 32, hipMemcpyDeviceToHost));
        }
    }
    return n;
}

extern "C" int svo_debug_get_redo_count(svo_ctx* ctx, uint32_t* pairs, int reset)
{
    if (ctx) use_device(ctx);
    if (!ctx || !pairs) return SVO_ERR_ARG;
    HIPCHECK(sync_all(ctx));
    HIPCHECK(hipMemcpy(pairs, ctx->dc.redo_n + 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (reset) HIPCHECK(hipMemset(ctx->dc.redo_n + 1, 0, sizeof(uint32_t)));
    return SVO_OK;
}

// SVO_TIMELINE=1: the launch hulls of the last SVO_TL_STEPS frames, 4096 records of (t0, t1) in 10 ns ticks of the device-wide wall clock
// (t0 = ~0: no such launch); record index = (frame % 16) * 256 + kind * 8 + aux (svo_device.h).  Each is the min / max over the launch's
// stamping waves.  Returns the frame counter; reset = 1 clears the table.  0 when the context was created without SVO_TIMELINE=1.
// Waits for the enqueued work.
extern "C" int svo_debug_timeline(svo_ctx* ctx, uint64_t* out, int cap_records, int reset)
{
    if (ctx) use_device(ctx);
    if (!ctx) return SVO_ERR_ARG;
    if (!ctx->dc.tl) return 0;
    const int n = SVO_TL_STEPS * 256;
    int rc = svo_wait(ctx); if (rc) return rc;
    std::vector<TlRec> all((size_t)n * SVO_TL_SUB);
    if (out) {
        if (cap_records < n) return SVO_ERR_ARG;
        HIPCHECK(hipMemcpy(all.data(), ctx->dc.tl, all.size() * sizeof(TlRec), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
            uint64_t t0 = ~0ull, t1 = 0;
            for (int j = 0; j < SVO_TL_SUB; j++) { const TlRec& r = all[(size_t)i * SVO_TL_SUB + j]; if (r.t0 != ~0ull) { if (r.t0 < t0) t0 = r.t0; if (r.t1 > t1) t1 = r.t1; } }
            out[2 * i] = t0; out[2 * i + 1] = t1;
        }
    }
    if (reset) {
        for (auto& r : all) { r.t0 = ~0ull; r.t1 = 0; }
        HIPCHECK(hipMemcpy(ctx->dc.tl, all.data(), all.size() * sizeof(TlRec), hipMemcpyHostToDevice));
    }
    return ctx->dc.tl_step;
}

extern "C" int svo_debug_get_status_word(svo_ctx* ctx, int lane, uint32_t* w)
{
    if (ctx) use_device(ctx);
    if (!ctx || !w || lane < 0 || lane >= ctx->cfg.n_lanes) return SVO_ERR_ARG;
    int rc = svo_wait(ctx); if (rc) return rc;
    HIPCHECK(hipMemcpy(w, ctx->dc.status + lane, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SVO_OK;
}
