// svo_batch.cpp -- include/svo_batch.h: the batched / pipelined and the frame-parallel schedules, on HIP streams and events,
// over the public C-ABI of svo_hip.h only (svo_set_stream, svo_process, svo_copy_results_async, svo_export / import_frame).
//
// No reference counterpart: the reference runs one estimator on one thread (demo-main.cpp:210-220).  What makes the lanes and
// contexts independent is that all estimator state is per instance (libstereo-odometry.h:732-831).
//
// Pipelined schedule, per step and context k (round 4):
//     detect stream:      [wait scratch_free[k] of the previous step]  detector of context k, AHEAD     -> det_done[k]
//     stage 3-5 stream:   [wait det_done[k]]  shift, NMS / row sort + description -> scratch_free[k];
//                         stages 3-5 of context k, result records -> records buffer                      -> rest_done[k]
// The detect call (SVO_FLAG_DETECT_AHEAD) writes the detector's per-image scratch only -- level-0 pointer table, pyramid,
// candidates, per-level winners -- and the last reader of that scratch is the description of the frame before, so the detector of
// frame t + 1 starts while the matchers, both RANSACs and the Gauss-Newton solve of frame t are still running on the context's own
// stream: a context's cycle is max(detector + NMS + description, its stage 3-5 chain) instead of their sum (rounds 1-3 waited for
// rest_done[k]: 2.915 ms per step = 0.904 ms of detector + 2.002 ms of the rest).  The prev/cur shift, the record
// clear and the status word move to the stage 3-5 call; the two feature slots are touched by that stream alone.
// post_mode 0 / 2 (the detect call describes too, or a third stream does) keep the old wait on rest_done[k].
// Default split (post_mode 1, one stage 3-5 stream per context): the detect stream carries the pyramid, FAST and the per-level
// selection; the reference's own NMS / row sort (one latency-bound block per image), the description of its survivors and stages
// 3-5 run on the context's own stream, so the latency-bound kernels of the contexts overlap each other as well as the next detect.
#define SVO_BATCH_NO_SIZED_CREATE      // this file defines the plain symbol as well
#include "../../include/svo_batch.h"
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string>
#include <vector>

namespace {
struct Err { std::string text; };
int hip_fail(std::string& dst, const char* what, hipError_t e) { dst = std::string(what) + ": " + hipGetErrorString(e); return SVO_ERR_HIP; }
}
#define BHIP(obj, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_fail((obj)->last_error, #expr, _e); } while (0)
#define BSVO(obj, c, expr) do { int _rc = (expr); if (_rc < 0) { (obj)->last_error = std::string(#expr) + ": " + svo_strerror(_rc) + " [" + svo_last_error(c) + "]"; return _rc; } } while (0)

static const uint32_t IMG_FLAGS = SVO_FLAG_DEVICE_IMAGES | SVO_FLAG_PINNED_IMAGES | SVO_FLAG_BGR_IMAGES;

struct svo_batch {
    svo_batch_config cfg;
    int NC = 0, Bc = 0, B = 0;
    bool pipelined = false, first = true, step_started = false;
    std::vector<svo_ctx*> ctx;
    std::vector<hipStream_t> own;                    // one per context: the stream it was created with (free schedule)
    std::vector<hipStream_t> s_dets, s_rests; hipStream_t s_post = nullptr;
    std::vector<hipEvent_t> det_done, rest_done, done, pre_done, scratch_free;
    bool ahead = false;                              // the detect calls run ahead of the previous frame's stages 3-5 (post_mode 1 / 3)
    std::vector<hipEvent_t> held;                    // svo_batch_hold_for_event: what the next step's record copies wait for (every one of them)
    uint8_t* rec = nullptr; uint8_t* own_rec = nullptr;
    std::string last_error;
};

extern "C" void svo_batch_config_defaults(svo_batch_config* c)
{
    if (!c) return;
    svo_config_defaults(&c->ctx);
    c->ctx.n_lanes = 96;                       // the measured default (2 x 96 since round 6, 3 x 64 before); SVO_MAX_LANES is the limit of the pointer tables in the kernel arguments
    c->n_contexts = 2; c->schedule = SVO_SCHED_PIPELINED; c->det_priority_high = 1; c->post_mode = 1; c->det_streams = 1; c->rest_streams = 0; c->no_detect_ahead = 0;
}

extern "C" const char* svo_batch_last_error(const svo_batch* b) { return b ? b->last_error.c_str() : ""; }
extern "C" int svo_batch_lanes(const svo_batch* b) { return b ? b->B : SVO_ERR_ARG; }
extern "C" int svo_batch_contexts(const svo_batch* b) { return b ? b->NC : SVO_ERR_ARG; }
extern "C" svo_ctx* svo_batch_context(svo_batch* b, int k) { return (b && k >= 0 && k < b->NC) ? b->ctx[(size_t)k] : nullptr; }

static int make_stream(std::string& err, hipStream_t* s, bool high)
{
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);                    // numerically lower = higher priority
    const hipError_t e = hipStreamCreateWithPriority(s, hipStreamNonBlocking, high ? greatest : 0);
    return e == hipSuccess ? SVO_OK : hip_fail(err, "hipStreamCreateWithPriority", e);
}

extern "C" void svo_batch_abi_sizes(int32_t* out3)
{
    if (!out3) return;
    out3[0] = (int32_t)sizeof(svo_batch_config); out3[1] = SVO_MAX_LANES; out3[2] = SVO_BATCH_ABI_VERSION;
}

// what a C / C++ host reaches through the header's svo_batch_create(): its OWN sizeof(svo_batch_config) travels with the call, so a
// host compiled against another version of the header is refused instead of having trailing fields read from whatever follows
extern "C" int svo_batch_create_sized(const svo_batch_config* cfg, size_t cfg_bytes, svo_batch** out)
{
    if (!cfg || !out) return SVO_ERR_ARG;
    if (cfg_bytes != sizeof(svo_batch_config)) {
        svo_batch* b = new svo_batch();
        *out = b;
        b->last_error = "svo_batch_config is " + std::to_string(cfg_bytes) + " bytes in the caller's header, " + std::to_string(sizeof(svo_batch_config)) + " in this library (ABI version " + std::to_string(SVO_BATCH_ABI_VERSION) + "): rebuild the host against include/svo_batch.h of this library";
        return SVO_ERR_ARG;
    }
    return svo_batch_create(cfg, out);
}

extern "C" int svo_batch_create(const svo_batch_config* cfg, svo_batch** out)
{
    if (!cfg || !out) return SVO_ERR_ARG;
    *out = nullptr;
    if (cfg->n_contexts < 1 || cfg->ctx.n_lanes < 1 || cfg->ctx.n_lanes > SVO_MAX_LANES || cfg->post_mode < 0 || cfg->post_mode > 3) return SVO_ERR_ARG;
    int ndev = 0;
    if (cfg->ctx.device < 0) return SVO_ERR_ARG;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->ctx.device >= ndev) return SVO_ERR_NO_DEVICE;
    svo_batch* b = new svo_batch();
    *out = b;                                                                    // so that the caller can read last_error and destroy
    b->cfg = *cfg; b->NC = cfg->n_contexts; b->Bc = cfg->ctx.n_lanes; b->B = b->NC * b->Bc;
    b->pipelined = b->NC > 1 && cfg->schedule == SVO_SCHED_PIPELINED;
    BHIP(b, hipSetDevice(cfg->ctx.device));
    b->ahead = b->pipelined && (cfg->post_mode == 1 || cfg->post_mode == 3) && !cfg->no_detect_ahead;
    // The pipelined schedule's streams first, and no stream that is never launched on: HIP multiplexes its streams onto a few hardware
    // queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share one never overlap (tools/ubench/queue_overlap.hip: with three
    // idle streams created first, the first two priority-0 streams made afterwards executed one after the other, always).  A context of the
    // pipelined schedule is therefore created ON its stage 3-5 stream; only the free schedule makes a stream per context.
    if (b->pipelined) {
        const int nd = cfg->det_streams > 1 ? cfg->det_streams : 1;
        for (int i = 0; i < nd; i++) { hipStream_t s = nullptr; int rc = make_stream(b->last_error, &s, cfg->det_priority_high != 0); if (rc) return rc; b->s_dets.push_back(s); }
        const int nr = cfg->rest_streams > 0 ? cfg->rest_streams : b->NC;
        for (int i = 0; i < nr; i++) { hipStream_t s = nullptr; int rc = make_stream(b->last_error, &s, cfg->det_priority_high == 0); if (rc) return rc; b->s_rests.push_back(s); }
        if (cfg->post_mode == 2) { int rc = make_stream(b->last_error, &b->s_post, cfg->det_priority_high == 0); if (rc) return rc; }
    }
    for (int k = 0; k < b->NC; k++) {
        hipStream_t s = nullptr;
        if (b->pipelined) s = b->s_rests[(size_t)k % b->s_rests.size()];
        else { BHIP(b, hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); b->own.push_back(s); }
        svo_config c = cfg->ctx; c.stream = s;
        svo_ctx* x = nullptr;
        const int rc = svo_create(&c, &x);
        if (rc != SVO_OK) { b->last_error = std::string("svo_create: ") + svo_strerror(rc) + (x ? std::string(" [") + svo_last_error(x) + "]" : std::string()); if (x) svo_destroy(x); return rc; }
        b->ctx.push_back(x);
        for (auto* v : { &b->det_done, &b->rest_done, &b->done, &b->pre_done, &b->scratch_free }) {   // pushed as created: a failure half way leaks nothing
            hipEvent_t e = nullptr;
            BHIP(b, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            v->push_back(e);
        }
    }
    BHIP(b, hipMalloc((void**)&b->own_rec, (size_t)b->B * sizeof(svo_result)));
    BHIP(b, hipMemset(b->own_rec, 0, (size_t)b->B * sizeof(svo_result)));
    b->rec = b->own_rec;
    return SVO_OK;
}

extern "C" void svo_batch_destroy(svo_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->cfg.ctx.device);
    (void)hipDeviceSynchronize();
    for (svo_ctx* c : b->ctx) { (void)svo_set_stream(c, nullptr); svo_destroy(c); }
    for (hipStream_t s : b->s_dets) (void)hipStreamDestroy(s);
    for (hipStream_t s : b->s_rests) (void)hipStreamDestroy(s);
    if (b->s_post) (void)hipStreamDestroy(b->s_post);
    for (hipStream_t s : b->own) (void)hipStreamDestroy(s);
    for (auto* v : { &b->det_done, &b->rest_done, &b->done, &b->pre_done, &b->scratch_free }) for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    if (b->own_rec) (void)hipFree(b->own_rec);
    delete b;
}

extern "C" int svo_batch_set_params(svo_batch* b, const svo_params* p)
{
    if (!b || !p) return SVO_ERR_ARG;
    for (svo_ctx* c : b->ctx) BSVO(b, c, svo_set_params(c, p));
    return SVO_OK;
}

extern "C" int svo_batch_set_camera(svo_batch* b, int lane, const svo_stereo_camera* cam)
{
    if (!b || !cam || lane < -1 || lane >= b->B) return SVO_ERR_ARG;
    if (lane < 0) { for (svo_ctx* c : b->ctx) BSVO(b, c, svo_set_camera(c, -1, cam)); return SVO_OK; }
    svo_ctx* c = b->ctx[(size_t)(lane / b->Bc)];
    BSVO(b, c, svo_set_camera(c, lane % b->Bc, cam));
    return SVO_OK;
}

extern "C" int svo_batch_set_results_buffer(svo_batch* b, void* dev_records, size_t bytes)
{
    if (!b || (dev_records && bytes < (size_t)b->B * sizeof(svo_result))) return SVO_ERR_ARG;
    int rc = svo_batch_synchronize(b); if (rc) return rc;
    b->rec = dev_records ? (uint8_t*)dev_records : b->own_rec;
    return SVO_OK;
}

extern "C" int svo_batch_switch_results_buffer(svo_batch* b, void* dev_records, size_t bytes)
{
    if (!b || (dev_records && bytes < (size_t)b->B * sizeof(svo_result))) return SVO_ERR_ARG;
    b->rec = dev_records ? (uint8_t*)dev_records : b->own_rec;
    return SVO_OK;
}

static int batch_step_impl(svo_batch* b, const svo_frame* frames, uint32_t flags);

extern "C" int svo_batch_step(svo_batch* b, const svo_frame* frames, uint32_t flags)
{
    if (!b) return SVO_ERR_ARG;
    b->step_started = false;
    const int rc = batch_step_impl(b, frames, flags);
    if (rc == SVO_OK || b->step_started) b->held.clear();      // the held events belong to the step that was ENQUEUED (wholly or in part): a call refused
                                                               // before anything was enqueued (NULL frames, unknown flags) keeps them for the caller's retry
    if (rc != SVO_OK && b->step_started) {    // (an argument refused before anything was enqueued leaves the event chain intact)
        // a step that failed half way has recorded some of its events and not others: let everything enqueued so far drain and start
        // the event chain over, so that the next step neither waits on a stale record (a no-op) nor overwrites scratch still being read
        (void)hipSetDevice(b->cfg.ctx.device);
        (void)hipDeviceSynchronize();
        b->first = true;
    }
    return rc;
}

static int batch_step_impl(svo_batch* b, const svo_frame* frames, uint32_t flags)
{
    if (!frames || (flags & ~IMG_FLAGS)) return SVO_ERR_ARG;
    BHIP(b, hipSetDevice(b->cfg.ctx.device));
    const size_t rsz = sizeof(svo_result);
    const uint32_t AH = b->ahead ? (uint32_t)SVO_FLAG_DETECT_AHEAD : 0u;
    const uint32_t REST = SVO_RUN_MATCH | SVO_RUN_TRACK | SVO_RUN_OPTIMIZE | (b->ahead ? AH : (uint32_t)SVO_FLAG_NO_SHIFT) | (b->cfg.post_mode == 1 ? (uint32_t)SVO_RUN_DETECT_POST : 0u)
                        | (b->cfg.post_mode == 3 ? (uint32_t)(SVO_RUN_DETECT_POST | SVO_FLAG_DETECT_SPLIT_AT_SELECT) : 0u);
    b->step_started = true;
    for (int k = 0; k < b->NC; k++) {
        svo_ctx* c = b->ctx[(size_t)k];
        const svo_frame* pk = frames + (size_t)k * b->Bc;
        uint8_t* dst = b->rec + (size_t)k * b->Bc * rsz;
        if (b->pipelined) {
            hipStream_t s_det = b->s_dets[(size_t)k % b->s_dets.size()], s_rest = b->s_rests[(size_t)k % b->s_rests.size()];
            if (!b->first) BHIP(b, hipStreamWaitEvent(s_det, b->ahead ? b->scratch_free[(size_t)k] : b->rest_done[(size_t)k], 0));
            BSVO(b, c, svo_set_stream(c, s_det));
            BSVO(b, c, svo_process(c, pk, SVO_RUN_DETECT | AH | (b->ahead ? (uint32_t)SVO_FLAG_NO_SHIFT : 0u) | (b->cfg.post_mode ? (uint32_t)SVO_FLAG_DETECT_NO_POST : 0u) | (b->cfg.post_mode == 3 ? (uint32_t)SVO_FLAG_DETECT_SPLIT_AT_SELECT : 0u) | flags));
            if (b->cfg.post_mode == 2) {
                BHIP(b, hipEventRecord(b->pre_done[(size_t)k], s_det));
                BHIP(b, hipStreamWaitEvent(b->s_post, b->pre_done[(size_t)k], 0));
                BSVO(b, c, svo_set_stream(c, b->s_post));
                BSVO(b, c, svo_process(c, nullptr, SVO_RUN_DETECT_POST | SVO_FLAG_NO_SHIFT));
                BHIP(b, hipEventRecord(b->det_done[(size_t)k], b->s_post));
            } else BHIP(b, hipEventRecord(b->det_done[(size_t)k], s_det));
            BHIP(b, hipStreamWaitEvent(s_rest, b->det_done[(size_t)k], 0));
            BSVO(b, c, svo_set_stream(c, s_rest));
            if (b->ahead) BSVO(b, c, svo_record_after_post(c, b->scratch_free[(size_t)k]));
            BSVO(b, c, svo_process(c, nullptr, REST));
            for (hipEvent_t h : b->held) BHIP(b, hipStreamWaitEvent(s_rest, h, 0));       // only the record copy waits for a reader of the records buffer
            BSVO(b, c, svo_copy_results_async(c, dst, (size_t)b->Bc * rsz));
            BHIP(b, hipEventRecord(b->rest_done[(size_t)k], s_rest));
        } else {
            BSVO(b, c, svo_set_stream(c, nullptr));
            BSVO(b, c, svo_process(c, pk, SVO_RUN_ALL | flags));
            for (hipEvent_t h : b->held) BHIP(b, hipStreamWaitEvent(b->own[(size_t)k], h, 0));
            BSVO(b, c, svo_copy_results_async(c, dst, (size_t)b->Bc * rsz));
            BHIP(b, hipEventRecord(b->done[(size_t)k], b->own[(size_t)k]));
        }
    }
    b->first = false;
    return SVO_OK;
}

extern "C" int svo_batch_wait_on_stream(svo_batch* b, void* stream)
{
    if (!b) return SVO_ERR_ARG;
    if (b->first) return SVO_OK;                                                 // nothing enqueued yet
    for (int k = 0; k < b->NC; k++) BHIP(b, hipStreamWaitEvent((hipStream_t)stream, b->pipelined ? b->rest_done[(size_t)k] : b->done[(size_t)k], 0));
    return SVO_OK;
}

extern "C" int svo_batch_hold_for_event(svo_batch* b, void* event)
{
    if (!b || !event) return SVO_ERR_ARG;
    b->held.push_back((hipEvent_t)event);   // the next svo_batch_step makes each context's record copy -- and nothing before it -- wait for all of them
    return SVO_OK;
}

extern "C" int svo_batch_synchronize(svo_batch* b)
{
    if (!b) return SVO_ERR_ARG;
    BHIP(b, hipSetDevice(b->cfg.ctx.device));
    for (hipStream_t s : b->s_dets) BHIP(b, hipStreamSynchronize(s));
    if (b->s_post) BHIP(b, hipStreamSynchronize(b->s_post));
    for (hipStream_t s : b->s_rests) BHIP(b, hipStreamSynchronize(s));
    for (svo_ctx* c : b->ctx) BSVO(b, c, svo_wait(c));
    return SVO_OK;
}

extern "C" int svo_batch_results(svo_batch* b, svo_result* res)
{
    if (!b || !res) return SVO_ERR_ARG;
    int rc = svo_batch_synchronize(b); if (rc) return rc;
    for (int k = 0; k < b->NC; k++) BSVO(b, b->ctx[(size_t)k], svo_get_results(b->ctx[(size_t)k], res + (size_t)k * b->Bc));
    return SVO_OK;
}

extern "C" int svo_batch_reset(svo_batch* b)
{
    if (!b) return SVO_ERR_ARG;
    int rc = svo_batch_synchronize(b); if (rc) return rc;
    for (svo_ctx* c : b->ctx) { BSVO(b, c, svo_set_stream(c, nullptr)); BSVO(b, c, svo_reset(c, -1)); }
    b->first = true;
    return SVO_OK;
}

// ---- frame-parallelism within one stream ---------------------------------------------------------------------------------
// Context g = t % G runs stages 2-3 of frame t on its own stream as soon as it is free -- overlapping stages 4-5 of frame t - 1
// on the context before it -- then imports the previous owner's hand-over record (svo_export_frame / svo_import_frame), runs
// stages 4-5 and exports its own.  Same results as one context fed sequentially; frames per second are bounded by
// max(stages 4-5 of one frame, a whole frame / G) instead of a whole frame.
struct svo_fpstream {
    svo_config cfg; int G = 0; long long t = 0;
    std::vector<svo_ctx*> ctx; std::vector<hipStream_t> st; std::vector<hipEvent_t> exported; std::vector<uint8_t*> blob;
    size_t nbytes = 0;
    std::string last_error;
};

extern "C" const char* svo_fpstream_last_error(const svo_fpstream* f) { return f ? f->last_error.c_str() : ""; }
extern "C" int svo_fpstream_contexts(const svo_fpstream* f) { return f ? f->G : SVO_ERR_ARG; }
extern "C" svo_ctx* svo_fpstream_context(svo_fpstream* f, int k) { return (f && k >= 0 && k < f->G) ? f->ctx[(size_t)k] : nullptr; }
extern "C" svo_ctx* svo_fpstream_last_owner(svo_fpstream* f) { return (f && f->t > 0) ? f->ctx[(size_t)((f->t - 1) % f->G)] : nullptr; }

extern "C" int svo_fpstream_create(const svo_config* cfg, int n_contexts, svo_fpstream** out)
{
    if (!cfg || !out || n_contexts < 1) return SVO_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) return SVO_ERR_NO_DEVICE;
    svo_fpstream* f = new svo_fpstream();
    *out = f;
    f->cfg = *cfg; f->G = n_contexts;
    BHIP(f, hipSetDevice(cfg->device));
    for (int g = 0; g < f->G; g++) {
        hipStream_t s = nullptr;
        BHIP(f, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        f->st.push_back(s);
        svo_config c = *cfg; c.stream = s;
        svo_ctx* x = nullptr;
        const int rc = svo_create(&c, &x);
        if (rc != SVO_OK) { f->last_error = std::string("svo_create: ") + svo_strerror(rc) + (x ? std::string(" [") + svo_last_error(x) + "]" : std::string()); if (x) svo_destroy(x); return rc; }
        f->ctx.push_back(x);
        hipEvent_t e; BHIP(f, hipEventCreateWithFlags(&e, hipEventDisableTiming)); f->exported.push_back(e);
    }
    f->nbytes = svo_handover_bytes(f->ctx[0]);
    for (int g = 0; g < f->G; g++) { uint8_t* p = nullptr; BHIP(f, hipMalloc((void**)&p, f->nbytes)); BHIP(f, hipMemset(p, 0, f->nbytes)); f->blob.push_back(p); }
    return SVO_OK;
}

extern "C" void svo_fpstream_destroy(svo_fpstream* f)
{
    if (!f) return;
    (void)hipSetDevice(f->cfg.device);
    (void)hipDeviceSynchronize();
    for (svo_ctx* c : f->ctx) svo_destroy(c);
    for (hipStream_t s : f->st) (void)hipStreamDestroy(s);
    for (hipEvent_t e : f->exported) (void)hipEventDestroy(e);
    for (uint8_t* p : f->blob) (void)hipFree(p);
    delete f;
}

extern "C" int svo_fpstream_set_params(svo_fpstream* f, const svo_params* p)
{
    if (!f || !p) return SVO_ERR_ARG;
    for (svo_ctx* c : f->ctx) BSVO(f, c, svo_set_params(c, p));
    return SVO_OK;
}

extern "C" int svo_fpstream_set_camera(svo_fpstream* f, int lane, const svo_stereo_camera* cam)
{
    if (!f || !cam) return SVO_ERR_ARG;
    for (svo_ctx* c : f->ctx) BSVO(f, c, svo_set_camera(c, lane, cam));
    return SVO_OK;
}

extern "C" int svo_fpstream_push(svo_fpstream* f, const svo_frame* frames, uint32_t flags)
{
    if (!f || !frames || (flags & ~IMG_FLAGS)) return SVO_ERR_ARG;
    BHIP(f, hipSetDevice(f->cfg.device));
    const int g = (int)(f->t % f->G);
    svo_ctx* c = f->ctx[(size_t)g]; hipStream_t s = f->st[(size_t)g];
    BSVO(f, c, svo_process(c, frames, SVO_RUN_DETECT | SVO_RUN_MATCH | flags));             // stages 2-3: independent of every other frame
    if (f->t > 0) {
        const int gp = (int)((f->t - 1) % f->G);
        BHIP(f, hipStreamWaitEvent(s, f->exported[(size_t)gp], 0));
        BSVO(f, c, svo_import_frame(c, f->blob[(size_t)gp], f->nbytes));
    }
    BSVO(f, c, svo_process(c, nullptr, SVO_RUN_TRACK | SVO_RUN_OPTIMIZE | SVO_FLAG_NO_SHIFT));
    BSVO(f, c, svo_export_frame(c, f->blob[(size_t)g], f->nbytes));
    BHIP(f, hipEventRecord(f->exported[(size_t)g], s));
    f->t++;
    return SVO_OK;
}

extern "C" int svo_fpstream_synchronize(svo_fpstream* f)
{
    if (!f) return SVO_ERR_ARG;
    BHIP(f, hipSetDevice(f->cfg.device));
    for (svo_ctx* c : f->ctx) BSVO(f, c, svo_wait(c));
    return SVO_OK;
}
