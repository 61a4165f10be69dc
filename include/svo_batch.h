/* svo_batch.h -- the batched / pipelined schedules of the MI355X stereo-VO hot path behind the C-ABI (libsvo_hip.so).
 *
 * The reference's caller is a C++ loop around one estimator (demo-stereo-odometry/demo-main.cpp:210-220); all of an
 * estimator's state is per instance (libstereo-odometry/include/libstereo-odometry.h:732-831), so any number of them run
 * side by side.  A context (svo_hip.h) drives up to SVO_MAX_LANES of them through every kernel launch; what fills an
 * MI355X is SEVERAL such contexts whose latency-bound stages 3-5 overlap the throughput kernels of stage 2 of the next
 * one.  That choreography -- two HIP streams with priorities, the events between them, the reuse hold-off of the result
 * buffer -- lives here, in the library, so that a C / C++ host gets the benchmarked schedule with one call per step:
 *
 *   svo_batch       n_contexts x lanes independent streams, one frame of every stream per svo_batch_step
 *   svo_fpstream    ONE stream (or lanes streams advancing together) whose consecutive frames are dealt round-robin to
 *                   several contexts: stages 2-3 of frame t overlap stages 4-5 of frame t-1 (SURVEY.md 8e "within one
 *                   stream"); results equal the sequential run list for list
 *
 * Conventions as in svo_hip.h: plain pointers and sizes, int status (0 ok, < 0 error), nothing throws, no CPU fallback.
 * Neither object is thread-safe; one host thread drives one batch (one batch per GPU and thread on a multi-GPU node).
 *
 * The default shape is two contexts of 96 streams: one detect stream + two stage 3-5 streams (round 6: 70.8 k pairs/s against 69.9 k for
 * 3 x 64 and 49.7 k for 4 x 64, profiles/r06_contexts_sweep.txt).  HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by
 * default) and two streams that share one never overlap: a host should export GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises
 * (the two-context 2048 x 1536 shape ran at 23.7 k or 36.7 k pairs/s on that alone).
 */
#ifndef SVO_BATCH_H
#define SVO_BATCH_H
#include "svo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { SVO_SCHED_PIPELINED = 0,   /* ONE stream carries stage 2 of the contexts back to back, ANOTHER stages 3-5 of each frame */
       SVO_SCHED_FREE = 1 };      /* every context runs its whole frame on its own stream, unsynchronised with the others */

typedef struct svo_batch svo_batch;

typedef struct svo_batch_config {
    svo_config ctx;           /* per CONTEXT: n_lanes = streams per context (<= SVO_MAX_LANES); .stream is ignored */
    int32_t n_contexts;       /* contexts side by side on ctx.device (total streams = n_contexts * ctx.n_lanes) */
    int32_t schedule;         /* SVO_SCHED_*; a single context always runs SVO_SCHED_FREE */
    int32_t det_priority_high;/* which side gets the high HIP stream priority.  1 (default): the detect stream -- with the NMS, the description
                                 and stages 3-5 spread over one stream per context (post_mode 1) the detect stream is the serial chain of a
                                 step, and its kernels should not queue behind three streams of everything else (66.4 k against 64.9 k
                                 pairs/s, r03); 0: the stage 3-5 streams (the better choice when ONE stream carries all of them) */
    int32_t post_mode;        /* where the reference's own post-processing of the detector output (NMS + row sort + describe) runs:
                                 0 on the detect stream, 1 (default, measured best) on the stage 3-5 stream, 2 on a third stream of its own, 3 on the stage 3-5 stream TOGETHER
                                 WITH the per-level selection (top-K, Harris, sort): the detect stream keeps the pyramid and the FAST kernel only */
    int32_t det_streams;      /* >= 1: detect phases of consecutive contexts alternate over this many streams (default 1) */
    int32_t rest_streams;     /* >= 1: stages 3-5 of consecutive contexts alternate over this many streams; 0 (default): one per context */
    int32_t no_detect_ahead;  /* 0 (default): with post_mode 1 / 3 the detector of a context's next frame starts as soon as the description of
                                 its current frame has read the detector's scratch (SVO_FLAG_DETECT_AHEAD, svo_record_after_post), i.e. it
                                 overlaps that context's own stages 3-5; 1: it waits for the whole frame, as rounds 1-3 did */
} svo_batch_config;

/* Bumped whenever svo_batch_config or SVO_MAX_LANES changes (2 = round 4: `no_detect_ahead` appended, SVO_MAX_LANES 64 -> 128). */
#define SVO_BATCH_ABI_VERSION 2
/* out3[0] = sizeof(svo_batch_config), out3[1] = SVO_MAX_LANES, out3[2] = SVO_BATCH_ABI_VERSION as this LIBRARY was compiled: a
 * binding checks its mirror against them (tests/test_abi.py does for the ctypes one). */
void svo_batch_abi_sizes(int32_t* out3);
void svo_batch_config_defaults(svo_batch_config* c);
/* C / C++ hosts call svo_batch_create(): the macro below passes the caller's own sizeof(svo_batch_config) along, and a host built
 * against another version of this header gets SVO_ERR_ARG (+ the two sizes in svo_batch_last_error) instead of a config whose
 * trailing fields are read from whatever follows the shorter struct.  Bindings that cannot use the macro (ctypes) call the plain
 * symbol after checking svo_batch_abi_sizes.
 * ON FAILURE *out MAY BE A HANDLE (rc != SVO_OK and *out != NULL): it exists so that svo_batch_last_error(*out) can say what went
 * wrong -- the caller owns it and must svo_batch_destroy() it.  *out == NULL on failure means nothing was allocated. */
int  svo_batch_create_sized(const svo_batch_config* cfg, size_t cfg_bytes, svo_batch** out);
int  svo_batch_create(const svo_batch_config* cfg, svo_batch** out);
#ifndef SVO_BATCH_NO_SIZED_CREATE
#define svo_batch_create(cfg, out) svo_batch_create_sized((cfg), sizeof(svo_batch_config), (out))
#endif
void svo_batch_destroy(svo_batch* b);
const char* svo_batch_last_error(const svo_batch* b);
int  svo_batch_lanes(const svo_batch* b);                       /* n_contexts * lanes per context */
int  svo_batch_contexts(const svo_batch* b);
/* context k (borrowed: per-context calls of svo_hip.h -- getters, thresholds, kernel times -- go through it; do not destroy it,
 * and do not call svo_set_stream on it between steps) */
svo_ctx* svo_batch_context(svo_batch* b, int k);
int  svo_batch_set_params(svo_batch* b, const svo_params* p);   /* every context (loadParamsFromConfigFile, H:554-663) */
int  svo_batch_set_camera(svo_batch* b, int lane, const svo_stereo_camera* cam);   /* global lane index, -1 = every stream */
/* Where every step leaves the result records: caller-owned DEVICE memory of svo_batch_lanes() * sizeof(svo_result) bytes, in
 * lane order (lane = context * lanes_per_context + lane_in_context) -- e.g. this rank's slot of an all-gather buffer.
 * NULL: a buffer of the batch's own (svo_batch_results reads it). */
int  svo_batch_set_results_buffer(svo_batch* b, void* dev_records, size_t bytes);
/* The same without waiting for the work in flight: takes effect for the steps enqueued AFTER the call (a step's copy is bound to the
 * buffer when it is enqueued).  For alternating between two buffers -- step t + 1 writes one while an all-gather still reads step t's
 * from the other; only step t + 2 has to wait for that all-gather (svo_batch_hold_for_event), which is long over by then. */
int  svo_batch_switch_results_buffer(svo_batch* b, void* dev_records, size_t bytes);
/* processNewImagePair for every stream: frames[lane], lane in [0, svo_batch_lanes).  `flags`: SVO_FLAG_DEVICE_IMAGES,
 * SVO_FLAG_PINNED_IMAGES or neither (pageable host images), | SVO_FLAG_BGR_IMAGES.  ENQUEUES and returns. */
int  svo_batch_step(svo_batch* b, const svo_frame* frames, uint32_t flags);
/* make the caller's hipStream_t wait for the last step's work of every context (e.g. before an all-gather of the records) */
int  svo_batch_wait_on_stream(svo_batch* b, void* stream);
/* the NEXT step's result copies -- and nothing ahead of them in that step -- wait for this hipEvent_t (e.g. the all-gather that
 * still reads the records buffer).  Several calls before one step add up (the step waits for every one of them); the list is
 * emptied by that step once it has enqueued anything, whether it then succeeds or not (a call refused for its ARGUMENTS, before
 * anything was enqueued, keeps the list: the caller may correct the argument and retry).  The events are only referenced: each must stay alive, and must not be
 * re-recorded, until that svo_batch_step has returned. */
int  svo_batch_hold_for_event(svo_batch* b, void* event);
int  svo_batch_synchronize(svo_batch* b);
int  svo_batch_results(svo_batch* b, svo_result* res /* svo_batch_lanes() entries */);    /* implies svo_batch_synchronize */
int  svo_batch_reset(svo_batch* b);                             /* every stream a freshly constructed estimator again (C:28-50) */

typedef struct svo_fpstream svo_fpstream;
/* cfg as for one context (n_lanes streams advancing together; .stream ignored); n_contexts >= 1 contexts take the frames in turn */
int  svo_fpstream_create(const svo_config* cfg, int n_contexts, svo_fpstream** out);
void svo_fpstream_destroy(svo_fpstream* f);
const char* svo_fpstream_last_error(const svo_fpstream* f);
int  svo_fpstream_contexts(const svo_fpstream* f);
svo_ctx* svo_fpstream_context(svo_fpstream* f, int k);          /* borrowed */
svo_ctx* svo_fpstream_last_owner(svo_fpstream* f);              /* the context that ran the frame pushed last (NULL before the first) */
int  svo_fpstream_set_params(svo_fpstream* f, const svo_params* p);
int  svo_fpstream_set_camera(svo_fpstream* f, int lane, const svo_stereo_camera* cam);
/* the next frame of the stream(s): frames[lane]; flags as for svo_batch_step.  Enqueues and returns; the frame's result records
 * (n_lanes of them) land in the owner context -- svo_get_result(s) on svo_fpstream_last_owner waits for them. */
int  svo_fpstream_push(svo_fpstream* f, const svo_frame* frames, uint32_t flags);
int  svo_fpstream_synchronize(svo_fpstream* f);

#ifdef __cplusplus
}
#endif
#endif
