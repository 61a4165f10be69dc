// k_handover.hip -- frame-parallelism WITHIN one stereo stream (SURVEY.md 8e "Within ONE stream", 8f-3).
//
// Stages 2-3 of a frame depend on nothing but its two images, so consecutive frames of one stream can be dealt
// round-robin to several contexts (on one GPU, or one per GPU).  Stages 4-5 of frame t need the lists of frame t-1 and
// the estimator members a call inherits from the call before it: m_error (the recovery rule, P:86-95),
// m_last_computed_pose (the warm start, S5:506-507, 720-721), the match-ID counters (H:735-742).  The owner of frame
// t-1 therefore EXPORTS, once its stages 4-5 are through, one contiguous record per lane -- "what the next call would
// find as its previous frame" -- and the owner of frame t IMPORTS it into its previous-frame slot before it runs
// stages 4-5.  The record is a flat device buffer: copy it device-to-device on one GPU, or ncclSend / ncclRecv it
// between GPUs.  The result is the sequential run's, list for list and pose for pose (nothing is dropped, not even the
// warm start); what overlaps is stage 2-3 of frame t with stages 2-5 of frame t-1.
//
// Which frame is "previous" for the next call (P:86-89): the frame just processed, unless that call ended in
// voecBadTracking / voecBadCondNumber, in which case the OLDER frame stays.  The export resolves that on the device.
#include "svo_device.h"
#include "svo_kernels.h"

// byte layout of one lane-octave's record (all offsets multiples of 16):
//   header (256 B): magic, version and the geometry the layout depends on (max_kps, max_h, oct_cap, n_lanes) -- a record from a
//       differently configured context (another rank's, say) has other offsets: the importer checks them and refuses;
//       int32 n_kps[2], n_matches, n_ids, present; LaneState (lane-level, octave 0 only carries it)
//   kps[2][max_kps] | desc[2][max_kps][32] | matches[max_kps] | ids[max_kps] | row_index[2][max_h] | mrow_index[max_h + 1]
#define SVO_HANDOVER_MAGIC 0x53564F48      // "SVOH"
#define SVO_HANDOVER_VERSION 2
struct HandoverHeader { int32_t magic, version, max_kps, max_h, oct_cap, n_lanes; int32_t n_kps[2], n_matches, n_ids, present, pad; LaneState ls; };
static_assert(sizeof(HandoverHeader) <= 256, "header slot");

static inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
size_t handover_record_bytes(const DevCtx& c)
{
    const size_t MK = (size_t)c.max_kps, H = (size_t)c.max_h;
    return 256 + align16(2 * MK * sizeof(svo_keypoint)) + 2 * MK * 32 + MK * sizeof(svo_dmatch) + align16(MK * 4) + align16(2 * H * 4) + align16((H + 1) * 4);
}

struct HandoverOffsets { size_t kps, desc, matches, ids, row, mrow, total; };
__host__ __device__ static inline HandoverOffsets handover_offsets(int max_kps, int max_h)
{
    const size_t MK = (size_t)max_kps, H = (size_t)max_h;
    HandoverOffsets o;
    o.kps = 256;
    o.desc = o.kps + ((2 * MK * sizeof(svo_keypoint) + 15) & ~(size_t)15);
    o.matches = o.desc + 2 * MK * 32;
    o.ids = o.matches + MK * sizeof(svo_dmatch);
    o.row = o.ids + ((MK * 4 + 15) & ~(size_t)15);
    o.mrow = o.row + ((2 * H * 4 + 15) & ~(size_t)15);
    o.total = o.mrow + (((H + 1) * 4 + 15) & ~(size_t)15);
    return o;
}

__device__ __forceinline__ bool handover_header_ok(const DevCtx& c, const HandoverHeader* h)
{
    return h->magic == SVO_HANDOVER_MAGIC && h->version == SVO_HANDOVER_VERSION && h->max_kps == c.max_kps && h->max_h == c.max_h &&
           h->oct_cap == c.oct_cap && h->n_lanes == c.n_lanes;
}

__device__ __forceinline__ void copy_words(const void* src, void* dst, size_t nbytes, int t, int nt)
{
    const uint32_t* sp = (const uint32_t*)src; uint32_t* dp = (uint32_t*)dst;
    for (size_t i = (size_t)t; i < nbytes / 4; i += (size_t)nt) dp[i] = sp[i];
}

// grid (n_vl, 8): block (vl, part) copies a share of the lane-octave's lists
__global__ void __launch_bounds__(256) k_export_frame(DevCtx c, uint8_t* blob)
{
    SVO_TL_SCOPE(c, TL_OTHER, 0);
    const int vl = blockIdx.x, lane = vl / c.oct_cap, oct = vl % c.oct_cap;
    const HandoverOffsets o = handover_offsets(c.max_kps, c.max_h);
    uint8_t* rec = blob + (size_t)vl * o.total;
    const LaneState& s = c.lane[lane];
    // the slot the NEXT call finds as "previous": the current one, or the older one after a failed call (P:86-89)
    const bool keep_old = s.m_error == SVO_VOEC_BAD_TRACKING || s.m_error == SVO_VOEC_BAD_COND_NUMBER;
    const int slot = (keep_old || !s.has_cur) ? s.prev_slot : 1 - s.prev_slot;
    const bool present = oct < c.n_oct && ((keep_old || !s.has_cur) ? s.has_prev != 0 : true);
    const int nl = present ? c.n_kps[feat_cnt_idx(vl, slot, 0)] : 0, nr = present ? c.n_kps[feat_cnt_idx(vl, slot, 1)] : 0;
    const int nm = present ? c.n_matches[vl * 2 + slot] : 0, ni = present ? c.n_ids[vl * 2 + slot] : 0;
    const int t = blockIdx.y * blockDim.x + threadIdx.x, nt = gridDim.y * blockDim.x;
    if (t == 0) {
        HandoverHeader* h = (HandoverHeader*)rec;
        h->magic = SVO_HANDOVER_MAGIC; h->version = SVO_HANDOVER_VERSION; h->max_kps = c.max_kps; h->max_h = c.max_h; h->oct_cap = c.oct_cap; h->n_lanes = c.n_lanes;
        h->n_kps[0] = nl; h->n_kps[1] = nr; h->n_matches = nm; h->n_ids = ni; h->present = present ? 1 : 0;
        h->ls = s;
    }
    const size_t MK = (size_t)c.max_kps, H = (size_t)c.max_h;
    copy_words(c.kps + feat_base(c, vl, slot, 0), rec + o.kps, (size_t)nl * sizeof(svo_keypoint), t, nt);
    copy_words(c.kps + feat_base(c, vl, slot, 1), rec + o.kps + MK * sizeof(svo_keypoint), (size_t)nr * sizeof(svo_keypoint), t, nt);
    copy_words(c.desc + feat_base(c, vl, slot, 0) * 32, rec + o.desc, (size_t)nl * 32, t, nt);
    copy_words(c.desc + feat_base(c, vl, slot, 1) * 32, rec + o.desc + MK * 32, (size_t)nr * 32, t, nt);
    copy_words(c.matches + match_base(c, vl, slot), rec + o.matches, (size_t)nm * sizeof(svo_dmatch), t, nt);
    copy_words(c.ids + match_base(c, vl, slot), rec + o.ids, (size_t)ni * 4, t, nt);
    copy_words(c.row_index + (long long)feat_cnt_idx(vl, slot, 0) * c.max_h, rec + o.row, H * 4, t, nt);
    copy_words(c.row_index + (long long)feat_cnt_idx(vl, slot, 1) * c.max_h, rec + o.row + H * 4, H * 4, t, nt);
    copy_words(c.mrow_index + (long long)(vl * 2 + slot) * (c.max_h + 1), rec + o.mrow, (H + 1) * 4, t, nt);
}

// The importing context has ALREADY run stages 2-3 of its frame (k_begin_frame shifted its own, stale, slots): the record
// replaces whatever sits in its previous-frame slot and the inherited estimator members; its current frame stays.
__global__ void __launch_bounds__(256) k_import_frame(DevCtx c, const uint8_t* blob)
{
    SVO_TL_SCOPE(c, TL_OTHER, 1);
    const int vl = blockIdx.x, lane = vl / c.oct_cap;
    const HandoverOffsets o = handover_offsets(c.max_kps, c.max_h);
    const uint8_t* rec = blob + (size_t)vl * o.total;
    const HandoverHeader* h = (const HandoverHeader*)rec;
    const int slot = c.lane[lane].prev_slot;                 // not modified below
    const int t = blockIdx.y * blockDim.x + threadIdx.x, nt = gridDim.y * blockDim.x;
    if (!handover_header_ok(c, h)) {                          // another layout (or not a record at all): nothing is copied, the lane is flagged
        if (t == 0) { atomicOr(&c.status[lane], SVO_ST_HANDOVER_MISMATCH); atomicOr(&c.results[lane].status, (int)SVO_ST_HANDOVER_MISMATCH); }
        return;
    }
    // counts are clamped to the lists' capacity: a damaged record must not write past the lane's lists
    const int nl = min(max(h->n_kps[0], 0), c.max_kps), nr = min(max(h->n_kps[1], 0), c.max_kps);
    const int nm = min(max(h->n_matches, 0), c.max_kps), ni = min(max(h->n_ids, 0), c.max_kps);
    const size_t MK = (size_t)c.max_kps, H = (size_t)c.max_h;
    copy_words(rec + o.kps, c.kps + feat_base(c, vl, slot, 0), (size_t)nl * sizeof(svo_keypoint), t, nt);
    copy_words(rec + o.kps + MK * sizeof(svo_keypoint), c.kps + feat_base(c, vl, slot, 1), (size_t)nr * sizeof(svo_keypoint), t, nt);
    copy_words(rec + o.desc, c.desc + feat_base(c, vl, slot, 0) * 32, (size_t)nl * 32, t, nt);
    copy_words(rec + o.desc + MK * 32, c.desc + feat_base(c, vl, slot, 1) * 32, (size_t)nr * 32, t, nt);
    copy_words(rec + o.matches, c.matches + match_base(c, vl, slot), (size_t)nm * sizeof(svo_dmatch), t, nt);
    copy_words(rec + o.ids, c.ids + match_base(c, vl, slot), (size_t)ni * 4, t, nt);
    copy_words(rec + o.row, c.row_index + (long long)feat_cnt_idx(vl, slot, 0) * c.max_h, H * 4, t, nt);
    copy_words(rec + o.row + H * 4, c.row_index + (long long)feat_cnt_idx(vl, slot, 1) * c.max_h, H * 4, t, nt);
    copy_words(rec + o.mrow, c.mrow_index + (long long)(vl * 2 + slot) * (c.max_h + 1), (H + 1) * 4, t, nt);
    if (t == 0) {
        c.n_kps[feat_cnt_idx(vl, slot, 0)] = nl; c.n_kps[feat_cnt_idx(vl, slot, 1)] = nr;
        c.n_matches[vl * 2 + slot] = nm; c.n_ids[vl * 2 + slot] = ni;
    }
}

// lane-level members, after the list copies of every octave are under way (separate tiny kernel: one writer per lane)
__global__ void k_import_state(DevCtx c, const uint8_t* blob)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= c.n_lanes) return;
    const HandoverOffsets o = handover_offsets(c.max_kps, c.max_h);
    const HandoverHeader* h = (const HandoverHeader*)(blob + (size_t)lane * c.oct_cap * o.total);
    if (!handover_header_ok(c, h)) return;                  // flagged by k_import_frame; the lane keeps what it had
    LaneState& s = c.lane[lane];
    const LaneState& e = h->ls;
    s.has_prev = h->present;
    // P:95: the new call starts with m_error cleared (k_begin_frame of this context did that already); what it inherits:
    for (int k = 0; k < 6; k++) s.last_pose[k] = e.last_pose[k];
    s.it_counter = e.it_counter + 1;
    s.reset_ids = e.reset_ids; s.last_match_id = e.last_match_id; s.last_kf_max_id = e.last_kf_max_id; s.num_tracked_last_kf = e.num_tracked_last_kf;
}

void launch_export_frame(const DevCtx& c, uint8_t* blob, hipStream_t st)
{
    hipLaunchKernelGGL(k_export_frame, dim3(c.n_lanes * c.oct_cap, 8), dim3(256), 0, st, c, blob);
}
void launch_import_frame(const DevCtx& c, const uint8_t* blob, hipStream_t st)
{
    hipLaunchKernelGGL(k_import_frame, dim3(c.n_lanes * c.oct_cap, 8), dim3(256), 0, st, c, blob);
    hipLaunchKernelGGL(k_import_state, dim3((c.n_lanes + 63) / 64), dim3(64), 0, st, c, blob);
}
