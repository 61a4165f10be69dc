// svo_device.h -- device-side data layout and small wave/block primitives shared by the HIP kernels.
//
// HBM layout (all context-owned, allocated once in svo_create):
//   pyr        [n_img][pyr_bytes]              levels 1..L-1 of every image, row pitch padded to 64 B
//   cand_keys  [n_img][cand_total] u32         FAST corners per level: score<<24 | (0xFFFFFF - (y*w+x))
//   lvl_pos/resp [n_img][raw_cap]              per-level winners (level-segmented, quota[l] slots each)
//   raw_kps/raw_desc [n_img][raw_cap]          cv::ORB output stand-in before the reference's own NMS
//   kps/desc   [n_lanes][2 slots][2 sides][max_kps]   final row-sorted features; slot = current/previous frame
//   matches    [n_lanes][2 slots][max_kps]     left-right pairings (cv::DMatch records)
//   ...        per-lane scratch of stages 4 and 5, lane state, result records
// n_img = 2 * n_lanes; image index = lane * 2 + side.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svo_hip.h"

#define SVO_EDGE 31
#define SVO_RANSAC_HYP 1000         // sample schedule of one F-matrix RANSAC (oracle: RANSAC_MAX_HYP): minimal samples of seven pairs
#define SVO_RANSAC_PAD 1024         // ... padded to whole regions
#define SVO_RANSAC_REG 16           // samples per region (one DPP row of k_ransac_hyp_thread, one block of k_ransac_hyp)
#define SVO_RANSAC_RSLOTS 48        // model slots per region: a sample has one or three models (the real roots of the 7-point cubic)
#define SVO_RANSAC_SLOTS (SVO_RANSAC_PAD / SVO_RANSAC_REG * SVO_RANSAC_RSLOTS)     // stride of the per-(lane, side) model arrays
#define SVO_RANSAC_CHUNK0 32        // samples [0, CHUNK0) are evaluated unconditionally, [CHUNK0, CHUNK1) and [CHUNK1, HYP) only as far as
#define SVO_RANSAC_FEW 320          // both chunk ends for a handful of lanes (DevCtx.rs_c0 / rs_c1): one hypothesis + count pair covers what the stop rule usually visits
#ifndef SVO_RANSAC_CHUNK1
#define SVO_RANSAC_CHUNK1 160       // the 0.99-confidence stop of the sequential algorithm can still reach (rs_bound)
#endif
#define SVO_SEL_MAX 2048     // >= 2 * quota[0]: corners per (image, level) ranked by their Harris response
#define SVO_FT_W 62          // k_fast tile (interior pixels; 64x64 score window with the NMS halo)
#define SVO_FT_H 62
#define SVO_CNT_STRIDE 32          // u32 stride between hot atomic counters = one 128-byte cache line each
#define SVO_RS_ATT 1152             // tabulated attempts of cv::findFundamentalMat's sampler per point count n (1000 samples + room for rejected attempts)
#define SVO_RS_SMALL_N 64           // ... below this many points 11008 of them: enough for 1000 samples and getSubset's 10000 attempts on the last
#define SVO_RS_ATT_SMALL 11008
#define SVO_LMEDS_MAX_N 14          // 8 .. 14 point pairs: cv::findFundamentalMat's LMedS registrator instead of the RANSAC (`npoints >= 15`)
#define SVO_LMEDS_ITERS 300          // ... its fixed budget: RANSACUpdateNumIters(0.99, outlier ratio 0.45, 7 points, 1000) = cvRound(ln 0.01 / ln(1 - 0.55^7))
#define SVO_RS_ST 12                // ints of schedule state per lane-octave (rs_sched)
#define SVO_RS_EXT 9216             // attempts a lane may draw beyond its table row, continuing cv::RNG on the device (k_match.hip, rs_schedule_block): table + this >= getSubset's 10000 + one chunk, so that "no sample in 10000 attempts" is decided as OpenCV decides it

// status-word bits (svo_debug_get_status_word)
#define SVO_ST_CAND_OVERFLOW 1u
#define SVO_ST_KPS_OVERFLOW 2u
#define SVO_ST_INTERNAL 8u            // an internal consistency bound was hit (a loop that must terminate did not, an index left its list): a bug, reported instead of a hang
#define SVO_ST_HANDOVER_MISMATCH 4u  // svo_import_frame was handed a record of another layout (magic / version / max_kps / max_h / octaves / lanes)

// division of a 32-bit unsigned by a launch-time constant (Granlund-Montgomery round-up): exact for every x, and on a
// wave-uniform x it compiles to s_mul_hi_u32 + three scalar ops -- the hardware has no integer divide, and the
// compiler's expansion costs ~25 VALU instructions per use
struct FastDiv { uint32_t m, s1, s2, d; };
static inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f; f.d = d;
    uint32_t l = 0; while ((1ull << l) < d) l++;
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.s1 = l < 1 ? l : 1; f.s2 = l > 1 ? l - 1 : 0;
    return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t fastdiv(uint32_t x, const FastDiv& f)
{
    const uint32_t t = __umulhi(f.m, x);
    return (t + ((x - t) >> f.s1)) >> f.s2;
}
#endif

struct LevelGeom {
    int w, h, pitch;
    int tiles_x, tiles_y, tile_off;   // FAST tiling of the [EDGE, w-EDGE) x [EDGE, h-EDGE) interior
    int quota, slot_off;              // per-level keypoint quota and its offset in the level-segmented arrays
    int cand_cap, cand_off;
    long long offset;                 // byte offset of the level in one image's pyramid block (levels >= 1)
    float scale;
    int rtab_off;                     // offset of this level's resize tables (x table then y table)
};

struct LaneState {
    int prev_slot;        // slot (0/1) holding the previous frame; the current frame is written to 1 - prev_slot
    int has_prev;         // m_prev_imgpair.present()
    int has_cur;
    int m_error;          // CStereoOdometryEstimator::m_error (C:38)
    double last_pose[6];  // m_last_computed_pose (C:49)
    unsigned it_counter;
    int reset_ids;        // m_reset (H:684, P:254-267)
    int last_match_id;    // m_last_match_ID (H:741)
    int last_kf_max_id;   // m_last_kf_max_id (H:738; uninitialised in the reference, 0 here)
    int num_tracked_last_kf;   // m_num_tracked_pairs_from_last_kf (S4:743-751)
    int pad;
};

// Octaves: in the FAST+ORB mode (stage2_detect.cpp:502-515) the reference works on nOctaves x1/2 images per eye and
// keeps one feature / pairing / track list per octave.  Every per-lane list below is therefore indexed by the
// "lane-octave" vl = lane * oct_cap + octave (oct_cap = svo_config.max_octaves; ORB mode uses octave 0 only).
struct DevCtx {
    int n_lanes, n_img, n_levels;
    int W, H;
    int oct_cap, n_oct;       // allocated / active octaves
    int ow[SVO_MAX_OCTAVES], oh[SVO_MAX_OCTAVES];      // octave image sizes
    int kps_to_detect[SVO_MAX_OCTAVES];                // stage2_detect.cpp:404-407
    int fast_orb;             // 1: detect_method == dmFAST_ORB (levels of lv[] are the octave images, keypoint size 7)
    int max_h;                // row-index table pitch
    int img0_pitch;
    int max_kps, raw_cap, cand_total, n_tiles, n_slots;
    int sel_max;              // capacity of a level's 2 * quota list (2048, or 4096 when max_kps > 4096)
    uint8_t* big_scratch;     // global stand-in for the large LDS arrays of k_nms_rowsort when lists exceed 4096 entries
    uint8_t* gn_scratch;      // the same for k_gauss_newton
    FastDiv div_tiles;        // / n_tiles
    int fast_th, orb_th;
    int debug_mode;           // SVO_DEBUG_MODE in the environment of svo_create; 0 in production.  Timing ablations (results are then
                              // meaningless) and forced kernel forms (results unchanged: the parity tests run them):
                              //   1 / 2 / 3 / 4  k_fast returns after staging / compaction / scores, or publishes nothing
                              //   5 / 6 / 7      k_resize without staging / without the blend / without the store
                              //   8              linear (not XCD-chunked) tile orders              [results unchanged]
                              //   9              detector order: describe every raw keypoint, then NMS   [results unchanged]
                              //   10 / 11        k_gauss_newton returns after its set-up / runs one iteration per phase
                              //   12             no speculative FAST threshold                     [results unchanged]
                              //   13 / 16        RANSAC count replays every verdict / never stops early   [results unchanged]
                              //   14, 50-54      forced RANSAC kernel forms (k_match.hip launchers)        [results unchanged]
                              //   21-26          k_nms_rowsort returns after a phase (tests/dev/nms_breakdown.py)
                              //   31, 41-45      k_select / k_select_sort return after a phase
                              //   60-63          k_gauss_newton without its per-track loop / solve / rotation update (tests/dev/gn_breakdown.py)
    long long pyr_bytes;
    LevelGeom lv[SVO_MAX_LEVELS];
    // buffers
    const uint8_t** img0;     // [n_img] level-0 pointers of the frame being processed
    uint8_t* pyr;
    int* rtab;                // resize tables: for level l: idx_x[w], frac_x[w], idx_y[h], frac_y[h]
    uint32_t* cand_keys;
    uint32_t* cand_cnt;       // [n_img][SVO_MAX_LEVELS] counters, ONE PER 128-BYTE LINE (atomics to one L2 line serialise)
    // speculative per-(image, level) FAST threshold (see k_select): th_dyn = what the next frame should try, th_used = what
    // this frame's k_fast ran with, redo_flag / redo_list / redo_n = the (image, level) pairs whose speculation failed
    uint32_t* fast_th_dyn;    // [n_img][SVO_MAX_LEVELS]   0 = no speculation (base threshold)
    uint32_t* fast_th_used;   // [n_img][SVO_MAX_LEVELS]
    const uint4* fast_tiles;  // [n_tiles] geometry of FAST tile k of the concatenated level tilings: x0 | y0 << 16, w | h << 16, level | pitch << 8, pyramid offset
    uint32_t* redo_flag;      // [n_img][SVO_MAX_LEVELS]
    uint32_t* redo_list;      // [n_img * SVO_MAX_LEVELS]  img * SVO_MAX_LEVELS + level
    uint32_t* redo_n;         // [1]
    uint32_t* lvl_pos;
    float* lvl_resp;
    uint32_t* sel_keys;             // [n_img][SVO_MAX_LEVELS][SVO_SEL_MAX]  k_select's winners (FAST key), input of k_harris
    unsigned long long* sel_resp;   // same shape: (Harris response, position) keys, input of k_select_sort
    int* sel_n;                     // [n_img][SVO_MAX_LEVELS]
    int* lvl_n;               // [n_img][SVO_MAX_LEVELS]
    svo_keypoint* raw_kps;
    uint8_t* raw_desc;
    int* raw_n;               // [n_img]
    uint2* desc_work;         // [n_img][raw_cap] describe-after-NMS work list: (x | y << 16 at its level, level) of final keypoint i
    int* desc_n;              // [n_img] its length
    svo_keypoint* kps;
    uint8_t* desc;
    int* final_slot;          // same index space as kps: detector slot of each final keypoint (describe-after-NMS path)
    uint8_t* mdesc;           // same shape as desc: descriptors of the paired features in pairing order (stage-4 BF input)
    int* n_kps;               // [n_lanes][2][2]
    svo_dmatch* matches;
    int* n_matches;           // [n_vl][2]
    int* row_index;           // [n_vl][2 slots][2 sides][max_h]   m_update_indexes table (stage2_detect.cpp:103-129)
    int* mrow_index;          // [n_vl][2 slots][max_h + 1]       matches_lr_row_index (stage3:425-445)
    int* ids;                 // [n_vl][2 slots][max_kps]         matches_IDs (H:794), only with vo_use_matches_ids
    int* n_ids;               // [n_vl][2]
    // stage 3/4 scratch, per lane
    int* bf_idx;              // [n_lanes][3][max_kps]  best train index for: LR, prevL->curL, prevR->curR
    int* bf_dist;             // [n_lanes][3][max_kps]
    int* trk_kq;              // [n_lanes][max_kps]  indices k that survive the joint filter (S4:145-160)
    int* trk_nk;              // [n_lanes]
    float* trk_pts;           // [n_lanes][2 sides][max_kps][4]  (x1,y1,x2,y2) for the F-matrix RANSAC
    double* rs_F;             // [n_lanes][2][SLOTS][9]   the models, packed per region of 16 samples (k_match.hip, K9)
    double* rs_guard;         // [n_lanes][2][SLOTS][2]   per model: the |l'|^2 and |l|^2 below which the matrix-core lines are not trusted
    int* rs_cnt;              // [n_lanes][2][SLOTS]      inlier count of each model (0: no model here / not scored)
    int* rs_k;                // [n_lanes][2][SLOTS]      the sample a slot's model came from
    int* rs_nvalid;           // [n_lanes][2][PAD / 16]   models in each region
    int* rs_bound;            // [n_lanes][2]  upper limit of the SAMPLES the sequential stop can still reach
    int* rs_gen;              // [n_lanes][2]  end of the samples the current chunk generated
    const unsigned short* rs_att;   // [n = 8 .. SVO_RS_SMALL_N - 1][SVO_RS_ATT_SMALL][8] then [n = SVO_RS_SMALL_N .. rs_att_nmax][SVO_RS_ATT][8]: the attempts of OpenCV's sampler for n = 8 .. rs_att_nmax points (host-built, k_ransac_schedule)
    int rs_att_nmax;
    const unsigned long long* rs_att_state;   // [rs_att_nmax + 1] cv::RNG's state after the last tabulated attempt of row n (where the device continues the stream)
    int rs_c1;                // end of chunk 1 (= what phase 0 of the schedule draws): SVO_RANSAC_CHUNK1, or more together with rs_c0 for a handful of lanes
    int rs_c0;                // end of chunk 0 of the sample schedule: SVO_RANSAC_CHUNK0, or SVO_RANSAC_CHUNK1 (chunk 1 empty, its two launches skipped) for a handful of lanes, where a launch costs more than the samples it saves (svo_api.hip)
    unsigned short* rs_smp;   // [n_lanes][2][SVO_RANSAC_PAD][8] the seven indices of every sample, as OpenCV's getSubset draws them
    int* rs_sched;            // [n_lanes][SVO_RS_ST] schedule state (k_match.hip, rs_schedule_block)
    int* rs_ticket;           // [n_lanes][2][SLOTS / 16] blocks of k_ransac_count_mfma16 that have added their share of a group's counts (0 between launches)
    int* rs_floor;            // [n_lanes][2][2] best inlier count of chunk 0 / of chunks 0-1: what a later model must exceed to matter
    svo_index_pair* tracked;  // [n_lanes][max_kps]
    int* n_tracked;           // [n_lanes]
    // stage 5
    double* gn_lmk;           // [n_lanes][max_kps][3]
    float* gn_obs;            // [n_lanes][max_kps][8]  l1.xy r1.xy l2.xy r2.xy
    double* residual;         // [n_lanes][max_kps]
    int* outliers;            // [n_lanes][max_kps]
    svo_stereo_camera* cams;  // [n_lanes]
    LaneState* lane;
    svo_result* results;
    uint32_t* status;         // [n_lanes]
    uint32_t* det_status;     // [n_lanes] capacity bits raised by a detect call that runs ahead (SVO_FLAG_DETECT_AHEAD): folded into status / the record by its post call
    int det_ahead;            // 1 while launching the kernels of such a call: k_fast / k_select raise their bits in det_status
    int rest_prio;            // SVO_REST_PRIO (0..3, default in svo_create): wave priority the per-lane latency chains of stages 3-5 run at (SVO_LATENCY_CHAIN)
    // SVO_TIMELINE=1 in the environment of svo_create (nullptr otherwise): what was really in flight, measured inside the kernels
    // (svo_debug_timeline).  A launch = (tl_step % SVO_TL_STEPS) * 256 + kind * 8 + aux owns SVO_TL_SUB records, one per stamping wave
    // (TlScope below): (time in, time out) on the device-wide 100 MHz wall clock.
    struct TlRec* tl;
    int tl_step;              // frame counter of the context (host side: advanced by every call that runs the detector)
    int tl_pad;
};

struct TlRec { unsigned long long t0, t1; };
#define SVO_TL_STEPS 16
#define SVO_TL_SUB 128
enum { TL_BEGIN = 0, TL_RESIZE, TL_FAST, TL_SELECT, TL_HARRIS, TL_SELECT_SORT, TL_NMS, TL_DESCRIBE, TL_HAMMING, TL_LR_FILTER, TL_TRK_FILTER,
       TL_RS_SCHED, TL_RS_HYP, TL_RS_COUNT, TL_TRK_FINAL, TL_MATCH_IDS, TL_GN, TL_OTHER, TL_KINDS };     // < 32

// The per-lane kernels of stages 3-5 are a few waves per lane walking dependent chains; in the batched schedule they share their
// SIMDs with the detector's tiles, which fill every issue slot they are given.  A raised wave priority lets the chain's next
// instruction win the arbitration instead of queueing behind thirty throughput waves (it costs those almost nothing: the chains
// issue rarely).  The value comes from DevCtx so that an A/B needs no rebuild; s_setprio takes an immediate, hence the switch.
#ifdef __HIPCC__
#define SVO_LATENCY_CHAIN(c) do { if ((c).rest_prio == 3) __builtin_amdgcn_s_setprio(3); else if ((c).rest_prio == 2) __builtin_amdgcn_s_setprio(2); else if ((c).rest_prio == 1) __builtin_amdgcn_s_setprio(1); } while (0)
#endif

#ifdef __HIPCC__
// One line at the top of a kernel: SVO_TL_SCOPE(c, kind, aux).  Off (c.tl == nullptr): one scalar compare.  On: a few waves of the launch
// each leave (time in, time out) in a record of their own -- plain 16-byte stores, one per stamping wave, on every way out of the kernel
// (the destructor runs at each return; a wave that leaves in several pieces overwrites its record with the latest time).  Stamping waves:
// wave 0 of every (gridDim.x / 32)-th block and of the last block (sub-records 0..63), and in grids of at most 64 blocks also the last wave
// of every block (64 + blockIdx.x).  The host takes min / max over a launch's sub-records.  (A first version did atomicMin / atomicMax on
// ONE record per launch from every wave: device-scope atomics to one address complete at the memory side at ~90 ns each, and a kernel
// is not complete before they are -- k_harris took 790 us instead of 60, the whole step 7.7 ms instead of 2.8.)
struct TlScope {
    // Off: ONE wave-uniform flag (an SGPR compare at the top of the kernel and one scalar branch per way out) -- k_fast's waves run ~600
    // instructions each, and a first form that decided per lane which record to stamp cost it 15 of them (2.4 %) with the knob off.
    TlRec* base; unsigned long long t0;
    __device__ __forceinline__ TlScope(const DevCtx& c, int kind, int aux) : base(nullptr), t0(0)
    {
        if (c.tl) {
            base = c.tl + ((((size_t)(c.tl_step & (SVO_TL_STEPS - 1)) << 8) | (size_t)((kind & 31) << 3) | (size_t)(aux & 7)) << 7);
            t0 = wall_clock64();
        }
    }
    __device__ __forceinline__ ~TlScope()
    {
        if (base) {
            const unsigned g = gridDim.x, stride = g > 32u ? g / 32u : 1u, wave = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
            int sub = -1;
            if (wave == 0u && blockIdx.x + 1u == g) sub = 63;
            else if (wave == 0u && blockIdx.x % stride == 0u) sub = (int)min(blockIdx.x / stride, 62u);
            else if (g <= 64u && wave + 1u == nw) sub = 64 + (int)blockIdx.x;
            const unsigned long long act = __ballot(1);
            if (sub >= 0 && (int)(threadIdx.x & 63u) == __ffsll((long long)act) - 1) *(ulonglong2*)(base + sub) = make_ulonglong2(t0, (unsigned long long)wall_clock64());
        }
    }
};
#define SVO_TL_SCOPE(c, kind, aux) TlScope _tl_scope((c), (kind), (aux))
#endif

// where a detect-phase kernel raises a capacity bit of its lane
__device__ __forceinline__ void raise_detect_status(const DevCtx& c, int lane, uint32_t bit)
{
    if (c.det_ahead) { atomicOr(&c.det_status[lane], bit); return; }
    atomicOr(&c.status[lane], bit); atomicOr(&c.results[lane].status, (int)bit);
}

__device__ __forceinline__ const uint8_t* level_ptr(const DevCtx& c, int img, int level, int& pitch)
{
    if (level == 0) { pitch = c.img0_pitch; return c.img0[img]; }
    pitch = c.lv[level].pitch;
    return c.pyr + (long long)img * c.pyr_bytes + c.lv[level].offset;
}

__device__ __forceinline__ long long feat_base(const DevCtx& c, int vl, int slot, int side)
{
    return (((long long)vl * 2 + slot) * 2 + side) * c.max_kps;
}
__device__ __forceinline__ int feat_cnt_idx(int lane, int slot, int side) { return (lane * 2 + slot) * 2 + side; }
__device__ __forceinline__ long long match_base(const DevCtx& c, int lane, int slot) { return ((long long)lane * 2 + slot) * c.max_kps; }

// total order on floats through their bit pattern (no NaNs on this path) -- same definition as the oracle's
__device__ __forceinline__ uint32_t ord32(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float inv_ord32(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// ---- block-wide bitonic sort of P (power of two) 64-bit keys held in LDS ---------------------------------
// DESC = true sorts descending.  All threads of the block must call it; keys beyond the live count must be
// padded by the caller (0 for descending, ~0 for ascending).
// make this wave's LDS writes visible to its own later reads (no s_barrier: waves do not share data here)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool DESC>
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* keys, int P)
{
    // One compare-exchange per work item t in [0, P/2): i = t with a zero bit inserted at log2(j), partner i | j -- every
    // thread is busy in every stage.  A wave's 64 consecutive work items stay inside one aligned 128-key segment while
    // j <= 64, so those stages need no block barrier, only the wave's own LDS ordering; for P = 2048 that leaves 14
    // block barriers out of 66 stages.
    const int tid = threadIdx.x, nt = blockDim.x, half = P >> 1;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 128 || (j == 64 && k >= 256)) __syncthreads(); else wave_lds_sync();
            for (int t = tid; t < half; t += nt) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                const unsigned long long a = keys[i], b = keys[ixj];
                const bool up = ((i & k) == 0);
                const bool lt = a < b;                                  // one 64-bit compare: for a != b, a > b == !lt; swapping equal keys is harmless
                const bool sw = DESC ? (lt == up) : (lt != up);
                if (sw) { keys[i] = b; keys[ixj] = a; }
            }
        }
    }
    __syncthreads();
}

// sum over the 64 lanes (all active), returned wave-uniform: four DPP steps inside each row of 16, then one
// v_readlane per row -- no LDS traffic, unlike the ds_bpermute butterflies of __shfl_xor
__device__ __forceinline__ int wave_sum_uniform(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);     // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ int wave_reduce_sum_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum over the 64 lanes (all active) on the DPP network: row_shr 1/2/4/8 scan each row of 16,
// row_bcast15 / row_bcast31 carry the row totals forward.  Six VALU instructions, no LDS (the __shfl_up version is
// six ds_bpermute round trips plus selects).
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2 and 3
    return v;
}

// exclusive prefix sum of one int per thread over the block (blockDim.x <= 1024, a multiple of 64, called by ALL
// threads); `scratch` has >= 17 ints.  Returns the exclusive prefix; *total receives the block sum.
// Two barriers: one so that the previous call's readers are done with `scratch`, one after the wave totals land;
// every wave then sums the totals below it itself (16 lanes, DPP) instead of waiting for a serial pass of thread 0.
__device__ __forceinline__ int block_exclusive_scan(int v, int* scratch, int* total)
{
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (blockDim.x + 63) >> 6;
    const int inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    const int t = lane < nw ? scratch[lane] : 0;
    *total = wave_sum_uniform(t);
    const int base = wave_sum_uniform(lane < wid ? t : 0);
    return base + inc - v;
}

// ---- the reference's greedy grid NMS (m_non_max_sup, stage2_detect.cpp:225-283 / 296-370), exact and block-parallel ----
// Reference: visit keypoints in response-descending rank order; accept one iff its grid cell is unmarked, then mark
// the cell and its 4 neighbours.  Equivalent parallel form used here:
//   * only the lowest-rank keypoint of a cell (its "representative") can ever be accepted: if it is accepted the
//     cell is marked for everybody behind it, and if it is rejected the accepted neighbour that rejected it marked
//     the cell before any later keypoint of the same cell;
//   * a representative is accepted iff no ACCEPTED representative of lower rank sits in one of the 4 neighbour
//     cells.  That recurrence is resolved in rounds: an undecided representative is rejected as soon as a
//     lower-rank neighbour is accepted, accepted as soon as all lower-rank neighbours are rejected.  The undecided
//     representative of globally lowest rank always decides, so the loop terminates; chains are short in practice.
// cellxy[i] = (sx << 16) | sy of the rank-i keypoint, 0xFFFFFFFF when outside the grid (such keypoints are skipped).
// hkey/hval: LDS hash of HSZ (power of two >= 2n) slots; state[i] ends 1 (accepted) or 0.  All threads of the
// block must call; `flag` points at THREE LDS ints.  The caller applies the num_out_points cap in rank order.
#define SVO_NMS_UNDECIDED 2
__device__ __forceinline__ uint32_t nms_hash_slot(uint32_t key, int HSZ) { return ((key * 2654435761u) >> 7) & (uint32_t)(HSZ - 1); }

__device__ __forceinline__ uint32_t nms_lookup(const uint32_t* hkey, const uint32_t* hval, int HSZ, uint32_t key)
{
    uint32_t h = nms_hash_slot(key, HSZ);
    for (;;) {
        const uint32_t k = hkey[h];
        if (k == key) return hval[h];
        if (k == 0xFFFFFFFFu) return 0xFFFFFFFFu;
        h = (h + 1) & (uint32_t)(HSZ - 1);
    }
}

// ITEMS > 0: n <= ITEMS * blockDim.x is guaranteed by the caller and the neighbour lookups are cached in registers;
// ITEMS = 0: any n, lookups repeated every round.
template <int ITEMS>
__device__ __forceinline__ void grid_nms_block(int n, unsigned gly, const uint32_t* cellxy, uint32_t* hkey, uint32_t* hval, int HSZ,
                                               volatile unsigned char* state, volatile int* flag)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < HSZ; i += nt) { hkey[i] = 0xFFFFFFFFu; hval[i] = 0xFFFFFFFFu; }
    __syncthreads();
    // representative (minimum rank) of every occupied cell
    for (int i = tid; i < n; i += nt) {
        const uint32_t c = cellxy[i];
        if (c == 0xFFFFFFFFu) continue;
        const uint32_t key = (c >> 16) * gly + (c & 0xFFFFu);
        uint32_t h = nms_hash_slot(key, HSZ);
        for (;;) {
            const uint32_t old = atomicCAS(&hkey[h], 0xFFFFFFFFu, key);
            if (old == 0xFFFFFFFFu || old == key) { atomicMin(&hval[h], (uint32_t)i); break; }
            h = (h + 1) & (uint32_t)(HSZ - 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const uint32_t c = cellxy[i];
        unsigned char st = 0;
        if (c != 0xFFFFFFFFu && nms_lookup(hkey, hval, HSZ, (c >> 16) * gly + (c & 0xFFFFu)) == (uint32_t)i) st = SVO_NMS_UNDECIDED;
        state[i] = st;
    }
    __syncthreads();
    if constexpr (ITEMS == 0) {
        if (tid < 3) flag[tid] = 0;
        __syncthreads();
        for (int round = 0; round <= n; round++) {                     // every round decides the lowest-rank undecided representative: n rounds always suffice
            // ONE barrier per round (three until round 4): the rounds rotate over three flag words -- this round's is raised before the
            // barrier and read after it, the next round's is cleared now (its last readers passed the previous barrier)
            volatile int* fl = flag + round % 3;
            if (tid == 0) flag[(round + 1) % 3] = 0;
            bool pending = false;
            for (int i = tid; i < n; i += nt) {
                if (state[i] != SVO_NMS_UNDECIDED) continue;
                const uint32_t c = cellxy[i];
                const int sx = (int)(c >> 16), sy = (int)(c & 0xFFFFu);
                bool any_acc = false, any_und = false;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int cx = sx + (q == 0) - (q == 1), cy = sy + (q == 2) - (q == 3);
                    if (cx < 0 || cy < 0 || cy >= (int)gly) continue;      // (cx, gly) would alias the key of (cx+1, 0)
                    const uint32_t j = nms_lookup(hkey, hval, HSZ, (uint32_t)cx * gly + (uint32_t)cy);
                    if (j < (uint32_t)i) { const unsigned char sj = state[j]; any_acc |= sj == 1; any_und |= sj == SVO_NMS_UNDECIDED; }
                }
                if (any_acc) state[i] = 0;
                else if (!any_und) state[i] = 1;
                else pending = true;
            }
            if (pending) *fl = 1;
            __syncthreads();
            if (!*fl) break;
        }
        __syncthreads();
        return;
    }
    // the lower-rank representatives of the 4 neighbour cells of each of this thread's undecided keypoints, looked up
    // ONCE (the hash probes are the expensive part); the rounds below only re-read their states
    constexpr int NI = ITEMS > 0 ? ITEMS : 1;
    unsigned short nb[NI][4];
#pragma unroll
    for (int it = 0; it < NI; it++) {
        const int i = tid + it * nt;
#pragma unroll
        for (int q = 0; q < 4; q++) nb[it][q] = 0xFFFFu;
        if (i < n && state[i] == SVO_NMS_UNDECIDED) {
            const uint32_t c = cellxy[i];
            const int sx = (int)(c >> 16), sy = (int)(c & 0xFFFFu);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cx = sx + (q == 0) - (q == 1), cy = sy + (q == 2) - (q == 3);
                if (cx < 0 || cy < 0 || cy >= (int)gly) continue;      // (cx, gly) would alias the key of (cx+1, 0)
                const uint32_t j = nms_lookup(hkey, hval, HSZ, (uint32_t)cx * gly + (uint32_t)cy);
                if (j < (uint32_t)i) nb[it][q] = (unsigned short)j;
            }
        }
    }
    if (tid < 3) flag[tid] = 0;
    __syncthreads();
    // (Round 6 handed the chains' tails -- what is still undecided after six rounds -- to ONE wave with wave-level LDS ordering instead of a
    // block barrier per link: lists equal, k_nms_rowsort's NMS phase 24.5 -> 25.0 us (tests/dev/nms_breakdown.py, r06n).  The phase's time is
    // the hash build and the cached neighbour look-ups above, not these rounds; removed again.)
    for (int round = 0; round <= n; round++) {
        volatile int* fl = flag + round % 3;                           // one barrier per round: see the ITEMS == 0 loop above
        if (tid == 0) flag[(round + 1) % 3] = 0;
        bool pending = false;
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int i = tid + it * nt;
            if (i >= n || state[i] != SVO_NMS_UNDECIDED) continue;
            bool any_acc = false, any_und = false;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (nb[it][q] != 0xFFFFu) { const unsigned char sj = state[nb[it][q]]; any_acc |= sj == 1; any_und |= sj == SVO_NMS_UNDECIDED; }
            if (any_acc) state[i] = 0;
            else if (!any_und) state[i] = 1;
            else pending = true;
        }
        if (pending) *fl = 1;
        __syncthreads();
        if (!*fl) break;
    }
    __syncthreads();
}
