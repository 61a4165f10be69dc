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
#define SVO_RANSAC_HYP 256
#define SVO_RANSAC_SEED 0x5EEDF00DCAFE1234ULL

// status-word bits (svo_debug_get_status_word)
#define SVO_ST_CAND_OVERFLOW 1u
#define SVO_ST_KPS_OVERFLOW 2u

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
    int pad;
};

struct DevCtx {
    int n_lanes, n_img, n_levels;
    int W, H;
    int img0_pitch;
    int max_kps, raw_cap, cand_total, n_tiles, n_slots;
    int fast_th, orb_th;
    int debug_mode;           // SVO_DEBUG_MODE env (kernel ablations while tuning; 0 in production)
    long long pyr_bytes;
    LevelGeom lv[SVO_MAX_LEVELS];
    // buffers
    const uint8_t** img0;     // [n_img] level-0 pointers of the frame being processed
    uint8_t* pyr;
    int* rtab;                // resize tables: for level l: idx_x[w], frac_x[w], idx_y[h], frac_y[h]
    uint32_t* cand_keys;
    uint32_t* cand_cnt;       // [n_img][SVO_MAX_LEVELS]
    uint32_t* lvl_pos;
    float* lvl_resp;
    int* lvl_n;               // [n_img][SVO_MAX_LEVELS]
    svo_keypoint* raw_kps;
    uint8_t* raw_desc;
    int* raw_n;               // [n_img]
    svo_keypoint* kps;
    uint8_t* desc;
    int* n_kps;               // [n_lanes][2][2]
    svo_dmatch* matches;
    int* n_matches;           // [n_lanes][2]
    // stage 3/4 scratch, per lane
    int* bf_idx;              // [n_lanes][3][max_kps]  best train index for: LR, prevL->curL, prevR->curR
    int* bf_dist;             // [n_lanes][3][max_kps]
    int* trk_kq;              // [n_lanes][max_kps]  indices k that survive the joint filter (S4:145-160)
    int* trk_nk;              // [n_lanes]
    float* trk_pts;           // [n_lanes][2 sides][max_kps][4]  (x1,y1,x2,y2) for the F-matrix RANSAC
    double* rs_F;             // [n_lanes][2][HYP][9]
    int* rs_cnt;              // [n_lanes][2][HYP]
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
};

__device__ __forceinline__ const uint8_t* level_ptr(const DevCtx& c, int img, int level, int& pitch)
{
    if (level == 0) { pitch = c.img0_pitch; return c.img0[img]; }
    pitch = c.lv[level].pitch;
    return c.pyr + (long long)img * c.pyr_bytes + c.lv[level].offset;
}

__device__ __forceinline__ long long feat_base(const DevCtx& c, int lane, int slot, int side)
{
    return (((long long)lane * 2 + slot) * 2 + side) * c.max_kps;
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
template <bool DESC>
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* keys, int P)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = tid; i < P; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = ((i & k) == 0);
                    const bool sw = DESC ? (up ? a < b : a > b) : (up ? a > b : a < b);
                    if (sw) { keys[i] = b; keys[ixj] = a; }
                }
            }
        }
    }
    __syncthreads();
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

// exclusive prefix sum of one int per thread over the block (blockDim.x <= 1024); `scratch` has >= 17 ints.
// returns the exclusive prefix; *total receives the block sum.
__device__ __forceinline__ int block_exclusive_scan(int v, int* scratch, int* total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < nw; w++) { int t = scratch[w]; scratch[w] = s; s += t; } scratch[16] = s; }
    __syncthreads();
    const int base = scratch[wid];
    *total = scratch[16];
    return base + inc - v;
}

// ---- the reference's greedy grid NMS (m_non_max_sup, stage2_detect.cpp:225-283 / 296-370), exact, wave-parallel ----
// Keypoints are visited in `order` (response-descending ranks).  An accepted keypoint blocks its own cell and
// the 4 neighbours; a keypoint is rejected iff an EARLIER ACCEPTED one lies within Manhattan cell distance 1.
// One wave walks the list in chunks of 64: earlier chunks are looked up in an LDS hash set of accepted cells,
// conflicts inside the chunk are resolved with ballot masks in rank order.  Must be called by exactly one full
// wave (64 lanes).  hash: table of HSZ (power of two) u32 slots, pre-filled with 0xFFFFFFFF.
// cellx/celly(i) give the grid cell of the i-th keypoint in rank order; accept(i, out_index) is called for
// survivors.  Returns the number accepted (<= cap).
template <typename CellFn, typename AcceptFn>
__device__ __forceinline__ int grid_nms_wave(int n, int cap, unsigned gly, uint32_t* hash, int HSZ, CellFn cell, AcceptFn accept)
{
    const int lane = threadIdx.x & 63;
    int n_acc = 0;
    for (int base = 0; base < n && n_acc < cap; base += 64) {
        const int i = base + lane;
        const bool valid = i < n;
        int sx = -4, sy = -4;
        bool in_grid = false;
        if (valid) in_grid = cell(i, sx, sy);
        // 1) blocked by a keypoint accepted in an earlier chunk?
        bool blocked = false;
        if (valid && in_grid) {
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int cx = sx + (q == 1) - (q == 2), cy = sy + (q == 3) - (q == 4);
                if (cx < 0 || cy < 0) continue;
                const uint32_t key = (uint32_t)cx * gly + (uint32_t)cy;
                uint32_t h = (key * 2654435761u) & (uint32_t)(HSZ - 1);
                for (;;) { const uint32_t v = hash[h]; if (v == key) { blocked = true; break; } if (v == 0xFFFFFFFFu) break; h = (h + 1) & (uint32_t)(HSZ - 1); }
            }
        }
        // 2) conflicts with earlier lanes of this chunk
        unsigned long long conf = 0;
        for (int j = 0; j < 63; j++) {
            const int ox = __shfl(sx, j, 64), oy = __shfl(sy, j, 64);
            const int d = abs(ox - sx) + abs(oy - sy);
            if (j < lane && d <= 1) conf |= 1ull << j;
        }
        bool undecided = valid && in_grid && !blocked;
        unsigned long long acc_mask = 0;
        for (;;) {
            const unsigned long long und = __ballot(undecided);
            if (!und) break;
            bool acc_now = false;
            if (undecided) {
                if (conf & acc_mask) undecided = false;                     // an earlier accepted keypoint blocks me
                else if (!(conf & und)) { acc_now = true; undecided = false; }   // nobody earlier can still block me
            }
            acc_mask |= __ballot(acc_now);
        }
        // 3) commit in rank order, honouring the cap (the reference stops at num_out_points)
        const bool is_acc = (acc_mask >> lane) & 1ull;
        const int my = n_acc + __popcll(acc_mask & ((1ull << lane) - 1ull));
        if (is_acc && my < cap) {
            const uint32_t key = (uint32_t)sx * gly + (uint32_t)sy;
            uint32_t h = (key * 2654435761u) & (uint32_t)(HSZ - 1);
            for (;;) { const uint32_t old = atomicCAS(&hash[h], 0xFFFFFFFFu, key); if (old == 0xFFFFFFFFu || old == key) break; h = (h + 1) & (uint32_t)(HSZ - 1); }
            accept(i, my);
        }
        n_acc += __popcll(acc_mask);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    return n_acc < cap ? n_acc : cap;
}
