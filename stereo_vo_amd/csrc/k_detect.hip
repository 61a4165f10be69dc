// k_detect.hip -- stage 2 on the device: pyramid, FAST-9/16 + 3x3 NMS, per-level top-K, Harris, orientation,
// steered BRIEF-256, then the reference's own grid NMS and row sort.
//
// Replaces cv::ORB::detectAndCompute as called at libstereo-odometry/src/stage2_detect.cpp:482-493 and the
// reference's m_non_max_sup (stage2_detect.cpp:296-370) + m_update_indexes (stage2_detect.cpp:65-130).
// Every formula is the one frozen in oracle/svo_oracle.c; integer work is bit-exact by construction and the few
// float expressions are written one IEEE operation per operator (compiled with -ffp-contract=off).
#include <algorithm>
#include "svo_device.h"
#include "svo_kernels.h"
#include "../../include/svo_orb_tables.h"

// device copies of the constant tables
__constant__ int c_umax[16];
// cv::ORB's 256 test pairs (bit_pattern_31_) as floats, (x0, x1, y0, y1) per pair -- the operand pairs of packed f32 operations:
// k_describe rotates them by the keypoint's angle
__device__ __attribute__((aligned(16))) float4 g_brief_patf[SVO_BRIEF_NPAIRS];
// The 7 x 7 Gaussian of k_describe as two banded matrices in matrix-core operand layout (v_mfma_i32_16x16x64_i8: lane l carries
// the 16 bytes of k-group l / 16 for row / column l % 16; byte b of a group in A meets byte b of the same group in B):
//   g_blur_gh[nt][l]  horizontal pass H = X Gh:  column n = 16 nt + l % 16 of Gh, byte b <-> window byte 16 (l / 16) + b
//   g_blur_gv[ot][l]  vertical pass  B^T = H^T Gv^T:  column = output row o = 16 ot + l % 16, byte s <-> H row
//                     16 (s / 4) + 4 (l / 16) + s % 4 for s < 12 (where the horizontal pass left its results), 0 for s >= 12
__device__ __attribute__((aligned(16))) uint4 g_blur_gh[3 * 64], g_blur_gv[3 * 64];
// the radius-15 disc by rows of the describe window, for v_dot4_u32_u8: entry e = (v + 15) * 8 + d covers the window
// bytes 8 + 4d .. 11 + 4d of row v + 21 (columns u = 4d - 15 .. 4d - 12); g_disc_m holds 1 per disc pixel,
// g_disc_x holds u + 15 per disc pixel (0 outside), one byte each
#define SVO_DISC_E 256
__device__ uint32_t g_disc_m[SVO_DISC_E], g_disc_x[SVO_DISC_E];
#define DP_REACH SVO_BRIEF_REACH          // 18: the blurred pixels a rotated pair can reach are (x, y) +- 18
#define DP_BW (2 * DP_REACH + 1)          // 37 x 37 blurred window
#define DP_RW (DP_BW + 6)                 // 43 raw rows y - 21 .. y + 21
#define DP_X0 (DP_REACH + 5)              // window byte 0 = column x - 23, so that the disc's first column x - 15 is byte 8 (a dword boundary)

hipError_t svo_upload_tables()
{
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_umax), svo_umax, sizeof(svo_umax));
    if (e != hipSuccess) return e;
    {
        static float4 pf[SVO_BRIEF_NPAIRS];
        for (int i = 0; i < SVO_BRIEF_NPAIRS; i++) pf[i] = make_float4((float)svo_brief_pat[i][0], (float)svo_brief_pat[i][2], (float)svo_brief_pat[i][1], (float)svo_brief_pat[i][3]);
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_brief_patf), pf, sizeof(pf));
        if (e != hipSuccess) return e;
    }
    {
        static uint8_t gh[3 * 64 * 16], gv[3 * 64 * 16];
        for (int t = 0; t < 3; t++)
            for (int l = 0; l < 64; l++)
                for (int b = 0; b < 16; b++) {
                    const int q = l >> 4, i = l & 15;
                    // horizontal: output column n (blurred column x - 18 + n) sums window bytes n + 2 .. n + 8 (columns x - 21 + n .. x - 15 + n)
                    const int n = 16 * t + i, kb = 16 * q + b, dh = kb - (n + DP_X0 - DP_REACH - 3);
                    gh[(t * 64 + l) * 16 + b] = (uint8_t)((n < DP_BW && dh >= 0 && dh < 7) ? svo_gauss7[dh] : 0);
                    // vertical: output row o (blurred row y - 18 + o) sums H rows o .. o + 6 (window rows y - 21 + o ..)
                    const int o = 16 * t + i, rho = 16 * (b >> 2) + 4 * q + (b & 3), dv = rho - o;
                    gv[(t * 64 + l) * 16 + b] = (uint8_t)((b < 12 && o < DP_BW && dv >= 0 && dv < 7) ? svo_gauss7[dv] : 0);
                }
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_blur_gh), gh, sizeof(gh));
        if (e != hipSuccess) return e;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_blur_gv), gv, sizeof(gv));
        if (e != hipSuccess) return e;
    }
    uint32_t dm[SVO_DISC_E], dx[SVO_DISC_E];
    for (int i = 0; i < SVO_DISC_E; i++) { dm[i] = 0; dx[i] = 0; }
    for (int v = -15; v <= 15; v++) {
        const int um = svo_umax[v < 0 ? -v : v];
        for (int u = -um; u <= um; u++) {
            const int en = (v + 15) * 8 + (u + 15) / 4, by = (u + 15) & 3;
            dm[en] |= 1u << (8 * by); dx[en] |= (uint32_t)(u + 15) << (8 * by);
        }
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_disc_m), dm, sizeof(dm));
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_disc_x), dx, sizeof(dx));
}

// ------------------------------------------------------------------------------------------------------------
// begin_frame: publish the level-0 pointers, clear the per-frame counters, and apply the prev/cur shift with
// the reference's recovery rule (process_new_image_pair.cpp:86-100): the previous frame is replaced by the
// current one unless this is a repeat or the last call ended in voecBadTracking / voecBadCondNumber.
// ------------------------------------------------------------------------------------------------------------
struct ImgPtrs { const uint8_t* p[2 * SVO_MAX_LANES]; };

__global__ void k_begin_frame(DevCtx c, ImgPtrs ptrs, unsigned flags)
{
    SVO_TL_SCOPE(c, TL_BEGIN, (flags & SVO_RUN_DETECT) ? 0 : 1);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // The brute-force result words are atomicMin keys and start from all ones: filled here, ahead of the matchers of the same call
    // (it was a launch of its own; a kernel and not hipMemsetAsync because a captured byte memset of more than 64 KB came back
    // incomplete on graph replay: two lanes x two octaves x 2048 keypoints was the first configuration to cross that size, its pairings
    // went wrong and the stale indices walked k_track_filter out of its tables -- test_adaptive_nms_after_fast_orb_matches_oracle[True]).  The three regions per lane-octave (left/right, track left, track
    // right) are written by different launches, so one fill per call serves them all.
    if (flags & (SVO_RUN_MATCH | SVO_RUN_TRACK)) {
        const size_t n_words = (size_t)c.n_lanes * c.oct_cap * 3 * c.max_kps, n16 = n_words / 4;
        uint4* p16 = (uint4*)c.bf_idx;
        for (size_t i = (size_t)t; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if ((size_t)t < n_words - 4 * n16) c.bf_idx[4 * n16 + t] = -1;
    }
    const bool detect = flags & SVO_RUN_DETECT, do_shift = !(flags & SVO_FLAG_NO_SHIFT), repeat = flags & SVO_FLAG_REPEAT;
    // SVO_FLAG_DETECT_AHEAD: a detect call that may overlap stages 3-5 of the frame before it -- it initialises the detector's
    // per-image scratch and nothing else; its post call (SVO_RUN_DETECT_POST with the same flag) does the lane part below
    const bool ahead = flags & SVO_FLAG_DETECT_AHEAD, ahead_post = ahead && !detect;
    if (detect) {
        if (t < c.n_img) { c.img0[t] = ptrs.p[t]; c.raw_n[t] = 0; }
        if (t < c.n_img * SVO_MAX_LEVELS) {
            c.cand_cnt[t * SVO_CNT_STRIDE] = 0; c.lvl_n[t] = 0;
            // this frame's FAST threshold per (image, level): the speculated one when there is one (ORB mode only; debug
            // mode 12 switches the speculation off), never below the caller's threshold
            const uint32_t dyn = (c.fast_orb || c.debug_mode == 12) ? 0u : c.fast_th_dyn[t];
            c.fast_th_used[t] = max((uint32_t)c.fast_th, min(dyn, 250u));
            c.redo_flag[t] = 0;
        }
        if (t == 0) *c.redo_n = 0;
        if (ahead && t < c.n_lanes) c.det_status[t] = 0;
    }
    if (t < c.n_lanes && !(ahead && detect)) {
        LaneState& s = c.lane[t];
        if (do_shift) {
            if (!repeat && s.m_error != SVO_VOEC_BAD_TRACKING && s.m_error != SVO_VOEC_BAD_COND_NUMBER) {
                if (s.has_cur) { s.prev_slot = 1 - s.prev_slot; s.has_prev = 1; }     // m_prev_imgpair = m_current_imgpair (P:86-89)
            }
            s.m_error = SVO_VOEC_NONE;                                                 // P:95
            s.has_cur = 1;                                                             // P:100
            const int cur = 1 - s.prev_slot;
            for (int o = 0; o < c.oct_cap; o++) {
                const int vl = t * c.oct_cap + o;
                c.n_kps[feat_cnt_idx(vl, cur, 0)] = 0; c.n_kps[feat_cnt_idx(vl, cur, 1)] = 0;
                c.n_matches[vl * 2 + cur] = 0; c.n_ids[vl * 2 + cur] = 0;
            }
            if (!repeat) s.it_counter++;                                               // P:380-381
        }
        if (flags & SVO_RUN_TRACK) for (int o = 0; o < c.oct_cap; o++) c.n_tracked[t * c.oct_cap + o] = 0;
        const uint32_t st0 = ahead_post ? c.det_status[t] : 0u;                        // what the ahead half of this frame raised
        if (detect || ahead_post) c.status[t] = st0;
        svo_result& r = c.results[t];
        r.error_code = SVO_VOEC_NONE;                                                  // P:50
        r.valid = 0; r.num_it = 0; r.num_it_final = 0; r.n_outliers = 0; r.n_residual = 0; r.n_octaves = c.n_oct;
        if (detect || ahead_post) r.status = (int)st0;
        r.tracked_feats_from_last_frame = 0; r.tracked_feats_from_last_KF = 0;
        for (int k = 0; k < 8; k++) r.track_stats[k] = 0;
        for (int k = 0; k < 6; k++) { r.outPose[k] = 0; r.delta[k] = 0; }
    }
}

// getValues (H:704-724) packed for ONE transfer: header {n_left, n_right, n_matches, n_ids} at byte 0, then at byte 64 the
// six lists at max_kps-strided offsets (left kps | right kps | left desc | right desc | pairings | pairing IDs); the
// previous / current slot is resolved here, on the device, so that the host needs no round trip to learn it.
__global__ void __launch_bounds__(256) k_pack_values(DevCtx c, int lane, int which, int octave, uint8_t* dst)
{
    const LaneState& s = c.lane[lane];
    const int slot = which ? s.prev_slot : 1 - s.prev_slot, vl = lane * c.oct_cap + octave;
    const bool present = which ? s.has_prev != 0 : s.has_cur != 0;
    const int nl = present ? c.n_kps[feat_cnt_idx(vl, slot, 0)] : 0, nr = present ? c.n_kps[feat_cnt_idx(vl, slot, 1)] : 0;
    const int nm = present ? c.n_matches[vl * 2 + slot] : 0, ni = present ? c.n_ids[vl * 2 + slot] : 0;
    const size_t MK = (size_t)c.max_kps;
    uint4* out = (uint4*)(dst + 64);
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (t == 0) { int32_t* h = (int32_t*)dst; h[0] = nl; h[1] = nr; h[2] = nm; h[3] = ni; }
    // everything is copied as 4-byte words (28-byte keypoints are not 16-byte multiples)
    auto copy_words = [&](const void* src, size_t dst_off, size_t nbytes) {
        const uint32_t* sp = (const uint32_t*)src; uint32_t* dp = (uint32_t*)((uint8_t*)out + dst_off);
        for (size_t i = (size_t)t; i < nbytes / 4; i += (size_t)nt) dp[i] = sp[i];
    };
    copy_words(c.kps + feat_base(c, vl, slot, 0), 0, (size_t)nl * sizeof(svo_keypoint));
    copy_words(c.kps + feat_base(c, vl, slot, 1), MK * sizeof(svo_keypoint), (size_t)nr * sizeof(svo_keypoint));
    copy_words(c.desc + feat_base(c, vl, slot, 0) * 32, 2 * MK * sizeof(svo_keypoint), (size_t)nl * 32);
    copy_words(c.desc + feat_base(c, vl, slot, 1) * 32, 2 * MK * sizeof(svo_keypoint) + MK * 32, (size_t)nr * 32);
    copy_words(c.matches + match_base(c, vl, slot), 2 * MK * (sizeof(svo_keypoint) + 32), (size_t)nm * sizeof(svo_dmatch));
    copy_words(c.ids + match_base(c, vl, slot), 2 * MK * (sizeof(svo_keypoint) + 32) + MK * sizeof(svo_dmatch), (size_t)ni * 4);
}
void launch_pack_values(const DevCtx& c, int lane, int which, int octave, uint8_t* dst, hipStream_t st)
{
    hipLaunchKernelGGL(k_pack_values, dim3(64), dim3(256), 0, st, c, lane, which, octave, dst);
}

// ------------------------------------------------------------------------------------------------------------
// K1: one pyramid level from the previous one: cv::resize's 8-bit INTER_LINEAR (oracle v7) with host-built 11-bit tap tables.
// HBM-bound on paper (~1.44 source bytes read + 1 written per output pixel), VALU-issue-bound in practice, so the
// blend is written for instruction count.  A 256-thread block produces a 128x32 destination tile: the <= 160x41
// source window is staged in LDS by LDS-DMA (global_load_lds_dwordx4: 42 rows x 11 chunks of 16 bytes, two wave-instructions
// per wave, no VGPR round trip), the tile's slice of the x / y tables sits in LDS too, every thread blends 4 rows x 4 adjacent pixels and stores one dword per row.
// Per pixel: the two taps of each source row come out of two aligned LDS dwords as a u16 pair by one v_perm (the
// selector is a per-column constant), v_dot2_u32_u16 applies the column's (a0, a1), the two row sums are cut, weighted, cut and rounded
// as OpenCV's uchar VResizeLinear does, and shifts + ors pack four results.
// (rounds 1-5: the two row sums went through two 24-bit multiply-adds with the row weights pre-scaled by 4, so that the ONCE-rounded result
// landed in byte 3.  Oracle v7 follows cv::resize's 8-bit path instead: ((b0 * (top >> 4)) >> 16) + ((b1 * (bot >> 4)) >> 16) + 2) >> 2,
// weights as OpenCV derives them in float -- four more VALU operations per pixel in a kernel that waits for memory.)
// ------------------------------------------------------------------------------------------------------------
// (Round 5 rebuilt this kernel twice for instruction count -- a wave owning eight rows and a lane two columns, the horizontal blend of a
// source row computed once and shared by the rows above and below it, row tables by v_readlane: 232 -> ~150 VALU instructions per wave,
// pyramid byte-identical -- and measured both forms SLOWER: 28.8 / 31.6 us per launch against 27.7, 69.7 k / 67.5 k pairs/s against
// 70.9 k / 69.8 k in the same gpurun call.  The kernel runs at 4.2 TB/s of combined traffic, i.e. it waits for memory, not for issue
// slots; what the rewrite changed was how many independent row chains a wave has in flight (four here, one there).  Kept as it was.)
#define RZ_W 128
#define RZ_H 32
#define RZ_SP 176     // LDS window pitch in bytes: 11 DMA chunks of 16 (>= 128 * 1.2 + 2 + 15, the window origin is 16-byte aligned in x)
#define RZ_SH 41      // >= 32 * 1.2 + 2
#define RZ_CHUNK 16   // consecutive tiles per XCD turn

typedef unsigned short rz_u16x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_resize(DevCtx c, int level, FastDiv div_img, FastDiv div_ntx)
{
    SVO_TL_SCOPE(c, TL_RESIZE, level);
    __shared__ __attribute__((aligned(16))) uint32_t win[(RZ_SH + 1) * (RZ_SP / 4)];
    __shared__ uint32_t xw[RZ_W], xr[RZ_W], yw[RZ_H], yr[RZ_H];
    const int tid = threadIdx.x;
    const LevelGeom& d = c.lv[level];
    const LevelGeom& s = c.lv[level - 1];
    // XCD-aware tile order (see k_fast): chunks of RZ_CHUNK consecutive tiles of the image-major tile list per XCD turn.
    // A 160-byte source row touches two or three 128-byte L2 lines, shared with the tiles left and right: through
    // one L2 they are fetched once, through eight they were fetched twice (measured 940 MB -> for 475 MB of source).
    const int ntx = (d.w + RZ_W - 1) / RZ_W, nty = (d.h + RZ_H - 1) / RZ_H, per_img = ntx * nty;
    const uint32_t slot = blockIdx.x >> 3;
    const uint32_t work = c.debug_mode == 8 ? blockIdx.x : (((slot / RZ_CHUNK) * 8 + (blockIdx.x & 7)) * RZ_CHUNK + slot % RZ_CHUNK);
    const int img = (int)fastdiv(work, div_img);
    if (img >= c.n_img) return;
    const int tt = (int)work - img * per_img, tby = (int)fastdiv((uint32_t)tt, div_ntx), tbx = tt - tby * ntx;
    const int dx0 = tbx * RZ_W, dy0 = tby * RZ_H;
    int spitch; const uint8_t* src = level_ptr(c, img, level - 1, spitch);
    uint8_t* dst = c.pyr + (long long)img * c.pyr_bytes + d.offset;
    const int* xi = c.rtab + d.rtab_off, *xf = xi + d.w, *yi = xf + d.w, *yf = yi + d.h;
    const int sx0 = xi[dx0] & ~15, sy0 = yi[dy0];                      // window origin (block-uniform)
    if (tid < RZ_W) {
        const int x = min(dx0 + tid, d.w - 1), rx = xi[x] - sx0;
        xw[tid] = (uint32_t)xf[x];                                                                 // dot2 weights: a0 | a1 << 16
        xr[tid] = (uint32_t)rx;
    } else if (tid < RZ_W + RZ_H) {
        const int y = min(dy0 + tid - RZ_W, d.h - 1);
        yw[tid - RZ_W] = (uint32_t)yf[y];                                                          // b0 | b1 << 16
        yr[tid - RZ_W] = (uint32_t)(yi[y] - sy0);
    }
    if (c.debug_mode == 5) { /* ablation: no staging */ }
    else if ((spitch & 15) == 0) {
        // LDS-DMA: chunk i = (row i / 11, piece i % 11) lands at LDS byte 16 i.  The source base may sit at any byte alignment
        // (tools/ubench/glds_align.hip); with the origin 16-aligned in x and a pitch that is a multiple of 16 no chunk
        // straddles the end of a row, so the clamp to the last chunk of the row only ever moves chunks that lie wholly
        // beyond it (never read: the taps stop at column s.w - 1).
        typedef const void __attribute__((address_space(1)))* gptr_t;
        typedef void __attribute__((address_space(3)))* lptr_t;
        const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
        auto chunk_src = [&](int i) -> const uint8_t* {
            const int r = (i * 745) >> 13, q = i - 11 * r;                  // i / 11, i % 11 for i < 512
            const int yy = min(sy0 + r, s.h - 1), xx = min(sx0 + 16 * q, spitch - 16);
            return src + (uint32_t)(yy * spitch + xx);
        };
        uint8_t* wb = (uint8_t*)win;
        __builtin_amdgcn_global_load_lds((gptr_t)chunk_src(tid), (lptr_t)(wb + wid * 1024), 16, 0, 0);
        if (tid + 256 < (RZ_SH + 1) * (RZ_SP / 16)) __builtin_amdgcn_global_load_lds((gptr_t)chunk_src(tid + 256), (lptr_t)(wb + 4096 + wid * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        uint8_t* wb = (uint8_t*)win;
        for (int i = tid; i < RZ_SH * RZ_SP; i += 256) {
            const int r = i / RZ_SP, q = i - r * RZ_SP;
            wb[i] = src[(long long)min(sy0 + r, s.h - 1) * spitch + min(sx0 + q, s.w - 1)];
        }
    }
    __syncthreads();
    const int gx = (tid & 31) * 4, x4 = dx0 + gx;                       // 4 adjacent pixels per row, 4 rows per thread
    if (x4 >= d.w || c.debug_mode == 6) return;
    // The taps of the thread's 4 adjacent pixels lie within 8 source bytes (3 x 1.2 px apart + the second tap): per source
    // row ONE 12-byte fetch from the aligned dword of the first tap, funnel-shifted to an 8-byte window starting at that
    // tap, serves all four -- LDS reads were what bounded this kernel (16 dword gathers per pixel quad before).
    uint32_t wx[4], sel[4];
    const uint32_t rx0 = xr[gx];
    const int wi0 = (int)(rx0 >> 2);
    const uint32_t sh0 = rx0 & 3u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = xr[gx + k] - rx0;                            // 0 .. 5 (< 7: both taps inside the window)
        wx[k] = xw[gx + k];
        sel[k] = 0x0c010c00u + o * 0x00010001u;                         // bytes (o, o + 1) of the 8-byte window -> u16 pair
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int gy = (tid >> 5) + 8 * j, y = dy0 + gy;
        if (y >= d.h) continue;
        const uint32_t wy = yw[gy];
        const uint32_t wy0 = wy & 0xFFFFu, wy1 = wy >> 16;
        const uint32_t* r0 = win + yr[gy] * (RZ_SP / 4) + wi0, *r1 = r0 + RZ_SP / 4;
        const uint32_t a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
        const uint32_t w0lo = __builtin_amdgcn_alignbyte(a1, a0, sh0), w0hi = __builtin_amdgcn_alignbyte(a2, a1, sh0);
        const uint32_t w1lo = __builtin_amdgcn_alignbyte(b1, b0, sh0), w1hi = __builtin_amdgcn_alignbyte(b2, b1, sh0);
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t t0 = __builtin_amdgcn_perm(w0hi, w0lo, sel[k]);
            const uint32_t t1 = __builtin_amdgcn_perm(w1hi, w1lo, sel[k]);
            const uint32_t top = __builtin_amdgcn_udot2(__builtin_bit_cast(rz_u16x2, t0), __builtin_bit_cast(rz_u16x2, wx[k]), 0u, false);
            const uint32_t bot = __builtin_amdgcn_udot2(__builtin_bit_cast(rz_u16x2, t1), __builtin_bit_cast(rz_u16x2, wx[k]), 0u, false);
            // cv::resize's uchar VResizeLinear (oracle v7): the row sums lose 4 bits, each product is cut to quarter grey levels, then one rounding:
            // ((b0 * (top >> 4)) >> 16) + ((b1 * (bot >> 4)) >> 16) + 2.  The two high halves come out of ONE v_perm, their sum + 2 out of one v_sad_u16
            const uint32_t m0 = __umul24(top >> 4, wy0), m1 = __umul24(bot >> 4, wy1);
            v[k] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_perm(m1, m0, 0x07060302u), 0u, 2u);       // <= 1022; >> 2 below, two at a time
        }
        const rz_u16x2 p01 = __builtin_bit_cast(rz_u16x2, v[0] | (v[1] << 16)) >> 2, p23 = __builtin_bit_cast(rz_u16x2, v[2] | (v[3] << 16)) >> 2;
        const uint32_t out = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
        if (c.debug_mode == 7 && out != 0x12345678u) continue;
        *(uint32_t*)(dst + (long long)y * d.pitch + x4) = out;   // pitch is a multiple of 64: the tail of the last dword is padding
    }
}

// LDS-DMA written as asm: 16 bytes per lane from the lane's global address to LDS byte `lds_base` + 16 * lane.  Through the
// builtin the compiler cannot tell which LDS bytes a transfer writes and puts s_waitcnt vmcnt(0) before EVERY later LDS access,
// i.e. a window in flight for the NEXT keypoint would be waited for at once.  As asm the transfer is invisible to that logic; the
// kernel waits for it itself (s_waitcnt vmcnt(0) at the top of the keypoint loop).  The compiler's own vmcnt arithmetic stays
// safe: these transfers only add to the outstanding count, so its waits can only get stricter.
__device__ __forceinline__ void glds16_asm(const uint8_t* gaddr, uint32_t lds_base)
{
    // m0 is a reserved register for hipcc: a clobber on it is ignored (with a warning), so the asm saves and restores it itself and the
    // compiler's view "m0 unchanged" stays true whatever else in the kernel uses it (the builtin LDS-DMA form, LDS-direct, s_movrel)
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_saved) : "v"(gaddr), "s"(lds_base) : "memory");
}

// ------------------------------------------------------------------------------------------------------------
// K2: FAST-9/16 score + 3x3 non-max suppression, all levels of all images in one launch.
// Integer VALU issue bounds this kernel (97-99 % of the issue slots, profiles/r03*_pmc.json), so the design minimises
// instructions per position first and a tile's footprint in wave slots and LDS second (fast_tile below: two waves per tile,
// 12 KB of LDS, geometry from a table).
// Tile = 62x62 interior pixels (FT_W x FT_H); with the 1-pixel NMS halo the score window is 64x64, one byte per position,
// entry = r * 64 + q.  The 80x70 source window (3 px circle radius around the score window, x origin x0 - 5 so that groups of
// four positions sit on LDS dword boundaries) is staged by LDS-DMA.  A three-step cascade keeps the expensive work dense:
//   (1) every position of the score window takes the 4-pixel cardinal test -- a 9-arc on the 16-circle always
//       contains two ADJACENT compass points, so a corner needs two adjacent compass pixels both brighter than
//       c+t or both darker than c-t.  It runs on FOUR positions per lane-op: the centre row / N / S come in as
//       aligned LDS dwords, E / W by v_alignbyte, and the comparisons are saturating packed-16-bit subtractions
//       ("nonzero" = true, AND = pk_min, OR = bitwise or).  A thread owns one group column and EIGHT CONSECUTIVE ROWS
//       (16 group columns x 8 row blocks = 128 threads): the centre dword of a row is the N operand of the row three below
//       and the S operand of the row three above, so 14 centre loads (and their 14 shifted copies) serve 8 rows, and the
//       whole index arithmetic of the phase is one base address.  Survivors are compacted into an LDS list by one scan;
//   (2) only the listed positions compute the arc-min score (both polarities through one packed min network);
//   (3) the 3x3 NMS also walks the list; survivors are appended to the level's candidate list with one global
//       atomic per tile, issued by a single wave.
// (Rounds 2-3 ran 64x56 tiles with 17 groups x 58 rows dealt to the threads as 493 row pairs: ~70 instructions of index
// arithmetic, bounds checks and address set-up per thread went with that, and 14 per listed survivor.)
// No position is bounds-checked before step (3): the staged window only ever holds readable memory, and a position
// outside [EDGE-1, dim-EDGE] is never a neighbour of an interior position, so whatever score it gets is never read.
// ------------------------------------------------------------------------------------------------------------
#define FT_W SVO_FT_W
#define FT_H SVO_FT_H
#define FT_LW 80              // LDS window pitch; window x origin = x0 - 5: position q sits at byte q + 4
#define FT_LH (FT_H + 8)      // window y origin = y0 - 4: position row r sits at window row r + 3
#define FT_SW 64              // score window: x = x0 - 1 + q, q in [0, 64); interior q in [1, 63)
#define FT_SH 64              // y = y0 - 1 + r, r in [0, 64); interior r in [1, 63)
#define FT_SP 64              // score map pitch: entry = r * FT_SP + q
#define FT_NG (FT_SW / 4)     // 16 groups of four positions per row
#define FT_CHUNK 32            // consecutive tiles per XCD turn
static_assert(FT_W + 2 == FT_SW && FT_H + 2 == FT_SH && FT_NG == 16, "k_fast's thread mapping is written for a 64x64 score window (16 group columns x 8 blocks of 8 rows)");

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// cardinal-pair test of two positions held in the 16-bit halves of each operand (either as v or as v << 8: the
// caller scales the threshold the same way).  A half of the result is nonzero when its position passes.
__device__ __forceinline__ uint32_t quick_half(uint32_t c, uint32_t n, uint32_t e, uint32_t s, uint32_t w, u16x2 t2)
{
    const u16x2 cc = as_u16x2(c), hi = __builtin_elementwise_add_sat(cc, t2), lo = __builtin_elementwise_sub_sat(cc, t2);
    const u16x2 N = as_u16x2(n), E = as_u16x2(e), S = as_u16x2(s), W = as_u16x2(w);
    // "two adjacent compass points bright" == (N or S bright) and (E or W bright): any N/S point is adjacent to any E/W point
    //   bright  <=>  min(max(N, S), max(E, W)) > c + t        dark  <=>  max(min(N, S), min(E, W)) < c - t
    const u16x2 b = __builtin_elementwise_min(__builtin_elementwise_max(N, S), __builtin_elementwise_max(E, W));
    const u16x2 d = __builtin_elementwise_max(__builtin_elementwise_min(N, S), __builtin_elementwise_min(E, W));
    return as_u32(__builtin_elementwise_sub_sat(b, hi) | __builtin_elementwise_sub_sat(lo, d));
}

typedef short i16x2 __attribute__((ext_vector_type(2)));

// FAST-9 score of the pixel at byte offset `off` of the LDS window: the largest t for which it is still a corner
// = max over the 16 arcs of 9 contiguous circle pixels of the min one-sided difference, minus 1; 0 when that is
// not above th.  Both polarities ride in the two halves of one register, v[i] = (p_i - c, c - p_i), so one min
// network serves both.  The 16 circular 9-windows come from block prefix/suffix minima (van Herk): with the circle
// cut into [0,8) and [8,16), window [i, i+8] = suffix_i of one block + prefix_i of the other: 28 + 16 + 15 packed ops.
__device__ __forceinline__ int fast_score_lds(const uint8_t* win, int off, int th)
{
    // immediate offsets from one base; the middle byte of the two 3-byte runs (rows -3 and +3) goes through a second,
    // opaque copy of the base: left alone the compiler fuses neighbouring bytes into ds_read_u16 at odd addresses,
    // which the LDS replays lane by lane
    const int base = off - 3 * FT_LW - 3;
    int basem = base; asm volatile("" : "+v"(basem));
#define FO(dx, dy) (((dy) + 3) * FT_LW + (dx) + 3)
    const int c = win[base + FO(0, 0)];
    int p[16];
    p[0] = win[basem + FO(0, -3)]; p[1] = win[base + FO(1, -3)]; p[2] = win[base + FO(2, -2)]; p[3] = win[base + FO(3, -1)];
    p[4] = win[base + FO(3, 0)];   p[5] = win[base + FO(3, 1)];  p[6] = win[base + FO(2, 2)];  p[7] = win[base + FO(1, 3)];
    p[8] = win[basem + FO(0, 3)];  p[9] = win[base + FO(-1, 3)]; p[10] = win[base + FO(-2, 2)]; p[11] = win[base + FO(-3, 1)];
    p[12] = win[base + FO(-3, 0)]; p[13] = win[base + FO(-3, -1)]; p[14] = win[base + FO(-2, -2)]; p[15] = win[base + FO(-1, -3)];
#undef FO
    const i16x2 K = { (short)(-c), (short)c };
    i16x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = __builtin_bit_cast(i16x2, __mul24(p[i], -65535)) + K;      // (p, -p) + (-c, c)
    i16x2 s0[8], s1[8], p0[8], p1[8];       // suffix / prefix minima of the blocks [0,8) and [8,16)
    s0[7] = v[7]; s1[7] = v[15]; p0[0] = v[0]; p1[0] = v[8];
#pragma unroll
    for (int j = 6; j >= 0; j--) { s0[j] = __builtin_elementwise_min(s0[j + 1], v[j]); s1[j] = __builtin_elementwise_min(s1[j + 1], v[8 + j]); }
#pragma unroll
    for (int j = 1; j < 8; j++) { p0[j] = __builtin_elementwise_min(p0[j - 1], v[j]); p1[j] = __builtin_elementwise_min(p1[j - 1], v[8 + j]); }
    i16x2 m = __builtin_elementwise_min(s0[0], p1[0]);                                  // window [0, 8]
#pragma unroll
    for (int i = 1; i < 8; i++) m = __builtin_elementwise_max(m, __builtin_elementwise_min(s0[i], p1[i]));      // [i, i+8]
#pragma unroll
    for (int i = 0; i < 8; i++) m = __builtin_elementwise_max(m, __builtin_elementwise_min(s1[i], p0[i]));      // [8+i, 16+i]
    const int best = max((int)m.x, (int)m.y);
    return best > th ? best - 1 : 0;
}

#define FT_NT 128
#define FT_ROWS 8                                    // consecutive score rows per thread
#define FT_LIST_CAP 512                              // LDS list of the cardinal test's survivors (~80 per tile on a textured scene); 10.8 KB per tile = 15 tiles per CU
#define FT_CHUNKS (FT_LH * 5)                        // 16-byte DMA chunks of the window
static_assert(FT_NG * (FT_SH / FT_ROWS) == FT_NT && FT_ROWS * 4 <= 32, "one thread per (group column, block of rows); a thread's verdicts must fit one 32-bit mask");
static_assert(FT_SH * FT_SP <= 65536, "list entries are 16 bits");

struct FastSmem {
    __attribute__((aligned(16))) uint8_t tile[FT_LH * FT_LW];
    __attribute__((aligned(16))) uint8_t score[FT_SH * FT_SP + 80];      // + a row: the 3x3 reads around a position of the last (halo) row
    unsigned short list[FT_LIST_CAP];
    unsigned s_count, s_nout;
};
// the NMS survivors' keys (3x3 NMS leaves at most one per 2x2 block) go where the window was: it is dead once the scores exist
static_assert((FT_W * FT_H / 4 + 64) * 4 <= FT_LH * FT_LW, "out_keys aliases the window");

// one 62x62 tile of one level of one image with FAST threshold th_fast (>= the caller's threshold, see k_select)
// (src, pitch, gw, gh: the level's image; x0, y0: the tile's interior origin -- the caller has them from the tile table)
// FT_NT = 128 threads per tile, two waves: the kernel is bound by VALU issue (SQ_INSTS_VALU x 4 cycles = 97 % of its duration,
// profiles/r03_pmc.json), and what does not scale with the pixels -- staging, the fixed part of the compaction, the
// publication -- is paid per tile.  The list of the cardinal test's survivors is capped (LDS per tile decides how many tiles a
// CU holds); what does not fit stays with the thread that found it and is scored / suppressed by its owner after the listed ones.
// the window [x0-5, x0+75) x [y0-4, y0+FT_H+4) by LDS-DMA (global_load_lds_dwordx4): a wave-instruction lands 64 x 16 bytes at LDS
// base + lane * 16 straight from the lanes' global addresses, no VGPR round trip and no ds_write.  The window is FT_LH rows x 5
// chunks of 16 bytes, chunk i at LDS byte 16 i (pitch 80): wave w takes chunks 128 j + 64 w + lane.  The source may sit at ANY byte
// alignment and the destination base at any dword (tools/ubench/glds_align.hip pins both on the hardware), so one path serves every
// pointer / stride.  Chunks are clamped to the image proper (x <= gw - 16, y <= gh - 1): a clamped chunk holds shifted bytes, but only
// columns >= gw - 16 can be affected and nothing right of column gw - 28 is ever read by an interior position (EDGE 31 - radius 3 -
// NMS halo 1).  ASM: the transfer as inline asm (glds16_asm), invisible to the compiler's LDS ordering: the persistent kernel keeps
// the NEXT tile's window in flight and waits for it itself.
template <bool ASM>
__device__ __forceinline__ void fast_stage(uint8_t* tile, const uint8_t* src, int pitch, int gw, int gh, int x0, int y0)
{
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto chunk_src = [&](int i) -> const uint8_t* {
        const int r = (i * 205) >> 10, q = i - 5 * r;                   // i / 5, i % 5 for i < 1024
        const int yy = min(y0 - 4 + r, gh - 1), xx = min(x0 - 5 + 16 * q, gw - 16);
        return src + (uint32_t)(yy * pitch + xx);                       // 32-bit offset from the level's uniform base
    };
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    const uint32_t lb = ASM ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)tile) : 0u;
#pragma unroll
    for (int j = 0; j < (FT_CHUNKS + FT_NT - 1) / FT_NT; j++) {
        if (j * FT_NT + FT_NT <= FT_CHUNKS || j * FT_NT + tid < FT_CHUNKS) {
            if (ASM) glds16_asm(chunk_src(j * FT_NT + tid), lb + (uint32_t)(j * (FT_NT * 16) + wid * 1024));
            else __builtin_amdgcn_global_load_lds((gptr_t)chunk_src(j * FT_NT + tid), (lptr_t)(tile + j * (FT_NT * 16) + wid * 1024), 16, 0, 0);
        }
    }
}

// counters and score map of a tile: cleared BEFORE its window is waited for (through the builtin the compiler orders every LDS
// store behind an outstanding LDS-DMA, so a store placed after the issue would wait out the whole fetch)
__device__ __forceinline__ void fast_reset(uint8_t* score, unsigned& s_count, unsigned& s_nout)
{
    const int tid = threadIdx.x;
    if (tid == 0) { s_nout = 0; s_count = 0; }
    constexpr int N16 = (FT_SH * FT_SP + 15) / 16;
#pragma unroll
    for (int i = 0; i < (N16 + FT_NT - 1) / FT_NT; i++) if (i * FT_NT + tid < N16) ((uint4*)score)[i * FT_NT + tid] = make_uint4(0, 0, 0, 0);
}

__device__ __forceinline__ void fast_compute(const DevCtx& c, uint8_t* tile, uint8_t* score, unsigned short* list, unsigned& s_count, unsigned& s_nout,
                                             int img, int level, int gw, int gh, int x0, int y0, int th_fast);

__device__ __forceinline__ void fast_tile(const DevCtx& c, FastSmem& sm, int img, int level, const uint8_t* src, int pitch, int gw, int gh, int x0, int y0, int th_fast)
{
    fast_reset(sm.score, sm.s_count, sm.s_nout);
    fast_stage<false>(sm.tile, src, pitch, gw, gh, x0, y0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's DMA chunks have landed; the barrier covers the other wave's
    __syncthreads();
    fast_compute(c, sm.tile, sm.score, sm.list, sm.s_count, sm.s_nout, img, level, gw, gh, x0, y0, th_fast);
}

// everything of a tile after its window is in LDS (block-uniform early exits only; the caller's next barrier is its own)
__device__ __forceinline__ void fast_compute(const DevCtx& c, uint8_t* tile, uint8_t* score, unsigned short* list, unsigned& s_count, unsigned& s_nout,
                                             int img, int level, int gw, int gh, int x0, int y0, int th_fast)
{
    uint32_t* out_keys = (uint32_t*)tile;
    const int tid = threadIdx.x;
    if (c.debug_mode == 1) return;
    // ---- (1) packed cardinal test: thread -> group column gq (positions q = 4 gq .. 4 gq + 3), score rows 8 rb .. 8 rb + 7 ----
    const uint32_t th = (uint32_t)th_fast;
    const u16x2 t2h = { (unsigned short)(th << 8), (unsigned short)(th << 8) };
    const int gq = tid & (FT_NG - 1), rb = tid >> 4;
    const int e_base = rb * (FT_ROWS * FT_SP) + 4 * gq;                     // entry of (row 8 rb, position 4 gq)
    uint32_t pe[FT_ROWS], po[FT_ROWS];                                      // nonzero halves = passing positions
    {
        // window row of score row r: r + 3; its N / S operands: window rows r, r + 6.  col[20 i + 1] = centre dword of window
        // row 8 rb + i (bytes 4 gq + 4 ..: the group's four positions), col[20 i] / col[20 i + 2] = the dwords left / right of it
        // (Bank conflicts: the row blocks are 8 x 20 = 160 dwords apart, 0 modulo the LDS's 32 banks, so the two row blocks of a half-wave
        // read the same 16 banks in every one of these loads -- a systematic 2-way conflict, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42.
        // The pitch cannot be padded away: the LDS-DMA lands 16-byte chunks back to back, and 8 rows x any multiple of 4 dwords is 0 modulo 32.
        // What it costs: DESIGN.md section 4, round 6 -- 19.9 M cycles of LDS instructions against 143 M of VALU per launch.)
        const uint32_t* col = (const uint32_t*)tile + rb * (FT_ROWS * (FT_LW / 4)) + gq;
        uint32_t cc[FT_ROWS + 6];
#pragma unroll
        for (int i = 0; i < FT_ROWS + 6; i++) cc[i] = col[i * (FT_LW / 4) + 1];
#pragma unroll
        for (int g = 0; g < FT_ROWS; g++) {
            const uint32_t Cp = col[(g + 3) * (FT_LW / 4)], C = cc[g + 3], Cn = col[(g + 3) * (FT_LW / 4) + 2], N = cc[g], S = cc[g + 6];
            // The test compares 16-bit halves whose HIGH byte is the pixel of interest; the low byte is whatever sits next to it
            // and only matters when the high bytes tie at exactly the threshold, where it can let a position through that the
            // exact score then zeroes again (a false alarm, never a miss: 16-bit max / min order by the high byte first).  So no
            // masking: positions 1, 3 are the high bytes of the registers as loaded, positions 0, 2 those of the same registers
            // one byte further left -- E / W come out of the funnel shifter in that alignment directly (W is simply Cp).
            const uint32_t Eo = __builtin_amdgcn_alignbyte(Cn, C, 3), Wo = __builtin_amdgcn_alignbyte(C, Cp, 1);
            const uint32_t Ee = __builtin_amdgcn_alignbyte(Cn, C, 2);
            pe[g] = quick_half(C << 8, N << 8, Ee, S << 8, Cp, t2h);         // positions 0 (low half), 2 (high half)
            po[g] = quick_half(C, N, Eo, S, Wo, t2h);                        // positions 1, 3
        }
    }
    // compaction: the thread's verdicts become one bit mask (bit 8 p + g: position p of row g), a DPP scan of
    // the popcounts gives every lane its first list slot, ONE LDS atomic per wave reserves the wave's range, and each lane
    // writes its own survivors.  (Ballots with their mbcnt pairs and conditional stores cost ~100 VALU instructions
    // per wave whether the tile held one candidate or fifty; this costs ~35 plus ~7 per survivor of the busiest lane, and
    // at the speculated thresholds 2 % of the positions survive.)  List order is irrelevant.  Entry = r * 64 + q.
    // Survivors beyond FT_LIST_CAP stay in m_left: their owner scores and suppresses them after the listed ones.
    auto entry_of = [&](int b) -> int { return e_base + ((b & 7) << 6) + (b >> 3); };
    uint32_t m_left = 0;
    {
        // "half nonzero" -> 0 / 1 by ONE v_pk_min_u16 per register, written as asm: left to itself hipcc turns min(x, 1) on
        // packed halves into two 16-bit compares, two selects and a v_perm (80 instructions per thread instead of 16)
        const uint32_t one2 = 0x00010001u;
        uint32_t me = 0, mo = 0;
#pragma unroll
        for (int g = 0; g < FT_ROWS; g++) {
            uint32_t fe, fo;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(fe) : "v"(pe[g]), "s"(one2));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(fo) : "v"(po[g]), "s"(one2));
            me |= fe << g; mo |= fo << g;                                  // positions 0, 2 from pe's halves; 1, 3 from po's
        }
        uint32_t m = me | (mo << 8);
        const int cnt = __popc(m), inc = wave_inclusive_scan(cnt);
        const int total = __builtin_amdgcn_readlane(inc, 63);
        if (total) {
            unsigned wbase = 0;
            if ((tid & 63) == 0) wbase = atomicAdd(&s_count, (unsigned)total);
            unsigned idx = (unsigned)__builtin_amdgcn_readfirstlane((int)wbase) + (unsigned)(inc - cnt);
            while (m && idx < FT_LIST_CAP) {
                const int b = __builtin_ctz(m); m &= m - 1;
                list[idx++] = (unsigned short)entry_of(b);
            }
            m_left = m;
        }
    }
    __syncthreads();
    const int ns_all = (int)s_count, ns = min(ns_all, FT_LIST_CAP);
    const bool overflow = ns_all > FT_LIST_CAP;                              // block-uniform
    if (c.debug_mode == 2) return;
    // ---- (2) score on the survivors only ----
    auto win_off = [&](int e) -> int { return (e >> 6) * FT_LW + (e & 63) + (3 * FT_LW + 4); };
    for (int i = tid; i < ns; i += FT_NT) {
        const int e = list[i];
        score[e] = (uint8_t)fast_score_lds(tile, win_off(e), th_fast);
    }
    if (__builtin_expect(overflow, 0)) {
        for (uint32_t m = m_left; m; m &= m - 1) {
            const int e = entry_of(__builtin_ctz(m));
            score[e] = (uint8_t)fast_score_lds(tile, win_off(e), th_fast);
        }
    }
    __syncthreads();
    if (c.debug_mode == 3) return;
    // ---- (3) 3x3 NMS on the listed interior positions; one global atomic per tile ----
    auto nms_round = [&](bool has, int e) {
        bool keep = false; uint32_t key = 0;
        if (has) {
            const int r = e >> 6, q = e & 63;
            // nine byte reads at immediate offsets, issued together; the middle column goes through an opaque copy of
            // the base so that no two fuse into a misaligned ds_read_u16.  (r = 0 reads below the map: rejected below.)
            const int nb = e - FT_SP - 1; int nbm = nb; asm volatile("" : "+v"(nbm));
            const int v = score[nbm + FT_SP + 1];
            const int n0 = score[nb], n1 = score[nbm + 1], n2 = score[nb + 2], n3 = score[nb + FT_SP], n4 = score[nb + FT_SP + 2];
            const int n5 = score[nb + 2 * FT_SP], n6 = score[nbm + 2 * FT_SP + 1], n7 = score[nb + 2 * FT_SP + 2];
            const int mx = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            const int x = x0 - 1 + q, y = y0 - 1 + r;
            keep = (v > mx) & (r >= 1) & (r <= FT_H) & (q >= 1) & (q <= FT_W) & (x < gw - SVO_EDGE) & (y < gh - SVO_EDGE);
            key = ((uint32_t)v << 24) | (0xFFFFFFu - (uint32_t)(y * gw + x));
        }
        const unsigned long long m = __ballot(keep);
        if (m) {
            unsigned b2 = 0;
            const int leader = __ffsll((long long)m) - 1;
            if ((tid & 63) == leader) b2 = atomicAdd(&s_nout, (unsigned)__popcll(m));
            b2 = __shfl(b2, leader, 64);
            if (keep) out_keys[b2 + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = key;
        }
    };
    // out_keys aliases the window, which the overflow path still scores from: every score is in the map by now (barrier above)
    for (int base = 0; base < ns; base += FT_NT) {
        const int i = base + tid;
        nms_round(i < ns, i < ns ? (int)list[i] : 0);
    }
    if (__builtin_expect(overflow, 0)) {
        uint32_t m = m_left;
        while (__ballot(m != 0)) {                                           // wave-uniform trip count: the ballots inside need every lane
            const bool has = m != 0;
            const int e = has ? entry_of(__builtin_ctz(m)) : 0;
            m &= m - 1;
            nms_round(has, e);
        }
    }
    __syncthreads();
    // only the first wave publishes: the other retires here, so the returning global atomic (a ~2-3 us round trip)
    // stalls one wave per tile instead of the whole workgroup
    if (tid < 64) {
        const unsigned nout = s_nout;
        if (nout != 0 && c.debug_mode != 4) {
            const LevelGeom& g = c.lv[level];
            unsigned gbase = 0;
            if (tid == 0) gbase = atomicAdd(&c.cand_cnt[(img * SVO_MAX_LEVELS + level) * SVO_CNT_STRIDE], nout);
            gbase = __shfl(gbase, 0, 64);
            uint32_t* dst = c.cand_keys + (long long)img * c.cand_total + g.cand_off;
            for (unsigned i = tid; i < nout; i += 64) {
                if (gbase + i < (unsigned)g.cand_cap) dst[gbase + i] = out_keys[i];
                else raise_detect_status(c, img >> 1, SVO_ST_CAND_OVERFLOW);
            }
        }
    }
}

__global__ void __launch_bounds__(FT_NT) __attribute__((amdgpu_waves_per_eu(8, 8))) k_fast(DevCtx c)
{
    SVO_TL_SCOPE(c, TL_FAST, 0);
    __shared__ FastSmem sm;
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2).  The global work list
    // (image-major, then the image's tiles of all levels) is cut into chunks of FT_CHUNK consecutive tiles and chunk k
    // goes to XCD k % 8, so the halo columns / rows that neighbouring tiles share are re-read from that XCD's L2 rather
    // than through another one.  (One contiguous eighth of every image per XCD shares more, but gives each XCD a
    // fixed set of pyramid levels -- their corner densities differ and the launch waits for the slowest XCD: measured.)
    //
    // A CU holds 8 of these workgroups and a workgroup lives ~3 us, so what a wave does BEFORE its loads are in flight is paid
    // in throughput: working the tile's level, origin and source address out of the kernel arguments was a chain of
    // six dependent scalar-load waits and a division.  The geometry of tile k is the same for every image and every frame:
    // it comes from a table (svo_api.hip fills it with the level geometry), next to the image's eight thresholds and its
    // level-0 pointer -- one round of scalar loads after the kernel arguments.
    const uint32_t slot = blockIdx.x >> 3;
    const uint32_t work = c.debug_mode == 8 ? blockIdx.x : (((slot / FT_CHUNK) * 8 + (blockIdx.x & 7)) * FT_CHUNK + slot % FT_CHUNK);
    const int img = (int)fastdiv(work, c.div_tiles), tile_id = (int)work - img * c.n_tiles;
    if (img >= c.n_img) return;
    const uint4 e = c.fast_tiles[tile_id];                  // x0 | y0 << 16, w | h << 16, level | pitch << 8, level offset in the pyramid
    const uint4 tha = ((const uint4*)(c.fast_th_used + img * SVO_MAX_LEVELS))[0], thb = ((const uint4*)(c.fast_th_used + img * SVO_MAX_LEVELS))[1];
    const uint8_t* base0 = c.img0[img];
    const int level = (int)(e.z & 0xFFu);
    const uint32_t thv[8] = { tha.x, tha.y, tha.z, tha.w, thb.x, thb.y, thb.z, thb.w };
    uint32_t th = thv[0];
#pragma unroll
    for (int l = 1; l < SVO_MAX_LEVELS; l++) th = level == l ? thv[l] : th;
    const uint8_t* src = level == 0 ? base0 : c.pyr + (long long)img * c.pyr_bytes + e.w;
    const int pitch = level == 0 ? c.img0_pitch : (int)(e.z >> 8);
    fast_tile(c, sm, img, level, src, pitch, (int)(e.y & 0xFFFFu), (int)(e.y >> 16), (int)(e.x & 0xFFFFu), (int)(e.x >> 16), (int)th);
}

// (A persistent form -- a workgroup working through 2 / 4 / 8 consecutive tiles with two window buffers, the next tile's window in
// flight by asm LDS-DMA while the current one is tested -- was built in round 4, bit-exact, and measured at 0.305 / 0.313 / 0.331 ms
// against 0.259 ms: 16.4 KB of LDS per workgroup leave 18 waves per CU where the one-tile form keeps 30, and this kernel lives on
// occupancy (profiles/r04k_*).  Dropped; the stage / compute split below is what is left of it.)
// The (image, level) pairs whose speculated threshold found fewer than 2 * quota corners (k_select) again, with the
// caller's threshold.  Normally the list is empty and the launch retires at once; otherwise the workgroups stride over
// the concatenated tile lists of the listed pairs.
__global__ void __launch_bounds__(FT_NT) k_fast_redo(DevCtx c)
{
    SVO_TL_SCOPE(c, TL_FAST, 1);
    __shared__ FastSmem sm;
    const unsigned n = *c.redo_n;
    if (n == 0) return;
    int first = (int)blockIdx.x;                                             // this workgroup's next tile, relative to entry e
    for (unsigned e = 0; e < n; e++) {
        const uint32_t il = c.redo_list[e];
        const int img = (int)(il / SVO_MAX_LEVELS), level = (int)(il % SVO_MAX_LEVELS);
        const int nt = c.lv[level].tiles_x * c.lv[level].tiles_y;
        int t = first;
        int pitch; const uint8_t* src = level_ptr(c, img, level, pitch);
        const int tx = c.lv[level].tiles_x, gw = c.lv[level].w, gh = c.lv[level].h;
        for (; t < nt; t += (int)gridDim.x) {
            const int by = t / tx, bx = t - by * tx;
            fast_tile(c, sm, img, level, src, pitch, gw, gh, SVO_EDGE + bx * FT_W, SVO_EDGE + by * FT_H, c.fast_th);
            __syncthreads();                                                 // the tile's LDS is reused by the next one
        }
        first = t - nt;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K3: per (image, level): keep the best 2*quota corners by (FAST score desc, position asc) with a 4-pass radix
// select on the unique 32-bit keys, compute the Harris response of each, sort by (response desc, position asc)
// and keep the best quota.  One 512-thread block per (image, level).
// ------------------------------------------------------------------------------------------------------------
// SEL_MAX = capacity of the 2 * quota list of a level (template parameter: 2048 in the usual configurations, 4096 for contexts
// with max_kps > 4096); the LDS tie list holds 2 * SEL_MAX u32 entries aliasing the u64 key array

__device__ __forceinline__ float harris_at(const uint8_t* img, int pitch, int x, int y)
{
    int a = 0, b = 0, cc = 0;
    for (int dy = -3; dy <= 3; dy++) {
        const uint8_t* pm = img + (long long)(y + dy - 1) * pitch + x, *p0 = pm + pitch, *pp = p0 + pitch;
#pragma unroll
        for (int dx = -3; dx <= 3; dx++) {
            const int Ix = ((int)p0[dx + 1] - (int)p0[dx - 1]) * 2 + ((int)pm[dx + 1] - (int)pm[dx - 1]) + ((int)pp[dx + 1] - (int)pp[dx - 1]);
            const int Iy = ((int)pp[dx] - (int)pm[dx]) * 2 + ((int)pp[dx - 1] - (int)pm[dx - 1]) + ((int)pp[dx + 1] - (int)pm[dx + 1]);
            a += Ix * Ix; b += Iy * Iy; cc += Ix * Iy;
        }
    }
    // cv::ORB's HarrisResponses, operator for operator (oracle: harris_response)
    const float s = 1.0f / (4.0f * 7.0f * 255.0f);
    const float s4 = s * s * s * s;
    const float fa = (float)a, fb = (float)b, fc = (float)cc;
    const float tr = fa + fb;
    return (fa * fb - fc * fc - 0.04f * tr * tr) * s4;
}

// Harris response from 27 aligned dword loads (9 rows x 12 bytes) instead of 81 byte loads; same integer sums and
// float expression as harris_at (the oracle's harris_response)
__device__ __forceinline__ float harris_at_dw(const uint8_t* img, int pitch, int x, int y)
{
    const int xa = (x - 4) & ~3, o = (x - 4) & 3;
    int px[9][9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        // one 12-byte load per row: three dword loads of 64 scattered lanes are three passes over the same 64 cache lines,
        // and with other waves in between the L1 has dropped them by then
        const uint3 pw = *(const uint3*)(img + (long long)(y - 4 + r) * pitch + xa);
        const uint32_t w0 = pw.x, w1 = pw.y, w2 = pw.z;
        const uint32_t A = __builtin_amdgcn_alignbyte(w1, w0, o), B = __builtin_amdgcn_alignbyte(w2, w1, o), C = w2 >> (8 * o);
#pragma unroll
        for (int k = 0; k < 4; k++) { px[r][k] = (A >> (8 * k)) & 0xFF; px[r][4 + k] = (B >> (8 * k)) & 0xFF; }
        px[r][8] = C & 0xFF;
    }
    int a = 0, b = 0, cc = 0;
#pragma unroll
    for (int r = 1; r < 8; r++) {
#pragma unroll
        for (int q = 1; q < 8; q++) {
            const int Ix = (px[r][q + 1] - px[r][q - 1]) * 2 + (px[r - 1][q + 1] - px[r - 1][q - 1]) + (px[r + 1][q + 1] - px[r + 1][q - 1]);
            const int Iy = (px[r + 1][q] - px[r - 1][q]) * 2 + (px[r + 1][q - 1] - px[r - 1][q - 1]) + (px[r + 1][q + 1] - px[r - 1][q + 1]);
            a += __mul24(Ix, Ix); b += __mul24(Iy, Iy); cc += __mul24(Ix, Iy);
        }
    }
    // cv::ORB's HarrisResponses, operator for operator (oracle: harris_response)
    const float s = 1.0f / (4.0f * 7.0f * 255.0f);
    const float s4 = s * s * s * s;
    const float fa = (float)a, fb = (float)b, fc = (float)cc;
    const float tr = fa + fb;
    return (fa * fb - fc * fc - 0.04f * tr * tr) * s4;
}

// Speculative FAST threshold (exact, with fallback).  Only the 2 * quota best corners of a level survive this kernel, and
// on a textured scene they are a small fraction of what FAST finds at the caller's threshold (1 300 of 12 000 at level 0
// of a 1280x960 frame at th = 20: the cut-off score sits near 80).  A corner's score, and whether it survives the 3x3
// NMS against weaker neighbours, do not depend on the threshold it was found with (a neighbour that a higher threshold
// misses has a lower score than anything it finds), so k_fast run with th' >= th yields exactly the candidates with
// score >= th' -- and if there are at least 2 * quota of them, the 2 * quota best of those ARE the 2 * quota best of
// all.  Each (image, level) therefore carries its own th' from frame to frame: this kernel sets the next frame's th' to
// the score that 1.5 x (2 * quota) + 32 of this frame's candidates reach, and when a frame's th' turns out too high
// (fewer than 2 * quota found) it discards the list and queues the pair for k_fast_redo at the caller's threshold,
// after which pass 1 of this kernel selects from the complete list.  Same lists as the oracle, bit for bit, either way.
// (the body of the selection: the LDS arrays are the caller's -- keys[SEL_MAX] u64 (the tie
// list), sel[SEL_MAX] u32, hist[256], scan_s[32], sv[5] -- and the K winners are LEFT IN sel[]; returns K (block-uniform), 0 = this block
// has nothing to rank: its pair was queued for the redo, is not flagged in the redo pass, or has no candidates)
template <int SEL_MAX>
__device__ __forceinline__ int select_block(const DevCtx& c, int redo_pass, int img, int level, unsigned long long* keys, uint32_t* sel, unsigned* hist, int* scan_s, unsigned* sv)
{
    constexpr int SEL_TIE_MAX = 2 * SEL_MAX;
    unsigned& s_prefix = sv[0]; unsigned& s_need = sv[1]; unsigned& s_sel = sv[2]; unsigned& s_tie = sv[3]; unsigned& s_ntie = sv[4];
    const LevelGeom& g = c.lv[level];
    const int tid = threadIdx.x;
    const int il = img * SVO_MAX_LEVELS + level;
    if (redo_pass && !c.redo_flag[il]) return 0;
    unsigned nc = c.cand_cnt[il * SVO_CNT_STRIDE];
    if (nc > (unsigned)g.cand_cap) nc = g.cand_cap;
    const unsigned K = min(nc, (unsigned)(2 * g.quota));
    const uint32_t th_used = c.fast_th_used[il], th_base = (uint32_t)c.fast_th;
    if (!redo_pass && g.quota > 0 && th_used > th_base && nc < (unsigned)(2 * g.quota)) {
        // the speculated threshold found too few corners: start this pair over at the caller's threshold
        if (tid == 0) {
            c.redo_flag[il] = 1; c.redo_list[atomicAdd(c.redo_n, 1u)] = (uint32_t)il;
            atomicAdd(c.redo_n + 1, 1u);                               // running total since svo_create (svo_debug_get_redo_count)
            c.cand_cnt[il * SVO_CNT_STRIDE] = 0; c.fast_th_used[il] = th_base; c.fast_th_dyn[il] = 0;
        }
        return 0;
    }
    if (K == 0 || g.quota <= 0) { if (tid == 0) { c.lvl_n[il] = 0; c.sel_n[il] = 0; c.fast_th_dyn[il] = 0; } return 0; }
    const uint32_t* ck = c.cand_keys + (long long)img * c.cand_total + g.cand_off;
    // ---- the K largest of the unique 32-bit keys (score << 24 | inverted position) ----
    // Two sweeps over the candidate list, eight independent loads in flight per thread (a one-load-per-iteration loop
    // pays the ~1 us global latency 64 times per sweep on level 0): (1) histogram of the score byte -> the score bin
    // b that holds the K-th key; (2) keys above b go straight to sel[], keys in bin b to an LDS tie list, which three
    // LDS-resident radix passes then cut at exactly K.  A tie list that does not fit falls back to global passes.
    uint32_t* tie = (uint32_t*)keys;                                   // SEL_TIE_MAX u32 alias the u64 key array (free until Harris)
    for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
    if (tid == 0) { s_sel = 0; s_tie = 0; }
    __syncthreads();
    for (unsigned base = tid; base < nc; base += 8 * 512) {
        uint32_t k[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const unsigned i = base + u * 512; k[u] = i < nc ? ck[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) if (base + u * 512 < nc) atomicAdd(&hist[k[u] >> 24], 1u);
    }
    __syncthreads();
    if (c.debug_mode == 31) return 0;
    unsigned prefix = 0, mask = 0, need = K;
    {
        const int mine = tid < 256 ? (int)hist[255 - tid] : 0;         // bins in DESCENDING order: thread t owns bin 255 - t
        int tot;
        const int before = block_exclusive_scan(mine, scan_s, &tot);
        if (tid < 256 && (unsigned)before < need && need <= (unsigned)(before + mine)) { s_prefix = (unsigned)(255 - tid) << 24; s_need = need - (unsigned)before; s_ntie = (unsigned)mine; }
        // the next frame's threshold for this pair: the largest score s that M candidates reach (k_fast at th' = s finds
        // exactly the corners with score >= s); with fewer than M in sight, step below this frame's threshold by the
        // deficit over the density at the visible edge.  0 = do not speculate.
        const unsigned M = 3u * (unsigned)g.quota + 32u;
        if (tid < 256 && (unsigned)before < M && M <= (unsigned)(before + mine)) c.fast_th_dyn[il] = (uint32_t)(255 - tid) > th_base ? (uint32_t)(255 - tid) : 0u;
        if (tid == 0 && (unsigned)tot < M) {
            const unsigned dens = max(hist[min(th_used, 255u)], 1u), lower = (M - (unsigned)tot) / dens + 1u;
            c.fast_th_dyn[il] = th_used > th_base + lower ? th_used - lower : 0u;
        }
        __syncthreads();
        prefix = s_prefix; need = s_need; mask = 0xFF000000u;
    }
    const unsigned n_tie = s_ntie;
    uint32_t cutoff;
    // cv::KeyPointsFilter::retainBest keeps EVERY candidate that ties with the K-th by response, and FAST scores are integers:
    // the whole score bin of the K-th key goes to the Harris ranking (K - need keys above the bin + all n_tie of it), as long as
    // the list holds them; a bin too large for it is cut at exactly K (position order) and reported in the result's status word
    const unsigned K_ties = (K - need) + n_tie;
    if (K_ties <= (unsigned)SEL_MAX) {
        cutoff = prefix;                                                   // score byte of the K-th key, low bits zero: the whole bin
        for (unsigned base = tid; base < nc; base += 8 * 512) {
            uint32_t k[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const unsigned i = base + u * 512; k[u] = i < nc ? ck[i] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; u++) if (base + u * 512 < nc && k[u] >= cutoff) sel[atomicAdd(&s_sel, 1u)] = k[u];
        }
        __syncthreads();
        return (int)K_ties;
    }
    if (tid == 0) raise_detect_status(c, img >> 1, SVO_ST_CAND_OVERFLOW);
    if (n_tie <= SEL_TIE_MAX) {
        for (unsigned base = tid; base < nc; base += 8 * 512) {
            uint32_t k[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const unsigned i = base + u * 512; k[u] = i < nc ? ck[i] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (base + u * 512 >= nc) continue;
                if ((k[u] & 0xFF000000u) > prefix) sel[atomicAdd(&s_sel, 1u)] = k[u];          // < K of these, K <= SEL_MAX
                else if ((k[u] & 0xFF000000u) == prefix) tie[atomicAdd(&s_tie, 1u)] = k[u];
            }
        }
        __syncthreads();
        for (int shift = 16; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
            __syncthreads();
            for (unsigned i = tid; i < n_tie; i += blockDim.x) { const uint32_t k = tie[i]; if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u); }
            __syncthreads();
            const int mine = tid < 256 ? (int)hist[255 - tid] : 0;
            int tot;
            const int before = block_exclusive_scan(mine, scan_s, &tot);
            if (tid < 256 && (unsigned)before < need && need <= (unsigned)(before + mine)) { s_prefix = prefix | ((unsigned)(255 - tid) << shift); s_need = need - (unsigned)before; }
            __syncthreads();
            prefix = s_prefix; need = s_need; mask |= 255u << shift;
            __syncthreads();
        }
        cutoff = prefix;     // exactly K keys are >= cutoff (keys are unique)
        for (unsigned i = tid; i < n_tie; i += blockDim.x) { const uint32_t k = tie[i]; if (k >= cutoff) sel[atomicAdd(&s_sel, 1u)] = k; }
        __syncthreads();
    } else {
        for (int shift = 16; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
            __syncthreads();
            for (unsigned i = tid; i < nc; i += blockDim.x) { const uint32_t k = ck[i]; if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u); }
            __syncthreads();
            const int mine = tid < 256 ? (int)hist[255 - tid] : 0;
            int tot;
            const int before = block_exclusive_scan(mine, scan_s, &tot);
            if (tid < 256 && (unsigned)before < need && need <= (unsigned)(before + mine)) { s_prefix = prefix | ((unsigned)(255 - tid) << shift); s_need = need - (unsigned)before; }
            __syncthreads();
            prefix = s_prefix; need = s_need; mask |= 255u << shift;
            __syncthreads();
        }
        cutoff = prefix;
        for (unsigned i = tid; i < nc; i += blockDim.x) { const uint32_t k = ck[i]; if (k >= cutoff) { const unsigned sl = atomicAdd(&s_sel, 1u); if (sl < SEL_MAX) sel[sl] = k; } }
        __syncthreads();
    }
    return (int)K;
}

// K3a: the selection; its winners go to k_harris through sel_keys
template <int SEL_MAX>
__global__ void __launch_bounds__(512) k_select(DevCtx c, int redo_pass)
{
    SVO_TL_SCOPE(c, TL_SELECT, redo_pass);
    __shared__ unsigned long long keys[SEL_MAX];
    __shared__ uint32_t sel[SEL_MAX];
    __shared__ unsigned hist[256];
    __shared__ int scan_s[32];
    __shared__ unsigned sv[5];
    // image index fastest: consecutive workgroups go to consecutive XCDs, so with the level fastest every level-0 block
    // (the heavy ones) landed on the same XCD
    const int level = blockIdx.y, img = blockIdx.x, tid = threadIdx.x;
    const int K = select_block<SEL_MAX>(c, redo_pass, img, level, keys, sel, hist, scan_s, sv);
    if (K <= 0) return;
    // hand the K winners to k_harris: one CU gathering 868 x 9 scattered rows was bound by its own outstanding-request
    // budget (55 us of this kernel's 108), the whole GPU does it in a few
    uint32_t* gsel = c.sel_keys + ((long long)img * SVO_MAX_LEVELS + level) * c.sel_max;
    for (int i = tid; i < K; i += blockDim.x) gsel[i] = sel[i];
    if (tid == 0) c.sel_n[img * SVO_MAX_LEVELS + level] = K;
}

// Harris response of every selected corner, one thread each, all (image, level) lists in one launch
__global__ void __launch_bounds__(256) k_harris(DevCtx c)
{
    SVO_TL_SCOPE(c, TL_HARRIS, 0);
    const int img = blockIdx.x, level = blockIdx.z, i = blockIdx.y * 256 + threadIdx.x;
    const LevelGeom& g = c.lv[level];
    const int K = g.quota > 0 ? c.sel_n[img * SVO_MAX_LEVELS + level] : 0;
    if ((int)(blockIdx.y * 256) >= K) return;
    if (i >= K) return;
    const long long o = ((long long)img * SVO_MAX_LEVELS + level) * c.sel_max + i;
    const uint32_t pos = 0xFFFFFFu - (c.sel_keys[o] & 0xFFFFFFu);
    const int x = (int)(pos % (uint32_t)g.w), y = (int)(pos / (uint32_t)g.w);
    int pitch; const uint8_t* lim = level_ptr(c, img, level, pitch);
    const bool aligned = (((uintptr_t)lim | (uintptr_t)pitch) & 3) == 0;
    const float r = aligned ? harris_at_dw(lim, pitch, x, y) : harris_at(lim, pitch, x, y);
    c.sel_resp[o] = ((unsigned long long)ord32(r) << 32) | (unsigned long long)(0xFFFFFFFFu - pos);
}

// order the K (Harris response desc, position asc) keys of one (image, level) and keep the best quota.
// Bucket sort instead of a bitonic network (1024 of these blocks run at once and the network was instruction bound,
// 29 us): 1024 buckets over [min, max] of the response's order-preserving bit pattern (float bit patterns are log
// spaced, which suits the heavy-tailed Harris response), bucket sizes by LDS atomics, scan, scatter, and each key ranks
// itself inside its bucket by full 64-bit compares.  Any bucket function monotone in the key gives the exact order.
#define SS_NB 1024
// (the body of the ranking: mykey[it] = the (response, position) key of entry
// tid + 512 it of the K entries (0 beyond K), in registers; LDS arrays are the caller's: keys / tmp [SEL_MAX] u64, cnt / off [SS_NB], scan_s[32],
// sv[4]; called by all 512 threads with K > 0)
template <int SEL_MAX>
__device__ __forceinline__ void rank_block(const DevCtx& c, int img, int level, int K, const unsigned long long (&mykey)[SEL_MAX / 512],
                                           unsigned long long* keys, unsigned long long* tmp, int* cnt, int* off, int* scan_s, unsigned* sv)
{
    unsigned& s_mn = sv[0]; unsigned& s_mx = sv[1]; unsigned& s_mnp = sv[2]; int& s_np = *(int*)(sv + 3);
    const int tid = threadIdx.x;
    const LevelGeom& g = c.lv[level];
    if (tid == 0) { s_mn = 0xFFFFFFFFu; s_mx = 0u; s_mnp = 0xFFFFFFFFu; s_np = 0; }
    for (int i = tid; i < SS_NB; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    int mypos[SEL_MAX / 512], myb[SEL_MAX / 512];
    // Only the best `quota` keys are emitted, so only they need their exact order: the buckets span the POSITIVE responses
    // (corners; the order-preserving pattern of a positive float is >= 0x80000000) when there are at least quota of them,
    // and everything below shares the last bucket, which then starts past the output range and is never ranked.
    unsigned lmn = 0xFFFFFFFFu, lmx = 0u, lmnp = 0xFFFFFFFFu; int lnp = 0;
#pragma unroll
    for (int it = 0; it < SEL_MAX / 512; it++) {
        const int i = tid + it * 512;
        if (i < K) {
            const unsigned h = (unsigned)(mykey[it] >> 32);
            lmn = min(lmn, h); lmx = max(lmx, h);
            if (h >= 0x80000000u) { lmnp = min(lmnp, h); lnp++; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lmn = min(lmn, (unsigned)__shfl_xor((int)lmn, o, 64)); lmx = max(lmx, (unsigned)__shfl_xor((int)lmx, o, 64));
        lmnp = min(lmnp, (unsigned)__shfl_xor((int)lmnp, o, 64)); lnp += __shfl_xor(lnp, o, 64);
    }
    if (c.debug_mode == 41) return;
    if ((tid & 63) == 0) { atomicMin(&s_mn, lmn); atomicMax(&s_mx, lmx); atomicMin(&s_mnp, lmnp); atomicAdd(&s_np, lnp); }
    __syncthreads();
    const unsigned mn = s_np >= g.quota ? s_mnp : s_mn;
    const float scale = (float)(SS_NB - 1) / ((float)(s_mx - mn) + 1.0f);
#pragma unroll
    for (int it = 0; it < SEL_MAX / 512; it++) {
        const int i = tid + it * 512;
        if (i < K) {
            // descending order: bucket 0 holds the largest responses; keys below mn all go to the last bucket
            const unsigned h = (unsigned)(mykey[it] >> 32);
            const int b = h < mn ? SS_NB - 1 : SS_NB - 1 - min((int)((float)(h - mn) * scale), SS_NB - 1);
            myb[it] = b; mypos[it] = atomicAdd(&cnt[b], 1);
        }
    }
    __syncthreads();
    if (c.debug_mode == 42) return;
    int run = 0;
    for (int base = 0; base < SS_NB; base += 512) {
        int tot;
        const int o = block_exclusive_scan(cnt[base + tid], scan_s, &tot);
        off[base + tid] = run + o;
        run += tot;
        __syncthreads();
    }
    if (c.debug_mode == 43) return;
#pragma unroll
    for (int it = 0; it < SEL_MAX / 512; it++) { const int i = tid + it * 512; if (i < K) tmp[off[myb[it]] + mypos[it]] = mykey[it]; }
    __syncthreads();
    if (c.debug_mode == 44) return;
#pragma unroll
    for (int it = 0; it < SEL_MAX / 512; it++) {
        const int i = tid + it * 512;
        if (i < K) {
            const int b0 = off[myb[it]], nb = cnt[myb[it]];
            if (b0 >= min(K, g.quota)) continue;                   // past the output range: order irrelevant, never read
            int greater = 0;
            for (int q = 0; q < nb; q++) greater += tmp[b0 + q] > mykey[it] ? 1 : 0;
            keys[b0 + greater] = mykey[it];
        }
    }
    __syncthreads();
    if (c.debug_mode == 45) return;
    const int nout = min(K, g.quota);
    for (int i = tid; i < nout; i += blockDim.x) {
        const unsigned long long k = keys[i];
        const uint32_t pos = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull);
        c.lvl_pos[(long long)img * c.raw_cap + g.slot_off + i] = (pos % (uint32_t)g.w) | ((pos / (uint32_t)g.w) << 16);   // x | y << 16
        c.lvl_resp[(long long)img * c.raw_cap + g.slot_off + i] = inv_ord32((uint32_t)(k >> 32));
    }
    if (tid == 0) c.lvl_n[img * SVO_MAX_LEVELS + level] = nout;
}


// K3c: order the keys k_harris left in sel_resp
template <int SEL_MAX>
__global__ void __launch_bounds__(512) k_select_sort(DevCtx c)
{
    SVO_TL_SCOPE(c, TL_SELECT_SORT, 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];          // keys[SEL_MAX] | tmp[SEL_MAX] (72 KB of LDS in all at SEL_MAX = 4096)
    unsigned long long* keys = (unsigned long long*)ss_smem, *tmp = keys + SEL_MAX;
    __shared__ int cnt[SS_NB], off[SS_NB], scan_s[32];
    __shared__ unsigned sv[4];
    const int level = blockIdx.y, img = blockIdx.x, tid = threadIdx.x;
    if (c.lv[level].quota <= 0) return;
    const int K = c.sel_n[img * SVO_MAX_LEVELS + level];
    if (K <= 0) return;                                      // lvl_n was zeroed by k_select
    const unsigned long long* gk = c.sel_resp + ((long long)img * SVO_MAX_LEVELS + level) * c.sel_max;
    unsigned long long mykey[SEL_MAX / 512];
#pragma unroll
    for (int it = 0; it < SEL_MAX / 512; it++) { const int i = tid + it * 512; mykey[it] = i < K ? gk[i] : 0ull; }
    rank_block<SEL_MAX>(c, img, level, K, mykey, keys, tmp, cnt, off, scan_s, sv);
}

// (Round 6 built the fused form -- selection, Harris responses and ranking of one (image, level) in ONE 512-thread block, three launches per
// frame instead of five -- on top of select_block / rank_block, lists bit-exact, and measured it SLOWER in both regimes: 65.5 k against
// 67.5 k pairs/s at 3 x 64 lanes, 0.3546 against 0.3511 ms per frame for one stream (gpurun r06l).  The level-0 block ranks 868 corners:
// their 9 x 12-byte patch reads from ONE CU cost more than the two launches saved, and at 128 VGPRs only two such blocks fit a CU.
// Removed; k_harris keeps spreading the patch reads over the whole chip.)

// ------------------------------------------------------------------------------------------------------------
// K4+K5: cv::ORB per keypoint (oracle: orb_angle_desc): orientation (intensity centroid, radius 15) and the 256 tests of
// OpenCV's learned pair table, steered by the keypoint's CONTINUOUS angle, on the 7x7 sigma-2 blur of its level.
// One wave per keypoint, 4 independent waves per block (no block barriers: every wave owns its LDS region).
// VALU issue bounds this kernel, so the Gaussian -- two banded matrix products -- runs on the matrix cores, which are idle here:
//   A  raw window rows y-21..y+21, columns x-23..x+24 by LDS-DMA (global_load_lds_dwordx4): 43 rows x 3 chunks of 16 bytes land
//      at LDS pitch 48 straight from their (byte-unaligned) global addresses, three wave-instructions per keypoint;
//   B  moments by v_dot4_u32_u8: the disc starts on a dword boundary (column x-15 = byte 8), one lane = one (row, dword) with a
//      0/1 weight dword (row sums -> m01) and a (u+15) weight dword (-> m10 + 15 * sum); wave sums by DPP + readlane;
//      angle = the oracle's atan2_deg, then its single-precision sine / cosine (svo_oracle_sincosf, operator for operator);
//   C  horizontal pass H = X Gh on v_mfma_i32_16x16x64_i8: operand A = 16 window rows x 64 window bytes (one ds_read_b128 per
//      lane, pixels re-biased to signed by xor 0x80), operand B = the banded tap matrix (g_blur_gh), accumulator seeded with 128
//      so that a result IS H - 32768 as a signed 16-bit number: 3 x 3 tiles, 9 MFMAs;
//   D  vertical pass B^T = H^T Gv^T: lane (n, q) of a horizontal tile already holds the H values of column n that make up
//      k-group q of the vertical product's operand A -- no cross-lane movement; their high bytes (signed) and low bytes (re-biased)
//      are two operands, 256 D_hi + D_lo is the exact integer sum; 18 MFMAs.  The accumulator of the low product starts at
//      257 * 32896 + 32768 (the two biases and the rounding), (t >> 16) saturated to a byte is the blurred pixel, and a lane's
//      four results are four consecutive columns of one row: one ds_write_b32 into the 37 x 48 blurred window (over the raw one);
//   E  256 tests: pair (px, py) -> (px a - py b, px b + py a) in single precision, round half to even by the 1.5 * 2^23 trick
//      (the mantissa then holds the integer), one byte gather per sample point, four wave ballots.
// ------------------------------------------------------------------------------------------------------------
#define DP_PW 12            // LDS row pitch of the raw / blurred window in dwords (3 DMA chunks of 16 bytes)
#define DP_LROWS 48         // rows the matrix-core tiles cover (43 hold pixels; the rest meet zero taps)
typedef int dp_v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float atan2_deg(float y, float x)
{
    // the oracle's octant formula, with its two branch divisions folded into one (same operands, same IEEE result)
    const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool steep = !(ax >= ay);
    const float cq = (steep ? ax : ay) / ((steep ? ay : ax) + 2.220446e-16f);
    const float c2 = cq * cq;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * cq;
    if (steep) a = 90.0f - a;
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

// svo_oracle_sincosf, operator for operator (x in [0, 2 pi]; compiled with -ffp-contract=off)
__device__ __forceinline__ void sincos_f32(float x, float& sn, float& cs)
{
    const int k = (int)(x * 0.63661977f + 0.5f);
    const float fk = (float)k;
    const float r = ((x - fk * 1.5703125f) - fk * 4.837512969970703125e-4f) - fk * 7.54978995489188216e-8f;
    const float z = r * r;
    const float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const bool odd = k & 1;
    const float s0 = odd ? cp : sp, c0 = odd ? sp : cp;
    sn = (k & 2) ? -s0 : s0;                                   // k & 3:  0 (sp, cp)   1 (cp, -sp)   2 (-sp, -cp)   3 (-cp, sp)
    cs = (((k + 1) & 2) != 0) ? -c0 : c0;
}

__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t acc)
{
    // the builtin, not inline asm: the hazard recogniser must see a DOT op to pad its result's wait states
    return __builtin_amdgcn_udot4(a, b, acc, false);
}

// NW waves per block, each working through `kpw` consecutive keypoints of one image (no block barrier: every wave owns its LDS
// region).  Round 4: the kernel sat at 0.67 of its VALU issue rate -- a wave lived ~5 us of which the window fetch, the ten
// 16-byte table loads per lane (Gaussian operands, test pairs) and the slot -> level -> geometry prologue were latency and
// set-up paid per keypoint.  Now a wave keeps the tables in registers across its keypoints and the window of keypoint i + 1 is
// in flight (LDS-DMA into the wave's second buffer) while keypoint i is computed.
// TL: the Gaussian operands too live in LDS and are fetched at the head of their pass (24 VGPRs less: 5 waves per SIMD instead of 4)
template <int NW, bool TL>
__global__ void __launch_bounds__(NW * 64) k_describe(DevCtx c, FastDiv gx_div, int pre, int kpw)
{
    SVO_TL_SCOPE(c, TL_DESCRIBE, pre);
    __shared__ __attribute__((aligned(16))) uint32_t raw32[NW][2][DP_LROWS * DP_PW + 4];      // + 16 bytes: the last row's fourth k-group
    __shared__ __attribute__((aligned(16))) float4 s_pat[SVO_BRIEF_NPAIRS];                   // the test pairs and the disc weights: one copy per block
    __shared__ uint32_t s_disc_m[SVO_DISC_E], s_disc_x[SVO_DISC_E];
    static_assert(NW * 64 == SVO_BRIEF_NPAIRS && NW * 64 == SVO_DISC_E, "one table entry per thread");
    __shared__ __attribute__((aligned(16))) uint4 s_gh[TL ? 3 * 64 : 1], s_gv[TL ? 3 * 64 : 1];
    s_pat[threadIdx.x] = g_brief_patf[threadIdx.x]; s_disc_m[threadIdx.x] = g_disc_m[threadIdx.x]; s_disc_x[threadIdx.x] = g_disc_x[threadIdx.x];
    if (TL && threadIdx.x < 3 * 64) { s_gh[threadIdx.x] = g_blur_gh[threadIdx.x]; s_gv[threadIdx.x] = g_blur_gv[threadIdx.x]; }
    __syncthreads();                                         // the only block barrier: before any wave can leave
    // the wave index is uniform but lives in a VGPR: readfirstlane moves the whole slot / level / geometry prologue to
    // the scalar unit
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // XCD affinity by IMAGE: workgroup ids go round-robin over the 8 XCDs, so id -> (image, slot group) is laid out so
    // that every XCD works through whole images (image = 8 * group + id % 8).  A keypoint window costs 43 rows x one or
    // two 128-byte lines and neighbouring keypoints share most of them; with an image's keypoints spread over eight
    // L2s each line was fetched again per XCD, through one L2 the image's pyramid (3.8 MB < 4 MB) is fetched about once.
    int img, bx;
    if (c.debug_mode == 8) { img = blockIdx.x / gx_div.d; bx = blockIdx.x - img * gx_div.d; }
    else { const uint32_t r = blockIdx.x >> 3, grp = fastdiv(r, gx_div); bx = (int)(r - grp * gx_div.d); img = (int)(grp * 8 + (blockIdx.x & 7)); }
    if (img >= c.n_img) return;
    // pre != 0: the NMS ran first (k_nms_rowsort, pre mode); work item = final keypoint fi of the image's current list, which
    // the NMS kernel wrote as ONE list (position, level) beside its length, and only the angle and the descriptor are left to
    // fill in.  pre == 0: the level-segmented detector slots (quota_l slots per level, lvl_n of them live).
    const int lane_id = img >> 1, vl0 = lane_id * c.oct_cap;
    const int n_end = __builtin_amdgcn_readfirstlane(pre ? c.desc_n[img] : c.n_slots);
    const int slot0 = (bx * NW + wid) * kpw;
    if (slot0 >= n_end) return;                                                              // wave-uniform
    const int slot1 = min(slot0 + kpw, n_end);
    const long long fo0 = pre ? feat_base(c, vl0, 1 - c.lane[lane_id].prev_slot, img & 1) : 0;
    // the constant operands of the two passes and this lane's four test pairs
    dp_v4i GH[3], GV[3];
#pragma unroll
    for (int t = 0; t < 3; t++) { if (!TL) { GH[t] = *(const dp_v4i*)&g_blur_gh[t * 64 + lane]; GV[t] = *(const dp_v4i*)&g_blur_gv[t * 64 + lane]; } else { GH[t] = dp_v4i{0, 0, 0, 0}; GV[t] = GH[t]; } }
    // the tables are "used" here, ahead of the loop: the compiler then waits for them once, here, instead of re-stating its
    // s_waitcnt vmcnt at their first uses inside the loop -- where it would wait out the next keypoint's window as well
#pragma unroll
    for (int t = 0; t < 3; t++) if (!TL) asm volatile("" : "+v"(GH[t]), "+v"(GV[t]));
    // the work items of this wave's slots: lane i holds slot0 + i (one vector load ahead of the loop; an item is then two v_readlane)
    uint32_t w_pos = 0, w_lvl = 0;
    if (slot0 + lane < slot1) {
        if (pre) { const uint2 w = c.desc_work[(long long)img * c.raw_cap + slot0 + lane]; w_pos = w.x; w_lvl = w.y; }
        else w_pos = c.lvl_pos[(long long)img * c.raw_cap + slot0 + lane];
    }
    asm volatile("" : "+v"(w_pos), "+v"(w_lvl));             // (waited for here, once: see the tables above)
    // (every field goes through readfirstlane / readlane: the values are uniform, and left to itself the compiler carries the
    // level across the loop in a VGPR and fetches its geometry with vector loads -- two exposed round trips per keypoint)
    auto item = [&](int slot, int& live, int& x, int& y, int& level) {
        const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)w_pos, slot - slot0);
        int lv = 0, ok = 1;
        if (pre) lv = __builtin_amdgcn_readlane((int)w_lvl, slot - slot0);
        else {
#pragma unroll
            for (int l = 1; l < SVO_MAX_LEVELS; l++) if (l < c.n_levels && slot >= c.lv[l].slot_off) lv = l;
            lv = __builtin_amdgcn_readfirstlane(lv);
            ok = __builtin_amdgcn_readfirstlane(slot - c.lv[lv].slot_off < c.lvl_n[img * SVO_MAX_LEVELS + lv] ? 1 : 0);
        }
        level = lv; live = ok; x = (int)(pos & 0xFFFFu); y = (int)(pos >> 16);
    };
    // ---- A: window rows y-21..y+21, columns x-23..x+24, LDS-DMA: chunk i = (row i / 3, 16-byte piece i % 3) lands at LDS byte 16 i ----
    //      every byte read lies inside the image: keypoints keep EDGE = 31 pixels from every border
    // level-0 pointer of the image (a table in memory) and its pyramid block: fetched once, kept in scalar registers
    const uint8_t* base0; const uint8_t* pyr0 = c.pyr + (long long)img * c.pyr_bytes;
    {
        const uint64_t b = (uint64_t)(uintptr_t)c.img0[img];
        base0 = (const uint8_t*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b));
    }
    const int cr0 = (lane * 171) >> 9, cq0 = lane - 3 * cr0, cr1 = ((lane + 64) * 171) >> 9, cq1 = lane + 64 - 3 * cr1;       // i / 3, i % 3 for i <= 128
    auto stage = [&](uint32_t* dst, int x, int y, int level) {
        const int pitch = level == 0 ? c.img0_pitch : c.lv[level].pitch;
        const uint8_t* lim = level == 0 ? base0 : pyr0 + c.lv[level].offset;
        const uint8_t* org = lim + (uint32_t)((y - (DP_REACH + 3)) * pitch + (x - DP_X0));
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)dst);
        glds16_asm(org + (uint32_t)(cr0 * pitch + 16 * cq0), lb);
        glds16_asm(org + (uint32_t)(cr1 * pitch + 16 * cq1), lb + 1024);
        if (lane < DP_RW * 3 - 128) glds16_asm(org + (uint32_t)(42 * pitch + 32), lb + 2048);      // chunk 128 = (row 42, piece 2)
    };
    static_assert(DP_RW * 3 - 128 == 1, "the third DMA instruction carries chunk 128 alone");
    int live, x, y, level;
    item(slot0, live, x, y, level);
    if (live) stage(raw32[wid][0], x, y, level);
    const int n16 = lane & 15, q4 = lane >> 4;
    for (int slot = slot0; slot < slot1; slot++) {
        uint32_t* R32 = raw32[wid][(slot - slot0) & 1];
        int live_n = 0, x_n = 0, y_n = 0, level_n = 0;
        if (slot + 1 < slot1) item(slot + 1, live_n, x_n, y_n, level_n);
        live_n = __builtin_amdgcn_readfirstlane(live_n); x_n = __builtin_amdgcn_readfirstlane(x_n); y_n = __builtin_amdgcn_readfirstlane(y_n); level_n = __builtin_amdgcn_readfirstlane(level_n);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this keypoint's window has landed (and the last one's stores are out)
        wave_lds_sync();
        // the next window goes to the other buffer, whose last reader was the test phase of the keypoint before this one
        if (live_n) stage(raw32[wid][(slot - slot0 + 1) & 1], x_n, y_n, level_n);
        if (live) {
            // ---- B: moments over the disc: 31 rows x 8 dwords = 248 lane-tasks ----
            uint32_t m10u = 0; int m01 = 0, msum = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int en = i * 64 + lane;                     // entries 248..255 have zero weights and read a row that exists
                const int vr = en >> 3, d = en & 7;               // disc row v = vr - 15, window row vr + 6
                const uint32_t px = R32[(vr + 6) * DP_PW + 2 + d];
                const uint32_t s = udot4(px, s_disc_m[en], 0u);
                m10u = udot4(px, s_disc_x[en], m10u);
                msum += (int)s;
                m01 += __mul24(vr - 15, (int)s);
            }
            const int m10 = wave_sum_uniform((int)m10u - 15 * msum);
            m01 = wave_sum_uniform(m01);
            float angle, sn, cs;
            if (c.debug_mode == 46) { angle = (float)(m10 & 255); sn = (float)(m01 & 255) * 0.001f; cs = 1.f - sn; }      /* ablation: no trigonometry (results differ) */
            else { angle = atan2_deg((float)m01, (float)m10); sincos_f32(angle * 0.017453292f, sn, cs); }
            // ---- C: horizontal pass on the matrix cores: S[mt][nt] = H - 32768 for window rows 16 mt + 4 q + r, blurred columns 16 nt + n ----
            dp_v4i S[3][3];
            {
                if (TL) {
#pragma unroll
                    for (int t = 0; t < 3; t++) GH[t] = *(const dp_v4i*)&s_gh[t * 64 + lane];
                }
                const dp_v4i seed = { 128, 128, 128, 128 };
#pragma unroll
                for (int mt = 0; mt < 3; mt++) {
                    dp_v4i X = *(const dp_v4i*)&R32[(16 * mt + n16) * DP_PW + 4 * q4];
                    X ^= (int)0x80808080;
#pragma unroll
                    for (int nt = 0; nt < 3; nt++) S[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(X, GH[nt], seed, 0, 0, 0);
                }
            }
            wave_lds_sync();                                       // every lane has read the raw window: the blurred one may overwrite it
            // ---- D: vertical pass; lane (n16, q4) owns k-group q4 of column n16 already ----
            uint8_t* Bl = (uint8_t*)R32;
            {
                if (TL) {
#pragma unroll
                    for (int t = 0; t < 3; t++) GV[t] = *(const dp_v4i*)&s_gv[t * 64 + lane];
                }
                const int c2 = 257 * 32896 + 32768;
                const dp_v4i seed_lo = { c2, c2, c2, c2 }, zero = { 0, 0, 0, 0 };
#pragma unroll
                for (int nt = 0; nt < 3; nt++) {
                    dp_v4i Alo, Ahi;
#pragma unroll
                    for (int mt = 0; mt < 3; mt++) {
                        const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)S[mt][nt][1], (uint32_t)S[mt][nt][0], 0x05010400u);     // lo0 lo1 hi0 hi1
                        const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)S[mt][nt][3], (uint32_t)S[mt][nt][2], 0x05010400u);
                        Alo[mt] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
                        Ahi[mt] = (int)__builtin_amdgcn_perm(p23, p01, 0x07060302u);
                    }
                    Alo[3] = 0; Ahi[3] = 0;
#pragma unroll
                    for (int ot = 0; ot < 3; ot++) {
                        const dp_v4i Dhi = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ahi, GV[ot], zero, 0, 0, 0);
                        const dp_v4i Dlo = __builtin_amdgcn_mfma_i32_16x16x64_i8(Alo, GV[ot], seed_lo, 0, 0, 0);
                        uint32_t t[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) t[r] = ((uint32_t)Dhi[r] << 8) + (uint32_t)Dlo[r];
                        // (t >> 16) is 0..257: saturate the 16-bit halves to bytes (v_sat_pk_u8_i16), four columns of one row per lane
                        uint32_t h01 = __builtin_amdgcn_perm(t[1], t[0], 0x07060302u), h23 = __builtin_amdgcn_perm(t[3], t[2], 0x07060302u), s01, s23;
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(s01) : "v"(h01));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(s23) : "v"(h23));
                        *(uint32_t*)&Bl[(16 * ot + n16) * (4 * DP_PW) + 16 * nt + 4 * q4] = __builtin_amdgcn_perm(s23, s01, 0x05040100u);
                    }
                }
            }
            wave_lds_sync();
            // ---- E: 256 tests, one byte gather per sample point, packed with four wave ballots ----
            // x + 1.5 * 2^23 rounds to the nearest integer, ties to even (cvRound), and leaves it in the mantissa: bits = 0x4B400000 + ix.
            // The offsets of the window centre and of this wave's LDS region ride in the magic constants; the products with the pitch
            // take the low 24 bits (v_mad_u32_u24), the sum's low 16 bits are the LDS address.
            const float MX = 12582912.0f + (float)(DP_REACH + (int)((uint8_t*)R32 - (uint8_t*)&raw32[0][0][0])), MY = 12582912.0f + (float)DP_REACH;
            const uint8_t* lds0 = (const uint8_t*)&raw32[0][0][0];
            unsigned long long bits[4];
            // the two points of a pair ride in the halves of packed single-precision operations (v_pk_mul_f32 / v_pk_add_f32: each half
            // is one IEEE operation, as the scalar form): pat = (x0, x1, y0, y1)
            typedef float dp_f2 __attribute__((ext_vector_type(2)));
            const dp_f2 cs2 = { cs, cs }, sn2 = { sn, sn }, MX2 = { MX, MX }, MY2 = { MY, MY };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 pt = s_pat[k * 64 + lane];
                const dp_f2 px = { pt.x, pt.y }, py = { pt.z, pt.w };
                const dp_f2 xs = (px * cs2 - py * sn2) + MX2, ys = (px * sn2 + py * cs2) + MY2;
                const uint32_t o0 = (__umul24(__float_as_uint(ys.x), 4 * DP_PW) + __float_as_uint(xs.x)) & 0xFFFFu;
                const uint32_t o1 = (__umul24(__float_as_uint(ys.y), 4 * DP_PW) + __float_as_uint(xs.y)) & 0xFFFFu;
                const int a = lds0[o0], b = lds0[o1];
                bits[k] = __ballot(a < b);
            }
            if (lane == 0) {
                if (pre) {
                    const long long fo = fo0 + slot;
                    unsigned long long* d = (unsigned long long*)(c.desc + fo * 32);
                    d[0] = bits[0]; d[1] = bits[1]; d[2] = bits[2]; d[3] = bits[3];
                    c.kps[fo].angle = angle;
                } else {
                    const LevelGeom& g = c.lv[level];
                    const long long o = (long long)img * c.raw_cap + slot;
                    unsigned long long* d = (unsigned long long*)(c.raw_desc + o * 32);
                    d[0] = bits[0]; d[1] = bits[1]; d[2] = bits[2]; d[3] = bits[3];
                    svo_keypoint k;
                    // ORB mode: level-0 coordinates, size 31*scale, octave = pyramid level.  FAST+ORB mode: cv::FAST keypoints
                    // (octave-image coordinates, size 7, octave 0) that cv::ORB::compute only orients and describes
                    k.x = (float)x * g.scale; k.y = (float)y * g.scale; k.size = c.fast_orb ? 7.0f : 31.0f * g.scale; k.angle = angle;
                    k.response = c.lvl_resp[o]; k.octave = c.fast_orb ? 0 : level; k.class_id = -1;
                    c.raw_kps[o] = k;
                }
            }
        }
        live = live_n; x = x_n; y = y_n; level = level_n;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K6: the reference's own post-processing of the detector output, one 1024-thread block per image:
//   m_non_max_sup (stage2_detect.cpp:296-370): visit in (response desc, index asc) order, greedy grid occupancy;
//   m_update_indexes(order=true) (stage2_detect.cpp:65-130): re-sort by (pt.y asc, rank asc).
// Writes the final keypoints + descriptors of the lane's current slot.
// ------------------------------------------------------------------------------------------------------------
// NI = keys per thread: 4 (lists up to 4096) or 8 (up to 8192)
template <int NI>
__global__ void __launch_bounds__(1024) k_nms_rowsort(DevCtx c, int do_nms, int min_distance, int NS_MAX, int pre, uint8_t* big)
{
    SVO_TL_SCOPE(c, TL_NMS, pre);
    SVO_LATENCY_CHAIN(c);
    // dynamic LDS only (G17): keys[NS_MAX] u64 | hkey[2*NS_MAX] | hval[2*NS_MAX] | cellxy[NS_MAX] | acc_idx[NS_MAX] u16 | state[NS_MAX] u8 | scan[32] | flag
    // Above 4096 keys that is more LDS than a CU has: the first four arrays (24 of the 27 bytes per key) then live in a global
    // scratch region of this (image, octave) -- `big` -- and only the small ones stay in LDS.  Same code either way: barriers
    // order a workgroup's global accesses as they order its LDS accesses, the atomics work on both.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NS_HASH = 2 * NS_MAX;
    const int img = blockIdx.x, oct = blockIdx.y, lane_id = img >> 1, side = img & 1, tid = threadIdx.x;
    const bool raster = do_nms == 3;          // FAST + ORB with the NMS switched off: every corner, in cv::FAST's raster order
    if (raster) do_nms = 0;
    unsigned char* wide = big ? big + ((size_t)img * c.oct_cap + oct) * ((size_t)NS_MAX * 28) : smem;
    unsigned long long* keys = (unsigned long long*)wide;
    uint32_t* hkey = (uint32_t*)(keys + NS_MAX);
    uint32_t* hval = hkey + NS_HASH;
    uint32_t* cellxy = hval + NS_HASH;
    unsigned short* acc_idx = (unsigned short*)(big ? smem : (unsigned char*)(cellxy + NS_MAX));          // raw index of the i-th survivor
    unsigned char* state = (unsigned char*)(acc_idx + NS_MAX);
    int* scan = (int*)(((uintptr_t)(state + NS_MAX) + 15) & ~(uintptr_t)15);
    int* flag = scan + 32;
    const int vl = lane_id * c.oct_cap + oct;
    const svo_keypoint* rk = c.raw_kps + (long long)img * c.raw_cap;
    // pre != 0 (ORB mode): this kernel runs BEFORE the describe kernel -- the reference's NMS needs positions and
    // responses only -- so that only its survivors are oriented and described (a third of the detector's output is
    // dropped here).  The keypoint attributes then come from k_select_sort's lists with the expressions the describe
    // kernel uses, the final records are written with a placeholder angle, and final_slot tells k_describe what to fill in.
    const uint32_t* lpos = c.lvl_pos + (long long)img * c.raw_cap;
    const float* lresp = c.lvl_resp + (long long)img * c.raw_cap;
    auto kp_at = [&](int slot) -> svo_keypoint {
        if (!pre) return rk[slot];
        int l = 0;
#pragma unroll
        for (int q = 1; q < SVO_MAX_LEVELS; q++) if (q < c.n_levels && slot >= c.lv[q].slot_off) l = q;
        const uint32_t pos = lpos[slot];
        const float sc = c.lv[l].scale;
        svo_keypoint k;
        k.x = (float)(pos & 0xFFFFu) * sc; k.y = (float)(pos >> 16) * sc; k.size = 31.0f * sc; k.angle = 0.0f;
        k.response = lresp[slot]; k.octave = l; k.class_id = -1;
        return k;
    };
    // ORB mode (one octave): all pyramid levels are gathered, level 0 first, each level by Harris rank, and the
    // reference's grid NMS runs here.  FAST+ORB mode: level == octave, the NMS already ran before the describe kernel
    // (k_fastorb_nms), so only this octave's segment is taken, in its accepted (response-descending) order.
    const int l_first = c.fast_orb ? oct : 0, l_last = c.fast_orb ? oct + 1 : c.n_levels;
    int lvl_base[SVO_MAX_LEVELS + 1];
    for (int l = 0; l <= SVO_MAX_LEVELS; l++) lvl_base[l] = 0;
    for (int l = l_first; l < SVO_MAX_LEVELS; l++) lvl_base[l + 1] = lvl_base[l] + (l < l_last ? c.lvl_n[img * SVO_MAX_LEVELS + l] : 0);
    const int n = lvl_base[l_last];
    for (int i = tid; i < NS_MAX; i += blockDim.x) keys[i] = 0;
    __syncthreads();
    // key = (response desc, raw index asc); the payload is the raw index, the slot is recovered from it
    for (int l = l_first; l < l_last; l++) {
        const int nl = lvl_base[l + 1] - lvl_base[l];
        for (int i = tid; i < nl; i += blockDim.x) {
            const int raw_i = lvl_base[l] + i;
            const float resp = pre ? lresp[c.lv[l].slot_off + i] : rk[c.lv[l].slot_off + i].response;
            keys[raw_i] = ((unsigned long long)ord32(resp) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)raw_i);
        }
    }
    __syncthreads();
    if (c.debug_mode == 21) return;
    int P = 64; while (P < n) P <<= 1;
    if (do_nms && !c.fast_orb) {
        // (response desc, raw index asc) order by bucket sort: buckets over the order-preserving bit pattern of the
        // response with the empty stretch between the largest negative and the smallest positive response cut out
        // (float bit patterns are log spaced: both populated bands spread evenly), sizes by LDS atomics, scan, scatter,
        // and every key ranks itself inside its bucket by full 64-bit compares.  Any bucket function monotone in the key
        // yields the exact order.  (A 4096-key bitonic network took 34 us here, merge ranks over the per-level segments
        // 19 us -- 70 scattered LDS reads per key.)
        const int NB = min(2048, NS_MAX);
        int* bcnt = (int*)hkey, *boff = bcnt + NB;                      // 8 * NB <= 8 * NS_MAX bytes: inside the hkey region
        unsigned long long* tmp = (unsigned long long*)hval;
        unsigned* red = (unsigned*)scan;                                   // [0] mn  [1] mx of negatives  [2] mn of positives  [3] mx
        if (tid == 0) { red[0] = 0xFFFFFFFFu; red[1] = 0u; red[2] = 0xFFFFFFFFu; red[3] = 0u; }
        for (int i = tid; i < NB; i += blockDim.x) bcnt[i] = 0;
        __syncthreads();
        unsigned long long mykey[NI]; int myb[NI], mypos[NI];
        unsigned l0 = 0xFFFFFFFFu, l1 = 0u, l2 = 0xFFFFFFFFu, l3 = 0u;
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int i = tid + it * (int)blockDim.x;
            mykey[it] = i < n ? keys[i] : 0ull;
            if (i < n) {
                const unsigned h = (unsigned)(mykey[it] >> 32);
                l0 = min(l0, h); l3 = max(l3, h);
                if (h < 0x80000000u) l1 = max(l1, h); else l2 = min(l2, h);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l0 = min(l0, (unsigned)__shfl_xor((int)l0, o, 64)); l1 = max(l1, (unsigned)__shfl_xor((int)l1, o, 64));
            l2 = min(l2, (unsigned)__shfl_xor((int)l2, o, 64)); l3 = max(l3, (unsigned)__shfl_xor((int)l3, o, 64));
        }
        if ((tid & 63) == 0) { atomicMin(&red[0], l0); atomicMax(&red[1], l1); atomicMin(&red[2], l2); atomicMax(&red[3], l3); }
        __syncthreads();
        const unsigned mn = red[0], mxn = red[1], mnp = red[2], mx = red[3];
        const bool has_neg = mn < 0x80000000u, has_pos = mx >= 0x80000000u;
        const unsigned wneg = has_neg ? mxn - mn + 1u : 0u;
        const float range = (float)wneg + (has_pos ? (float)(mx - mnp) + 1.0f : 0.0f);
        const float bscale = (float)(NB - 1) / range;
        __syncthreads();                                                  // red[] is scan[]: free it before the scans below
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int i = tid + it * (int)blockDim.x;
            if (i < n) {
                const unsigned h = (unsigned)(mykey[it] >> 32);
                const float code = h < 0x80000000u ? (float)(h - mn) : (float)(h - mnp) + (float)wneg;
                const int b = NB - 1 - min((int)(code * bscale), NB - 1);      // descending: bucket 0 = largest responses
                myb[it] = b; mypos[it] = atomicAdd(&bcnt[b], 1);
            }
        }
        __syncthreads();
        int run = 0;
        for (int base = 0; base < NB; base += blockDim.x) {
            const int bi = base + tid;
            int tot;
            const int o = block_exclusive_scan(bi < NB ? bcnt[bi] : 0, scan, &tot);
            if (bi < NB) boff[bi] = run + o;
            run += tot;
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < NI; it++) { const int i = tid + it * (int)blockDim.x; if (i < n) tmp[boff[myb[it]] + mypos[it]] = mykey[it]; }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int i = tid + it * (int)blockDim.x;
            if (i < n) {
                const int b0 = boff[myb[it]], nb = bcnt[myb[it]];
                int greater = 0;
                for (int q = 0; q < nb; q++) greater += tmp[b0 + q] > mykey[it] ? 1 : 0;
                keys[b0 + greater] = mykey[it];
            }
        }
        __syncthreads();
    } else if (do_nms) bitonic_sort_lds<true>(keys, P);
    if (c.debug_mode == 22) return;
    auto slot_of = [&](int raw_i) { int l = l_first; for (int q = l_first + 1; q < SVO_MAX_LEVELS; q++) if (q < l_last && raw_i >= lvl_base[q]) l = q; return c.lv[l].slot_off + (raw_i - lvl_base[l]); };
    const int W = c.ow[oct], H = c.oh[oct], num_out_points = c.kps_to_detect[oct];
    int nacc = 0;
    if (do_nms == 2) {
        // ---- m_adaptive_non_max_sup (stage2_detect.cpp:141-215; oracle: svo_oracle_anms_copy) ----
        // keys[] holds the keypoints in (response desc, raw index asc) order.  radius^2 of rank i = min over the ranks
        // k2 in [1, i) with resp_i < 0.9 * resp_k2 (double compare, S2:183) of the float squared distance, bounded by
        // the distance to rank 0 whatever its response (S2:176).  0.9 * resp is non-increasing along the ranks, so the
        // qualifying k2 are a prefix [1, t]: t by binary search, then a compare-free min loop.
        float2* sxy = (float2*)hkey;                       // [NS_MAX]   (the grid-hash arrays are free in this mode)
        double* thr = (double*)(hkey + 2 * NS_MAX);        // [NS_MAX]
        float* rad = (float*)cellxy;                       // [NS_MAX]
        for (int i = tid; i < n; i += blockDim.x) {
            const int raw_i = (int)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull));
            const svo_keypoint k = kp_at(slot_of(raw_i));
            sxy[i] = make_float2(k.x, k.y);
            thr[i] = 0.9 * (double)k.response;
        }
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            float r = __builtin_inff();                                               // S2:167
            if (i > 0) {
                const float2 a = sxy[i], s0 = sxy[0];
                float dx = a.x - s0.x, dy = a.y - s0.y;
                r = fabsf(dx * dx + dy * dy);                                         // S2:176
                const double resp = (double)inv_ord32((uint32_t)(keys[i] >> 32));
                int lo = 1, hi = i;                                                   // first k2 in [1, i) with !(resp < thr[k2])
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (resp < thr[mid]) lo = mid + 1; else hi = mid; }
                for (int k2 = 1; k2 < lo; k2++) {                                     // S2:179-189
                    const float2 b = sxy[k2];
                    dx = a.x - b.x; dy = a.y - b.y;
                    r = fminf(r, fabsf(dx * dx + dy * dy));
                }
            }
            rad[i] = r;
        }
        __syncthreads();
        for (int i = tid; i < P; i += blockDim.x) keys[i] = i < n ? (((unsigned long long)ord32(rad[i]) << 32) | (keys[i] & 0xFFFFFFFFull)) : 0ull;
        bitonic_sort_lds<true>(keys, P);                                              // S2:196: (radius desc, raw index asc)
        const int actual = min(num_out_points, n);                                    // S2:151
        for (int base = 0; base < actual; base += blockDim.x) {                       // S2:207-214, min_radius_th = 0
            const int i = base + tid;
            const int keep = (i < actual && inv_ord32((uint32_t)(keys[i] >> 32)) > 0.0f) ? 1 : 0;
            int tot;
            const int off = block_exclusive_scan(keep, scan, &tot);
            if (keep) acc_idx[nacc + off] = (unsigned short)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull));
            nacc += tot;
            __syncthreads();
        }
    } else if (do_nms) {
        const unsigned cell = (unsigned)((double)min_distance / 2.0);            // S2:331
        const float inv = 1.0f / (float)cell;                                    // S2:332
        const unsigned glx = (unsigned)(1 + (float)W * inv), gly = (unsigned)(1 + (float)H * inv);   // S2:334-335
        for (int i = tid; i < n; i += blockDim.x) {
            const int raw_i = (int)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull));
            const svo_keypoint k = kp_at(slot_of(raw_i));
            const size_t ux = (size_t)(k.x * inv), uy = (size_t)(k.y * inv);     // S2:348-349
            cellxy[i] = (ux < glx && uy < gly) ? (((uint32_t)ux << 16) | (uint32_t)uy) : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (c.debug_mode == 26) return;
        grid_nms_block<NI>(n, gly, cellxy, hkey, hval, NS_HASH, state, flag);
        if (c.debug_mode == 23) return;
        // survivors in rank order, at most num_out_points of them (S2:342)
        for (int base = 0; base < n && nacc < num_out_points; base += blockDim.x) {
            const int i = base + tid;
            const int keep = (i < n && state[i] == 1) ? 1 : 0;
            int tot;
            const int off = block_exclusive_scan(keep, scan, &tot);
            if (keep && nacc + off < num_out_points) acc_idx[nacc + off] = (unsigned short)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull));
            nacc += tot;
            __syncthreads();
        }
        if (nacc > num_out_points) nacc = num_out_points;
    } else if (raster) {
        // no NMS after FAST: the reference keeps cv::FAST's output as it comes, in raster order (S2:613-614), which is what the
        // row sort below then keeps inside a row.  k_fastorb_nms delivered the corners by score: back to (y, x) order.
        for (int i = tid; i < P; i += blockDim.x) {
            unsigned long long k = ~0ull;
            if (i < n) { const svo_keypoint kp = kp_at(slot_of(i)); k = ((unsigned long long)((uint32_t)kp.y * (uint32_t)W + (uint32_t)kp.x) << 32) | (unsigned)i; }
            keys[i] = k;
        }
        bitonic_sort_lds<false>(keys, P);
        for (int i = tid; i < n; i += blockDim.x) acc_idx[i] = (unsigned short)(keys[i] & 0xFFFFFFFFull);
        nacc = n;
        __syncthreads();
    } else {
        for (int i = tid; i < n; i += blockDim.x) acc_idx[i] = (unsigned short)i;
        nacc = n;
        __syncthreads();
    }
    if (c.debug_mode == 24) return;
    if (nacc > c.max_kps) { nacc = c.max_kps; if (tid == 0) { atomicOr(&c.status[lane_id], SVO_ST_KPS_OVERFLOW); atomicOr(&c.results[lane_id].status, (int)SVO_ST_KPS_OVERFLOW); } }
    // row sort: (pt.y asc, survivor rank asc)
    __syncthreads();
    for (int i = tid; i < NS_MAX; i += blockDim.x) keys[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < nacc; i += blockDim.x) keys[i] = ((unsigned long long)ord32(kp_at(slot_of(acc_idx[i])).y) << 32) | (unsigned)i;
    __syncthreads();
    if (H <= NS_MAX) {
        // counting sort on the integer row (pt.y >= 0, so ord32 order == numeric order and the row is monotone in the key):
        // bucket sizes by LDS atomics, exclusive scan over the rows, scatter, then each key ranks itself inside its (tiny)
        // bucket.  Six barriers instead of the 66 stages of a 2048-key bitonic network.
        int* rcnt = (int*)hkey, *roff = rcnt + H;                         // the grid-hash arrays are free again
        unsigned long long* tmp = (unsigned long long*)hval;
        for (int r = tid; r < H; r += blockDim.x) rcnt[r] = 0;
        __syncthreads();
        int myrow[NI], mypos[NI];                                         // nacc <= NI * 1024 keys, 1024 threads
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int i = tid + it * (int)blockDim.x;
            myrow[it] = -1; mypos[it] = 0;
            if (i < nacc) { const int r = min(max((int)inv_ord32((uint32_t)(keys[i] >> 32)), 0), H - 1); myrow[it] = r; mypos[it] = atomicAdd(&rcnt[r], 1); }
        }
        __syncthreads();
        int run = 0;
        for (int base = 0; base < H; base += blockDim.x) {
            const int r = base + tid;
            const int v = r < H ? rcnt[r] : 0;
            int tot;
            const int off = block_exclusive_scan(v, scan, &tot);
            if (r < H) roff[r] = run + off;
            run += tot;
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int i = tid + it * (int)blockDim.x;
            if (i < nacc) tmp[roff[myrow[it]] + mypos[it]] = keys[i];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const int i = tid + it * (int)blockDim.x;
            if (i < nacc) {
                const int r = myrow[it], b0 = roff[r], n_r = rcnt[r];
                const unsigned long long key = tmp[b0 + mypos[it]];
                int less = 0;
                for (int q = 0; q < n_r; q++) less += tmp[b0 + q] < key ? 1 : 0;
                keys[b0 + less] = key;
            }
        }
        __syncthreads();
    } else {
        P = 64; while (P < nacc) P <<= 1;
        bitonic_sort_lds<false>(keys, P);
    }
    if (c.debug_mode == 25) return;
    const int cur = 1 - c.lane[lane_id].prev_slot;
    const long long ob = feat_base(c, vl, cur, side);
    for (int i = tid; i < nacc; i += blockDim.x) {
        const int s = slot_of(acc_idx[(int)(keys[i] & 0xFFFFFFFFull)]);
        const svo_keypoint kp = kp_at(s);
        c.kps[ob + i] = kp;
        if (pre) {      // k_describe's work item i of this image: everything it needs to start loading, in one place
            c.final_slot[ob + i] = s;
            c.desc_work[(long long)img * c.raw_cap + i] = make_uint2(lpos[s], (uint32_t)kp.octave);
            continue;
        }
        const uint4* sd = (const uint4*)(c.raw_desc + ((long long)img * c.raw_cap + s) * 32);
        uint4* dd = (uint4*)(c.desc + (ob + i) * 32);
        dd[0] = sd[0]; dd[1] = sd[1];
    }
    // m_update_indexes row table (stage2_detect.cpp:103-129): idx[r] = #keypoints with (int)y <= r for
    // first_row <= r < last_row, 0 elsewhere (the reference never fills the tail).  keys are sorted by y: binary search.
    {
        int* ridx = c.row_index + (long long)feat_cnt_idx(vl, cur, side) * c.max_h;
        int first_row = 0, last_row = 0;
        if (nacc > 0) { first_row = (int)inv_ord32((uint32_t)(keys[0] >> 32)); last_row = (int)inv_ord32((uint32_t)(keys[nacc - 1] >> 32)); }
        for (int r = tid; r < H; r += blockDim.x) {
            int v = 0;
            if (nacc > 0 && r >= first_row && r < last_row) {
                int lo = 0, hi = nacc;                                         // first index whose (int)y > r
                while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)inv_ord32((uint32_t)(keys[mid] >> 32)) <= r) lo = mid + 1; else hi = mid; }
                v = lo;
            }
            ridx[r] = v;
        }
    }
    if (tid == 0) {
        c.n_kps[feat_cnt_idx(vl, cur, side)] = nacc;
        c.raw_n[img] = n;
        if (pre) c.desc_n[img] = nacc;
        if (side == 0) c.results[lane_id].detected_left[oct] = nacc; else c.results[lane_id].detected_right[oct] = nacc;
    }
}

// ------------------------------------------------------------------------------------------------------------
// FAST+ORB mode (stage2_detect.cpp:502-515, configs[4]): the x1/2 octave pyramid of mrpt CImagePyramid (S1:82-83)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_half(DevCtx c, int level)
{
    SVO_TL_SCOPE(c, TL_RESIZE, level);
    const int img = blockIdx.z;
    const LevelGeom& d = c.lv[level];
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= d.w || y >= d.h) return;
    int spitch; const uint8_t* src = level_ptr(c, img, level - 1, spitch);
    uint8_t* dst = c.pyr + (long long)img * c.pyr_bytes + d.offset;
    const uint8_t* r0 = src + (long long)(2 * y) * spitch + 2 * x4, *r1 = r0 + spitch;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x4 + k < d.w) out |= (uint32_t)(((int)r0[2 * k] + r0[2 * k + 1] + r1[2 * k] + r1[2 * k + 1] + 2) >> 2) << (8 * k);
    *(uint32_t*)(dst + (long long)y * d.pitch + x4) = out;
}

// FAST+ORB: the reference's grid NMS (m_non_max_sup, S2:296-370, cap kps_to_detect[octave]) applied to ALL FAST corners
// of one (image, octave) in (score desc, raster position asc) order -- exactly the descending order of the unique
// candidate keys.  The corners are consumed in chunks of the NS_MAX largest remaining keys (radix select, bitonic sort,
// block-parallel NMS).  Between chunks only the ACCEPTED cells survive (a compact list re-seeds the hash with the
// marker FO_ACCEPTED, which every later candidate sees as "accepted, lower rank"); a cell whose representative was
// rejected needs no memory: whatever rejected it still rejects every later candidate of that cell.
// Survivors go to the level-segmented arrays that k_describe reads (describing only survivors gives the same
// descriptors as the reference's describe-then-suppress order).
#define FO_ACCEPTED 0xFFFFFFFEu
__global__ void __launch_bounds__(1024) k_fastorb_nms(DevCtx c, int min_distance, int do_nms, int NS_MAX, int ACC_MAX)
{
    SVO_TL_SCOPE(c, TL_NMS, 2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NS_HASH = 4 * NS_MAX;
    unsigned long long* keys = (unsigned long long*)smem;                  // NS_MAX
    uint32_t* hkey = (uint32_t*)(keys + NS_MAX);                           // 4*NS_MAX
    uint32_t* hval = hkey + NS_HASH;                                       // 4*NS_MAX
    uint32_t* cellxy = hval + NS_HASH;                                     // NS_MAX
    uint32_t* acc_cells = cellxy + NS_MAX;                                 // ACC_MAX: cell keys of everything accepted so far
    unsigned* hist = acc_cells + ACC_MAX;                                  // 256
    int* scan = (int*)(hist + 256);                                        // 32
    int* flag = scan + 32;
    unsigned* sh = (unsigned*)(flag + 1);                                  // s_prefix, s_need, s_sel
    unsigned char* state = (unsigned char*)(sh + 4);                       // NS_MAX
    const int level = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const LevelGeom& g = c.lv[level];
    unsigned nc = c.cand_cnt[(img * SVO_MAX_LEVELS + level) * SVO_CNT_STRIDE];
    if (nc > (unsigned)g.cand_cap) nc = g.cand_cap;
    // quota = slots reserved for this octave; the reference's cap kps_to_detect belongs to its NMS (S2:342), without NMS every corner stays
    const int cap = do_nms ? min(min(c.kps_to_detect[level], g.quota), ACC_MAX) : min(g.quota, ACC_MAX);
    if (!do_nms && nc > (unsigned)cap && tid == 0) raise_detect_status(c, img >> 1, SVO_ST_KPS_OVERFLOW);
    const uint32_t* ck = c.cand_keys + (long long)img * c.cand_total + g.cand_off;
    const unsigned cell = (unsigned)((double)min_distance / 2.0);
    const float inv = 1.0f / (float)cell;
    const unsigned glx = (unsigned)(1 + (float)g.w * inv), gly = (unsigned)(1 + (float)g.h * inv);
    auto slot_of_key = [&](uint32_t key) {                                 // find-or-insert
        uint32_t h = nms_hash_slot(key, NS_HASH);
        for (;;) {
            const uint32_t old = atomicCAS(&hkey[h], 0xFFFFFFFFu, key);
            if (old == 0xFFFFFFFFu || old == key) return h;
            h = (h + 1) & (uint32_t)(NS_HASH - 1);
        }
    };
    int nacc = 0;
    uint32_t upper = 0xFFFFFFFFu;             // keys >= upper are already consumed
    unsigned remaining = nc;
    while (remaining > 0 && nacc < cap) {
        const unsigned K = min(remaining, (unsigned)NS_MAX);
        // ---- the K largest keys below `upper` (radix select on the unique keys) ----
        unsigned prefix = 0, mask = 0, need = K;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
            __syncthreads();
            for (unsigned i = tid; i < nc; i += blockDim.x) { const uint32_t k = ck[i]; if (k < upper && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u); }
            __syncthreads();
            const int mine = tid < 256 ? (int)hist[255 - tid] : 0;
            int tot;
            const int before = block_exclusive_scan(mine, scan, &tot);
            if (tid < 256 && (unsigned)before < need && need <= (unsigned)(before + mine)) { sh[0] = prefix | ((unsigned)(255 - tid) << shift); sh[1] = need - (unsigned)before; }
            __syncthreads();
            prefix = sh[0]; need = sh[1]; mask |= 255u << shift;
            __syncthreads();
        }
        const uint32_t cutoff = prefix;
        if (tid == 0) sh[2] = 0;
        for (int i = tid; i < NS_MAX; i += blockDim.x) keys[i] = 0;
        __syncthreads();
        for (unsigned base = 0; base < nc; base += blockDim.x) {
            const unsigned i = base + tid;
            uint32_t k = 0;
            const bool take = i < nc && (k = ck[i]) >= cutoff && k < upper;
            const unsigned long long m = __ballot(take);
            if (m) {
                unsigned b0 = 0;
                const int leader = __ffsll((long long)m) - 1;
                if ((tid & 63) == leader) b0 = atomicAdd(&sh[2], (unsigned)__popcll(m));
                b0 = __shfl(b0, leader, 64);
                const unsigned slot = b0 + __popcll(m & ((1ull << (tid & 63)) - 1ull));
                if (take && slot < (unsigned)NS_MAX) keys[slot] = (unsigned long long)k;
            }
        }
        __syncthreads();
        int P = 64; while (P < (int)K) P <<= 1;
        bitonic_sort_lds<true>(keys, P);                                   // rank order of this chunk
        if (do_nms) {
            // ---- re-seed the hash with the accepted cells, then the chunk's representatives ----
            for (int i = tid; i < NS_HASH; i += blockDim.x) { hkey[i] = 0xFFFFFFFFu; hval[i] = 0xFFFFFFFFu; }
            __syncthreads();
            for (int i = tid; i < nacc; i += blockDim.x) hval[slot_of_key(acc_cells[i])] = FO_ACCEPTED;
            __syncthreads();
            for (int i = tid; i < (int)K; i += blockDim.x) {
                const uint32_t pos = 0xFFFFFFu - ((uint32_t)keys[i] & 0xFFFFFFu);
                const float fx = (float)(pos % (uint32_t)g.w), fy = (float)(pos / (uint32_t)g.w);
                const size_t ux = (size_t)(fx * inv), uy = (size_t)(fy * inv);
                const uint32_t cxy = (ux < glx && uy < gly) ? (((uint32_t)ux << 16) | (uint32_t)uy) : 0xFFFFFFFFu;
                cellxy[i] = cxy;
                if (cxy != 0xFFFFFFFFu) {
                    const uint32_t h = slot_of_key((cxy >> 16) * gly + (cxy & 0xFFFFu));
                    if (hval[h] != FO_ACCEPTED) atomicMin(&hval[h], (uint32_t)i);      // markers are only written before this phase
                }
            }
            __syncthreads();
            for (int i = tid; i < (int)K; i += blockDim.x) {
                const uint32_t cxy = cellxy[i];
                unsigned char st = 0;
                if (cxy != 0xFFFFFFFFu && nms_lookup(hkey, hval, NS_HASH, (cxy >> 16) * gly + (cxy & 0xFFFFu)) == (uint32_t)i) st = SVO_NMS_UNDECIDED;
                state[i] = st;                                             // non-representatives and already-accepted cells: rejected
            }
            __syncthreads();
            for (;;) {
                if (tid == 0) *flag = 0;
                __syncthreads();
                bool pending = false;
                for (int i = tid; i < (int)K; i += blockDim.x) {
                    if (((volatile unsigned char*)state)[i] != SVO_NMS_UNDECIDED) continue;
                    const uint32_t cxy = cellxy[i];
                    const int sx = (int)(cxy >> 16), sy = (int)(cxy & 0xFFFFu);
                    bool any_acc = false, any_und = false;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int cx = sx + (q == 0) - (q == 1), cy = sy + (q == 2) - (q == 3);
                        if (cx < 0 || cy < 0 || cy >= (int)gly) continue;
                        const uint32_t j = nms_lookup(hkey, hval, NS_HASH, (uint32_t)cx * gly + (uint32_t)cy);
                        if (j == FO_ACCEPTED) any_acc = true;
                        else if (j < (uint32_t)i) { const unsigned char sj = ((volatile unsigned char*)state)[j]; any_acc |= sj == 1; any_und |= sj == SVO_NMS_UNDECIDED; }
                    }
                    if (any_acc) ((volatile unsigned char*)state)[i] = 0;
                    else if (!any_und) ((volatile unsigned char*)state)[i] = 1;
                    else pending = true;
                }
                if (pending) *flag = 1;
                __syncthreads();
                const int again = *flag;
                __syncthreads();
                if (!again) break;
            }
        } else {
            for (int i = tid; i < (int)K; i += blockDim.x) state[i] = 1;
            __syncthreads();
        }
        // ---- emit the chunk's survivors in rank order, up to the cap, and remember their cells ----
        int chunk_acc = 0;
        for (int base = 0; base < (int)K && nacc + chunk_acc < cap; base += blockDim.x) {
            const int i = base + tid;
            const int keep = (i < (int)K && state[i] == 1) ? 1 : 0;
            int tot;
            const int off = block_exclusive_scan(keep, scan, &tot);
            const int o_idx = nacc + chunk_acc + off;
            if (keep && o_idx < cap) {
                const uint32_t k = (uint32_t)keys[i];
                const long long o = (long long)img * c.raw_cap + g.slot_off + o_idx;
                const uint32_t pos = 0xFFFFFFu - (k & 0xFFFFFFu);
                c.lvl_pos[o] = (pos % (uint32_t)g.w) | ((pos / (uint32_t)g.w) << 16);                      // x | y << 16
                c.lvl_resp[o] = (float)(k >> 24);                          // cv::FAST response = score
                if (do_nms) { const uint32_t cxy = cellxy[i]; acc_cells[o_idx] = (cxy >> 16) * gly + (cxy & 0xFFFFu); }
            }
            chunk_acc += tot;
            __syncthreads();
        }
        nacc = min(nacc + chunk_acc, cap);
        upper = cutoff;
        remaining -= K;
        __syncthreads();
    }
    if (tid == 0) c.lvl_n[img * SVO_MAX_LEVELS + level] = nacc;
}

// FAST+ORB with nmsmAdaptive (stage2_detect.cpp:599-606 applies m_adaptive_non_max_sup, S2:141-215, to whatever the
// detector produced -- here ALL FAST corners of an octave, tens of thousands).  One 1024-thread block per (octave, image).
// Reference: sort by (response desc, index asc); radius^2 of rank k1 = min over rank 0 and over the ranks k2 in [1, k1) with
// resp_k1 < 0.9 resp_k2 of the float squared distance; sort by (radius desc, index asc); keep the first min(num_out, N).
// No rank order is needed for the radii: 0.9 resp_k2 > resp_k1 already implies resp_k2 > resp_k1, i.e. k2 < k1, so the set
// is "rank 0, plus every corner whose score passes the 0.9 test" -- and FAST responses are small integers, so a counting
// sort by score lines those corners up as a PREFIX whose length depends on the score alone.  Squared pixel distances stay
// below 2^24: the float arithmetic of the reference is exact integer arithmetic.  The first min(num_out, N) of the radius
// order come from a radix select on the radius (ties at the cut-off by position), then one LDS sort of <= max_kps keys.
// Scratch (global, per image): by_score[cand_total] (key of every corner, score-descending buckets), xy (their x | y << 16),
// radius[cand_total].
__global__ void __launch_bounds__(1024) k_fastorb_anms(DevCtx c, uint32_t* by_score_all, uint32_t* radius_all, uint32_t* xy_all, int KMAX)
{
    SVO_TL_SCOPE(c, TL_NMS, 3);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = (unsigned long long*)smem;                  // KMAX (power of two >= cap)
    __shared__ unsigned hist[256], start[256], prefix_len[256];
    __shared__ int scan[40];
    __shared__ unsigned sh[8];
    const int level = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const LevelGeom& g = c.lv[level];
    unsigned nc = c.cand_cnt[(img * SVO_MAX_LEVELS + level) * SVO_CNT_STRIDE];
    if (nc > (unsigned)g.cand_cap) nc = g.cand_cap;
    const int cap = min(min(c.kps_to_detect[level], g.quota), c.max_kps);
    const int actual = min((int)nc, cap);                                  // S2:151
    if (actual <= 0) { if (tid == 0) c.lvl_n[img * SVO_MAX_LEVELS + level] = 0; return; }
    const uint32_t* ck = c.cand_keys + (long long)img * c.cand_total + g.cand_off;
    uint32_t* bs = by_score_all + (long long)img * c.cand_total + g.cand_off;
    uint32_t* rad = radius_all + (long long)img * c.cand_total + g.cand_off;
    uint32_t* xy = xy_all + (long long)img * c.cand_total + g.cand_off;
    const uint32_t gw = (uint32_t)g.w;
    // ---- counting sort by score, descending buckets; the strongest corner (largest key) found on the way ----
    for (int i = tid; i < 256; i += 1024) hist[i] = 0;
    if (tid == 0) sh[0] = 0;
    __syncthreads();
    uint32_t kmax = 0;
    for (unsigned i = tid; i < nc; i += 1024) { const uint32_t k = ck[i]; atomicAdd(&hist[k >> 24], 1u); kmax = max(kmax, k); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
    if ((tid & 63) == 0) atomicMax(&sh[0], kmax);
    __syncthreads();
    {
        const int mine = tid < 256 ? (int)hist[255 - tid] : 0;            // thread t owns score 255 - t
        int tot;
        const int before = block_exclusive_scan(mine, scan, &tot);
        if (tid < 256) start[255 - tid] = (unsigned)before;
    }
    __syncthreads();
    // prefix_len[s] = number of corners b with (double)s < 0.9 * (double)score_b (S2:183): whole buckets, from the top
    if (tid < 256) {
        unsigned len = 0;
        for (int sb = 255; sb >= 0; sb--) if ((double)(float)tid < 0.9 * (double)(float)sb) len = start[sb] + hist[sb]; else break;
        prefix_len[tid] = len;
    }
    __syncthreads();
    for (int i = tid; i < 256; i += 1024) hist[i] = 0;                     // reused as the buckets' fill counters
    __syncthreads();
    for (unsigned i = tid; i < nc; i += 1024) {
        const uint32_t k = ck[i], pos = 0xFFFFFFu - (k & 0xFFFFFFu);
        const unsigned sl = start[k >> 24] + atomicAdd(&hist[k >> 24], 1u);
        bs[sl] = k; xy[sl] = (pos % gw) | ((pos / gw) << 16);              // the divisions once per corner, not once per distance
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t top = sh[0], tpos = 0xFFFFFFu - (top & 0xFFFFFFu);
    const int tx = (int)(tpos % gw), ty = (int)(tpos / gw);
    // ---- radius^2 of every corner (index = position in by_score) ----
    for (unsigned i = tid; i < nc; i += 1024) {
        const uint32_t k = bs[i];
        uint32_t r = 0xFFFFFFFFu;                                          // the strongest corner: infinite radius (S2:167)
        if (k != top) {
            const uint32_t me = xy[i];
            const int x = (int)(me & 0xFFFFu), y = (int)(me >> 16);
            int best = (x - tx) * (x - tx) + (y - ty) * (y - ty);           // S2:176
            const unsigned len = prefix_len[k >> 24];
            for (unsigned j = 0; j < len; j++) {                           // every lane of a wave reads the same entry: broadcast loads
                const uint32_t pj = xy[j];
                const int dx = x - (int)(pj & 0xFFFFu), dy = y - (int)(pj >> 16);
                best = min(best, dx * dx + dy * dy);                       // S2:185 (the strongest corner may be among them: same minimum)
            }
            r = (uint32_t)best;
        }
        rad[i] = r;
    }
    __threadfence_block();
    __syncthreads();
    // ---- the first `actual` of the (radius desc, position asc) order: radix select on the radius ----
    unsigned prefix = 0, mask = 0, need = (unsigned)actual;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        for (unsigned i = tid; i < nc; i += 1024) { const uint32_t r = rad[i]; if ((r & mask) == prefix) atomicAdd(&hist[(r >> shift) & 255u], 1u); }
        __syncthreads();
        const int mine = tid < 256 ? (int)hist[255 - tid] : 0;
        int tot;
        const int before = block_exclusive_scan(mine, scan, &tot);
        if (tid < 256 && (unsigned)before < need && need <= (unsigned)(before + mine)) { sh[1] = prefix | ((unsigned)(255 - tid) << shift); sh[2] = need - (unsigned)before; }
        __syncthreads();
        prefix = sh[1]; need = sh[2]; mask |= 255u << shift;
        __syncthreads();
    }
    const uint32_t rcut = prefix;          // radii > rcut are all in; `need` of the corners with radius == rcut, smallest positions first
    // among the ties, the `need` smallest positions = the `need` LARGEST inverted positions (key & 0xFFFFFF): radix select again
    unsigned p2 = 0, m2 = 0, need2 = need;
    for (int shift = 16; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        for (unsigned i = tid; i < nc; i += 1024) if (rad[i] == rcut) { const uint32_t ip = bs[i] & 0xFFFFFFu; if ((ip & m2) == p2) atomicAdd(&hist[(ip >> shift) & 255u], 1u); }
        __syncthreads();
        const int mine = tid < 256 ? (int)hist[255 - tid] : 0;
        int tot;
        const int before = block_exclusive_scan(mine, scan, &tot);
        if (tid < 256 && (unsigned)before < need2 && need2 <= (unsigned)(before + mine)) { sh[3] = p2 | ((unsigned)(255 - tid) << shift); sh[4] = need2 - (unsigned)before; }
        __syncthreads();
        p2 = sh[3]; need2 = sh[4]; m2 |= 255u << shift;
        __syncthreads();
    }
    const uint32_t ipcut = p2;             // ties with inverted position >= ipcut are in (positions are unique)
    for (int i = tid; i < KMAX; i += 1024) keys[i] = 0;
    if (tid == 0) sh[5] = 0;
    __syncthreads();
    for (unsigned i = tid; i < nc; i += 1024) {
        const uint32_t r = rad[i], k = bs[i];
        if (r > rcut || (r == rcut && (k & 0xFFFFFFu) >= ipcut)) {
            const unsigned sl = atomicAdd(&sh[5], 1u);
            // (radius desc, position asc): position asc == inverted position desc; the score rides along in the low byte
            if (sl < (unsigned)KMAX) keys[sl] = ((unsigned long long)r << 32) | ((unsigned long long)(k & 0xFFFFFFu) << 8) | (k >> 24);
        }
    }
    __syncthreads();
    int P = 64; while (P < actual) P <<= 1;
    bitonic_sort_lds<true>(keys, P);
    for (int i = tid; i < actual; i += 1024) {
        const unsigned long long e = keys[i];
        const uint32_t pos = 0xFFFFFFu - (uint32_t)((e >> 8) & 0xFFFFFFu);
        const long long o = (long long)img * c.raw_cap + g.slot_off + i;
        c.lvl_pos[o] = (pos % gw) | ((pos / gw) << 16);
        c.lvl_resp[o] = (float)(uint32_t)(e & 0xFFu);                      // cv::FAST response = score
    }
    if (tid == 0) c.lvl_n[img * SVO_MAX_LEVELS + level] = actual;
}

void launch_fastorb_anms(const DevCtx& c, uint32_t* scratch3, hipStream_t st)
{
    int kmax = 64; while (kmax < c.max_kps) kmax <<= 1;
    const size_t one = (size_t)c.n_img * c.cand_total;
    hipLaunchKernelGGL(k_fastorb_anms, dim3(c.n_levels, c.n_img), dim3(1024), (size_t)kmax * 8, st, c, scratch3, scratch3 + one, scratch3 + 2 * one, kmax);
}

// ------------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------------
void launch_begin_frame(const DevCtx& c, const uint8_t* const* ptrs, unsigned flags, hipStream_t st)
{
    ImgPtrs ip;
    for (int i = 0; i < 2 * SVO_MAX_LANES; i++) ip.p[i] = (ptrs && i < c.n_img) ? ptrs[i] : nullptr;
    const int n = c.n_img * SVO_MAX_LEVELS;
    int blocks = (n + 255) / 256;
    if (flags & (SVO_RUN_MATCH | SVO_RUN_TRACK)) {          // + the fill of the brute-force result words: 4 x 16 bytes per thread
        const size_t n16 = (size_t)c.n_lanes * c.oct_cap * 3 * c.max_kps / 4;
        blocks = std::max(blocks, (int)std::min<size_t>((n16 + 1023) / 1024, 2048));
    }
    hipLaunchKernelGGL(k_begin_frame, dim3(blocks), dim3(256), 0, st, c, ip, flags);
}

void launch_resize(const DevCtx& c, int level, hipStream_t st)
{
    const LevelGeom& d = c.lv[level];
    const int ntx = (d.w + RZ_W - 1) / RZ_W, per_img = ntx * ((d.h + RZ_H - 1) / RZ_H);
    const long long total = (long long)per_img * c.n_img, unit = 8 * RZ_CHUNK;
    hipLaunchKernelGGL(k_resize, dim3((unsigned)((total + unit - 1) / unit * unit)), dim3(256), 0, st, c, level, make_fastdiv((uint32_t)per_img), make_fastdiv((uint32_t)ntx));
}

void launch_fast(const DevCtx& c, hipStream_t st)
{
    if (c.n_tiles <= 0) return;
    const long long total = (long long)c.n_tiles * c.n_img, unit = 8 * FT_CHUNK;
    hipLaunchKernelGGL(k_fast, dim3((unsigned)((total + unit - 1) / unit * unit)), dim3(FT_NT), 0, st, c);
}

void launch_select(const DevCtx& c, hipStream_t st)
{
    const bool big = c.sel_max > 2048;
    if (big) hipLaunchKernelGGL(k_select<4096>, dim3(c.n_img, c.n_levels), dim3(512), 0, st, c, 0);
    else hipLaunchKernelGGL(k_select<2048>, dim3(c.n_img, c.n_levels), dim3(512), 0, st, c, 0);
    // the pairs whose speculated FAST threshold was too high (normally none: both launches retire at once)
    hipLaunchKernelGGL(k_fast_redo, dim3(4096), dim3(FT_NT), 0, st, c);
    if (big) hipLaunchKernelGGL(k_select<4096>, dim3(c.n_img, c.n_levels), dim3(512), 0, st, c, 1);
    else hipLaunchKernelGGL(k_select<2048>, dim3(c.n_img, c.n_levels), dim3(512), 0, st, c, 1);
    hipLaunchKernelGGL(k_harris, dim3(c.n_img, c.sel_max / 256, c.n_levels), dim3(256), 0, st, c);
    if (big) hipLaunchKernelGGL(k_select_sort<4096>, dim3(c.n_img, c.n_levels), dim3(512), (size_t)4096 * 16, st, c);
    else hipLaunchKernelGGL(k_select_sort<2048>, dim3(c.n_img, c.n_levels), dim3(512), (size_t)2048 * 16, st, c);
}

void launch_describe(const DevCtx& c, int pre, hipStream_t st)
{
    if (c.n_slots <= 0) return;
    // keypoints per wave: SVO_DESC_KPW overrides the default for an A/B (1 = a wave per keypoint, as rounds 1-3)
    static int kpw = 0;
    if (!kpw) { const char* e = getenv("SVO_DESC_KPW"); const int v = e ? atoi(e) : 0; kpw = (v >= 1 && v <= 64) ? v : 8; }       // <= 64: a wave keeps its work items one per lane
    const int per = 4 * kpw, gx = (c.n_slots + per - 1) / per, img8 = (c.n_img + 7) / 8 * 8;
    static int tl = -1;
    if (tl < 0) { const char* e = getenv("SVO_DESC_TL"); tl = (e && atoi(e) == 0) ? 0 : 1; }       // default: operands in LDS (0.204 -> 0.198 ms at 64 lanes, profiles/r04i); SVO_DESC_TL=0 keeps them in registers
    if (tl) hipLaunchKernelGGL((k_describe<4, true>), dim3((unsigned)((long long)gx * img8)), dim3(256), 0, st, c, make_fastdiv((uint32_t)gx), (pre && !c.fast_orb) ? 1 : 0, kpw);
    else hipLaunchKernelGGL((k_describe<4, false>), dim3((unsigned)((long long)gx * img8)), dim3(256), 0, st, c, make_fastdiv((uint32_t)gx), (pre && !c.fast_orb) ? 1 : 0, kpw);
}

#define FO_PMAX 2048     // chunk size of k_fastorb_nms (LDS: 45 B per entry)
static int nms_pmax(const DevCtx& c)
{
    // keys per block: ORB mode gathers every level of an image (n_slots), FAST+ORB mode one octave's quota
    int n = 64;
    if (c.fast_orb) { for (int l = 0; l < c.n_levels; l++) if (c.lv[l].quota > n) n = c.lv[l].quota; } else n = c.n_slots;
    int p = 64; while (p < n) p <<= 1; return p;
}
static size_t nms_rowsort_smem(int pmax) { return pmax > 4096 ? (size_t)pmax * 3 + 16 + 4 * 40 : (size_t)pmax * (8 + 8 + 8 + 4 + 2 + 1) + 16 + 4 * 40; }
static size_t fastorb_nms_smem(int pmax, int accmax) { return (size_t)pmax * (8 + 16 + 16 + 4 + 1) + (size_t)accmax * 4 + 4 * (256 + 32 + 8) + 16; }

size_t nms_rowsort_scratch_bytes(const DevCtx& c)          // global scratch of k_nms_rowsort for lists above 4096 keys (0: everything fits LDS)
{
    const int pmax = nms_pmax(c);
    return pmax > 4096 ? (size_t)c.n_img * c.oct_cap * (size_t)pmax * 28 : 0;
}

hipError_t configure_nms_rowsort(const DevCtx& c)
{
    hipError_t e = svo_raise_dyn_smem((const void*)k_fastorb_nms, fastorb_nms_smem(FO_PMAX, c.max_kps));
    if (e != hipSuccess) return e;
    int kmax = 64; while (kmax < c.max_kps) kmax <<= 1;
    e = svo_raise_dyn_smem((const void*)k_fastorb_anms, (size_t)kmax * 8);
    if (e != hipSuccess) return e;
    e = svo_raise_dyn_smem((const void*)k_select_sort<4096>, 4096 * 16);
    if (e != hipSuccess) return e;
    const int pmax = nms_pmax(c);
    if (pmax > 8192) return hipErrorInvalidValue;
    e = svo_raise_dyn_smem((const void*)k_nms_rowsort<8>, nms_rowsort_smem(8192));
    if (e != hipSuccess) return e;
    return svo_raise_dyn_smem((const void*)k_nms_rowsort<4>, nms_rowsort_smem(pmax > 4096 ? 4096 : pmax));
}

void launch_nms_rowsort(const DevCtx& c, int do_nms, int min_distance, int pre, hipStream_t st)
{
    const int pmax = nms_pmax(c);
    // threads per image: every phase strides by blockDim.x and a thread holds NI = 4 keys, so lists of up to 2048 keys also run on 512
    // threads (8 waves instead of 16 to find room for on a CU that the detector's tiles fill); SVO_NMS_NT = 512 / 1024 for an A/B
    static int nt_knob = -1;
    if (nt_knob < 0) { const char* e = getenv("SVO_NMS_NT"); nt_knob = e ? atoi(e) : 0; }
    const int nt = (pmax <= 2048 && nt_knob == 512) ? 512 : 1024;
    if (pmax > 4096) hipLaunchKernelGGL(k_nms_rowsort<8>, dim3(c.n_img, c.n_oct), dim3(1024), nms_rowsort_smem(pmax), st, c, c.fast_orb ? (do_nms == 3 ? 3 : 0) : do_nms, min_distance, pmax, (pre && !c.fast_orb) ? 1 : 0, c.big_scratch);
    else hipLaunchKernelGGL(k_nms_rowsort<4>, dim3(c.n_img, c.n_oct), dim3(nt), nms_rowsort_smem(pmax), st, c, c.fast_orb ? (do_nms == 3 ? 3 : 0) : do_nms, min_distance, pmax, (pre && !c.fast_orb) ? 1 : 0, (uint8_t*)nullptr);
}

void launch_half(const DevCtx& c, int level, hipStream_t st)
{
    const LevelGeom& d = c.lv[level];
    hipLaunchKernelGGL(k_half, dim3(((d.w + 3) / 4 + 255) / 256, d.h, c.n_img), dim3(256), 0, st, c, level);
}

void launch_fastorb_nms(const DevCtx& c, int do_nms, int min_distance, hipStream_t st)
{
    hipLaunchKernelGGL(k_fastorb_nms, dim3(c.n_levels, c.n_img), dim3(1024), fastorb_nms_smem(FO_PMAX, c.max_kps), st, c, min_distance, do_nms, FO_PMAX, c.max_kps);
}
