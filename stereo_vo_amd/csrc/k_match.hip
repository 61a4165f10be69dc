// k_match.hip -- stages 3 and 4 on the device: brute-force Hamming matching with 256-bit descriptor tiles in LDS,
// the reference's left-right filters, the sequential joint collision filter of the tracker, the fundamental-matrix
// RANSAC and the final consistency check.
//
// Replaces stage3_match_left_right (libstereo-odometry/src/stage3_match_left_right.cpp:83-178, smDescBF) and
// stage4_track (libstereo-odometry/src/stage4_match_consecutive.cpp:88-329, ifmDescBF).  cv::BFMatcher and
// cv::findFundamentalMat are replaced by the algorithms frozen in oracle/svo_oracle.c; all index lists come out
// bit-exact, the RANSAC arithmetic is IEEE double in the oracle's operation order (-ffp-contract=off).
#include "svo_device.h"
#include "svo_kernels.h"

// ------------------------------------------------------------------------------------------------------------
// K7: Hamming brute force, first minimum.  mode 0: current left -> current right (stage 3).  mode 1: previous
// pairings -> current pairings, blockIdx.z / nsplit = side (left-left / right-right) on the descriptor rows that
// k_gather_mdesc laid out contiguously in pairing order (the gather of S4:105-131).
// The distances are computed on the MATRIX CORES (see "the brute force on the matrix cores" below): this is the one
// genuinely GEMM-shaped piece of the path -- N x N x 256 -- and the VALU formulation (8 v_xor + 8 v_bcnt + 2 per pair,
// ~80 cycles per 64 distances whether the train row comes from LDS, scalar loads or SGPR constants:
// tools/ubench/ham_loop.hip) was its floor.  Partial results of the train splits merge by atomicMin on the packed
// (distance << 16 | index) word, as before.
// ------------------------------------------------------------------------------------------------------------
#define HM_TILE 256

#ifdef SVO_AB_KERNELS      // an A/B anchor, not a product form (svo_kernels.h): compiled into libsvo_hip_ab.so only
__global__ void __launch_bounds__(256) k_gather_mdesc(DevCtx c)
{
    // blockIdx.z: bit 0 = side, bit 1 = 0 previous / 1 current slot; 8 threads per descriptor (one dword each)
    const int vl = blockIdx.y, lane_id = vl / c.oct_cap;
    if (vl % c.oct_cap >= c.n_oct) return;
    const LaneState& ls = c.lane[lane_id];
    if (!ls.has_prev) return;
    const int side = blockIdx.z & 1, slot = (blockIdx.z & 2) ? 1 - ls.prev_slot : ls.prev_slot;
    const int n = c.n_matches[vl * 2 + slot];
    const int m = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, w = threadIdx.x & 7;
    if (m >= n) return;
    const svo_dmatch* mm = c.matches + match_base(c, vl, slot);
    const int row = side ? mm[m].trainIdx : mm[m].queryIdx;
    const uint32_t* src = (const uint32_t*)(c.desc + (feat_base(c, vl, slot, side) + row) * 32);
    uint32_t* dst = (uint32_t*)(c.mdesc + (feat_base(c, vl, slot, side) + m) * 32);
    dst[w] = src[w];
}
#endif

// ---- the brute force on the matrix cores ------------------------------------------------------------------------
// Hamming(q, t) over 256 bits IS a dot product: with bit b encoded as q' = +64 / -64 (b = 0 / 1) for the query and
// t' = -64 / +64 for the train row, sum_k t'_k q'_k = -4096 * (256 - 2 * ham).  v_mfma_i32_32x32x32_i8 accumulates
// that on top of C, and C is initialised with the train index, so one accumulator of the chain over the 8 k-steps is
//        D = t - 2^20 + 8192 * ham         (t < 8192)
// -- already the packed first-minimum key: min over D = smallest distance, then smallest train index, exactly the
// rule of cv::BFMatcher.  No popcount, no xor: the epilogue of a 32-train tile is 8 v_min3 per 32 queries.
// Tile shapes: A = 32 train rows (M), B = 32 query columns (N), K = 32 bits expanded to bytes per MFMA, 8 MFMAs per
// tile.  A wave keeps TWO sets of 32 queries expanded in registers (64 VGPRs) and streams train tiles from LDS, where
// the block expands each 1 KB tile of packed rows once for its four waves (layout [k-step][half][row][16 B]: a wave's
// ds_read_b128 is 1 KB contiguous).  Output layout of the 32x32 MFMA: lane l holds column l % 32, VGPR v holds row
// 8 * (v / 4) + v % 4 + 4 * (l / 32)  (tools/ubench/mfma_layout.hip).
typedef int hm_v4i __attribute__((ext_vector_type(4)));
typedef int hm_v16i __attribute__((ext_vector_type(16)));
#define HM_QB 256          // queries per block (4 waves x 2 sets x 32)

// 4 bits -> 4 bytes of 0 / 1 (LSB first), then -> bytes of base ^ 0x80 where the bit is set
// (v_mul_u32_u24: both factors are below 2^24 and the product below 2^32 -- the full 32-bit multiply issues at a quarter of the rate)
__device__ __forceinline__ uint32_t hm_spread4(uint32_t nib, uint32_t base)
{
    uint32_t m;                                                     // (as asm: hipcc turns __umul24 of a 4-bit value back into v_mul_lo_u32)
    asm("v_mul_u32_u24_e32 %0, 0x204081, %1" : "=v"(m) : "v"(nib));
    // base 0x40: bytes 0x40 / 0xC0 for bit 0 / 1 = 0x40 + (bit << 7); base 0xC0: 0xC0 / 0x40 = 0x40 + (!bit << 7) -- one and, one v_lshl_add
    return base == 0x40404040u ? (((m & 0x01010101u) << 7) + 0x40404040u) : (((~m & 0x01010101u) << 7) + 0x40404040u);
}
// 16 bits (low half of `bits`) -> 16 bytes
__device__ __forceinline__ hm_v4i hm_expand16(uint32_t bits, uint32_t base)
{
    hm_v4i r;
    r.x = (int)hm_spread4(bits & 15u, base); r.y = (int)hm_spread4((bits >> 4) & 15u, base);
    r.z = (int)hm_spread4((bits >> 8) & 15u, base); r.w = (int)hm_spread4((bits >> 12) & 15u, base);
    return r;
}

__device__ __forceinline__ void hamming_body(const DevCtx& c, int mode, int nsplit)
{
    __shared__ __attribute__((aligned(16))) hm_v4i tileA[2][16 * 32];          // [buffer][k-step * 64 + ((half * 32 + row) ^ k-step)]
    // grid = (lane-octave, query block, side x split): the query blocks past nq exit at once, and with the lane index
    // fastest they sit at the END of the dispatch order.  (With the query block fastest, live and dead workgroups
    // alternate, the dispatcher hands them to the two halves of each XCD in turn, and half the CUs idle: measured 2x.)
    const int vl = blockIdx.x, lane_id = vl / c.oct_cap;
    if (vl % c.oct_cap >= c.n_oct) return;
    const int side = mode ? (blockIdx.z / nsplit) : 0, split = blockIdx.z % nsplit;
    const LaneState& ls = c.lane[lane_id];
    const int cur = 1 - ls.prev_slot, prev = ls.prev_slot;
    int nq, nt; const uint8_t* qd, *td;
    if (mode == 0) {
        nq = c.n_kps[feat_cnt_idx(vl, cur, 0)]; nt = c.n_kps[feat_cnt_idx(vl, cur, 1)];
        qd = c.desc + feat_base(c, vl, cur, 0) * 32; td = c.desc + feat_base(c, vl, cur, 1) * 32;
    } else {
        if (!ls.has_prev) return;
        nq = c.n_matches[vl * 2 + prev]; nt = c.n_matches[vl * 2 + cur];
        qd = c.mdesc + feat_base(c, vl, prev, side) * 32; td = c.mdesc + feat_base(c, vl, cur, side) * 32;
    }
    if ((int)(blockIdx.y * HM_QB) >= nq || nt <= 0) return;                  // block-uniform
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    // ---- this wave's 2 x 32 queries, expanded once: Bq[set][k-step] = the 16 bits (k-step, half) of query column col ----
    hm_v4i Bq[2][8];
#pragma unroll
    for (int st = 0; st < 2; st++) {
        const int q = min((int)blockIdx.y * HM_QB + wid * 64 + st * 32 + col, nq - 1);        // past-the-end columns redo the last query, never stored
        const uint32_t* qp = (const uint32_t*)(qd + (long long)q * 32);
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = qp[k];
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) Bq[st][s8] = hm_expand16(w[s8] >> (16 * half), 0x40404040u);       // chunk s8 * 2 + half = bytes 4 s8 + 2 half ..
    }
    // ---- this block's share of the train rows, in tiles of 32 ----
    const int per = ((nt + nsplit - 1) / nsplit + 31) & ~31;
    const int j_begin = split * per, j_end = min(nt, j_begin + per);
    if (j_begin >= j_end) return;
    // staging role of this thread: row sr of the tile, packed dword sg -> chunks 2 sg, 2 sg + 1
    const int sr = tid >> 3, sg = tid & 7;
    const uint32_t* tw = (const uint32_t*)td;
    auto fetch = [&](int j0) -> uint32_t { return tw[(long long)min(j0 + sr, nt - 1) * 8 + sg]; };
    auto stage = [&](int buf, uint32_t bits) {
        // slot of (k-step sg, half h, row r) = sg * 64 + ((h * 32 + r) ^ sg): a ds_write_b128 is served in groups of 8 contiguous
        // lanes over 32 banks, and those 8 lanes hold the 8 k-steps of ONE row -- unswizzled they are 1 KB apart, i.e. on the
        // same four banks (8-way: 70 % of the kernel's LDS cycles were conflicts, profiles/r03e_pmc.json).  The wave's reads
        // stay a permutation of one contiguous KB per k-step.
        tileA[buf][sg * 64 + (sr ^ sg)] = hm_expand16(bits, 0xC0C0C0C0u);
        tileA[buf][sg * 64 + ((32 + sr) ^ sg)] = hm_expand16(bits >> 16, 0xC0C0C0C0u);
    };
    // C operand = the row of each accumulator register WITHIN its tile, the same for every tile (the matrix core reads C and
    // writes D to other registers: nothing to re-initialise): D = row + 8192 * ham - 2^20, the tile's minimum over a query
    // column's 16 rows is 8 v_min3, and the tile origin j0 is added to that ONE value before it meets the running minimum
    // (rounds 1-3 advanced sixteen train-index registers per tile and copied them into both accumulators)
    hm_v16i tc0;
#pragma unroll
    for (int v = 0; v < 16; v++) tc0[v] = 8 * (v >> 2) + (v & 3) + 4 * half;
    int best[2] = { 0x7FFFFFFF, 0x7FFFFFFF };
    uint32_t nxt = fetch(j_begin);
    stage(0, nxt);
    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += 32, buf ^= 1) {
        __syncthreads();                                           // tile `buf` is staged; the other buffer is free again
        const bool more = j0 + 32 < j_end;
        if (more) nxt = fetch(j0 + 32);
        hm_v16i acc0 = tc0, acc1 = tc0;
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
            const hm_v4i a = tileA[buf][s8 * 64 + (lane ^ s8)];
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bq[0][s8], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bq[1][s8], acc1, 0, 0, 0);
        }
        if (j0 + 32 > j_end) {                                     // ragged last tile: rows past the end can never win
#pragma unroll
            for (int v = 0; v < 16; v++) { const bool in = j0 + tc0[v] < j_end; acc0[v] = in ? acc0[v] : 0x3FFFFFFF; acc1[v] = in ? acc1[v] : 0x3FFFFFFF; }
        }
        auto min3 = [](int a, int b, int c) -> int { return min(min(a, b), c); };        // v_min3_i32
        auto min16 = [&](const hm_v16i& a) -> int {
            const int t0 = min3(a[0], a[1], a[2]), t1 = min3(a[3], a[4], a[5]), t2 = min3(a[6], a[7], a[8]);
            const int t3 = min3(a[9], a[10], a[11]), t4 = min3(a[12], a[13], a[14]);
            return min(min3(t0, t1, a[15]), min3(t2, t3, t4));
        };
        best[0] = min(best[0], min16(acc0) + j0);
        best[1] = min(best[1], min16(acc1) + j0);
        if (more) stage(buf ^ 1, nxt);
    }
    // ---- per query column: min over the two lane halves; D -> (distance << 16 | index) ----
#pragma unroll
    for (int st = 0; st < 2; st++) {
        int m = best[st];
        m = min(m, __shfl_xor(m, 32, 64));
        const int q = (int)blockIdx.y * HM_QB + wid * 64 + st * 32 + col;
        if (half == 0 && q < nq && m < 0x30000000) {
            const unsigned u = (unsigned)(m + (1 << 20));
            const unsigned packed = ((u >> 13) << 16) | (u & 8191u);
            unsigned* out = (unsigned*)c.bf_idx + ((long long)vl * 3 + (mode ? 1 + side : 0)) * c.max_kps + q;
            if (nsplit > 1) atomicMin(out, packed); else *out = packed;
        }
    }
}

// (a build capped at 128 VGPRs = 4 waves per SIMD instead of 3 pays one 16-byte spill reloaded per train tile: 50.5 against 45.4 us
// per launch at 64 lanes, profiles/r04f -- dropped)
#ifdef SVO_AB_KERNELS      // an A/B anchor, not a product form (svo_kernels.h): compiled into libsvo_hip_ab.so only
__global__ void __launch_bounds__(256) k_hamming(DevCtx c, int mode, int nsplit) { hamming_body(c, mode, nsplit); }
#endif

// ---- the same brute force on gfx950's block-scaled FP4 matrix path (round 4) ------------------------------------------------
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands FP4 (E2M1) runs at twice the int8 rate on half the operand bytes, and E2M1 holds
// +1 (0x2) and -1 (0xA) exactly: bit b of a query -> -1 / +1 ... no: query bit 0 / 1 -> +1 / -1, train bit 0 / 1 -> -1 / +1, so a
// product is -1 where the bits agree and the sum over the 256 bits is 2 ham - 256, exact in the f32 accumulator; both block scales are
// 2^0 (E8M0 127).  The C operand is row / 8192 (the row inside the 32-row tile), so D = 2 ham - 256 + row / 8192 orders by distance
// first, train index second -- cv::BFMatcher's first minimum again -- with 22 significant bits, inside f32's 24; the tile origin
// j0 / 8192 is added to the tile's minimum (v_min3_f32 x 8), exactly.  Four MFMAs per 32 x 32 tile instead of eight.
// Expansion: a byte of descriptor bits -> eight nibbles by ONE LDS table read (256 x 4 B, built by the block), where the int8
// form spends four VALU instructions per nibble.  Operand element (lane, nibble n) meets element (lane', nibble n) of the other
// operand for lanes of the same k-block (lane / 32): query and train rows are expanded by the same code, so the order of the bits
// inside a k-block does not matter.
typedef int hm_v8i __attribute__((ext_vector_type(8)));
typedef float hm_v16f __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_hamming_f4(DevCtx c, int mode, int nsplit)
{
    SVO_TL_SCOPE(c, TL_HAMMING, mode);
    SVO_LATENCY_CHAIN(c);
    __shared__ __attribute__((aligned(16))) hm_v4i tileA[2][8 * 32];           // [buffer][(2 s + kb) * 32 + (row ^ (2 s + kb))]: MFMA s, k-block kb
    __shared__ uint32_t lut[256];                                             // byte of TRAIN bits -> 8 nibbles (bit 1 -> +1 = 0x2, bit 0 -> -1 = 0xA); a query byte goes in complemented
    const int vl = blockIdx.x, lane_id = vl / c.oct_cap;
    if (vl % c.oct_cap >= c.n_oct) return;
    const int side = mode ? (blockIdx.z / nsplit) : 0, split = blockIdx.z % nsplit;
    const LaneState& ls = c.lane[lane_id];
    const int cur = 1 - ls.prev_slot, prev = ls.prev_slot;
    // mode 1 reads the descriptors of the PAIRED features straight through the pairing lists (row m of a side = the descriptor of
    // matches[m].queryIdx / .trainIdx: the gather of S4:105-131); rounds 1-3 had a kernel of its own lay them out first
    // (k_gather_mdesc, still what the int8 form reads)
    int nq, nt; const uint8_t* qd, *td; const svo_dmatch* qm = nullptr, *tm = nullptr;
    if (mode == 0) {
        nq = c.n_kps[feat_cnt_idx(vl, cur, 0)]; nt = c.n_kps[feat_cnt_idx(vl, cur, 1)];
        qd = c.desc + feat_base(c, vl, cur, 0) * 32; td = c.desc + feat_base(c, vl, cur, 1) * 32;
    } else {
        if (!ls.has_prev) return;
        nq = c.n_matches[vl * 2 + prev]; nt = c.n_matches[vl * 2 + cur];
        qd = c.desc + feat_base(c, vl, prev, side) * 32; td = c.desc + feat_base(c, vl, cur, side) * 32;
        qm = c.matches + match_base(c, vl, prev); tm = c.matches + match_base(c, vl, cur);
    }
    auto row_of = [&](const svo_dmatch* mm, int m) -> int { return mode ? (side ? mm[m].trainIdx : mm[m].queryIdx) : m; };
    if ((int)(blockIdx.y * HM_QB) >= nq || nt <= 0) return;                  // block-uniform
    const int per = ((nt + nsplit - 1) / nsplit + 31) & ~31;
    const int j_begin = split * per, j_end = min(nt, j_begin + per);
    if (j_begin >= j_end) return;                                            // block-uniform
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kb = lane >> 5;
    {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) v |= (((tid >> i) & 1) ? 0x2u : 0xAu) << (4 * i);
        lut[tid] = v;
    }
    __syncthreads();
    auto expand32 = [&](uint32_t bits) -> hm_v4i {                           // 32 bits -> 32 nibbles
        hm_v4i r;
        r.x = (int)lut[bits & 255u]; r.y = (int)lut[(bits >> 8) & 255u]; r.z = (int)lut[(bits >> 16) & 255u]; r.w = (int)lut[bits >> 24];
        return r;
    };
    // ---- this wave's 2 x 32 queries: Bq[set][s] = dword 2 s + kb of query column col, complemented, expanded ----
    hm_v4i Bq[2][4];
#pragma unroll
    for (int st = 0; st < 2; st++) {
        const int q = min((int)blockIdx.y * HM_QB + wid * 64 + st * 32 + col, nq - 1);        // past-the-end columns redo the last query, never stored
        const uint32_t* qp = (const uint32_t*)(qd + (long long)row_of(qm, q) * 32);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) Bq[st][s4] = expand32(~qp[2 * s4 + kb]);
    }
    // staging role of this thread: row sr of the tile, packed dword sg = 2 s + kb
    const int sr = tid >> 3, sg = tid & 7;
    const uint32_t* tw = (const uint32_t*)td;
    // the row index of a tile's staging role is fetched TWO tiles ahead, its dword one tile ahead: neither load waits for the other
    auto fetch_row = [&](int j0) -> int { return row_of(tm, min(j0 + sr, nt - 1)); };
    auto fetch = [&](int row) -> uint32_t { return tw[(long long)row * 8 + sg]; };
    auto stage = [&](int buf, uint32_t bits) { tileA[buf][sg * 32 + (sr ^ sg)] = expand32(bits); };      // (the XOR: see k_hamming's staging)
    hm_v16f tc0;
#pragma unroll
    for (int v = 0; v < 16; v++) tc0[v] = (float)(8 * (v >> 2) + (v & 3) + 4 * kb) * (1.0f / 8192.0f);
    float best[2] = { 3.0e38f, 3.0e38f };
    const int one = 127;                                                      // E8M0 block scale 2^0
    // Software pipeline (round 5): the dword of tile t + 1 is in a register when tile t starts, so its expansion and LDS write are
    // issued right behind tile t's operand reads and complete under tile t's MFMAs (they sat behind the minima, on the serial chain
    // barrier -> reads -> MFMAs -> minima -> four table reads -> write -> barrier, until round 4); rows are fetched three tiles ahead,
    // dwords two.  The other buffer is free from the barrier on: every wave finished tile t - 1's reads before it arrived there.
    stage(0, fetch(fetch_row(j_begin)));
    uint32_t nxt = j_begin + 32 < j_end ? fetch(fetch_row(j_begin + 32)) : 0u;
    int nrow = fetch_row(min(j_begin + 64, j_end - 1));
    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += 32, buf ^= 1) {
        __syncthreads();                                           // tile `buf` is staged; the other buffer is free again
        hm_v4i a4s[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) a4s[s4] = tileA[buf][(2 * s4 + kb) * 32 + (col ^ (2 * s4 + kb))];
        if (j0 + 32 < j_end) stage(buf ^ 1, nxt);
        if (j0 + 64 < j_end) { nxt = fetch(nrow); nrow = fetch_row(min(j0 + 96, j_end - 1)); }
        hm_v16f acc0 = tc0, acc1 = tc0;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const hm_v4i a4 = a4s[s4];
            const hm_v8i a = { a4.x, a4.y, a4.z, a4.w, 0, 0, 0, 0 };
            const hm_v8i b0 = { Bq[0][s4].x, Bq[0][s4].y, Bq[0][s4].z, Bq[0][s4].w, 0, 0, 0, 0 }, b1 = { Bq[1][s4].x, Bq[1][s4].y, Bq[1][s4].z, Bq[1][s4].w, 0, 0, 0, 0 };
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, acc0, 4, 4, 0, one, 0, one);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, acc1, 4, 4, 0, one, 0, one);
        }
        if (j0 + 32 > j_end) {                                     // ragged last tile: rows past the end can never win
#pragma unroll
            for (int v = 0; v < 16; v++) { const bool in = j0 + 8 * (v >> 2) + (v & 3) + 4 * kb < j_end; acc0[v] = in ? acc0[v] : 1.0e9f; acc1[v] = in ? acc1[v] : 1.0e9f; }
        }
        // (as asm: there are no NaNs here, and fminf's quieting adds a v_max_f32 x, x per operand)
        auto min3 = [](float a, float b, float c) -> float { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
        auto min2 = [](float a, float b) -> float { float r; asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
        auto min16 = [&](const hm_v16f& a) -> float {
            const float t0 = min3(a[0], a[1], a[2]), t1 = min3(a[3], a[4], a[5]), t2 = min3(a[6], a[7], a[8]);
            const float t3 = min3(a[9], a[10], a[11]), t4 = min3(a[12], a[13], a[14]);
            return min2(min3(t0, t1, a[15]), min3(t2, t3, t4));
        };
        const float o = (float)j0 * (1.0f / 8192.0f);
        best[0] = min2(best[0], min16(acc0) + o);
        best[1] = min2(best[1], min16(acc1) + o);
    }
    // ---- per query column: min over the two lane halves; D = 2 ham - 256 + index / 8192 -> (distance << 16 | index) ----
#pragma unroll
    for (int st = 0; st < 2; st++) {
        float m = best[st];
        m = fminf(m, __shfl_xor(m, 32, 64));
        const int q = (int)blockIdx.y * HM_QB + wid * 64 + st * 32 + col;
        if (kb == 0 && q < nq && m < 1.0e8f) {
            const float fl = floorf(m);
            const unsigned ham = (unsigned)((int)fl + 256) >> 1, idx = (unsigned)(int)((m - fl) * 8192.0f);
            const unsigned packed = (ham << 16) | idx;
            unsigned* out = (unsigned*)c.bf_idx + ((long long)vl * 3 + (mode ? 1 + side : 0)) * c.max_kps + q;
            if (nsplit > 1) atomicMin(out, packed); else *out = packed;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K8a: stage-3 filters (S3:124-175), one 1024-thread block per lane.
//   1-to-1: per right feature keep the left with the smallest (distance, left index)   [S3:127-147]
//   epipolar / threshold / disparity with the reference's int truncations                [S3:159-168]
// then an order-preserving compaction (pairings stay in ascending left index = ascending row).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_match_lr_filter(DevCtx c, int one_to_one, double max_y_diff)
{
    SVO_TL_SCOPE(c, TL_LR_FILTER, 0);
    SVO_LATENCY_CHAIN(c);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* right_best = (unsigned*)smem;              // max_kps
    int* scan = (int*)(right_best + c.max_kps);          // 32
    const int vl = blockIdx.x, lane_id = vl / c.oct_cap, oct = vl % c.oct_cap, tid = threadIdx.x;
    if (oct >= c.n_oct) return;
    const LaneState& ls = c.lane[lane_id];
    const int cur = 1 - ls.prev_slot;
    const int nl = c.n_kps[feat_cnt_idx(vl, cur, 0)], nr = c.n_kps[feat_cnt_idx(vl, cur, 1)];
    const svo_keypoint* kl = c.kps + feat_base(c, vl, cur, 0), *kr = c.kps + feat_base(c, vl, cur, 1);
    const unsigned* packed = (const unsigned*)c.bf_idx + ((long long)vl * 3 + 0) * c.max_kps;
    svo_dmatch* out = c.matches + match_base(c, vl, cur);
    for (int j = tid; j < nr; j += blockDim.x) right_best[j] = 0xFFFFFFFFu;
    __syncthreads();
    const bool any = nl > 0 && nr > 0;
    if (any && one_to_one)
        for (int i = tid; i < nl; i += blockDim.x) { const unsigned p = packed[i]; atomicMin(&right_best[p & 0xFFFFu], (p & 0xFFFF0000u) | (unsigned)i); }
    __syncthreads();
    int m_total = 0;
    const int n_iter = any ? (nl + (int)blockDim.x - 1) / (int)blockDim.x : 0;
    for (int it = 0; it < n_iter; it++) {
        const int i = it * blockDim.x + tid;
        int keep = 0; unsigned p = 0;
        if (i < nl) {
            p = packed[i];
            const int j = (int)(p & 0xFFFFu);
            const float distance = (float)(p >> 16);
            keep = 1;
            if (one_to_one && (int)(right_best[j] & 0xFFFFu) != i) keep = 0;
            const int diff = (int)(kl[i].y - kr[j].y);                                   // S3:162
            const int disp = (int)(kl[i].x - kr[j].x);                                   // S3:163
            if ((double)abs(diff) > max_y_diff || distance > (float)c.orb_th || (double)disp < 1.0 || (double)disp > (double)c.ow[oct]) keep = 0;
        }
        int tot;
        const int off = block_exclusive_scan(keep, scan, &tot);
        if (keep) { svo_dmatch d; d.queryIdx = i; d.trainIdx = (int)(p & 0xFFFFu); d.imgIdx = 0; d.distance = (float)(p >> 16); out[m_total + off] = d; }
        m_total += tot;
        __syncthreads();
    }
    if (tid == 0) { c.n_matches[vl * 2 + cur] = m_total; c.results[lane_id].stereo_matches[oct] = m_total; }
    // matches_lr_row_index (stage3_match_left_right.cpp:425-445): ri[y] = #pairings whose left keypoint has y <= y - 1
    // (pairings are in ascending row order: binary search); ri[H] = M (documented deviation from S3:443, oracle too)
    __threadfence_block();
    __syncthreads();
    {
        int* ri = c.mrow_index + (long long)(vl * 2 + cur) * (c.max_h + 1);
        const int H = c.oh[oct];
        for (int y = tid; y <= H; y += blockDim.x) {
            int v = m_total;
            if (y < H) {
                int lo = 0, hi = m_total;                      // first pairing with left y > y - 1
                const float lim = (float)(y - 1);
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (kl[out[mid].queryIdx].y <= lim) lo = mid + 1; else hi = mid; }
                v = y == 0 ? 0 : lo;
            }
            ri[y] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K8a': smDescRbR -- row-by-row left-right matching (stage3_match_left_right.cpp:185-419) with its quirks kept
// (SURVEY.md appendix A #10): the left keypoints of row r are visited in iteration y = r - 1 and meet the right
// keypoints of rows (y - d, y + d]; byte-wise popcount accumulated in a uint8_t (256 wraps to 0); the ratio test has
// no effect; keypoints of the last occupied row never match (their row-table range is empty).
// One 256-thread block per lane-octave.  The reference's sequential assignment (first claimant, or best claimant with
// enable_robust_1to1_match) is order-independent once written as a minimum over the claimants of a right feature.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_match_lr_rbr(DevCtx c, int one_to_one, double max_y_diff, double minimum_response, int max_distance)
{
    SVO_TL_SCOPE(c, TL_LR_FILTER, 1);
    SVO_LATENCY_CHAIN(c);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* right_best = (unsigned*)smem;              // max_kps: per right feature, min over its claimants
    unsigned* left_pick = right_best + c.max_kps;        // max_kps: per left feature (min_idx << 8 | min_1) or ~0
    int* scan = (int*)(left_pick + c.max_kps);           // 32
    const int vl = blockIdx.x, lane_id = vl / c.oct_cap, oct = vl % c.oct_cap, tid = threadIdx.x;
    if (oct >= c.n_oct) return;
    const LaneState& ls = c.lane[lane_id];
    const int cur = 1 - ls.prev_slot;
    const int nl = c.n_kps[feat_cnt_idx(vl, cur, 0)], nr = c.n_kps[feat_cnt_idx(vl, cur, 1)];
    const svo_keypoint* kl = c.kps + feat_base(c, vl, cur, 0), *kr = c.kps + feat_base(c, vl, cur, 1);
    const uint8_t* dl = c.desc + feat_base(c, vl, cur, 0) * 32, *dr = c.desc + feat_base(c, vl, cur, 1) * 32;
    const int* idxL = c.row_index + (long long)feat_cnt_idx(vl, cur, 0) * c.max_h, *idxR = c.row_index + (long long)feat_cnt_idx(vl, cur, 1) * c.max_h;
    svo_dmatch* out = c.matches + match_base(c, vl, cur);
    const int W = c.ow[oct], H = c.oh[oct];
    const int max_disparity = (int)((double)W * 0.7);                          // S3:247
    const int d_round = (int)round(max_y_diff);                                 // S3:254-255
    for (int j = tid; j < nr; j += blockDim.x) right_best[j] = 0xFFFFFFFFu;
    for (int i = tid; i < nl; i += blockDim.x) left_pick[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int iL = tid; iL < nl; iL += blockDim.x) {
        const svo_keypoint fL = kl[iL];
        const int y = (int)fL.y - 1;                                            // the iteration that visits this keypoint
        if (y < 0 || y + 1 > H - 1) continue;
        if (!(idxL[y] <= iL && iL < idxL[y + 1])) continue;                     // S3:253, 265 (empty / wrapped ranges)
        const int mrr = y - d_round, xrr = y + d_round;
        const int R0 = idxR[mrr > 0 ? mrr : 0], R1 = idxR[xrr < H - 1 ? xrr : H - 1];   // S3:254-256
        const ulonglong2 qa = ((const ulonglong2*)(dl + (long long)iL * 32))[0], qb = ((const ulonglong2*)(dl + (long long)iL * 32))[1];
        unsigned min_1 = 0xFFFFFFFFu; int min_idx = -1;
        for (int iR = R0; iR < R1; iR++) {                                      // S3:274 (R1 <= R0: no candidates)
            const svo_keypoint fR = kr[iR];
            if ((double)fL.response < minimum_response || (double)fR.response < minimum_response) continue;   // S3:279
            const int disparity = (int)(fL.x - fR.x);                           // S3:283
            if (disparity < 1 || disparity > max_disparity) continue;
            const ulonglong2 ta = ((const ulonglong2*)(dr + (long long)iR * 32))[0], tb = ((const ulonglong2*)(dr + (long long)iR * 32))[1];
            const unsigned dist = (unsigned)(__popcll(qa.x ^ ta.x) + __popcll(qa.y ^ ta.y) + __popcll(qb.x ^ tb.x) + __popcll(qb.y ^ tb.y)) & 0xFFu;   // uint8_t accumulator (S3:321-331)
            if ((int)dist > max_distance) continue;                             // S3:334
            if (dist < min_1) { min_1 = dist; min_idx = iR; }                   // S3:338-343 (first minimum)
        }
        if (min_idx >= 0) {
            left_pick[iL] = ((unsigned)min_idx << 8) | min_1;
            // S3:359-387: best claimant (robust) or first claimant wins the right feature
            atomicMin(&right_best[min_idx], one_to_one ? ((min_1 << 16) | (unsigned)iL) : (unsigned)iL);
        }
    }
    __syncthreads();
    int m_total = 0;
    const int n_iter = (nl + (int)blockDim.x - 1) / (int)blockDim.x;
    for (int it = 0; it < n_iter; it++) {
        const int i = it * blockDim.x + tid;
        int keep = 0; unsigned pk = 0xFFFFFFFFu;
        if (i < nl && (pk = left_pick[i]) != 0xFFFFFFFFu) keep = (int)(right_best[pk >> 8] & 0xFFFFu) == i;
        int tot;
        const int off = block_exclusive_scan(keep, scan, &tot);
        if (keep) { svo_dmatch d; d.queryIdx = i; d.trainIdx = (int)(pk >> 8); d.imgIdx = -1; d.distance = (float)(pk & 0xFFu); out[m_total + off] = d; }   // DMatch(i, fr, d): S3:404
        m_total += tot;
        __syncthreads();
    }
    if (tid == 0) { c.n_matches[vl * 2 + cur] = m_total; c.results[lane_id].stereo_matches[oct] = m_total; }
    __threadfence_block();
    __syncthreads();
    {
        int* ri = c.mrow_index + (long long)(vl * 2 + cur) * (c.max_h + 1);
        for (int y = tid; y <= H; y += blockDim.x) {
            int v = m_total;
            if (y < H) {
                int lo = 0, hi = m_total;
                const float lim = (float)(y - 1);
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (kl[out[mid].queryIdx].y <= lim) lo = mid + 1; else hi = mid; }
                v = y == 0 ? 0 : lo;
            }
            ri[y] = v;
        }
    }
}

// (defined with the RANSAC kernels below) phase 0 of the sample schedule runs at the end of the tracker kernels: same block shape, one launch less
__device__ __forceinline__ void rs_schedule_block(const DevCtx& c, int vl, int phase, int n, int* scan);
// Eight to fourteen pairs: cv::findFundamentalMat runs its LMedS registrator, not the RANSAC (oracle v6: lmeds_fundamental) -- a fixed
// budget of SVO_LMEDS_ITERS samples from the same generator, every model of every one of them ranked by its median error (k_track_finalize).
__device__ __forceinline__ bool rs_is_lmeds(int n) { return n >= 8 && n <= SVO_LMEDS_MAX_N; }
__device__ __forceinline__ int rs_first_bound(int n) { return rs_is_lmeds(n) ? SVO_LMEDS_ITERS : SVO_RANSAC_HYP; }      // rs_bound before any count

// ------------------------------------------------------------------------------------------------------------
// K8c: ifmDescWin -- window tracker (stage4_match_consecutive.cpp:435-738) with its quirks kept (appendix A #12):
// ifm_win_w is the VERTICAL half-size and ifm_win_h the horizontal one; left descriptors only, uint8_t accumulator,
// no distance threshold, best = strict "<" from 255.  A previous pairing of row iteration y meets the current pairings
// of rows [y - WIN_W, y + WIN_W]; a current pairing keeps its best claimant (first on ties).  Survivors are listed in
// ascending current index and handed to the same RANSAC kernels as the brute-force tracker.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_track_win(DevCtx c, int WIN_W, int WIN_H)
{
    SVO_TL_SCOPE(c, TL_TRK_FILTER, 1);
    SVO_LATENCY_CHAIN(c);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* cur_best = (unsigned*)smem;                // max_kps: (dist << 16 | pi) min over claimants
    int* scan = (int*)(cur_best + c.max_kps);
    const int vl = blockIdx.x, lane_id = vl / c.oct_cap, oct = vl % c.oct_cap, tid = threadIdx.x;
    if (oct >= c.n_oct) return;
    const LaneState& ls = c.lane[lane_id];
    if (!ls.has_prev) { if (tid == 0) c.trk_nk[vl] = 0; return; }
    const int cur = 1 - ls.prev_slot, prev = ls.prev_slot;
    const int npm = c.n_matches[vl * 2 + prev], ncm = c.n_matches[vl * 2 + cur];
    const svo_dmatch* pm = c.matches + match_base(c, vl, prev), *cm = c.matches + match_base(c, vl, cur);
    const svo_keypoint* pkl = c.kps + feat_base(c, vl, prev, 0), *pkr = c.kps + feat_base(c, vl, prev, 1);
    const svo_keypoint* ckl = c.kps + feat_base(c, vl, cur, 0), *ckr = c.kps + feat_base(c, vl, cur, 1);
    const uint8_t* pdl = c.desc + feat_base(c, vl, prev, 0) * 32, *cdl = c.desc + feat_base(c, vl, cur, 0) * 32;
    const int* ri_p = c.mrow_index + (long long)(vl * 2 + prev) * (c.max_h + 1), *ri_c = c.mrow_index + (long long)(vl * 2 + cur) * (c.max_h + 1);
    const int W = c.ow[oct], H = c.oh[oct];
    const int awx = W - 1, awy = H - 1;                                          // S4:489-490 (descriptor variant)
    for (int i = tid; i < ncm; i += blockDim.x) cur_best[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int pi = tid; pi < npm; pi += blockDim.x) {
        const svo_keypoint pl = pkl[pm[pi].queryIdx], pr = pkr[pm[pi].trainIdx];
        const int y = (int)ceilf(pl.y);                                          // the row iteration whose range [ri[y], ri[y+1]) holds pi
        if (y < 0 || y >= H - 1) continue;                                       // S4:514
        if (!(ri_p[y] <= pi && pi < ri_p[y + 1])) continue;
        const int wy_min = (y - WIN_W) > 0 ? (y - WIN_W) : 0, wy_max = awy < (y + WIN_W) ? awy : (y + WIN_W);   // S4:525-526
        const int c0 = ri_c[wy_min], c1 = ri_c[wy_max + 1];                      // S4:529-530
        const int a = (int)(pl.x - (float)WIN_H), b = (int)(pl.x + (float)WIN_H), cc = (int)(pr.x - (float)WIN_H), d = (int)(pr.x + (float)WIN_H);
        const int wxl0 = a > 0 ? a : 0, wxl1 = awx < b ? awx : b, wxr0 = cc > 0 ? cc : 0, wxr1 = awx < d ? awx : d;   // S4:552-555
        const ulonglong2 qa = ((const ulonglong2*)(pdl + (long long)pm[pi].queryIdx * 32))[0], qb = ((const ulonglong2*)(pdl + (long long)pm[pi].queryIdx * 32))[1];
        int best_c = -1; unsigned best_orb = 255;                                // S4:543-545
        for (int ci = c0; ci < c1; ci++) {
            const svo_dmatch mc = cm[ci];
            const svo_keypoint fl = ckl[mc.queryIdx], fr = ckr[mc.trainIdx];
            if (fl.x < (float)wxl0 || fl.x > (float)wxl1 || fr.x < (float)wxr0 || fr.x > (float)wxr1) continue;   // S4:567
            const ulonglong2 ta = ((const ulonglong2*)(cdl + (long long)mc.queryIdx * 32))[0], tb = ((const ulonglong2*)(cdl + (long long)mc.queryIdx * 32))[1];
            const unsigned orb_l = (unsigned)(__popcll(qa.x ^ ta.x) + __popcll(qa.y ^ ta.y) + __popcll(qb.x ^ tb.x) + __popcll(qb.y ^ tb.y)) & 0xFFu;   // S4:596-609
            if (orb_l < best_orb) { best_orb = orb_l; best_c = ci; }             // S4:614-618
        }
        if (best_c >= 0) atomicMin(&cur_best[best_c], (best_orb << 16) | (unsigned)pi);   // S4:622-636
    }
    __syncthreads();
    int* kq = c.trk_kq + (long long)vl * c.max_kps;                              // previous pairing index of survivor i
    int* cq = c.bf_idx + ((long long)vl * 3 + 1) * c.max_kps;                    // current pairing index of survivor i
    float* ptsL = c.trk_pts + ((long long)vl * 2 + 0) * c.max_kps * 4, *ptsR = c.trk_pts + ((long long)vl * 2 + 1) * c.max_kps * 4;
    int np_total = 0;
    const int n_iter = (ncm + (int)blockDim.x - 1) / (int)blockDim.x;
    for (int it = 0; it < n_iter; it++) {                                        // S4:640-679: ascending current index
        const int i = it * blockDim.x + tid;
        unsigned v = 0xFFFFFFFFu;
        const int keep = (i < ncm && (v = cur_best[i]) != 0xFFFFFFFFu) ? 1 : 0;
        int tot;
        const int off = block_exclusive_scan(keep, scan, &tot);
        if (keep) {
            const int pi = (int)(v & 0xFFFFu), o = np_total + off;
            kq[o] = pi; cq[o] = i;
            const svo_keypoint a = pkl[pm[pi].queryIdx], b = ckl[cm[i].queryIdx], e = pkr[pm[pi].trainIdx], f = ckr[cm[i].trainIdx];
            ptsL[o * 4] = a.x; ptsL[o * 4 + 1] = a.y; ptsL[o * 4 + 2] = b.x; ptsL[o * 4 + 3] = b.y;
            ptsR[o * 4] = e.x; ptsR[o * 4 + 1] = e.y; ptsR[o * 4 + 2] = f.x; ptsR[o * 4 + 3] = f.y;
        }
        np_total += tot;
        __syncthreads();
    }
    if (tid == 0) { atomicAdd(&c.results[lane_id].track_stats[SVO_TS_THRESHOLD], np_total); atomicAdd(&c.results[lane_id].track_stats[SVO_TS_COLLISION], np_total); }
    if (tid == 0) { c.trk_nk[vl] = np_total; c.rs_bound[vl * 2] = c.rs_bound[vl * 2 + 1] = rs_first_bound(np_total); c.rs_floor[vl * 4] = c.rs_floor[vl * 4 + 1] = c.rs_floor[vl * 4 + 2] = c.rs_floor[vl * 4 + 3] = 6; }
    __syncthreads();                                                       // the point pairs are written: the sampler's collinearity test reads them
    rs_schedule_block(c, vl, 0, np_total, scan);
}

// ------------------------------------------------------------------------------------------------------------
// K8b: the tracker's joint sequential filter (S4:145-160), exact, one 256-thread block per lane-octave.
// Reference: walk k = 0, 1, ...; k is kept iff dL <= th and dR <= th and neither its left nor its right train index was
// taken by an EARLIER KEPT k.  That is a greedy matching in index order, and it parallelises by rounds: among the still
// undecided candidates, every k that is the smallest undecided claimant of BOTH its train indices (and whose train indices
// are free) is kept at once -- everything earlier that competes for them has been decided, and decided means rejected,
// or the index would not be free -- then every undecided k whose left or right index was just taken is rejected (by a
// smaller kept k, as in the walk).  The smallest undecided k overall always qualifies, so a round always makes progress;
// chains of conflicts are a handful long.  A thread owns k = tid, tid + 256, ... in registers; the per-train-index
// minima live in LDS.  (One wave walking 64 entries at a time took 70 us; the rounds take a few.)
// Also gathers the pixel pairs for the two RANSACs (S4:181-189, 216-224).
// ------------------------------------------------------------------------------------------------------------
// TF_ITEMS x 256 threads >= max_kps: 16 (lists up to 4096) or 32
template <int TF_ITEMS>
__global__ void __launch_bounds__(256) k_track_filter(DevCtx c)
{
    SVO_TL_SCOPE(c, TL_TRK_FILTER, 0);
    SVO_LATENCY_CHAIN(c);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* firstL = (unsigned*)smem;                    // max_kps: smallest undecided k claiming left train index i
    unsigned* firstR = firstL + c.max_kps;                 // max_kps
    unsigned* takenL = firstR + c.max_kps;                 // max_kps / 32 bits
    unsigned* takenR = takenL + c.max_kps / 32;
    __shared__ int scan[40];
    __shared__ int s_und, s_th;
    const int vl = blockIdx.x, lane_id = vl / c.oct_cap, tid = threadIdx.x;
    if (vl % c.oct_cap >= c.n_oct) return;
    const LaneState& ls = c.lane[lane_id];
    if (!ls.has_prev) { if (tid == 0) c.trk_nk[vl] = 0; return; }
    const int cur = 1 - ls.prev_slot, prev = ls.prev_slot;
    const int npm = c.n_matches[vl * 2 + prev], ncm = c.n_matches[vl * 2 + cur];
    if (npm <= 0 || ncm <= 0) { if (tid == 0) c.trk_nk[vl] = 0; return; }
    const unsigned* gL = (const unsigned*)c.bf_idx + ((long long)vl * 3 + 1) * c.max_kps;
    const unsigned* gR = (const unsigned*)c.bf_idx + ((long long)vl * 3 + 2) * c.max_kps;
    // state per owned k: 0 rejected, 1 kept, 2 undecided
    // a thread's candidates k = tid + 256 j: (train indices, state).  Up to 16 per thread they live in registers; the 32 of a max_kps = 8192
    // context spilled there (512 VGPRs + 1.4 KB of scratch per thread, hipcc's report): those live in LDS behind the tables above
    unsigned tlr_r[TF_ITEMS <= 16 ? TF_ITEMS : 1]; unsigned char st_r[TF_ITEMS <= 16 ? TF_ITEMS : 1];
    unsigned* tlr_s = takenR + c.max_kps / 32; unsigned char* st_s = (unsigned char*)(tlr_s + (TF_ITEMS <= 16 ? 0 : TF_ITEMS * 256));
    auto TLR = [&](int j) -> unsigned& { if constexpr (TF_ITEMS <= 16) return tlr_r[j]; else return tlr_s[j * 256 + tid]; };
    auto ST = [&](int j) -> unsigned char& { if constexpr (TF_ITEMS <= 16) return st_r[j]; else return st_s[j * 256 + tid]; };
#pragma unroll
    for (int j = 0; j < TF_ITEMS; j++) {
        const int k = tid + 256 * j;
        TLR(j) = 0; ST(j) = 0;
        if (k < npm) {
            const unsigned a = gL[k], b = gR[k];
            TLR(j) = (a & 0xFFFFu) | (b << 16);                                   // tl | tr << 16
            ST(j) = ((float)(a >> 16) > (float)c.orb_th || (float)(b >> 16) > (float)c.orb_th) ? 0 : 2;      // S4:149
            if ((int)(a & 0xFFFFu) >= ncm || (int)(b & 0xFFFFu) >= ncm) { ST(j) = 0; atomicOr(&c.status[lane_id], SVO_ST_INTERNAL); }   // no match was written for k: never the case on a sound frame
        }
    }
    {   // svo_result.track_stats[SVO_TS_THRESHOLD]: candidates that pass the distance threshold on both sides
        int nth = 0;
#pragma unroll
        for (int j = 0; j < TF_ITEMS; j++) nth += ST(j) == 2 ? 1 : 0;
        if (tid == 0) s_th = 0;
        __syncthreads();
        nth = wave_sum_uniform(nth);
        if ((tid & 63) == 0) atomicAdd(&s_th, nth);
    }
    for (int i = tid; i < c.max_kps / 32; i += 256) { takenL[i] = 0; takenR[i] = 0; }
    const int n_items = (npm + 255) / 256;
    // every round decides at least the smallest undecided candidate, so npm rounds always suffice: the bound only matters if the
    // lists handed to this kernel are corrupt (a train index outside [0, ncm) would walk out of the LDS tables) -- then the lane is
    // flagged (SVO_ST_INTERNAL) and the loop left instead of spinning for ever
    for (int round = 0;; round++) {
        if (round > npm) { if (tid == 0) { atomicOr(&c.status[lane_id], SVO_ST_INTERNAL); atomicOr(&c.results[lane_id].status, (int)SVO_ST_INTERNAL); } break; }
        for (int i = tid; i < ncm; i += 256) { firstL[i] = 0xFFFFFFFFu; firstR[i] = 0xFFFFFFFFu; }
        if (tid == 0) s_und = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TF_ITEMS; j++)
            if (j < n_items && ST(j) == 2) { atomicMin(&firstL[TLR(j) & 0xFFFFu], (unsigned)(tid + 256 * j)); atomicMin(&firstR[TLR(j) >> 16], (unsigned)(tid + 256 * j)); }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TF_ITEMS; j++)
            if (j < n_items && ST(j) == 2) {
                const unsigned k = (unsigned)(tid + 256 * j), tl = TLR(j) & 0xFFFFu, tr = TLR(j) >> 16;
                if (firstL[tl] == k && firstR[tr] == k) { ST(j) = 1; atomicOr(&takenL[tl >> 5], 1u << (tl & 31)); atomicOr(&takenR[tr >> 5], 1u << (tr & 31)); }
            }
        __syncthreads();
        bool und = false;
#pragma unroll
        for (int j = 0; j < TF_ITEMS; j++)
            if (j < n_items && ST(j) == 2) {
                const unsigned tl = TLR(j) & 0xFFFFu, tr = TLR(j) >> 16;
                if (((takenL[tl >> 5] >> (tl & 31)) & 1u) || ((takenR[tr >> 5] >> (tr & 31)) & 1u)) ST(j) = 0; else und = true;
            }
        if (und) s_und = 1;
        __syncthreads();
        const int again = s_und;
        __syncthreads();
        if (!again) break;
    }
    // kept entries in ascending k: k = tid + 256 j is j-major, so one block scan per j
    const svo_dmatch* pm = c.matches + match_base(c, vl, prev), *cm = c.matches + match_base(c, vl, cur);
    const svo_keypoint* pkl = c.kps + feat_base(c, vl, prev, 0), *pkr = c.kps + feat_base(c, vl, prev, 1);
    const svo_keypoint* ckl = c.kps + feat_base(c, vl, cur, 0), *ckr = c.kps + feat_base(c, vl, cur, 1);
    int* kq = c.trk_kq + (long long)vl * c.max_kps;
    float* ptsL = c.trk_pts + ((long long)vl * 2 + 0) * c.max_kps * 4, *ptsR = c.trk_pts + ((long long)vl * 2 + 1) * c.max_kps * 4;
    int nk = 0;
#pragma unroll
    for (int j = 0; j < TF_ITEMS; j++) {
        if (j >= n_items) break;                                                    // block-uniform
        const int keep = ST(j) == 1 ? 1 : 0;
        int tot;
        const int off = block_exclusive_scan(keep, scan, &tot);
        if (keep) {
            const int o = nk + off, k = tid + 256 * j;
            const int tl = (int)(TLR(j) & 0xFFFFu), tr = (int)(TLR(j) >> 16);
            kq[o] = k;
            const svo_dmatch mp = pm[k];
            const svo_keypoint a = pkl[mp.queryIdx], b = ckl[cm[tl].queryIdx];
            ptsL[o * 4] = a.x; ptsL[o * 4 + 1] = a.y; ptsL[o * 4 + 2] = b.x; ptsL[o * 4 + 3] = b.y;
            const svo_keypoint e = pkr[mp.trainIdx], f = ckr[cm[tr].trainIdx];
            ptsR[o * 4] = e.x; ptsR[o * 4 + 1] = e.y; ptsR[o * 4 + 2] = f.x; ptsR[o * 4 + 3] = f.y;
        }
        nk += tot;
        __syncthreads();
    }
    if (tid == 0) { atomicAdd(&c.results[lane_id].track_stats[SVO_TS_THRESHOLD], s_th); atomicAdd(&c.results[lane_id].track_stats[SVO_TS_COLLISION], nk); }
    if (tid == 0) { c.trk_nk[vl] = nk; c.rs_bound[vl * 2] = c.rs_bound[vl * 2 + 1] = rs_first_bound(nk); c.rs_floor[vl * 4] = c.rs_floor[vl * 4 + 1] = c.rs_floor[vl * 4 + 2] = c.rs_floor[vl * 4 + 3] = 6; }
    __syncthreads();                                                       // the point pairs are written: the sampler's collinearity test reads them
    rs_schedule_block(c, vl, 0, nk, scan);
}

// ------------------------------------------------------------------------------------------------------------
// K9: fundamental-matrix RANSAC -- cv::findFundamentalMat(FM_RANSAC, 1.0, 0.99) as called at S4:202, 237 (oracle:
// svo_oracle_ransac_fundamental).  Fixed schedule of SVO_RANSAC_HYP seeded MINIMAL samples of seven pairs; each sample yields
// one or three models (the 7-point algorithm: null space of the 7x9 system + the cubic det = 0), every model is scored, and
// the adaptive stop of the sequential loop is emulated afterwards by scanning the counts in (sample, model) order.
//
// Where the models live: SAMPLES are grouped in regions of SVO_RANSAC_REG = 16 consecutive ones, a region owns
// SVO_RANSAC_RSLOTS = 48 model SLOTS (3 per sample), and the models of a region's samples are packed to the front of its slots
// in (sample, root) order -- rs_nvalid[region] of them, rs_k[slot] naming the sample of each.  The count kernels work on groups
// of 4 / 16 consecutive slots and drop a group that starts beyond rs_nvalid; the unused slots of a live group of 16 hold a filler
// matrix that no pair fits (count 0), and every count of a region is zeroed when it is generated, so a slot that was never
// scored reads as 0 = "no record".  Budgets and bounds (rs_bound, rs_gen, K(count)) are in SAMPLES, as the oracle's loop counter.
// ------------------------------------------------------------------------------------------------------------

// The sequential RANSAC's stop rule (oracle: svo_oracle_ransac_fundamental): samples are visited in order; a model whose count
// exceeds the best so far (and 6 = modelPoints - 1) becomes the result and shrinks the iteration budget to
// cv::RANSACUpdateNumIters(0.99, 1 - count / n, 7, budget).  svo_ln / ransac_niters repeat the oracle's functions operation
// by operation (+, -, *, / only, no contraction), so both sides round alike.
__device__ __forceinline__ double svo_ln(double x)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 0.058823529411764705;
    p = p * s2 + 0.066666666666666666;
    p = p * s2 + 0.076923076923076927;
    p = p * s2 + 0.090909090909090912;
    p = p * s2 + 0.1111111111111111;
    p = p * s2 + 0.14285714285714285;
    p = p * s2 + 0.2;
    p = p * s2 + 0.33333333333333331;
    p = p * s2 + 1.0;
    return (double)e * 0.69314718055994529 + 2.0 * (s * p);
}
__device__ __forceinline__ int ransac_niters(int cnt, int n, int max_iters)
{
    const double w = (double)cnt / (double)n, w2 = w * w, w4 = w2 * w2, w7 = (w4 * w2) * w;
    const double denom = 1.0 - w7;
    if (denom < 2.2250738585072014e-308) return 0;
    const double num = -4.6051701859880909;
    const double d = svo_ln(denom);
    if (d >= 0.0 || -num >= (double)max_iters * (-d)) return max_iters;
    return (int)rint(num / d);
}

// oracle: ransac_get_subset -- the seven indices of sample h, drawn by cv::RNG under getSubset's rules: k_ransac_schedule below wrote them
__device__ __forceinline__ void ransac_sample7(const DevCtx& c, int vl, int side, int h, int (&s)[7])
{
    const uint4 w = *(const uint4*)(c.rs_smp + (((long long)vl * 2 + side) * SVO_RANSAC_PAD + h) * 8);
    s[0] = (int)(w.x & 0xFFFFu); s[1] = (int)(w.x >> 16); s[2] = (int)(w.y & 0xFFFFu); s[3] = (int)(w.y >> 16);
    s[4] = (int)(w.z & 0xFFFFu); s[5] = (int)(w.z >> 16); s[6] = (int)(w.w & 0xFFFFu);
}

// ------------------------------------------------------------------------------------------------------------
// The sample schedule of cv::findFundamentalMat's RANSAC (oracle v5: cv_rng_next, ransac_get_subset, have_collinear).
// OpenCV draws from ONE multiply-with-carry stream per call, seeded (uint64)-1: draw after draw, an index that repeats inside an attempt
// is drawn again, an attempt whose last point is collinear with two earlier ones (in either image) is drawn afresh -- so WHERE in the
// stream sample k starts depends on everything before it.  But the stream is the same for every call, and the ATTEMPTS (which draws
// form attempt a, duplicates rejected) depend on the point count n alone: svo_create tabulates them on the host, once, for every n a
// context can meet -- c.rs_att[n][a] = the seven indices of attempt a (SVO_RS_ATT attempts per n; built and measured on the device first:
// a parallel orbit search over the raw stream, 32-90 us per frame and lane; the table makes it a lookup).  What is left per frame is the
// part that depends on the data: the collinearity test of each attempt on the lane's point pairs, per side, and the numbering of the
// attempts that pass -- sample k of a side = its k-th passing attempt.  A block per lane-octave, an attempt per thread.
// Phase 0 (before chunk 0) covers SVO_RANSAC_CHUNK1 samples per side, phase 1 (before chunk 2) continues to SVO_RANSAC_HYP for the lanes
// whose budget still reaches that far.  Running out of tabulated attempts before that (point sets on which nearly every sample is
// collinear) ends the schedule early: the hypothesis kernels generate what exists, and SVO_ST_INTERNAL is raised if the sequential
// algorithm could have gone further.
// ------------------------------------------------------------------------------------------------------------
// Is the last of seven points collinear with two earlier ones (haveCollinearPoints)?  The decision is the oracle's double-precision one;
// a single-precision screen settles the pairs that are nowhere near the FLT_EPSILON-relative threshold first (|cross| above 1e-3 of the
// scale, where a float's 6e-8 relative rounding cannot matter) -- on real point sets all of them.
__device__ __forceinline__ bool rs_collinear7(const float (&x)[7], const float (&y)[7])
{
    bool maybe = false;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const float dx1 = x[j] - x[6], dy1 = y[j] - y[6];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < j) {
            const float dx2 = x[k] - x[6], dy2 = y[k] - y[6];
            if (!(fabsf(dx2 * dy1 - dy2 * dx1) > 1.0e-3f * (fabsf(dx1) + fabsf(dy1) + fabsf(dx2) + fabsf(dy2)))) maybe = true;
        }
    }
    if (!maybe) return false;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double dx1 = (double)(x[j] - x[6]), dy1 = (double)(y[j] - y[6]);          // Point2f differences: float, then widened (oracle v7)
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < j) {
            const double dx2 = (double)(x[k] - x[6]), dy2 = (double)(y[k] - y[6]);
            if (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) bad = true;
        }
    }
    return bad;
}
// one phase of the schedule for lane-octave vl; called by ALL threads of a 256-thread block (`scan`: 40 ints of LDS)
__device__ __forceinline__ void rs_schedule_block(const DevCtx& c, int vl, int phase, int n, int* scan)
{
    const int tid = threadIdx.x;
    int* st = c.rs_sched + vl * SVO_RS_ST;     // last accepted attempt (left) | attempts consumed | samples left | samples right | last accepted (right) | ended left | ended right
    if (n < 8 || n > c.rs_att_nmax) { if (phase == 0 && tid < SVO_RS_ST) st[tid] = 0; return; }
    const bool lmeds = rs_is_lmeds(n);                                     // its 300 samples are all drawn in phase 0 (nothing shortens that budget)
    if (phase == 1 && (lmeds || max(c.rs_bound[vl * 2], c.rs_bound[vl * 2 + 1]) <= c.rs_c1)) return;      // neither side's budget reaches chunk 2
    const int target = lmeds ? SVO_LMEDS_ITERS : (phase ? SVO_RANSAC_HYP : c.rs_c1);
    const int max_attempts = lmeds ? 1000 : 10000;                         // getSubset's maxAttempts: the RANSAC's run() passes 10000, LMedS's takes the default (oracle v7)
    int attempts = phase ? st[1] : 0, ns[2] = { phase ? st[2] : 0, phase ? st[3] : 0 };
    int last_ok[2] = { phase ? st[0] : -1, phase ? st[4] : -1 }, ended[2] = { phase ? st[5] : 0, phase ? st[6] : 0 };
    const float4* ptsL = (const float4*)(c.trk_pts + ((long long)vl * 2 + 0) * c.max_kps * 4);
    const float4* ptsR = (const float4*)(c.trk_pts + ((long long)vl * 2 + 1) * c.max_kps * 4);
    // small point counts keep the long table: a handful of points in a narrow image are collinear more often than not (a 65 x 97 image
    // leaves its keypoints a strip three pixels wide: three attempts in four are drawn afresh), and getSubset gives a sample up to
    // 10000 attempts before the run ends
    const int n_att = n < SVO_RS_SMALL_N ? SVO_RS_ATT_SMALL : SVO_RS_ATT;
    const uint4* att = (const uint4*)c.rs_att + (n < SVO_RS_SMALL_N ? (long long)(n - 8) * SVO_RS_ATT_SMALL
                                                                     : (long long)(SVO_RS_SMALL_N - 8) * SVO_RS_ATT_SMALL + (long long)(n - SVO_RS_SMALL_N) * SVO_RS_ATT);
    // Beyond the tabulated attempts (point sets on which nearly every sample is collinear: integer coordinates along a few edges, a narrow
    // strip) the stream is CONTINUED here: thread 0 carries cv::RNG on from the state the table row ended in (rs_att_state[n]) and draws
    // the next attempts one after the other into LDS, 256 at a time; the other threads test them as they test tabulated ones.  Serial and
    // slow (~75 us per 256 attempts) but exact, and bounded: SVO_RS_EXT more attempts, then the schedule ends and says so (SVO_ST_INTERNAL).
    __shared__ uint4 ext[256];
    unsigned long long xs = 0; bool xs_loaded = false;                     // thread 0: the generator's state past the table
    bool exhausted = false;
    while ((!ended[0] && ns[0] < target) || (!ended[1] && ns[1] < target)) {
        if (attempts >= n_att + SVO_RS_EXT) { exhausted = true; break; }
        if (attempts + 256 > n_att) {                                      // (block-uniform) this chunk reaches past the table
            if (tid == 0) {
                if (!xs_loaded) {
                    xs = (phase && st[1] > n_att) ? (((unsigned long long)(unsigned)st[9] << 32) | (unsigned)st[8]) : c.rs_att_state[n];
                    xs_loaded = true;
                }
                const uint32_t un = (uint32_t)n;
                for (int a = max(attempts, n_att); a < attempts + 256; a++) {
                    uint32_t s7[7];
#pragma unroll
                    for (int i = 0; i < 7; i++) {
                        uint32_t v; bool dup;
                        do {
                            xs = (unsigned long long)(uint32_t)xs * 4164903690ull + (uint32_t)(xs >> 32);      // cv::RNG::next
                            v = (uint32_t)xs % un;                                                         // rng.uniform(0, n)
                            dup = false;
#pragma unroll
                            for (int k = 0; k < 7; k++) if (k < i) dup = dup || s7[k] == v;
                        } while (dup);
                        s7[i] = v;
                    }
                    ext[a - attempts] = make_uint4(s7[0] | (s7[1] << 16), s7[2] | (s7[3] << 16), s7[4] | (s7[5] << 16), s7[6]);
                }
            }
            __syncthreads();
        }
        const int a = attempts + tid;
        int flags = 0;                                                    // bit 0: passes on the left side, bit 10: on the right
        uint4 w = make_uint4(0, 0, 0, 0);
        {
            w = a < n_att ? att[a] : ext[tid];
            const int s[7] = { (int)(w.x & 0xFFFFu), (int)(w.x >> 16), (int)(w.y & 0xFFFFu), (int)(w.y >> 16), (int)(w.z & 0xFFFFu), (int)(w.z >> 16), (int)(w.w & 0xFFFFu) };
            float4 pa[7], pb[7];
#pragma unroll
            for (int i = 0; i < 7; i++) { pa[i] = ptsL[s[i]]; pb[i] = ptsR[s[i]]; }
            float x1[7], y1[7], x2[7], y2[7];
#pragma unroll
            for (int i = 0; i < 7; i++) { x1[i] = pa[i].x; y1[i] = pa[i].y; x2[i] = pa[i].z; y2[i] = pa[i].w; }
            const int okl = !(rs_collinear7(x1, y1) || rs_collinear7(x2, y2));
#pragma unroll
            for (int i = 0; i < 7; i++) { x1[i] = pb[i].x; y1[i] = pb[i].y; x2[i] = pb[i].z; y2[i] = pb[i].w; }
            const int okr = !(rs_collinear7(x1, y1) || rs_collinear7(x2, y2));
            flags = (ended[0] ? 0 : okl) | ((ended[1] ? 0 : okr) << 10);
        }
        int tt;
        const int pre = block_exclusive_scan(flags, scan, &tt);           // both counts ride in one scan (<= 256 each)
        const int tot[2] = { tt & 1023, tt >> 10 }, mypre[2] = { pre & 1023, pre >> 10 }, mine[2] = { flags & 1, flags >> 10 };
        // getSubset's maxAttempts: a sample that needs more attempts than that ends the run.  Gaps inside these 256
        // attempts are shorter, so only the FIRST passing attempt of the chunk can be too late; the last one is what the next chunk measures from
        int* fl = scan + 33;                                               // first | last passing attempt of the chunk, per side
        __syncthreads();
        if (tid < 4) fl[tid] = -1;
        __syncthreads();
#pragma unroll
        for (int sd = 0; sd < 2; sd++) if (mine[sd]) { if (mypre[sd] == 0) fl[2 * sd] = a; if (mypre[sd] == tot[sd] - 1) fl[2 * sd + 1] = a; }
        __syncthreads();
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
            if (ended[sd]) continue;
            const int first = fl[2 * sd], last = fl[2 * sd + 1];
            if ((first >= 0 ? first : attempts + 256) - last_ok[sd] > max_attempts) { ended[sd] = 1; continue; }      // (no passing attempt here and none for max_attempts: ended as well)
            if (first < 0) continue;
            const int idx = ns[sd] + mypre[sd];
            if (mine[sd] && idx < SVO_RANSAC_PAD) *(uint4*)(c.rs_smp + (((long long)vl * 2 + sd) * SVO_RANSAC_PAD + idx) * 8) = w;
            ns[sd] += tot[sd]; last_ok[sd] = last;
        }
        attempts += 256;
        __syncthreads();
    }
    if (tid == 0) {
        st[0] = last_ok[0]; st[1] = attempts; st[2] = min(ns[0], SVO_RANSAC_PAD); st[3] = min(ns[1], SVO_RANSAC_PAD); st[4] = last_ok[1]; st[5] = ended[0]; st[6] = ended[1];
        if (xs_loaded) { st[8] = (int)(uint32_t)xs; st[9] = (int)(uint32_t)(xs >> 32); }      // (valid for phase 1 when st[1] > n_att)
        // The attempts ran out (table + SVO_RS_EXT) while a side still wanted samples: remembered per side (bit 0 left, bit 1 right).  Whether
        // that MATTERS is only known once the counts are in -- the chunks are drawn ahead of the stop rule, and a lane whose budget collapses
        // after a handful of samples never needed the rest -- so k_track_finalize raises SVO_ST_INTERNAL if the sequential algorithm would have
        // visited a sample that was never drawn (in either phase, never silently).
        st[10] = exhausted ? ((!ended[0] && ns[0] < target ? 1 : 0) | (!ended[1] && ns[1] < target ? 2 : 0)) : 0;
    }
}
__global__ void __launch_bounds__(256) k_ransac_schedule(DevCtx c, int phase)
{
    SVO_TL_SCOPE(c, TL_RS_SCHED, phase);
    SVO_LATENCY_CHAIN(c);
    __shared__ int scan[40];
    const int vl = blockIdx.x;
    if (vl % c.oct_cap >= c.n_oct) return;
    rs_schedule_block(c, vl, phase, c.trk_nk[vl], scan);
}

// oracle: cbrt_rough (exponent / 3 on the bit pattern; integer arithmetic, hence the same bits)
__device__ __forceinline__ double cbrt_rough(double x)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    u = (unsigned long long)((unsigned)(u >> 32) / 3u + 715094163u) << 32;
    return __longlong_as_double((long long)u);
}

// The tail of the oracle's seven_point, operation for operation: g = f1 - f2 and f2 span the null space of the normalised 7x9
// system; the cubic det(lambda g + f2) = 0 (cofactors along the first row), made monic and depressed, its root of largest
// magnitude by twelve Newton steps from an upper bound, the other two (when the discriminant allows) by deflation + two
// polishing steps; each root's matrix denormalised F = T2^T f T1.  Returns the number of models (1 or 3), Fm[k][0..8].
__device__ __forceinline__ int seven_point_models(const double (&g)[9], const double (&f2)[9], double s1, double s2, double c1x, double c1y, double c2x, double c2y, double (&Fm)[3][9])
{
    const double g00 = g[4] * g[8] - g[5] * g[7], g01 = g[3] * g[8] - g[5] * g[6], g02 = g[3] * g[7] - g[4] * g[6];
    const double h00 = f2[4] * f2[8] - f2[5] * f2[7], h01 = f2[3] * f2[8] - f2[5] * f2[6], h02 = f2[3] * f2[7] - f2[4] * f2[6];
    const double m00 = (g[4] * f2[8] + f2[4] * g[8]) - (g[5] * f2[7] + f2[5] * g[7]);
    const double m01 = (g[3] * f2[8] + f2[3] * g[8]) - (g[5] * f2[6] + f2[5] * g[6]);
    const double m02 = (g[3] * f2[7] + f2[3] * g[7]) - (g[4] * f2[6] + f2[4] * g[6]);
    const double a3 = (g[0] * g00 - g[1] * g01) + g[2] * g02;
    const double a0 = (f2[0] * h00 - f2[1] * h01) + f2[2] * h02;
    const double a2 = ((f2[0] * g00 - f2[1] * g01) + f2[2] * g02) + ((g[0] * m00 - g[1] * m01) + g[2] * m02);
    const double a1 = ((g[0] * h00 - g[1] * h01) + g[2] * h02) + ((f2[0] * m00 - f2[1] * m01) + f2[2] * m02);
    if (__builtin_expect(a3 == 0.0, 0)) {
        // oracle v7 (seven_point): the cubic has lost its leading term -- the matrix g itself (lambda -> infinity) is a solution and comes
        // first, then the roots of what is left, in cv::solveCubic's order for its quadratic / linear branch.  Same expressions as the oracle.
        double lam0 = 0.0, lam1 = 0.0; int nq = 0;
        if (a2 == 0.0) { if (a1 != 0.0) { lam0 = -a0 / a1; nq = 1; } }
        else {
            double d = a1 * a1 - (4.0 * a2) * a0;
            if (d >= 0.0) {
                const bool two = d > 0.0;
                d = sqrt(d);
                const double q1 = (-a1 + d) * 0.5, q2 = (a1 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { lam0 = q1 / a2; lam1 = a0 / q1; } else { lam0 = q2 / a2; lam1 = a0 / q2; }
                nq = two ? 2 : 1;
            }
        }
        const double d1x = -(s1 * c1x), d1y = -(s1 * c1y), d2x = -(s2 * c2x), d2y = -(s2 * c2y);
#pragma unroll
        for (int k = 0; k < 3; k++) {                                  // (unrolled: a run-time index into Fm would put the caller's array in scratch)
            const double lam = k == 1 ? lam0 : lam1;
            double f[9];
#pragma unroll
            for (int i = 0; i < 9; i++) f[i] = k == 0 ? g[i] : g[i] * lam + f2[i];
            double M[3][3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                M[r][0] = f[3 * r] * s1; M[r][1] = f[3 * r + 1] * s1;
                M[r][2] = (f[3 * r] * d1x + f[3 * r + 1] * d1y) + f[3 * r + 2];
            }
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                Fm[k][cc] = s2 * M[0][cc]; Fm[k][3 + cc] = s2 * M[1][cc];
                Fm[k][6 + cc] = (d2x * M[0][cc] + d2y * M[1][cc]) + M[2][cc];
            }
        }
        return nq + 1;
    }
    const double Am = a2 / a3, Bm = a1 / a3, Cm = a0 / a3;
    const double sh = Am / 3.0;
    const double p = Bm - Am * sh;
    const double q = ((2.0 * sh) * sh) * sh - sh * Bm + Cm;
    const double Q = fabs(q), pn = p < 0.0 ? -p : 0.0;
    double u = cbrt_rough(2.0 * Q);
    const double ub = sqrt(2.0 * pn);
    if (ub > u) u = ub;
    u = u * 1.1;
    bool go = true;
#pragma unroll 1
    for (int it = 0; it < 12; it++) {
        const double f = (u * u + p) * u - Q, d = (3.0 * u) * u + p;
        go = go && d > 0.0;                                                // the oracle's `break`: once stopped, u stays
        if (go) u = u - f / d;
    }
    const double t1 = q > 0.0 ? -u : u;
    const double disc = (-3.0 * t1) * t1 - 4.0 * p;
    double t[3]; int n = 1;
    t[0] = t1; t[1] = 0.0; t[2] = 0.0;
    if (disc >= 0.0) {
        const double sq = sqrt(disc);
        t[1] = (sq - t1) * 0.5; t[2] = (-sq - t1) * 0.5;
#pragma unroll
        for (int k = 1; k < 3; k++) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const double f = (t[k] * t[k] + p) * t[k] + q, d = (3.0 * t[k]) * t[k] + p;
                if (d != 0.0) t[k] = t[k] - f / d;
            }
        }
        n = 3;
    }
    const double t1x = -(s1 * c1x), t1y = -(s1 * c1y), t2x = -(s2 * c2x), t2y = -(s2 * c2y);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double lam = t[k] - sh;
        double f[9];
#pragma unroll
        for (int i = 0; i < 9; i++) f[i] = g[i] * lam + f2[i];
        double M[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            M[r][0] = f[3 * r] * s1; M[r][1] = f[3 * r + 1] * s1;
            M[r][2] = (f[3 * r] * t1x + f[3 * r + 1] * t1y) + f[3 * r + 2];
        }
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            Fm[k][cc] = s2 * M[0][cc]; Fm[k][3 + cc] = s2 * M[1][cc];
            Fm[k][6 + cc] = (t2x * M[0][cc] + t2y * M[1][cc]) + M[2][cc];
        }
    }
    return n;
}

// one model into its slot: the matrix, the sample it came from, and the guards of the matrix-core line evaluation
__device__ __forceinline__ void store_model(const DevCtx& c, int vl, int side, int slot, int sample, const double (&Fv)[9])
{
    const long long o = ((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + slot;
    double* F = c.rs_F + o * 9;
#pragma unroll
    for (int j = 0; j < 9; j++) F[j] = Fv[j];
    c.rs_k[o] = sample;
    // guards of the matrix-core line evaluation in k_ransac_count (see there): E bounds how far its numerators can be from
    // the oracle's operation order, given coordinates inside the image; a side is trusted when |l| >= 2^28 E
    double aF[9];
#pragma unroll
    for (int j = 0; j < 9; j++) aF[j] = fabs(Fv[j]);
    const double X = (double)c.W, Y = (double)c.H, u = 7.105427357601002e-15;           // 2^-47
    const double EB = u * (X * (aF[0] * X + aF[1] * Y + aF[2]) + Y * (aF[3] * X + aF[4] * Y + aF[5]) + (aF[6] * X + aF[7] * Y + aF[8]));
    const double EA = u * (X * (aF[0] * X + aF[3] * Y + aF[6]) + Y * (aF[1] * X + aF[4] * Y + aF[7]) + (aF[2] * X + aF[5] * Y + aF[8]));
    double* Gd = c.rs_guard + o * 2;
    const double gA = 268435456.0 * EA, gB = 268435456.0 * EB;
    Gd[0] = gA * gA; Gd[1] = gB * gB;            // an overflow or a NaN here makes every comparison against it false: the side is never trusted
}
// The filler of an unused slot inside a live group of sixteen: l = F x1 = (1, 1, 0) for every x1, so the distance of x2 from it is
// (x2 + y2) / sqrt 2 -- tens of pixels for anything a tracker hands over (keypoints keep 31 px from every border): "outlier" on the
// fast path of every count kernel, count 0, never a record.
__device__ __forceinline__ void store_filler(const DevCtx& c, int vl, int side, int slot)
{
    const double Fd[9] = { 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0 };
    store_model(c, vl, side, slot, 0x3FFFFFFF, Fd);
}

// Evaluating models OUT OF ORDER still bounds what the sequential scan visits.  The scan visits the samples [0, E), E = the first
// k that is no longer below the budget (a record in sample k may cut the budget below k itself: k is still visited -- all its
// models are --, so E can exceed the final budget).  A visited model of sample h with count c > 6 leaves a budget <= K(c) whether
// it is a record or not (a record before it had a count >= c, and K falls with the count), so the scan ends by max(h + 1, K(c));
// a model of an unvisited sample has E <= h.  Hence E <= max(h + 1, K(c)) for EVERY scored model, and the minimum of those over
// whatever has been scored so far (rs_bound) is a safe limit: samples at or beyond it are never visited, every model of every
// sample below it is scored.
// (The tightening uses K(c - 1): one inlier less moves num / d by far more than svo_ln's rounding error, so the
// "K falls with the count" step holds for the computed values too, not only in exact arithmetic.)
// (the chunk ends are run-time values: c.rs_c0 / c.rs_c1 = SVO_RANSAC_CHUNK0 / SVO_RANSAC_CHUNK1 in the batched shapes; for a handful of lanes both are
// SVO_RANSAC_FEW -- chunk 1 empty and never launched, chunk 2 rarely reached -- because there a launch costs more than the samples the early bound
// saves; `c` = the DevCtx in scope)
#define RS_CHUNK_BEGIN(ch) ((ch) == 0 ? 0 : ((ch) == 1 ? c.rs_c0 : c.rs_c1))
#define RS_CHUNK_END(ch) ((ch) == 0 ? c.rs_c0 : ((ch) == 1 ? c.rs_c1 : SVO_RANSAC_HYP))
#define RS_SLOT_END(ch) (((RS_CHUNK_END(ch) + SVO_RANSAC_REG - 1) / SVO_RANSAC_REG) * SVO_RANSAC_RSLOTS)      // end of the chunk's model slots (whole regions)
static_assert(SVO_RANSAC_FEW % SVO_RANSAC_REG == 0 && SVO_RANSAC_FEW >= SVO_RANSAC_CHUNK1 && SVO_RANSAC_FEW <= SVO_RANSAC_HYP, "the few-lanes chunk end is a whole number of regions");
static_assert(SVO_RANSAC_CHUNK0 % SVO_RANSAC_REG == 0 && SVO_RANSAC_CHUNK1 % SVO_RANSAC_REG == 0 && SVO_RANSAC_RSLOTS == 3 * SVO_RANSAC_REG && SVO_RANSAC_REG == 16,
              "chunks are whole regions; a region is one DPP row of samples");

// What a group of consecutive slots starting at s0 (a multiple of 4 inside one region) still has to offer: the number of models
// at or after s0 in its region, 0 when the region was not generated by this chunk's hypothesis kernel (beyond rs_gen), when the
// group starts beyond the region's models, or when its first sample is one the sequential stop can no longer reach.
__device__ __forceinline__ int rs_group_live(const DevCtx& c, int vl, int side, int s0)
{
    const int reg = s0 / SVO_RANSAC_RSLOTS, off = s0 - reg * SVO_RANSAC_RSLOTS;
    if (reg * SVO_RANSAC_REG >= c.rs_gen[vl * 2 + side]) return 0;
    const int nv = c.rs_nvalid[((long long)vl * 2 + side) * (SVO_RANSAC_PAD / SVO_RANSAC_REG) + reg];
    if (off >= nv) return 0;
    if (c.rs_k[((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + s0] >= *(volatile int*)(c.rs_bound + vl * 2 + side)) return 0;
    return nv - off;
}
// the part of rs_group_live that cannot change while a count launch runs: models at or after s0 in a region this chunk generated
__device__ __forceinline__ int rs_group_generated(const DevCtx& c, int vl, int side, int s0)
{
    const int reg = s0 / SVO_RANSAC_RSLOTS, off = s0 - reg * SVO_RANSAC_RSLOTS;
    if (reg * SVO_RANSAC_REG >= c.rs_gen[vl * 2 + side]) return 0;
    const int nv = c.rs_nvalid[((long long)vl * 2 + side) * (SVO_RANSAC_PAD / SVO_RANSAC_REG) + reg];
    return off >= nv ? 0 : nv - off;
}
// after a block has its counts: its best model tightens rs_bound and raises the floors of the later chunks
__device__ __forceinline__ void rs_publish_best(const DevCtx& c, int vl, int side, int chunk, int n, int best, int best_slot)
{
    int* bound = c.rs_bound + vl * 2 + side;
    if (best > 7) {
        const int cur = *(volatile int*)bound, k = c.rs_k[((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + best_slot];
        if (k + 1 < cur) {
            const int K = ransac_niters(best - 1, n, cur);
            if (max(k + 1, K) < cur) atomicMin(bound, max(k + 1, K));
        }
    }
    // the floors of the later chunks: best count of chunk 0 (for chunk 1), of chunks 0 and 1 (for chunk 2)
    if (chunk == 0 && best > 6) atomicMax(&c.rs_floor[(vl * 2 + side) * 2], best);
    if (chunk <= 1 && best > 6) atomicMax(&c.rs_floor[(vl * 2 + side) * 2 + 1], best);
}

// chunk 0: samples [0, CHUNK0) always; chunks 1, 2: only below rs_bound.
//
// One sample = 16 lanes (a DPP row), four samples per wave, sixteen per 256-thread block = one region.  Lane c < 9 of a group
// holds COLUMN c of the 7x9 system in registers; the oracle's Gauss-Jordan with full pivoting (seven_point) then runs without any
// memory traffic: the pivot search is a per-lane scan plus a 16-lane all-reduce on the DPP network, rows are swapped in
// registers, columns are swapped VIRTUALLY (vcol = a lane's current column position; the data never moves), the pivot
// column's entries reach the other lanes by ds_bpermute.  Every arithmetic step is element-wise and uses the oracle's
// expression, so the matrices agree bit for bit; ties in the pivot search resolve to the oracle's scan order (row, then
// column position).  The group's first lane then solves the cubic and writes the models.
typedef struct { double v; int key; } PivotCand;
__device__ __forceinline__ double dpp_f64(double v, const int ctrl_tag)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_tag) {
        case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
        case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false); break; // row_mirror
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int dpp_i32(int v, const int ctrl_tag)
{
    switch (ctrl_tag) {
        case 0: return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
        default: return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    }
}
// value of `v` in lane `src` (0..15) of this lane's group of 16
__device__ __forceinline__ double group_bcast(double v, int src) { return __shfl(v, src, 16); }

// the models of one region, packed: nm[i] models of sample i (0 for a sample that was not generated) at the running offset.
// Called by the lane that owns sample i of the region (all sixteen owners take part); `prefix` = models of the samples before it.
__device__ __forceinline__ void store_region_models(const DevCtx& c, int vl, int side, int reg, int sample, int prefix, int nm, const double (&Fm)[3][9])
{
#pragma unroll
    for (int j = 0; j < 3; j++) if (j < nm) store_model(c, vl, side, reg * SVO_RANSAC_RSLOTS + prefix + j, sample, Fm[j]);
}
// ... and what the owners do together once the region's total is known (lane i of the sixteen): zero every count, fill the tail
// of the last live group of sixteen slots, publish the total
__device__ __forceinline__ void finish_region(const DevCtx& c, int vl, int side, int reg, int i16, int total)
{
    const long long o = ((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + (long long)reg * SVO_RANSAC_RSLOTS;
    c.rs_cnt[o + i16] = 0; c.rs_cnt[o + 16 + i16] = 0; c.rs_cnt[o + 32 + i16] = 0;
    const int s = total + i16;
    if (s < ((total + 15) & ~15)) store_filler(c, vl, side, reg * SVO_RANSAC_RSLOTS + s);
    if (i16 == 0) c.rs_nvalid[((long long)vl * 2 + side) * (SVO_RANSAC_PAD / SVO_RANSAC_REG) + reg] = total;
}

__global__ void __launch_bounds__(256) k_ransac_hyp(DevCtx c, int chunk)
{
    SVO_TL_SCOPE(c, TL_RS_HYP, chunk);
    SVO_LATENCY_CHAIN(c);
    __shared__ int nm_s[16];
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    const int h = RS_CHUNK_BEGIN(chunk) + blockIdx.x * 16 + grp, side = blockIdx.y, vl = blockIdx.z;
    if (vl % c.oct_cap >= c.n_oct) return;
    const int n = c.trk_nk[vl];
    if (n < 8) return;                  // (exactly seven pairs: findFundamentalMat's direct path, see k_track_finalize)
    // rs_bound is stable while this kernel runs (only k_ransac_count lowers it, and the previous chunk's has finished): what
    // this chunk generates is [begin, gen) with gen = min(end, rs_bound); k_ransac_count must not trust anything beyond it
    const int gen = min(chunk ? min(RS_CHUNK_END(chunk), c.rs_bound[vl * 2 + side]) : RS_CHUNK_END(chunk), c.rs_sched[vl * SVO_RS_ST + 2 + side]);      // (... and no further than the schedule's samples)
    if (blockIdx.x == 0 && threadIdx.x == 0) c.rs_gen[vl * 2 + side] = gen;
    if (RS_CHUNK_BEGIN(chunk) + (int)blockIdx.x * 16 >= gen) return;          // block-uniform; inside a live block every lane stays (DPP)
    const float4* pts = (const float4*)(c.trk_pts + ((long long)vl * 2 + side) * c.max_kps * 4);
    // ---- the sample (oracle: ransac_sample), computed redundantly by the 16 lanes of the group ----
    int s[7];
    ransac_sample7(c, vl, side, h, s);
    float4 P[7];
#pragma unroll
    for (int i = 0; i < 7; i++) P[i] = pts[s[i]];
    double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) { c1x += (double)P[i].x; c1y += (double)P[i].y; c2x += (double)P[i].z; c2y += (double)P[i].w; }
    c1x = c1x / 7.0; c1y = c1y / 7.0; c2x = c2x / 7.0; c2y = c2y / 7.0;
    // the fourteen square roots of the mean-distance normalisation: lane i < 7 of the group takes point i, the sums then run over
    // the lanes' values in the oracle's order (a double-precision sqrt is ~30 instructions)
    double d1 = 0, d2 = 0;
    {
        float4 Pm = P[0];
#pragma unroll
        for (int i = 1; i < 7; i++) if ((gl & 7) == i) Pm = P[i];
        const double ax = (double)Pm.x - c1x, ay = (double)Pm.y - c1y, bx = (double)Pm.z - c2x, by = (double)Pm.w - c2y;
        const double r1 = sqrt(ax * ax + ay * ay), r2 = sqrt(bx * bx + by * by);
#pragma unroll
        for (int i = 0; i < 7; i++) { d1 += group_bcast(r1, i); d2 += group_bcast(r2, i); }
    }
    const double s1 = 9.8994949366116654 / d1, s2 = 9.8994949366116654 / d2;
    // ---- column gl of the system: A[i][0..8] = x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1 ----
    double col[7];
    const int cu = gl / 3, cv = gl - 3 * cu;                                  // column = (x2 | y2 | 1) * (x1 | y1 | 1)
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double x1 = ((double)P[i].x - c1x) * s1, y1 = ((double)P[i].y - c1y) * s1, x2 = ((double)P[i].z - c2x) * s2, y2 = ((double)P[i].w - c2y) * s2;
        const double u = cu == 0 ? x2 : (cu == 1 ? y2 : 1.0), v = cv == 0 ? x1 : (cv == 1 ? y1 : 1.0);
        col[i] = gl < 9 ? u * v : 0.0;                                          // x * 1.0 == x exactly: same entries as the oracle's
    }
    int vcol = gl < 9 ? gl : 64;                                              // lanes 9..15 hold no column
#pragma unroll
    for (int k = 0; k < 7; k++) {
        // pivot = first maximum of |A[i][j]|, i >= k, j >= k, in (i, j) scan order
        double bv = -1.0; int bi = k;
#pragma unroll
        for (int i = k; i < 7; i++) { const double a = fabs(col[i]); if (a > bv) { bv = a; bi = i; } }
        const bool active = vcol >= k && vcol <= 8;
        if (!active) bv = -2.0;
        int bkey = (bi << 8) | ((vcol & 63) << 4) | gl;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            const double ov = dpp_f64(bv, st); const int ok = dpp_i32(bkey, st);
            const bool take = ov > bv || (ov == bv && ok < bkey);
            bv = take ? ov : bv; bkey = take ? ok : bkey;
        }
        const int pi = bkey >> 8, pjv = (bkey >> 4) & 15, plane = bkey & 15;
        // row swap k <-> pi in every column
#pragma unroll
        for (int i = k + 1; i < 7; i++) if (i == pi) { const double t = col[i]; col[i] = col[k]; col[k] = t; }
        // column swap k <-> pjv, virtual
        if (vcol == pjv) vcol = k; else if (vcol == k) vcol = pjv;
        const double piv = group_bcast(col[k], plane);
        const bool act2 = vcol >= k && vcol <= 8;
        if (act2) col[k] = col[k] / piv;
        double f[7];
#pragma unroll
        for (int i = 0; i < 7; i++) f[i] = i == k ? 0.0 : group_bcast(col[i], plane);       // A[i][k], read before anything below changes it
#pragma unroll
        for (int i = 0; i < 7; i++) if (i != k && act2) col[i] = col[i] - f[i] * col[k];
    }
    // the free unknowns sit at positions 7 and 8: g[perm[7]] = 1, g[perm[8]] = -1, g[perm[i]] = A[i][8] - A[i][7];
    // f2[perm[7]] = 0, f2[perm[8]] = 1, f2[perm[i]] = -A[i][8].  Lane L (original column L) sits at position vcol.
    const unsigned rowmask_shift = 16 * (threadIdx.x >> 4 & 3);
    const int lane7 = __ffs((unsigned)((__ballot(vcol == 7) >> rowmask_shift) & 0xFFFFu)) - 1;
    const int lane8 = __ffs((unsigned)((__ballot(vcol == 8) >> rowmask_shift) & 0xFFFFu)) - 1;
    double a7[7], a8[7];
#pragma unroll
    for (int i = 0; i < 7; i++) { a7[i] = group_bcast(col[i], lane7); a8[i] = group_bcast(col[i], lane8); }
    double gmine = vcol == 7 ? 1.0 : -1.0, fmine = vcol == 7 ? 0.0 : 1.0;
#pragma unroll
    for (int i = 0; i < 7; i++) if (vcol == i) { gmine = a8[i] - a7[i]; fmine = -a8[i]; }
    double gv[9], fv[9];
#pragma unroll
    for (int j = 0; j < 9; j++) { gv[j] = group_bcast(gmine, j); fv[j] = group_bcast(fmine, j); }
    double Fm[3][9];
    const bool live = h < gen;
    const int nm = seven_point_models(gv, fv, s1, s2, c1x, c1y, c2x, c2y, Fm);
    if (gl == 0) nm_s[grp] = live ? nm : 0;
    __syncthreads();
    int prefix = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { const int v = nm_s[i]; if (i < grp) prefix += v; total += v; }
    const int reg = (RS_CHUNK_BEGIN(chunk) >> 4) + blockIdx.x;
    if (gl == 0 && live) store_region_models(c, vl, side, reg, h, prefix, nm, Fm);
    if (threadIdx.x < 16) finish_region(c, vl, side, reg, threadIdx.x, total);
}

// step K of the oracle's Gauss-Jordan elimination with full pivoting (seven_point) on a thread's own registers
template <int K>
__device__ __forceinline__ void gj_step(double (&A)[7][9], int (&perm)[9])
{
    // pivot = first maximum of |A[i][j]|, i >= K, j >= K, in (i, j) scan order
    // (the maximum first -- fmax skips NaNs as the oracle's `v > best` does --, then the first entry that equals it: the scan runs
    // backwards so that the last assignment is the first position; three instructions per entry instead of four)
    double best = -1.0; int pij = K * 16 + K;
#pragma unroll
    for (int i = K; i < 7; i++) {
#pragma unroll
        for (int j = K; j < 9; j++) best = fmax(best, fabs(A[i][j]));
    }
#pragma unroll
    for (int i = 6; i >= K; i--) {
#pragma unroll
        for (int j = 8; j >= K; j--) if (fabs(A[i][j]) == best) pij = i * 16 + j;
    }
    const int pi = pij >> 4, pj = pij & 15;
#pragma unroll
    for (int i = K + 1; i < 7; i++) if (pi == i) {
#pragma unroll
        for (int j = K; j < 9; j++) { const double t = A[K][j]; A[K][j] = A[i][j]; A[i][j] = t; }
    }
#pragma unroll
    for (int j = K + 1; j < 9; j++) if (pj == j) {
#pragma unroll
        for (int i = 0; i < 7; i++) { const double t = A[i][K]; A[i][K] = A[i][j]; A[i][j] = t; }
        const int t = perm[K]; perm[K] = perm[j]; perm[j] = t;
    }
    const double piv = A[K][K];
#pragma unroll
    for (int j = K + 1; j < 9; j++) A[K][j] = A[K][j] / piv;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        if (i == K) continue;
        const double f = A[i][K];
#pragma unroll
        for (int j = K + 1; j < 9; j++) A[i][j] = A[i][j] - f * A[K][j];
    }
}

// The same samples with ONE THREAD each, for launches that fill the GPU anyway (many lanes): the 16-lane form above spends
// ~1500 wave-instructions on four samples (nine of sixteen lanes hold a column, every step pays its DPP all-reduce and
// sixteen cross-lane broadcasts) -- it is built for the latency of one stream.  Here the 7x9 system sits in the thread's own
// registers (every loop unrolled, no dynamic index), the oracle's Gauss-Jordan runs on it literally -- physical row and column
// swaps under the pivot's predicate (exec-masked register swaps), the elimination restricted to the columns that are still
// read (column k of the other rows becomes an exact 0 that nothing looks at again; columns left of k are dead).  A DPP row of
// sixteen threads is one region: its models are packed by a row scan of the model counts.
__global__ void __launch_bounds__(64) k_ransac_hyp_thread(DevCtx c, int chunk)
{
    SVO_TL_SCOPE(c, TL_RS_HYP, chunk);
    SVO_LATENCY_CHAIN(c);
    const int h = RS_CHUNK_BEGIN(chunk) + blockIdx.x * 64 + threadIdx.x, side = blockIdx.y, vl = blockIdx.z;
    if (vl % c.oct_cap >= c.n_oct) return;
    const int n = c.trk_nk[vl];
    if (n < 8) return;                  // (exactly seven pairs: findFundamentalMat's direct path, see k_track_finalize)
    const int gen = min(chunk ? min(RS_CHUNK_END(chunk), c.rs_bound[vl * 2 + side]) : RS_CHUNK_END(chunk), c.rs_sched[vl * SVO_RS_ST + 2 + side]);      // (... and no further than the schedule's samples)
    if (blockIdx.x == 0 && threadIdx.x == 0) c.rs_gen[vl * 2 + side] = gen;
    if (RS_CHUNK_BEGIN(chunk) + (int)blockIdx.x * 64 >= gen) return;          // wave-uniform: inside a live wave every lane stays (DPP scan)
    const bool live = h < gen;
    const float4* pts = (const float4*)(c.trk_pts + ((long long)vl * 2 + side) * c.max_kps * 4);
    int s[7];
    ransac_sample7(c, vl, side, live ? h : 0, s);
    float4 P[7];
#pragma unroll
    for (int i = 0; i < 7; i++) P[i] = pts[s[i]];
    double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) { c1x += (double)P[i].x; c1y += (double)P[i].y; c2x += (double)P[i].z; c2y += (double)P[i].w; }
    c1x = c1x / 7.0; c1y = c1y / 7.0; c2x = c2x / 7.0; c2y = c2y / 7.0;
    double d1 = 0, d2 = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double ax = (double)P[i].x - c1x, ay = (double)P[i].y - c1y, bx = (double)P[i].z - c2x, by = (double)P[i].w - c2y;
        d1 += sqrt(ax * ax + ay * ay); d2 += sqrt(bx * bx + by * by);
    }
    const double s1 = 9.8994949366116654 / d1, s2 = 9.8994949366116654 / d2;
    double A[7][9];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double x1 = ((double)P[i].x - c1x) * s1, y1 = ((double)P[i].y - c1y) * s1, x2 = ((double)P[i].z - c2x) * s2, y2 = ((double)P[i].w - c2y) * s2;
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2; A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2; A[i][6] = x1; A[i][7] = y1; A[i][8] = 1.0;
    }
    int perm[9];
#pragma unroll
    for (int j = 0; j < 9; j++) perm[j] = j;
    gj_step<0>(A, perm); gj_step<1>(A, perm); gj_step<2>(A, perm); gj_step<3>(A, perm);
    gj_step<4>(A, perm); gj_step<5>(A, perm); gj_step<6>(A, perm);
    // g[perm[7]] = 1, g[perm[8]] = -1, g[perm[i]] = A[i][8] - A[i][7];  f2[perm[7]] = 0, f2[perm[8]] = 1, f2[perm[i]] = -A[i][8]
    double gv[9], fv[9];
#pragma unroll
    for (int j = 0; j < 9; j++) {
        double gj = perm[7] == j ? 1.0 : -1.0, fj = perm[7] == j ? 0.0 : 1.0;
#pragma unroll
        for (int i = 0; i < 7; i++) if (perm[i] == j) { gj = A[i][8] - A[i][7]; fj = -A[i][8]; }
        gv[j] = gj; fv[j] = fj;
    }
    double Fm[3][9];
    int nm = seven_point_models(gv, fv, s1, s2, c1x, c1y, c2x, c2y, Fm);
    if (!live) nm = 0;
    // inclusive scan of the model counts inside the DPP row of sixteen = the region
    int inc = nm;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);      // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);      // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);      // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);      // row_shr:8
    const int total = __shfl(inc, 15, 16);
    const int reg = (h >> 4);
    if (reg >= SVO_RANSAC_PAD / SVO_RANSAC_REG) return;                       // (lanes past the padded schedule: nm = 0, nothing to say)
    store_region_models(c, vl, side, reg, h, inc - nm, nm, Fm);
    finish_region(c, vl, side, reg, threadIdx.x & 15, total);
}

// Symmetric epipolar test e = max(dA^2 / |lA|^2, dB^2 / |lB|^2) <= 1 with the oracle's arithmetic, minus its two f64
// divisions in all but borderline cases: the oracle's e_X = fl(fl(d*d) * fl(1 / den)) is within 2 ulp of dd / den, so
// dd <= den * (1 - 2^-40) decides "inlier" with certainty; OpenCV stores the error as a FLOAT and compares that with 1.0f
// (computeError / findInliers), so an error up to 1 + 2^-24 still rounds to an inlier and "outlier" is certain from
// dd >= den * (1 + 2^-22) on; only a point inside that band (or a degenerate line, den <= 0 / NaN) replays the exact
// expression.  Same decisions, bit for bit.
__device__ __forceinline__ int fm_inlier(const double* F, float fx1, float fy1, float fx2, float fy2)
{
    const double x1 = (double)fx1, y1 = (double)fy1, x2 = (double)fx2, y2 = (double)fy2;
    double a = (F[0] * x1 + F[1] * y1) + F[2], b = (F[3] * x1 + F[4] * y1) + F[5], cc = (F[6] * x1 + F[7] * y1) + F[8];
    const double denB = a * a + b * b, dB = (x2 * a + y2 * b) + cc;
    a = (F[0] * x2 + F[3] * y2) + F[6]; b = (F[1] * x2 + F[4] * y2) + F[7]; cc = (F[2] * x2 + F[5] * y2) + F[8];
    const double denA = a * a + b * b, dA = (x1 * a + y1 * b) + cc;
    const double ddA = dA * dA, ddB = dB * dB;
    const double lo = 1.0 - 9.094947017729282e-13, hi = 1.0 + 2.384185791015625e-07;             // 1 - 2^-40, 1 + 2^-22
    const bool inA = ddA <= denA * lo, inB = ddB <= denB * lo, outA = ddA >= denA * hi, outB = ddB >= denB * hi;
    const bool sure = (denA > 0.0) & (denB > 0.0) & (inA | outA) & (inB | outB);
    if (__builtin_expect(sure, 1)) return inA & inB;
    const double sB = 1.0 / denB, sA = 1.0 / denA;
    const double eA = ddA * sA, eB = ddB * sB;
    const double e = eA < eB ? eB : eA;                   // std::max(eA, eB)
    return (float)e <= 1.0f;
}
// the error itself, as computeError stores it (oracle: fm_error)
__device__ __forceinline__ float fm_error(const double* F, float fx1, float fy1, float fx2, float fy2)
{
    const double x1 = (double)fx1, y1 = (double)fy1, x2 = (double)fx2, y2 = (double)fy2;
    double a = (F[0] * x1 + F[1] * y1) + F[2], b = (F[3] * x1 + F[4] * y1) + F[5], cc = (F[6] * x1 + F[7] * y1) + F[8];
    const double sB = 1.0 / (a * a + b * b), dB = (x2 * a + y2 * b) + cc;
    a = (F[0] * x2 + F[3] * y2) + F[6]; b = (F[1] * x2 + F[4] * y2) + F[7]; cc = (F[2] * x2 + F[5] * y2) + F[8];
    const double sA = 1.0 / (a * a + b * b), dA = (x1 * a + y1 * b) + cc;
    const double eA = (dA * dA) * sA, eB = (dB * dB) * sB;
    return (float)(eA < eB ? eB : eA);
}
// the order nth_element gives the errors: their bit patterns as ints, a NaN in x86's default (negative) form (oracle: float_bits_x86)
__device__ __forceinline__ int fm_error_key(float e) { return e != e ? (int)0xFFC00000u : __float_as_int(e); }

// Inlier counts on the MATRIX CORES: the two epipolar lines of every (model, pair), l = F x1 and l' = F^T x2, are small
// dense products -- [16 rows = 4 models x (a, b, c, -)] x [4 = (x, y, 1, 0)] x [16 pairs] -- i.e. one
// v_mfma_f64_16x16x4_f64 each per wave and tile of 16 pairs, after which lane (g, j) holds (a, b, c) of model g for pair j
// in its own registers (result layout, pinned on the hardware: row i of the 16x16 tile sits in register i / 4 of lane
// j + 16 (i % 4); operands: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k).  What is left per test on the VALU is the two
// norms, the two numerators and the comparisons: about 45 instructions against 78.
// The matrix core sums its four products in an order of its own, so a, b, c can differ from the oracle's ((F0 x + F1 y) + F2)
// in the last bits.  E (store_model, per model and side) bounds the resulting shift of a numerator for coordinates
// inside the image; a side is trusted only when |l| >= 2^28 E, which keeps d to 2^-28 |l| and |l|^2 to 2^-28 relatively, and a
// verdict is given only when d^2 and |l|^2 differ by more than 2^-26: anything closer, and any failed guard or NaN, replays the
// oracle's own expression (fm_inlier).  In practice nothing is ever that close; the replay is there so that the counts are
// the oracle's by construction.
typedef double rc_d4 __attribute__((ext_vector_type(4)));
#ifdef SVO_AB_KERNELS      // an A/B anchor, not a product form (svo_kernels.h): compiled into libsvo_hip_ab.so only
__global__ void __launch_bounds__(256) k_ransac_count_mfma(DevCtx c, int chunk)
{
    SVO_TL_SCOPE(c, TL_RS_COUNT, chunk);
    SVO_LATENCY_CHAIN(c);
    __shared__ int cnt_s[16];
    const int side = blockIdx.y, vl = blockIdx.z, h0 = 3 * RS_CHUNK_BEGIN(chunk) + blockIdx.x * 16, tid = threadIdx.x;      // first SLOT of the block
    if (vl % c.oct_cap >= c.n_oct) return;
    const int n = c.trk_nk[vl];
    if (n <= SVO_LMEDS_MAX_N) return;   // (exactly seven pairs: findFundamentalMat's direct path; eight to fourteen: LMedS ranks medians, not counts -- see k_track_finalize)
    if (h0 >= RS_SLOT_END(chunk)) return;
    // ONE thread decides for the block (rs_bound moves while the launch runs: threads reading it themselves could disagree, and a
    // block of which some waves have left no longer fills its shared arrays)
    __shared__ int s_nlive;
    if (tid == 0) s_nlive = rs_group_live(c, vl, side, h0);
    __syncthreads();
    const int nlive = s_nlive;
    if (nlive <= 0) return;                                                   // nothing generated here, or samples the sequential stop never reaches
    const float4* pts = (const float4*)(c.trk_pts + ((long long)vl * 2 + side) * c.max_kps * 4);
    const double* F = c.rs_F + (((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + h0) * 9;
    const double* Gd = c.rs_guard + (((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + h0) * 2;
    const int w = tid >> 6, l = tid & 63, k = l >> 4, j = l & 15;
    // operand A of the two products: lane l holds row i = l % 16 = 4 r + g (component r of model g of this wave), column k
    const int ga = l & 3, ra = (l & 15) >> 2;
    const double* Fa = F + 9 * (4 * w + ga);
    const bool liveA = ra < 3 && k < 3;
    const double a1 = liveA ? Fa[3 * ra + k] : 0.0;            // F[r][k]:   l  = F   (x1 y1 1)
    const double a2 = liveA ? Fa[3 * k + ra] : 0.0;            // F[k][r]:   l' = F^T (x2 y2 1)
    // the model this lane evaluates: g = l / 16
    const double* Fe = F + 9 * (4 * w + k);
    const double dminA = Gd[2 * (4 * w + k)], dminB = Gd[2 * (4 * w + k) + 1];
    const double lo = 1.0 - 1.4901161193847656e-08, hi = 1.0 + 2.384185791015625e-07;         // 1 - 2^-26, 1 + 2^-22 (the error is compared AS A FLOAT: up to 1 + 2^-24 it rounds to 1.0f = inlier)
    // A model matters only if it is a RECORD (its count exceeds that of every earlier model, k_track_finalize), and
    // every model of chunks 1, 2 comes after all of chunk 0 (chunk 2: after all of chunk 1 as well), whose best count is
    // known by now: once a model cannot exceed that floor even if every remaining pair were an inlier, its exact count is
    // of no consequence -- the partial count it leaves is below the floor too, so it is no record, and as an under-estimate it
    // only loosens the rs_bound it feeds.  A wave stops when all four of its models are there (a contaminated sample
    // explains 10-20 % of the pairs against the floor's 50-60 %: about half way through the list).
    const int floor_cnt = chunk ? c.rs_floor[(vl * 2 + side) * 2 + (chunk - 1)] : 0x7FFFFFFF;
    int cnt = 0;
    for (int base = 0; base < n; base += 16) {
        if (chunk && base && (base & 127) == 0) {
            int gsum = cnt;                                              // the group's count so far, in all its 16 lanes
            gsum += __builtin_amdgcn_update_dpp(0, gsum, 0xB1, 0xF, 0xF, false);
            gsum += __builtin_amdgcn_update_dpp(0, gsum, 0x4E, 0xF, 0xF, false);
            gsum += __builtin_amdgcn_update_dpp(0, gsum, 0x141, 0xF, 0xF, false);
            gsum += __builtin_amdgcn_update_dpp(0, gsum, 0x140, 0xF, 0xF, false);
            if (__ballot(gsum + (n - base) <= floor_cnt) == ~0ull && c.debug_mode != 16) break;
        }
        const int pi = base + j;
        const float4 p = pts[min(pi, n - 1)];
        const double x1 = (double)p.x, y1 = (double)p.y, x2 = (double)p.z, y2 = (double)p.w;
        const double b1 = k == 0 ? x1 : (k == 1 ? y1 : (k == 2 ? 1.0 : 0.0));
        const double b2 = k == 0 ? x2 : (k == 1 ? y2 : (k == 2 ? 1.0 : 0.0));
        rc_d4 z = { 0.0, 0.0, 0.0, 0.0 };
        const rc_d4 LB = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, z, 0, 0, 0);
        const rc_d4 LA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, z, 0, 0, 0);
        const double denB = LB[0] * LB[0] + LB[1] * LB[1], dB = (x2 * LB[0] + y2 * LB[1]) + LB[2];
        const double denA = LA[0] * LA[0] + LA[1] * LA[1], dA = (x1 * LA[0] + y1 * LA[1]) + LA[2];
        const double ddA = dA * dA, ddB = dB * dB;
        const bool okA = denA >= dminA, okB = denB >= dminB;
        const bool inA = okA & (ddA <= denA * lo), inB = okB & (ddB <= denB * lo);
        const bool outA = okA & (ddA >= denA * hi), outB = okB & (ddB >= denB * hi);
        int v = (inA & inB) ? 1 : 0;
        if (__builtin_expect(!((inA & inB) | outA | outB) || c.debug_mode == 13, 0)) v = fm_inlier(Fe, p.x, p.y, p.z, p.w);
        cnt += pi < n ? v : 0;
    }
    // the 16 lanes of a group hold the partial counts of one model
    cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xF, 0xF, false);
    cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xF, 0xF, false);
    cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x141, 0xF, 0xF, false);
    cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x140, 0xF, 0xF, false);
    if (j == 0) { cnt_s[4 * w + k] = cnt; c.rs_cnt[((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + h0 + 4 * w + k] = cnt; }
    __syncthreads();
    if (tid == 0) {
        int best = 0, best_h = h0;
        for (int h = 0; h < 16; h++) if (h < nlive && cnt_s[h] > best) { best = cnt_s[h]; best_h = h0 + h; }
        rs_publish_best(c, vl, side, chunk, n, best, best_h);
    }
}
#endif

// The count for launches that fill the GPU (many lanes): SIXTEEN models x sixteen pairs per matrix-core tile, and the
// numerator on the matrix cores as well.  d = x2^T F x1 is one bilinear form -- the oracle's dA and dB are two roundings of it --
// so with phi = (x1 x2, y1 x2, x2, x1 y2 | y1 y2, y2, x1, y1 | 1) it is a [16 models x 9] x [9 x 16 pairs] product (three
// chained v_mfma_f64_16x16x4_f64), and the four line components a, b of l = F x1 and l' = F^T x2 are four more [16 x 3] x
// [3 x 16] products.  What is left on the VALU per test is the two norms, d^2, the band products and the comparisons: ~19
// instructions against ~45 of k_ransac_count_mfma (whose tiles are 4 models x 4 line components, numerators on the VALU),
// and the B operands (built once per block and 256 pairs in LDS, in operand layout) serve 64 slots.  The step is bound by
// VALU issue, the matrix pipe is otherwise idle: the instructions move to where there is room.
// Exactness as in k_ransac_count_mfma: E (store_model) bounds |d - dA|, |d - dB| for coordinates inside the image whatever
// the summation order (the oracle's dA / dB carry <= 6 roundings of the nine terms, three chained matrix-core products <= 12 if
// they are fused multiply-adds and <= 24 if products and sums round separately: 30 u_53 <= 2^-47 x 2^53 = 64, the factor there), a side is trusted only
// when |l| >= 2^28 E, a verdict is given only when d^2 and |l|^2 differ by more than 2^-26 relatively, and everything else
// -- a failed guard, a NaN, a borderline pair -- replays the oracle's own expression (fm_inlier).
// Result layout of the tile (pinned on the hardware, see above): lane l = j + 16 q, register r  <->  model 4 r + q, pair j.
#define RC16_SUPER 256
// Round 5: the PAIRS of a lane are split over `nsplit` blocks as well.  A block used to walk all n pairs (45-70 tiles of 16, ~1 us
// each: 45 us for the two blocks per lane and side of chunk 0, with 3/4 of the GPU idle); now it walks its quarter, adds its partial
// counts to rs_cnt (zeroed by the hypothesis kernel) and takes a ticket; the block that takes the last ticket of its 64 slots reads
// the sums and publishes the best one.  Counts are integers: the order of the additions does not matter.  A block that finds its
// groups out of reach (rs_bound moves while the launch runs, so the blocks of one group need not agree) still takes its ticket; the
// sums it leaves incomplete belong to samples at or beyond rs_bound, which the finalize never visits, and a best count published from
// incomplete sums is an under-estimate, which only loosens the bound it feeds (as the early exit's partial counts did).
#define RC16_NSPLIT0 1         // blocks the pairs of a lane are split over in chunk 0 (chunks 1, 2: one; launch_ransac_count)
__global__ void __launch_bounds__(256) k_ransac_count_mfma16(DevCtx c, int chunk, int nsplit)
{
    SVO_TL_SCOPE(c, TL_RS_COUNT, chunk);
    SVO_LATENCY_CHAIN(c);
    __shared__ double ops[(RC16_SUPER / 16) * 256];            // per tile of 16 pairs: B1 | B2 | phi[0..3] | phi[4..7], each [k][j]
    const int sblk = blockIdx.x / nsplit, split = blockIdx.x % nsplit;                                          // the splits of a group are neighbours in dispatch order
    const int side = blockIdx.y, vl = blockIdx.z, h0 = 3 * RS_CHUNK_BEGIN(chunk) + sblk * 64, tid = threadIdx.x;     // first SLOT of the block
    if (vl % c.oct_cap >= c.n_oct) return;
    const int n = c.trk_nk[vl];
    if (n <= SVO_LMEDS_MAX_N) return;   // (exactly seven pairs: findFundamentalMat's direct path; eight to fourteen: LMedS ranks medians, not counts -- see k_track_finalize)
    if (h0 >= RS_SLOT_END(chunk)) return;
    const float4* pts = (const float4*)(c.trk_pts + ((long long)vl * 2 + side) * c.max_kps * 4);
    const int w = tid >> 6, l = tid & 63, q = l >> 4, j = l & 15;
    const int hw = h0 + 16 * w;                                 // this wave's sixteen slots (one group: never across a region)
    // the block leaves when none of its four groups has anything to score.  Four threads decide, one per group, and the block reads
    // their verdicts from LDS: rs_bound moves while the launch runs, so waves looking for themselves could disagree, and a block of
    // which some waves have left no longer fills `ops`
    __shared__ int s_nlive[4], s_gen[4];
    if (tid < 4) {
        const int s0 = h0 + 16 * tid;
        const bool in = s0 < RS_SLOT_END(chunk) && s0 + 16 <= SVO_RANSAC_SLOTS;
        s_nlive[tid] = in ? rs_group_live(c, vl, side, s0) : 0;
        s_gen[tid] = in ? rs_group_generated(c, vl, side, s0) : 0;
    }
    __syncthreads();
    const int nlive_w = s_nlive[w];
    const bool block_dead = s_nlive[0] <= 0 && s_nlive[1] <= 0 && s_nlive[2] <= 0 && s_nlive[3] <= 0;      // (block-uniform: read from LDS)
    // a block none of whose groups was even GENERATED leaves at once, ticket and all: rs_gen and rs_nvalid do not move during the launch,
    // so the blocks of the group agree on it (rs_bound does move: a block that finds its groups merely out of reach stays for the ticket)
    if (s_gen[0] <= 0 && s_gen[1] <= 0 && s_gen[2] <= 0 && s_gen[3] <= 0) return;
    bool dead = nlive_w <= 0;
    // this block's share of the pairs: whole tiles of 16
    const int tiles_all = (n + 15) >> 4, tiles_per = (tiles_all + nsplit - 1) / nsplit;
    const int p0 = min(n, 16 * tiles_per * split), p1 = min(n, 16 * tiles_per * (split + 1));
    const int hs = min(hw, SVO_RANSAC_SLOTS - 16);
    const double* F = c.rs_F + (((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + hs) * 9;
    const double* Gd = c.rs_guard + (((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + hs) * 2;
    // operand A of the seven products: lane l holds row i = l % 16 (a model), column k = l / 16
    const double* Fa = F + 9 * j;
    const bool k3 = q < 3;
    double a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    double dminA[4] = { 0, 0, 0, 0 }, dminB[4] = { 0, 0, 0, 0 };
    if (!dead) {                                                // (a dead group's slots hold leftovers of earlier frames: not even read)
        a1 = k3 ? Fa[q] : 0.0; a2 = k3 ? Fa[3 + q] : 0.0;                 // (F0 F1 F2), (F3 F4 F5): a, b of l  = F (x1 y1 1)
        a3 = k3 ? Fa[3 * q] : 0.0; a4 = k3 ? Fa[3 * q + 1] : 0.0;         // (F0 F3 F6), (F1 F4 F7): a, b of l' = F^T (x2 y2 1)
        a5 = Fa[q]; a6 = Fa[4 + q]; a7 = q == 0 ? Fa[8] : 0.0;            // F0..F3 | F4..F7 | F8: the bilinear form
        // the four models this lane gets verdicts for: 4 r + q
#pragma unroll
        for (int r = 0; r < 4; r++) { dminA[r] = Gd[2 * (4 * r + q)]; dminB[r] = Gd[2 * (4 * r + q) + 1]; }
    }
    const double b7 = q == 0 ? 1.0 : 0.0;
    const double lo = 1.0 - 1.4901161193847656e-08, hi = 1.0 + 2.384185791015625e-07;         // 1 - 2^-26, 1 + 2^-22 (the error is compared AS A FLOAT: up to 1 + 2^-24 it rounds to 1.0f = inlier)
    bool dead_late = false;                                    // this wave stopped early: its partial counts are still published (below the floor)
    const int floor_cnt = (chunk && nsplit == 1) ? c.rs_floor[(vl * 2 + side) * 2 + (chunk - 1)] : 0x7FFFFFFF;
    int cnt[4] = { 0, 0, 0, 0 };
    for (int sb = p0; sb < p1 && !block_dead; sb += RC16_SUPER) {
        __syncthreads();                                                         // the previous 256 pairs have been consumed
        {
            const float4 p = pts[min(sb + tid, p1 - 1)];
            const double x1 = (double)p.x, y1 = (double)p.y, x2 = (double)p.z, y2 = (double)p.w;
            double* o = ops + (tid >> 4) * 256 + (tid & 15);
            o[0] = x1; o[16] = y1; o[32] = 1.0; o[48] = 0.0;
            o[64] = x2; o[80] = y2; o[96] = 1.0; o[112] = 0.0;
            o[128] = x1 * x2; o[144] = y1 * x2; o[160] = x2; o[176] = x1 * y2;     // products of two floats: exact in double
            o[192] = y1 * y2; o[208] = y2; o[224] = x1; o[240] = y1;
        }
        __syncthreads();
        if (dead || dead_late) continue;
        const int ntile = min(RC16_SUPER / 16, (p1 - sb + 15) >> 4);
        for (int t = 0; t < ntile; t++) {
            const int base = sb + 16 * t;
            if (nsplit == 1 && chunk && base && (t & 7) == 0) {
                // records only (see k_ransac_count_mfma): a wave that walks ALL pairs of its lane stops once none of its sixteen models can
                // exceed the floor even if every remaining pair were an inlier; the partial counts it leaves are below the floor (no
                // record) and, as under-estimates, only loosen the bound they feed
                bool hopeless = true;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int gsum = cnt[r];
                    gsum += __builtin_amdgcn_update_dpp(0, gsum, 0xB1, 0xF, 0xF, false);
                    gsum += __builtin_amdgcn_update_dpp(0, gsum, 0x4E, 0xF, 0xF, false);
                    gsum += __builtin_amdgcn_update_dpp(0, gsum, 0x141, 0xF, 0xF, false);
                    gsum += __builtin_amdgcn_update_dpp(0, gsum, 0x140, 0xF, 0xF, false);
                    hopeless = hopeless && (gsum + (n - base) <= floor_cnt);
                }
                if (__ballot(hopeless) == ~0ull && c.debug_mode != 16) { dead_late = true; break; }
            }
            const double* ot = ops + t * 256 + l;
            const double b1 = ot[0], b2 = ot[64], b5 = ot[128], b6 = ot[192];
            const rc_d4 z = { 0.0, 0.0, 0.0, 0.0 };
            const rc_d4 aB = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, z, 0, 0, 0);
            const rc_d4 bB = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b1, z, 0, 0, 0);
            const rc_d4 aA = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b2, z, 0, 0, 0);
            const rc_d4 bA = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b2, z, 0, 0, 0);
            rc_d4 D = __builtin_amdgcn_mfma_f64_16x16x4f64(a5, b5, z, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(a6, b6, D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(a7, b7, D, 0, 0, 0);
            const bool valid = base + j < p1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                // (the screen's norms may round any way they like -- fused here --: its band is 2^-26 wide, an ulp 2^-53.  Deciding both sides
                // by ONE band test on min(|l|^2, |l'|^2) and only when both guards hold was tried in round 5: 101 us per launch instead of 58 --
                // one side's guard fails often, and "out" by the other side alone is what keeps those tests off the replay path)
                const double denB = __builtin_fma(aB[r], aB[r], bB[r] * bB[r]), denA = __builtin_fma(aA[r], aA[r], bA[r] * bA[r]), dd = D[r] * D[r];
                const bool okA = denA >= dminA[r], okB = denB >= dminB[r];
                const bool inA = okA & (dd <= denA * lo), inB = okB & (dd <= denB * lo);
                const bool outA = okA & (dd >= denA * hi), outB = okB & (dd >= denB * hi);
                int v = (inA & inB) ? 1 : 0;
                if (__builtin_expect((valid & !((inA & inB) | outA | outB)) || c.debug_mode == 13 || c.debug_mode == 54, 0)) {
                    const float4 p = pts[min(base + j, n - 1)];
                    v = fm_inlier(F + 9 * (4 * r + q), p.x, p.y, p.z, p.w);
                }
                cnt[r] += valid ? v : 0;
            }
        }
    }
    // the 16 lanes of a DPP row hold the partial counts of models 4 r + q
    int* gcnt = c.rs_cnt + ((long long)vl * 2 + side) * SVO_RANSAC_SLOTS;
    // No __threadfence() anywhere here: an agent-scope fence writes back and invalidates the XCD's whole L2 on this part (eight
    // L2s, not coherent with each other), and 17 000 blocks doing that made the launch 2.3 ms.  Everything the blocks tell each other
    // goes through device-scope ATOMICS, which are performed at the memory side: the partial counts are RETURNING atomic adds (the wave
    // holds their results, i.e. they have been performed, before it reaches the barrier), the ticket is taken after the barrier, and the
    // last block reads the sums with atomic loads.
    int seen = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int v = cnt[r];
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
        // a real model of a live group (not filler, not a leftover); everything else keeps the zero the hypothesis kernel left
        if (j == 0 && !dead && 4 * r + q < nlive_w && v > 0) seen |= __hip_atomic_fetch_add(&gcnt[hw + 4 * r + q], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (seen < 0) atomicOr(&c.status[vl / c.oct_cap], SVO_ST_INTERNAL);          // (counts are never negative: this only makes the adds above returning ones)
    // the ticket of this block's 64 slots: every one of the nsplit blocks takes exactly one, computed or not
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
        int* ticket = c.rs_ticket + ((long long)vl * 2 + side) * (SVO_RANSAC_SLOTS / 16) + h0 / 16;
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == nsplit - 1;
        if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // nobody else touches it until the next launch
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < 64) {
        // best count, FIRST slot that has it: max over (count << 6 | 63 - slot), over the slots this block still sees alive
        const int gw = tid >> 4, nl = s_nlive[gw];
        const int cv = (nl > 0 && (tid & 15) < nl) ? __hip_atomic_load(gcnt + h0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int key = cv > 0 ? ((cv << 6) | (63 - tid)) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o, 64));
        if (tid == 0 && key > 0) rs_publish_best(c, vl, side, chunk, n, key >> 6, h0 + 63 - (key & 63));
    }
}

// inlier counts: RC_HB models per 256-thread block, points streamed once per thread.  The F matrices are read through
// a wave-uniform address (scalar loads into SGPRs: a VALU operand each, no LDS round trip per use).  The block's best
// model then tightens rs_bound (see above), so that later blocks of the launch and the next chunk stop earlier.
// RC_HB models per block: 16 when many lanes fill the GPU anyway (one fetch of the points serves 16 matrices), 4 when a
// few lanes leave it empty and the block's own latency is what a frame waits for
template <int RC_HB>
__global__ void __launch_bounds__(256) k_ransac_count(DevCtx c, int chunk)
{
    SVO_TL_SCOPE(c, TL_RS_COUNT, chunk);
    SVO_LATENCY_CHAIN(c);
    __shared__ int cnt_s[RC_HB];
    const int side = blockIdx.y, vl = blockIdx.z, h0 = 3 * RS_CHUNK_BEGIN(chunk) + blockIdx.x * RC_HB, tid = threadIdx.x;      // first SLOT of the block
    if (vl % c.oct_cap >= c.n_oct) return;
    const int n = c.trk_nk[vl];
    if (n <= SVO_LMEDS_MAX_N) return;   // (exactly seven pairs: findFundamentalMat's direct path; eight to fourteen: LMedS ranks medians, not counts -- see k_track_finalize)
    if (h0 >= RS_SLOT_END(chunk)) return;
    __shared__ int s_nlive;                                                   // one thread decides for the block (see k_ransac_count_mfma)
    if (tid == 0) s_nlive = rs_group_live(c, vl, side, h0);
    if (tid < RC_HB) cnt_s[tid] = 0;
    __syncthreads();
    const int nlive = s_nlive;
    if (nlive <= 0) return;                                                   // nothing generated here, or samples the sequential stop never reaches
    const float4* pts = (const float4*)(c.trk_pts + ((long long)vl * 2 + side) * c.max_kps * 4);
    const double* F = c.rs_F + (((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + h0) * 9;
    int cnt[RC_HB];
#pragma unroll
    for (int h = 0; h < RC_HB; h++) cnt[h] = 0;
    // four points per thread in registers, models in the outer loop: one scalar fetch of a model's matrix serves 1024
    // point tests (with the points outermost the sixteen matrices, 288 SGPRs' worth, were fetched again for every point)
    for (int base = 0; base < n; base += 4 * 256) {
        float4 p[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = base + q * 256 + tid; p[q] = i < n ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
        const int live = min(4, (n - base - tid + 255) / 256);            // this thread's points of the tile
#pragma unroll
        for (int h = 0; h < RC_HB; h++) {
            const double* Fh = F + 9 * h;
            int a = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { const int v = fm_inlier(Fh, p[q].x, p[q].y, p[q].z, p[q].w); a += q < live ? v : 0; }
            cnt[h] += a;
        }
    }
#pragma unroll
    for (int h = 0; h < RC_HB; h++) { const int v = wave_sum_uniform(cnt[h]); if ((tid & 63) == 0) atomicAdd(&cnt_s[h], v); }
    __syncthreads();
    if (tid < RC_HB) c.rs_cnt[((long long)vl * 2 + side) * SVO_RANSAC_SLOTS + h0 + tid] = tid < nlive ? cnt_s[tid] : 0;
    if (tid == 0) {
        int best = 0, best_h = h0;
        for (int h = 0; h < RC_HB; h++) if (h < nlive && cnt_s[h] > best) { best = cnt_s[h]; best_h = h0 + h; }
        rs_publish_best(c, vl, side, chunk, n, best, best_h);
    }
}

// pick the model a sequential RANSAC with the 0.99-confidence stop would have returned, apply both masks
// (S4:243-255), the consistency check (S4:282), and write tracked_pairs; then the bad-tracking gate (P:326-330)
// and the first-frame rule (P:348-352).
// gate_th >= 0 (single-octave contexts): the bad-tracking gate and the first-frame rule of k_track_gate, applied by this block itself
// (one launch less per frame); gate_th < 0: k_track_gate follows, after the blocks of all octaves
__global__ void __launch_bounds__(256) k_track_finalize(DevCtx c, int win_mode, int gate_th)
{
    SVO_TL_SCOPE(c, TL_TRK_FINAL, 0);
    SVO_LATENCY_CHAIN(c);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* in_l = smem;                         // max_kps
    unsigned char* in_r = in_l + c.max_kps;             // max_kps
    int* scan = (int*)(in_r + c.max_kps);               // 32
    __shared__ int s_best[2], s_cnt[2], s_vis[2], s_both;
    const int vl = blockIdx.x, lane_id = vl / c.oct_cap, oct = vl % c.oct_cap, tid = threadIdx.x;
    if (oct >= c.n_oct) return;
    LaneState& ls = c.lane[lane_id];
    if (!ls.has_prev) {
        if (tid == 0) { c.n_tracked[vl] = 0; if (gate_th >= 0) { svo_result& res = c.results[lane_id]; res.error_code = SVO_VOEC_FIRST_ITERATION; res.valid = 0; } }       // P:348-352
        return;
    }
    const int n = c.trk_nk[vl];
    // The sequential scan (records in (sample, model) order, each shrinking the budget) without its serial cost: the counts of
    // every model of every sample below rs_bound are all there (a slot that holds no model reads 0); a model is a RECORD when its
    // count exceeds every earlier one (and 6); only records can change the result or the budget.  Waves 0-1 / 2-3 take the two
    // sides: strict prefix maxima by a wave scan over chunks of 128 slots, the records' budgets K(count) computed in parallel,
    // then one thread walks the handful of records in order.
    __shared__ int rec_k[2][64], rec_c[2][64], rec_K[2][64], rec_n[2];
    __shared__ float s_thr[2];
    const bool lmeds = rs_is_lmeds(n);
    if (lmeds) {
        // Eight to fourteen pairs: LMeDSPointSetRegistrator::run (oracle v6: lmeds_fundamental).  The hypothesis kernels have solved all
        // SVO_LMEDS_ITERS samples (rs_first_bound; nothing shortens this budget); a thread per model: its n float errors, their median =
        // the element of rank n / 2 in nth_element's order; the winner is the smallest finite median, the earliest model among equals
        // (the sequential loop replaces its best on a STRICTLY smaller one): one 64-bit minimum over (median bits, slot).  Then the
        // mask threshold sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median), at least 0.001.
        __shared__ unsigned long long s_med[2];
        __shared__ int s_lkey[SVO_LMEDS_MAX_N * 256];
        const int side = tid >> 7, t = tid & 127;
        if (t == 0) s_med[side] = ~0ull;
        if (tid == 0) s_both = 0;
        __syncthreads();
        const long long sb = (long long)vl * 2 + side;
        const int lim_k = min(SVO_LMEDS_ITERS, c.rs_sched[vl * SVO_RS_ST + 2 + side]);          // the samples there are (getSubset may give up earlier)
        if (tid == 0 && ((c.rs_sched[vl * SVO_RS_ST + 10] >> side) & 1)) {                     // LMedS visits all 300: a schedule cut short by the attempt limit is a wrong answer
            atomicOr(&c.status[vl / c.oct_cap], SVO_ST_INTERNAL); atomicOr(&c.results[vl / c.oct_cap].status, (int)SVO_ST_INTERNAL);
        }
        const int lim = (lim_k + SVO_RANSAC_REG - 1) / SVO_RANSAC_REG * SVO_RANSAC_RSLOTS;
        const float4* pts = (const float4*)(c.trk_pts + sb * c.max_kps * 4);
        int* key = s_lkey + tid;                                                                // this thread's keys: key[i * 256]
        for (int sl = t; sl < lim; sl += 128) {
            const int reg = sl / SVO_RANSAC_RSLOTS;
            if (sl - reg * SVO_RANSAC_RSLOTS >= c.rs_nvalid[sb * (SVO_RANSAC_PAD / SVO_RANSAC_REG) + reg]) continue;      // no model in this slot
            if (c.rs_k[sb * SVO_RANSAC_SLOTS + sl] >= lim_k) continue;
            const double* Fg = c.rs_F + (sb * SVO_RANSAC_SLOTS + sl) * 9;
            double F[9];
#pragma unroll
            for (int i = 0; i < 9; i++) F[i] = Fg[i];
#pragma unroll 1
            for (int i = 0; i < n; i++) { const float4 p = pts[i]; key[i * 256] = fm_error_key(fm_error(F, p.x, p.y, p.z, p.w)); }
            int med = -1;
#pragma unroll 1
            for (int i = 0; i < n; i++) {
                const int ki = key[i * 256];
                int below = 0, equal = 0;
#pragma unroll 1
                for (int q = 0; q < n; q++) { const int kq = key[q * 256]; below += kq < ki; equal += kq == ki; }
                if (below <= n / 2 && n / 2 < below + equal) { med = ki; break; }
            }
            // `median < minMedian` from DBL_MAX down: an infinite or NaN median never wins; the errors are non-negative
            if (med >= 0 && med < 0x7F800000) atomicMin(&s_med[side], ((unsigned long long)(unsigned)med << 32) | (unsigned)sl);
        }
        __syncthreads();
        if (t == 0) {
            const unsigned long long best = s_med[side];
            int cnt = 0, slot = -1; float thr = 0.f;
            if (best != ~0ull) {
                slot = (int)(best & 0xFFFFFFFFull);
                const double m = (double)__uint_as_float((unsigned)(best >> 32));
                double sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (double)(n - 7)) * sqrt(m);
                if (!(sigma > 0.001)) sigma = 0.001;
                thr = (float)(sigma * sigma);
                const double* Fg = c.rs_F + (sb * SVO_RANSAC_SLOTS + slot) * 9;
                double F[9];
#pragma unroll
                for (int i = 0; i < 9; i++) F[i] = Fg[i];
#pragma unroll 1
                for (int i = 0; i < n; i++) { const float4 p = pts[i]; cnt += fm_error(F, p.x, p.y, p.z, p.w) <= thr; }
            }
            s_best[side] = slot; s_cnt[side] = cnt; s_thr[side] = thr; s_vis[side] = lim_k;
        }
        __syncthreads();
    } else {
        const int side = tid >> 7, t = tid & 127, wv = (tid >> 6) & 1, ln = tid & 63;
        __shared__ int run_max[2], wave_max[2][2];
        if (tid < 2) { rec_n[tid] = 0; run_max[tid] = 6; }
        if (tid == 0) s_both = 0;
        __syncthreads();
        // samples the scan can still reach -> the slots of their regions (all generated and zeroed this frame: see rs_bound)
        int lims[2];
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
            const int lk = n >= 8 ? min(min(max(c.rs_bound[vl * 2 + sd], c.rs_c0), SVO_RANSAC_HYP), c.rs_sched[vl * SVO_RS_ST + 2 + sd]) : 0;
            lims[sd] = lk > 0 ? ((lk - 1) / SVO_RANSAC_REG + 1) * SVO_RANSAC_RSLOTS : 0;
        }
        const int lim = lims[side];
        const int lim_both = max(lims[0], lims[1]);                          // block-uniform: the scan below stops where neither side has slots left (it used to walk all SVO_RANSAC_SLOTS)
        const int* gc = c.rs_cnt + ((long long)vl * 2 + side) * SVO_RANSAC_SLOTS;
        const int* gk = c.rs_k + ((long long)vl * 2 + side) * SVO_RANSAC_SLOTS;
        for (int base = 0; base < lim_both; base += 128) {                  // uniform trip count: barriers inside
            const int k = base + t;
            const int v = k < lim ? gc[k] : 0;
            // inclusive prefix max inside the wave, then across the side's two waves
            int m = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(m, o, 64); if (ln >= o) m = max(m, u); }
            if (ln == 63) wave_max[side][wv] = m;
            __syncthreads();
            const int before_wave = wv ? max(run_max[side], wave_max[side][0]) : run_max[side];
            int prev = __shfl_up(m, 1, 64); if (ln == 0) prev = 0;
            const int excl = max(before_wave, prev);                         // max of everything before k (and 6)
            if (v > excl) { const int slot = atomicAdd(&rec_n[side], 1); if (slot < 64) { rec_k[side][slot] = k; rec_c[side][slot] = v; } }
            __syncthreads();
            if (t == 0) run_max[side] = max(run_max[side], max(wave_max[side][0], wave_max[side][1]));
            __syncthreads();
        }
        // records are few (each at least one more inlier than the last; in practice ~ln(lim)); sort the <= 64 by index
        const int nr = min(rec_n[side], 64);
        if (t < nr) rec_K[side][t] = ransac_niters(rec_c[side][t], n, SVO_RANSAC_HYP);
        __syncthreads();
        // The walk, in the oracle's terms: the budget is tested once per SAMPLE, before its models are scored, so a record in
        // sample k counts iff k is below the budget left by the records of the samples before k (a record of the same sample
        // does not stop its later models).  `start` = that budget for the sample of the record at hand.
        if (t == 0 && rec_n[side] > 64) {                                   // more records than slots (a count creeping up one by one): the plain scan
            int best_s = -1, best_cnt = 0, niters = SVO_RANSAC_HYP, ks = -1, start = SVO_RANSAC_HYP;
            for (int sl = 0; sl < lim; sl++) {
                const int cnt = gc[sl];
                if (cnt <= (best_cnt > 6 ? best_cnt : 6)) continue;
                const int k = gk[sl];
                if (k != ks) start = niters;
                if (k >= start) break;
                best_cnt = cnt; best_s = sl; ks = k; niters = ransac_niters(cnt, n, niters);
            }
            s_best[side] = best_s; s_cnt[side] = best_s >= 0 ? best_cnt : 0;
            s_vis[side] = n >= 8 ? min(max(ks + 1, niters), c.rs_sched[vl * SVO_RS_ST + 2 + side]) : 0;
            if (n >= 8 && ((c.rs_sched[vl * SVO_RS_ST + 10] >> side) & 1) && max(ks + 1, niters) > c.rs_sched[vl * SVO_RS_ST + 2 + side]) {      // a sample the sampler could not draw would have been visited
                atomicOr(&c.status[vl / c.oct_cap], SVO_ST_INTERNAL); atomicOr(&c.results[vl / c.oct_cap].status, (int)SVO_ST_INTERNAL);
            }
        } else if (t == 0) {
            int best_s = -1, best_cnt = 0, niters = SVO_RANSAC_HYP, last = -1, ks = -1, start = SVO_RANSAC_HYP;
            for (int r = 0; r < nr; r++) {                                   // next record in slot order = smallest slot above `last`
                int sel = -1, sels = 0x7FFFFFFF;
                for (int q = 0; q < nr; q++) if (rec_k[side][q] > last && rec_k[side][q] < sels) { sels = rec_k[side][q]; sel = q; }
                if (sel < 0) break;
                const int k = gk[sels];
                if (k != ks) start = niters;
                if (k >= start) break;
                last = sels; best_s = sels; best_cnt = rec_c[side][sel]; ks = k;
                niters = min(niters, rec_K[side][sel]);
            }
            s_best[side] = best_s; s_cnt[side] = best_s >= 0 ? best_cnt : 0;
            // samples the sequential loop visits: it leaves at the first k that is no longer below the budget, and a record may
            // cut the budget below its own sample (oracle: svo_oracle_ransac_fundamental's n_hyp_used)
            s_vis[side] = n >= 8 ? min(max(ks + 1, niters), c.rs_sched[vl * SVO_RS_ST + 2 + side]) : 0;
            if (n >= 8 && ((c.rs_sched[vl * SVO_RS_ST + 10] >> side) & 1) && max(ks + 1, niters) > c.rs_sched[vl * SVO_RS_ST + 2 + side]) {      // (see the plain scan above)
                atomicOr(&c.status[vl / c.oct_cap], SVO_ST_INTERNAL); atomicOr(&c.results[vl / c.oct_cap].status, (int)SVO_ST_INTERNAL);
            }
        }
        // exactly seven pairs: cv::findFundamentalMat runs the 7-point kernel directly and sets the whole mask -- seven "inliers", no sample
        // visited, below the eight that S4:205, 240 ask for whichever model comes out (oracle: svo_oracle_ransac_fundamental, n == 7)
        if (t == 0 && n == 7) { s_best[side] = -1; s_cnt[side] = 7; s_vis[side] = 0; }
        if (t == 0) s_thr[side] = 1.0f;
        __syncthreads();
    }
    const bool goodFL = s_cnt[0] >= 8, goodFR = s_cnt[1] >= 8;       // S4:205, 240
    const bool use_f = goodFL && goodFR;                             // S4:243
    if (use_f) {
        // findInliers of both winners: the float error against 1.0f (RANSAC) or LMedS's own threshold; one side after the other
        // (the two matrices need not be live together)
#pragma unroll 1
        for (int sd = 0; sd < 2; sd++) {
            const double* Fg = c.rs_F + (((long long)vl * 2 + sd) * SVO_RANSAC_SLOTS + s_best[sd]) * 9;
            double f[9];
#pragma unroll
            for (int i = 0; i < 9; i++) f[i] = Fg[i];
            const float4* pp = (const float4*)(c.trk_pts + ((long long)vl * 2 + sd) * c.max_kps * 4);
            unsigned char* dst = sd ? in_r : in_l;
            const float thr = s_thr[sd];
            for (int i = tid; i < n; i += blockDim.x) { const float4 a = pp[i]; dst[i] = (unsigned char)(fm_error(f, a.x, a.y, a.z, a.w) <= thr); }
        }
    }
    __syncthreads();
    const unsigned* pL = (const unsigned*)c.bf_idx + ((long long)vl * 3 + 1) * c.max_kps;
    const unsigned* pR = (const unsigned*)c.bf_idx + ((long long)vl * 3 + 2) * c.max_kps;
    const int* kq = c.trk_kq + (long long)vl * c.max_kps;
    svo_index_pair* out = c.tracked + (long long)vl * c.max_kps;
    int t_total = 0;
    const int n_iter = (n + (int)blockDim.x - 1) / (int)blockDim.x;
    for (int it = 0; it < n_iter; it++) {
        const int i = it * blockDim.x + tid;
        int keep = 0, k = 0, tl = 0;
        if (i < n) {
            k = kq[i];
            keep = 1;
            if (use_f && (in_l[i] == 0 || in_r[i] == 0)) keep = 0;
            if (keep) atomicAdd(&s_both, 1);                          // svo_result.track_stats[SVO_TS_BOTH_MASKS]
            if (win_mode) tl = (int)pL[i];                            // survivor list of k_track_win: (previous, current) pairing (S4:708-714)
            else {
                tl = (int)(pL[k] & 0xFFFFu);
                const int tr = (int)(pR[k] & 0xFFFFu);
                if (tl != tr) keep = 0;                               // S4:282
            }
        }
        int tot;
        const int off = block_exclusive_scan(keep, scan, &tot);
        if (keep) { svo_index_pair p; p.first = k; p.second = tl; out[t_total + off] = p; }
        t_total += tot;
        __syncthreads();
    }
    if (tid == 0) {
        c.n_tracked[vl] = t_total;
        int* ts = c.results[lane_id].track_stats;                    // summed over the octaves (k_begin_frame zeroed them)
        atomicAdd(&ts[SVO_TS_INLIERS_L], s_cnt[0]); atomicAdd(&ts[SVO_TS_INLIERS_R], s_cnt[1]);
        atomicAdd(&ts[SVO_TS_HYP_L], s_vis[0]); atomicAdd(&ts[SVO_TS_HYP_R], s_vis[1]);
        atomicAdd(&ts[SVO_TS_BOTH_MASKS], s_both); atomicAdd(&ts[SVO_TS_TRACKED], t_total);
        if (gate_th >= 0 && t_total < gate_th) { ls.m_error = SVO_VOEC_BAD_TRACKING; c.results[lane_id].error_code = SVO_VOEC_BAD_TRACKING; }      // P:326-330
    }
}

// m_num_tracked_pairs_from_last_frame = sum over octaves (S4:743-752), then the bad-tracking gate (P:326-330) and the
// first-frame rule (P:348-352).  One thread per lane.
__global__ void k_track_gate(DevCtx c, int bad_tracking_th)
{
    SVO_TL_SCOPE(c, TL_TRK_FINAL, 1);
    const int lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane_id >= c.n_lanes) return;
    LaneState& ls = c.lane[lane_id];
    svo_result& res = c.results[lane_id];
    if (!ls.has_prev) { res.error_code = SVO_VOEC_FIRST_ITERATION; res.valid = 0; return; }
    int t = 0;
    for (int o = 0; o < c.n_oct; o++) t += c.n_tracked[lane_id * c.oct_cap + o];
    if (t < bad_tracking_th) { ls.m_error = SVO_VOEC_BAD_TRACKING; res.error_code = SVO_VOEC_BAD_TRACKING; }
}

// ------------------------------------------------------------------------------------------------------------
// Match-ID bookkeeping (params_general.vo_use_matches_ids; H:735-742): IDs follow a pairing through time.
//   reset (P:254-267): previous IDs renumbered 0..N-1, the frame becomes the key frame;
//   first frame (S3:67, 172-173): every pairing receives a fresh ID, octave by octave;
//   later frames (S4:268-305 / 716-733): tracked pairings inherit the previous ID, the others receive fresh ones in
//   index order; the key-frame counter is the number of current IDs <= m_last_kf_max_id (S4:747-751).
// One 256-thread block per lane; the fresh IDs are numbered with block scans so that the order is the reference's.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_match_ids(DevCtx c, unsigned flags)
{
    SVO_TL_SCOPE(c, TL_MATCH_IDS, 0);
    SVO_LATENCY_CHAIN(c);
    __shared__ int scan[32];
    __shared__ int s_next, s_kf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tracked_flag = smem;                  // max_kps
    const int lane_id = blockIdx.x, tid = threadIdx.x;
    LaneState& ls = c.lane[lane_id];
    const int cur = 1 - ls.prev_slot, prev = ls.prev_slot;
    int next_id = ls.last_match_id, kf_max = ls.last_kf_max_id;
    if (ls.reset_ids) {                                                          // P:254-267
        next_id = 0;
        if (ls.has_prev)
            for (int o = 0; o < c.n_oct; o++) {
                const int vl = lane_id * c.oct_cap + o, n = c.n_ids[vl * 2 + prev];
                int* pid = c.ids + ((long long)vl * 2 + prev) * c.max_kps;
                for (int m = tid; m < n; m += blockDim.x) pid[m] = next_id + m;
                next_id += n;
            }
        kf_max = next_id - 1;
    }
    __syncthreads();
    int kf_count = 0;
    if (!ls.has_prev && (flags & SVO_RUN_MATCH)) {                               // first frame: S3:172-173
        for (int o = 0; o < c.n_oct; o++) {
            const int vl = lane_id * c.oct_cap + o, n = c.n_matches[vl * 2 + cur];
            int* cid = c.ids + ((long long)vl * 2 + cur) * c.max_kps;
            for (int m = tid; m < n; m += blockDim.x) cid[m] = next_id + m;
            if (tid == 0) c.n_ids[vl * 2 + cur] = n;
            next_id += n;
        }
    } else if (ls.has_prev && (flags & SVO_RUN_TRACK)) {
        for (int o = 0; o < c.n_oct; o++) {
            const int vl = lane_id * c.oct_cap + o, ncm = c.n_matches[vl * 2 + cur], T = c.n_tracked[vl], npid = c.n_ids[vl * 2 + prev];
            const svo_index_pair* trk = c.tracked + (long long)vl * c.max_kps;
            const int* pid = c.ids + ((long long)vl * 2 + prev) * c.max_kps;
            int* cid = c.ids + ((long long)vl * 2 + cur) * c.max_kps;
            for (int k = tid; k < ncm; k += blockDim.x) tracked_flag[k] = 0;
            __syncthreads();
            for (int k = tid; k < T; k += blockDim.x) { const svo_index_pair p = trk[k]; cid[p.second] = p.first < npid ? pid[p.first] : 0; tracked_flag[p.second] = 1; }   // S4:290-291
            __syncthreads();
            for (int base = 0; base < ncm; base += blockDim.x) {                 // S4:302-304: fresh IDs in index order
                const int k = base + tid;
                const int fresh = (k < ncm && !tracked_flag[k]) ? 1 : 0;
                int tot;
                const int off = block_exclusive_scan(fresh, scan, &tot);
                if (fresh) cid[k] = next_id + off;
                next_id += tot;
                __syncthreads();
            }
            if (tid == 0) c.n_ids[vl * 2 + cur] = ncm;
            __threadfence_block();
            __syncthreads();
            for (int k = tid; k < ncm; k += blockDim.x) kf_count += cid[k] <= kf_max ? 1 : 0;   // S4:747-751
        }
        int tot;
        block_exclusive_scan(kf_count, scan, &tot);
        kf_count = tot;
    }
    (void)s_next; (void)s_kf;
    if (tid == 0) { ls.reset_ids = 0; ls.last_match_id = next_id; ls.last_kf_max_id = kf_max; ls.num_tracked_last_kf = kf_count; }
}

// ------------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------------
// contexts with max_kps > 4096 need more than the default 64 KB of dynamic LDS in some of the per-lane kernels
hipError_t configure_match(int max_kps)
{
    if (max_kps <= 4096) return hipSuccess;
    hipError_t e = svo_raise_dyn_smem((const void*)k_track_filter<32>, (size_t)(max_kps / 32) * 8 + (size_t)max_kps * 8 + (size_t)32 * 256 * 5);
    if (e != hipSuccess) return e;
    e = svo_raise_dyn_smem((const void*)k_match_lr_rbr, sizeof(unsigned) * 2 * max_kps + sizeof(int) * 32);
    return e;
}

// Which kernel forms this library holds: the product forms always; the A/B anchors (the int8 matcher SVO_HAM_FP4=0, the RANSAC count forms
// SVO_DEBUG_MODE=14 / 52) only when it was compiled with -DSVO_AB_KERNELS (libsvo_hip_ab.so: tests and A/B runs load it through SVO_HIP_LIB).
// svo_create refuses a knob that asks for a form the library does not hold.
bool svo_ab_kernels_built()
{
#ifdef SVO_AB_KERNELS
    return true;
#else
    return false;
#endif
}
bool svo_ab_form_requested(int debug_mode)
{
    const char* e = getenv("SVO_HAM_FP4");
    return (e && atoi(e) == 0) || debug_mode == 14 || debug_mode == 52;
}
// the FP4 form of the matrix-core brute force (k_hamming_f4) unless SVO_HAM_FP4=0
static bool hamming_fp4()
{
    static int f4 = -1;
    if (f4 < 0) { const char* e = getenv("SVO_HAM_FP4"); f4 = (e && atoi(e) == 0) ? 0 : 1; }      // default: FP4 (32.1 against 45.0 us per launch at 64 lanes, profiles/r04n, r04o); SVO_HAM_FP4=0 = the int8 form
    return f4 != 0;
}

void launch_hamming(const DevCtx& c, int mode, int nsplit, hipStream_t st)
{
    const dim3 grid(c.n_lanes * c.oct_cap, (c.max_kps + HM_QB - 1) / HM_QB, (mode ? 2 : 1) * nsplit);
#ifdef SVO_AB_KERNELS
    // (the int8 form reads the paired descriptors from a list laid out by a kernel of its own; the FP4 form gathers through the pairing lists)
    if (!hamming_fp4()) {
        if (mode) hipLaunchKernelGGL(k_gather_mdesc, dim3((c.max_kps * 8 + 255) / 256, c.n_lanes * c.oct_cap, 4), dim3(256), 0, st, c);
        hipLaunchKernelGGL(k_hamming, grid, dim3(256), 0, st, c, mode, nsplit);
        return;
    }
#endif
    hipLaunchKernelGGL(k_hamming_f4, grid, dim3(256), 0, st, c, mode, nsplit);       // (svo_create refuses SVO_HAM_FP4=0 in a library without the A/B forms)
}

void launch_match_lr_filter(const DevCtx& c, int one_to_one, double max_y_diff, hipStream_t st)
{
    const size_t sm = sizeof(unsigned) * c.max_kps + sizeof(int) * 32;
    hipLaunchKernelGGL(k_match_lr_filter, dim3(c.n_lanes * c.oct_cap), dim3(1024), sm, st, c, one_to_one, max_y_diff);
}

void launch_match_lr_rbr(const DevCtx& c, int one_to_one, double max_y_diff, double minimum_response, int max_distance, hipStream_t st)
{
    const size_t sm = sizeof(unsigned) * 2 * c.max_kps + sizeof(int) * 32;
    hipLaunchKernelGGL(k_match_lr_rbr, dim3(c.n_lanes * c.oct_cap), dim3(256), sm, st, c, one_to_one, max_y_diff, minimum_response, max_distance);
}
void launch_track_win(const DevCtx& c, int win_w, int win_h, hipStream_t st)
{
    hipLaunchKernelGGL(k_track_win, dim3(c.n_lanes * c.oct_cap), dim3(256), sizeof(unsigned) * c.max_kps + sizeof(int) * 40, st, c, win_w, win_h);      // (40: the sample schedule at its end uses scan[33..36])
}
void launch_match_ids(const DevCtx& c, unsigned flags, hipStream_t st)
{
    hipLaunchKernelGGL(k_match_ids, dim3(c.n_lanes), dim3(256), (size_t)c.max_kps, st, c, flags);
}
void launch_track_filter(const DevCtx& c, hipStream_t st)
{
    const size_t sm = (size_t)(c.max_kps / 32) * 2 * sizeof(unsigned) + (size_t)c.max_kps * 2 * sizeof(unsigned);
    if (c.max_kps > 4096) hipLaunchKernelGGL(k_track_filter<32>, dim3(c.n_lanes * c.oct_cap), dim3(256), sm + (size_t)32 * 256 * 5, st, c);      // + the per-thread candidates (4 + 1 bytes each) in LDS
    else hipLaunchKernelGGL(k_track_filter<16>, dim3(c.n_lanes * c.oct_cap), dim3(256), sm, st, c);
}
void launch_ransac_hyp(const DevCtx& c, int chunk, hipStream_t st)
{
    const int nh = RS_CHUNK_END(chunk) - RS_CHUNK_BEGIN(chunk);
    // one stream (few lanes): 16 lanes per sample, for latency; many lanes: one thread per sample, for instruction count
    // (debug_mode 50 forces the 16-lane form, 51 the one-thread form: tests/test_gpu_parity.py runs both against the oracle)
    const bool per_thread = c.debug_mode == 51 || (c.debug_mode != 50 && c.n_lanes * c.n_oct > 8);
    // (the samples of chunks 0-1 were numbered at the end of the tracker kernel) chunk 2: the rest, for the lanes whose budget reaches it
    if (chunk == 2) hipLaunchKernelGGL(k_ransac_schedule, dim3(c.n_lanes * c.oct_cap), dim3(256), 0, st, c, 1);
    if (per_thread) hipLaunchKernelGGL(k_ransac_hyp_thread, dim3((nh + 63) / 64, 2, c.n_lanes * c.oct_cap), dim3(64), 0, st, c, chunk);
    else hipLaunchKernelGGL(k_ransac_hyp, dim3((nh + 15) / 16, 2, c.n_lanes * c.oct_cap), dim3(256), 0, st, c, chunk);
}
void launch_ransac_count(const DevCtx& c, int chunk, hipStream_t st)
{
    // one stream (few lanes): four models per block on the VALU, for latency; many lanes: 16 x 16 matrix-core tiles.
    // debug modes (tests/test_gpu_parity.py runs each against the oracle): 14 = sixteen models per block on the VALU,
    // 52 = the 4 x 4-component matrix-core tiles, 53 = the 16 x 16 tiles whatever the lane count, 54 = the same with every
    // verdict replayed through the oracle's own expression.  The grids cover the chunk's model SLOTS (three per sample).
    const int ns = 3 * (((RS_CHUNK_END(chunk) + SVO_RANSAC_REG - 1) / SVO_RANSAC_REG) * SVO_RANSAC_REG - RS_CHUNK_BEGIN(chunk)), dm = c.debug_mode;
    const dim3 g16((ns + 15) / 16, 2, c.n_lanes * c.oct_cap);
    const bool many = c.n_lanes * c.n_oct > 8;
#ifdef SVO_AB_KERNELS
    if (dm == 14) { hipLaunchKernelGGL(k_ransac_count<16>, g16, dim3(256), 0, st, c, chunk); return; }
    if (dm == 52) { hipLaunchKernelGGL(k_ransac_count_mfma, g16, dim3(256), 0, st, c, chunk); return; }
#endif
    if (many || dm == 53 || dm == 54) {
        // The pairs of a lane CAN be split over several blocks per chunk (SVO_RC_SPLIT = s0[,s1[,s2]]): chunk 0 is two blocks per lane and side,
        // a chain of dependent f64 verdicts with one wave per SIMD (43 us alone, 29 us on four blocks each).  In the batched schedule it buys
        // nothing: 70.6 k pairs/s with (4, 1, 1), 71.0 k unsplit, 67.9 k with four everywhere (r05f, r05r).  Default: unsplit.
        static int split[3] = { 0, 0, 0 };
        if (!split[0]) {
            split[0] = RC16_NSPLIT0; split[1] = split[2] = 1;
            const char* e = getenv("SVO_RC_SPLIT");
            if (e) { int v[3] = { 0, 0, 0 }; const int k = sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]); for (int i = 0; i < 3; i++) { const int x = i < k ? v[i] : (k > 0 ? v[k - 1] : 0); if (x >= 1 && x <= 8) split[i] = x; } }
        }
        const int nsplit = split[chunk];
        hipLaunchKernelGGL(k_ransac_count_mfma16, dim3(((ns + 63) / 64) * nsplit, 2, c.n_lanes * c.oct_cap), dim3(256), 0, st, c, chunk, nsplit);
    }
    else hipLaunchKernelGGL(k_ransac_count<4>, dim3((ns + 3) / 4, 2, c.n_lanes * c.oct_cap), dim3(256), 0, st, c, chunk);
}
void launch_track_finalize(const DevCtx& c, int bad_tracking_th, int win_mode, hipStream_t st)
{
    const bool fold = c.oct_cap == 1;
    hipLaunchKernelGGL(k_track_finalize, dim3(c.n_lanes * c.oct_cap), dim3(256), (size_t)c.max_kps * 2 + sizeof(int) * 32, st, c, win_mode, fold ? bad_tracking_th : -1);
    if (!fold) hipLaunchKernelGGL(k_track_gate, dim3((c.n_lanes + 63) / 64), dim3(64), 0, st, c, bad_tracking_th);
}

// ------------------------------------------------------------------------------------------------------------
// standalone brute-force matcher on caller arrays (svo_hamming_match): same inner loop, plain row addressing
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_hamming_plain(const uint8_t* qd, int nq, const uint8_t* td, int nt, unsigned* out, int nsplit)
{
    __shared__ __attribute__((aligned(16))) unsigned long long tile[HM_TILE * 4];
    const int q = blockIdx.x * blockDim.x + threadIdx.x, split = blockIdx.y;
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (q < nq) { const ulonglong2* p = (const ulonglong2*)(qd + (long long)q * 32); const ulonglong2 a = p[0], b = p[1]; q0 = a.x; q1 = a.y; q2 = b.x; q3 = b.y; }
    const int tiles = (nt + HM_TILE - 1) / HM_TILE, t_per = (tiles + nsplit - 1) / nsplit;
    const int t_begin = split * t_per, t_end = min(tiles, t_begin + t_per);
    unsigned best = 0xFFFFFFFFu;
    for (int t = t_begin; t < t_end; t++) {
        const int j0 = t * HM_TILE, jn = min(HM_TILE, nt - j0);
        __syncthreads();
        if ((int)threadIdx.x < jn) {
            const ulonglong2* p = (const ulonglong2*)(td + (long long)(j0 + threadIdx.x) * 32);
            ((ulonglong2*)tile)[threadIdx.x * 2] = p[0]; ((ulonglong2*)tile)[threadIdx.x * 2 + 1] = p[1];
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < jn; j++) {
            const ulonglong2 a = ((const ulonglong2*)tile)[j * 2], b = ((const ulonglong2*)tile)[j * 2 + 1];
            const unsigned d = __popcll(q0 ^ a.x) + __popcll(q1 ^ a.y) + __popcll(q2 ^ b.x) + __popcll(q3 ^ b.y);
            best = min(best, (d << 16) | (unsigned)(j0 + j));
        }
    }
    if (q < nq && best != 0xFFFFFFFFu) { if (nsplit > 1) atomicMin(&out[q], best); else out[q] = best; }
}

void launch_hamming_plain(const uint8_t* q, int nq, const uint8_t* t, int nt, unsigned* out, int nsplit, hipStream_t st)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_hamming_plain, dim3((nq + 255) / 256, nsplit), dim3(256), 0, st, q, nq, t, nt, out, nsplit);
}
