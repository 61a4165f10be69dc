// k_gn.hip -- stage 5 on the device: one persistent 384-thread workgroup per lane (256 / 512 by SVO_GN_NT) runs the whole optimiser
// (gather, grid-NMS mask, triangulation, both Gauss-Newton phases, residual gating, pose inverse) in ONE launch
// with no host round trip.
//
// Replaces stage5_optimize (libstereo-odometry/src/stage5_optimization.cpp:392-736), m_evalRGN (:275-390) and
// m_pinhole_stereo_projection (:35-257).  Double precision throughout, like the reference.  The 28 sums of one
// iteration (21 of J^T J, 6 of J^T r, cost) are reduced inside each wave on the permlane-swap / DPP network and across the
// waves through 28 doubles of LDS each (gn_wave_sums): the Jacobian block is 4T x 6 with a 6x6 output, far too thin for an
// MFMA tile.  LDS per block: 33 KB (lists of up to 1024 tracked pairs; longer ones use DevCtx::gn_scratch).
// The reduction order differs from the reference's sequential loop, so results agree with the oracle to rounding
// (tests: 1e-4 rad / 1e-3 m), not bit for bit.
#include <mutex>
#include <map>
#include <utility>
#include "svo_device.h"
#include "svo_kernels.h"
#include <float.h>
#include <stdlib.h>

// Reciprocal, reciprocal square root, sine and cosine for the ONE thread that solves the normal equations and rebuilds the rotation
// between two barriers of every Gauss-Newton iteration (4.5 of the 8.3 us of an iteration were that thread: tests/dev/gn_breakdown.py).
// An IEEE f64 division is ~10 dependent instructions and the device library's sin / cos ~100 each with their range reduction;
// v_rcp_f64 / v_rsq_f64 plus two Newton steps are 5, and below 0.5 rad the Taylor series to x^17 / x^16 is exact to 1e-19.
// Stage 5 is held to the oracle by a tolerance (1e-3 m / 1e-4 rad, residuals 1e-6 relative), not bit for bit: these agree with the
// correctly rounded results to an ulp or two.
__device__ __forceinline__ double gn_rcp(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    return __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
}
__device__ __forceinline__ double gn_rsqrt(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    y = y * __builtin_fma(-0.5 * a * y, y, 1.5);
    return y * __builtin_fma(-0.5 * a * y, y, 1.5);
}
__device__ __forceinline__ void gn_sincos(double x, double& sn, double& cs)
{
    if (__builtin_expect(!(fabs(x) < 0.5), 0)) { sn = sin(x); cs = cos(x); return; }
    const double z = x * x;
    double ps = -2.8114572543455206e-15;                  // -1/17!
    ps = __builtin_fma(ps, z, 7.6471637318198164e-13);    //  1/15!
    ps = __builtin_fma(ps, z, -1.6059043836821613e-10);   // -1/13!
    ps = __builtin_fma(ps, z, 2.5052108385441720e-08);    //  1/11!
    ps = __builtin_fma(ps, z, -2.7557319223985893e-06);   // -1/9!
    ps = __builtin_fma(ps, z, 1.9841269841269841e-04);    //  1/7!
    ps = __builtin_fma(ps, z, -8.3333333333333332e-03);   // -1/5!
    ps = __builtin_fma(ps, z, 1.6666666666666666e-01);    //  1/3!
    sn = __builtin_fma(-x * z, ps, x);
    double pc = 4.7794773323873853e-14;                   //  1/16!
    pc = __builtin_fma(pc, z, -1.1470745597729725e-11);   // -1/14!
    pc = __builtin_fma(pc, z, 2.0876756987868100e-09);    //  1/12!
    pc = __builtin_fma(pc, z, -2.7557319223985888e-07);   // -1/10!
    pc = __builtin_fma(pc, z, 2.4801587301587302e-05);    //  1/8!
    pc = __builtin_fma(pc, z, -1.3888888888888889e-03);   // -1/6!
    pc = __builtin_fma(pc, z, 4.1666666666666664e-02);    //  1/4!
    pc = __builtin_fma(pc, z, -0.5);
    cs = __builtin_fma(pc, z, 1.0);
}

struct Rot { double r[9]; double dr[3][9]; int small_angle; };

// S5:45-163 (formulas kept as written, including the (w2^2+w3^2) factor of dr22dw3 at S5:162)
template <bool FAST = true>
__device__ void rodrigues_with_derivs(const double* dp, Rot& R)
{
    const double w1 = dp[0], w2 = dp[1], w3 = dp[2];
    const double w12 = w1 * w1, w22 = w2 * w2, w32 = w3 * w3;
    const double ss = w1 * w1 + w2 * w2 + w3 * w3;
    // FAST: inside the Gauss-Newton iteration (tolerance-checked); otherwise the correctly rounded library forms, as the oracle's
    // (getProjectedCoords is compared bit for bit)
    double itt_, tt, sin_tt, cos_tt;
    if (FAST) { itt_ = ss > 0.0 ? gn_rsqrt(ss) : 0.0; tt = ss * itt_; gn_sincos(tt, sin_tt, cos_tt); }
    else { tt = sqrt(ss); itt_ = 1.0 / tt; sin_tt = sin(tt); cos_tt = cos(tt); }
    const double tt2 = tt * tt, tt3 = tt2 * tt, tt4 = tt3 * tt;
    double* r = R.r;
    for (int i = 0; i < 9; i++) { R.dr[0][i] = 0; R.dr[1][i] = 0; R.dr[2][i] = 0; }
    if (tt < 1e-5) {
        R.small_angle = 1;
        r[0] = 1; r[1] = -w3; r[2] = w2; r[3] = w3; r[4] = 1; r[5] = -w1; r[6] = -w2; r[7] = w1; r[8] = 1;
        return;
    }
    R.small_angle = 0;
    // same expressions as S5:102-110 with the divisions by tt, tt2, tt3, tt4 folded into one reciprocal (an f64 divide
    // is a ~15-instruction sequence and this runs on one thread, on the critical path of every iteration)
    const double itt = itt_, itt2 = itt * itt, itt3 = itt2 * itt, itt4 = itt2 * itt2;
    const double u = (cos_tt - 1) * itt2;
    const double dudw1 = ((-sin_tt * w1 * itt) * tt2 - (cos_tt - 1) * 2 * w1) * itt4;
    const double dudw2 = ((-sin_tt * w2 * itt) * tt2 - (cos_tt - 1) * 2 * w2) * itt4;
    const double dudw3 = ((-sin_tt * w3 * itt) * tt2 - (cos_tt - 1) * 2 * w3) * itt4;
    const double v = sin_tt * itt;
    const double dvdw1 = w1 * (tt * cos_tt - sin_tt) * itt3;
    const double dvdw2 = w2 * (tt * cos_tt - sin_tt) * itt3;
    const double dvdw3 = w3 * (tt * cos_tt - sin_tt) * itt3;
    (void)tt3; (void)tt4;
    r[0] = (w22 + w32) * u + 1; r[1] = -w3 * v - w1 * w2 * u; r[2] = w2 * v - w1 * w3 * u;
    r[3] = w3 * v - w1 * w2 * u; r[4] = (w12 + w32) * u + 1; r[5] = -w1 * v - w2 * w3 * u;
    r[6] = -w2 * v - w1 * w3 * u; r[7] = w1 * v - w2 * w3 * u; r[8] = (w12 + w22) * u + 1;
    double (*d)[9] = R.dr;
    d[0][0] = (w22 + w32) * dudw1; d[1][0] = 2 * w2 * u + (w22 + w32) * dudw2; d[2][0] = 2 * w3 * u + (w22 + w32) * dudw3;
    d[0][1] = -w3 * dvdw1 - (w2 * u + w1 * w2 * dudw1); d[1][1] = -w3 * dvdw2 - (w1 * u + w1 * w2 * dudw2); d[2][1] = -(v + w3 * dvdw3) - w1 * w2 * dudw3;
    d[0][2] = w2 * dvdw1 - (w3 * u + w1 * w3 * dudw1); d[1][2] = (v + w2 * dvdw2) - w1 * w3 * dudw2; d[2][2] = w2 * dvdw3 - (w1 * u + w1 * w3 * dudw3);
    d[0][3] = w3 * dvdw1 - (w2 * u + w1 * w2 * dudw1); d[1][3] = w3 * dvdw2 - (w1 * u + w1 * w2 * dudw2); d[2][3] = (v + w3 * dvdw3) - w1 * w2 * dudw3;
    d[0][4] = 2 * w1 * u + (w12 + w32) * dudw1; d[1][4] = (w12 + w32) * dudw2; d[2][4] = 2 * w3 * u + (w12 + w32) * dudw3;
    d[0][5] = -(v + w1 * dvdw1) - w2 * w3 * dudw1; d[1][5] = -w1 * dvdw2 - (w3 * u + w2 * w3 * dudw2); d[2][5] = -w1 * dvdw3 - (w2 * u + w2 * w3 * dudw3);
    d[0][6] = -w2 * dvdw1 - (w3 * u + w1 * w3 * dudw1); d[1][6] = -(v + w2 * dvdw2) - w1 * w3 * dudw2; d[2][6] = -w2 * dvdw3 - (w1 * u + w1 * w3 * dudw3);
    d[0][7] = (v + w1 * dvdw1) - w2 * w3 * dudw1; d[1][7] = w1 * dvdw2 - (w3 * u + w2 * w3 * dudw2); d[2][7] = w1 * dvdw3 - (w2 * u + w2 * w3 * dudw3);
    d[0][8] = 2 * w1 * u + (w12 + w22) * dudw1; d[1][8] = 2 * w2 * u + (w12 + w22) * dudw2; d[2][8] = (w22 + w32) * dudw3;
}

// 6x6 symmetric solve (Eigen::JacobiSVD(H).solve(g) stand-in, S5:375-388).  Well-conditioned systems go through
// a Cholesky factorisation; anything else falls back to the oracle's Jacobi pseudo-inverse with Eigen's rank
// threshold.  Returns 0 when the condition number would be NaN (voecBadCondNumber).
__device__ int solve_sym6(const double* H, const double* g, double* x)
{
    double dmax = 0; bool bad = false;
    for (int i = 0; i < 36; i++) if (isnan(H[i]) || isinf(H[i])) bad = true;
    for (int i = 0; i < 6; i++) { if (isnan(g[i])) bad = true; dmax = fmax(dmax, fabs(H[i * 7])); }
    if (bad) return 0;
    double L[36];
    bool spd = dmax > 0;
    if (spd) {
        for (int j = 0; j < 6 && spd; j++) {
            double s = H[j * 6 + j];
            for (int k = 0; k < j; k++) s -= L[j * 6 + k] * L[j * 6 + k];
            if (!(s > 1e-13 * dmax)) { spd = false; break; }
            const double ljj = sqrt(s);
            L[j * 6 + j] = ljj;
            for (int i = j + 1; i < 6; i++) {
                double t = H[i * 6 + j];
                for (int k = 0; k < j; k++) t -= L[i * 6 + k] * L[j * 6 + k];
                L[i * 6 + j] = t / ljj;
            }
        }
    }
    if (spd) {
        double y[6];
        for (int i = 0; i < 6; i++) { double t = g[i]; for (int k = 0; k < i; k++) t -= L[i * 6 + k] * y[k]; y[i] = t / L[i * 6 + i]; }
        for (int i = 5; i >= 0; i--) { double t = y[i]; for (int k = i + 1; k < 6; k++) t -= L[k * 6 + i] * x[k]; x[i] = t / L[i * 6 + i]; }
        return 1;
    }
    // rank-deficient or indefinite: cyclic Jacobi eigen-decomposition + pseudo-inverse (oracle's solve_sym6)
    double A[36], V[36];
    for (int i = 0; i < 36; i++) { A[i] = H[i]; V[i] = (i % 7 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, dg = 0;
        for (int p = 0; p < 6; p++) { dg += A[p * 7] * A[p * 7]; for (int q = p + 1; q < 6; q++) off += A[p * 6 + q] * A[p * 6 + q]; }
        if (!(off > 1e-32 * dg)) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[p * 6 + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 6; k++) { const double akp = A[k * 6 + p], akq = A[k * 6 + q]; A[k * 6 + p] = cs * akp - sn * akq; A[k * 6 + q] = sn * akp + cs * akq; }
                for (int k = 0; k < 6; k++) { const double apk = A[p * 6 + k], aqk = A[q * 6 + k]; A[p * 6 + k] = cs * apk - sn * aqk; A[q * 6 + k] = sn * apk + cs * aqk; }
                for (int k = 0; k < 6; k++) { const double vkp = V[k * 6 + p], vkq = V[k * 6 + q]; V[k * 6 + p] = cs * vkp - sn * vkq; V[k * 6 + q] = sn * vkp + cs * vkq; }
            }
    }
    double smax = 0, smin = DBL_MAX;
    for (int i = 0; i < 6; i++) { const double s = fabs(A[i * 7]); if (s > smax) smax = s; if (s < smin) smin = s; }
    const double cond = smax / smin;
    if (isnan(cond)) return 0;
    const double thr = 6.0 * DBL_EPSILON * smax;
    for (int i = 0; i < 6; i++) x[i] = 0;
    for (int i = 0; i < 6; i++) {
        const double lam = A[i * 7];
        if (!(fabs(lam) > thr)) continue;
        double dot = 0;
        for (int k = 0; k < 6; k++) dot += V[k * 6 + i] * g[k];
        const double coef = dot / lam;
        for (int k = 0; k < 6; k++) x[k] += coef * V[k * 6 + i];
    }
    return 1;
}

// The general path out of line, fed from the 28 block sums in LDS and answering into LDS: rarely taken, and kept away from the
// iteration body so that neither its code nor a memory copy of H (an array whose address escapes lives in scratch) sits in it.
__device__ __attribute__((noinline)) int solve_sym6_from_sums(const double* tot, double* x_out)
{
    double H[36], g[6], x[6] = { 0, 0, 0, 0, 0, 0 };
    int h = 0;
    for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { const double v = tot[h++]; H[a * 6 + b] = v; H[b * 6 + a] = v; }
    for (int a = 0; a < 6; a++) g[a] = tot[21 + a];
    const int ok = solve_sym6(H, g, x);
    for (int a = 0; a < 6; a++) x_out[a] = x[a];
    return ok;
}

// CPose3D(CPose3DRotVec(delta).getInverse()) -> x y z yaw pitch roll (S5:717-718; oracle's svo_oracle_delta_to_pose)
__device__ void delta_to_pose(const double* dp, double* pose)
{
    const double w1 = dp[0], w2 = dp[1], w3 = dp[2];
    const double th = sqrt(w1 * w1 + w2 * w2 + w3 * w3);
    double R[9];
    if (th < 1e-10) { R[0] = 1; R[1] = -w3; R[2] = w2; R[3] = w3; R[4] = 1; R[5] = -w1; R[6] = -w2; R[7] = w1; R[8] = 1; }
    else {
        const double a = sin(th) / th, b = (1.0 - cos(th)) / (th * th);
        R[0] = 1 - b * (w2 * w2 + w3 * w3); R[1] = -a * w3 + b * w1 * w2; R[2] = a * w2 + b * w1 * w3;
        R[3] = a * w3 + b * w1 * w2; R[4] = 1 - b * (w1 * w1 + w3 * w3); R[5] = -a * w1 + b * w2 * w3;
        R[6] = -a * w2 + b * w1 * w3; R[7] = a * w1 + b * w2 * w3; R[8] = 1 - b * (w1 * w1 + w2 * w2);
    }
    const double Ri[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };
    pose[0] = -(Ri[0] * dp[3] + Ri[1] * dp[4] + Ri[2] * dp[5]);
    pose[1] = -(Ri[3] * dp[3] + Ri[4] * dp[4] + Ri[5] * dp[5]);
    pose[2] = -(Ri[6] * dp[3] + Ri[7] * dp[4] + Ri[8] * dp[5]);
    const double pitch = atan2(-Ri[6], hypot(Ri[0], Ri[3]));
    double yaw, roll;
    if (fabs(Ri[7]) + fabs(Ri[8]) < 10 * DBL_EPSILON) { roll = 0.0; yaw = pitch > 0 ? atan2(Ri[5], Ri[2]) : atan2(-Ri[5], -Ri[2]); }
    else { roll = atan2(Ri[7], Ri[8]); yaw = atan2(Ri[3], Ri[0]); }
    pose[3] = yaw; pose[4] = pitch; pose[5] = roll;
}

#define GN_NSUM 28
#define GN_NT_MAX 512        // the kernel is built for 256, 384 and 512 threads per lane (template parameter NT): SVO_GN_NT picks, see launch_gauss_newton
// Lists of up to GN_LCAP tracked pairs keep the sort / hash arrays of the stage-5 NMS mask and the two byte arrays in LDS
// (30 bytes per entry); longer lists use the lane's region of DevCtx::gn_scratch in global memory.  The LDS a block asks for
// therefore depends on what a frame of the workload tracks (a few hundred pairs), not on the capacity of the context's lists:
// 33 KB instead of 123 KB at max_kps 4096, so that a lane's block finds room on a CU beside the detector's tiles (round 5).
#define GN_LCAP 1024
struct GnShared {
    double part[GN_NT_MAX / 64][GN_NSUM];      // the 28 sums of every wave (reduced inside the wave on the DPP / permlane network)
    double tot[GN_NSUM];
    double step[6];
    double delta[6];
    double cost;
    Rot R;
    int ok;
    int n_non_masked;
};

// ---- the 28 sums of a wave without LDS ----------------------------------------------------------------------------------
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd rows of one register with the even rows of another (the upper
// half with the lower half): for a PAIR of sums (a, b) one swap per dword and one add leave "a over both rows" in one row and "b
// over both rows" in the other -- each step halves the number of live values instead of doubling the work: 28 -> 14 -> 7 values,
// then four DPP steps inside the rows of 16.  147 instructions per wave and iteration; a butterfly per sum would be 504.
template <bool ROW16>
__device__ __forceinline__ double gn_fold(double a, double b)
{
    const uint32_t a0 = (uint32_t)__double2loint(a), a1 = (uint32_t)__double2hiint(a), b0 = (uint32_t)__double2loint(b), b1 = (uint32_t)__double2hiint(b);
    if (ROW16) {
        const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false), r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false), r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
}
template <int CTRL>
__device__ __forceinline__ double gn_dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return v + __hiloint2double(hi, lo);
}
// acc[28] of every lane -> out[28] = the wave's sums (written by one lane per row of 16: 7 sums each)
__device__ __forceinline__ void gn_wave_sums(const double* acc, double* out)
{
    double n[14], m[7];
#pragma unroll
    for (int j = 0; j < 14; j++) n[j] = gn_fold<true>(acc[j], acc[j + 14]);       // rows 0, 2: sum j over two rows; rows 1, 3: sum j + 14
#pragma unroll
    for (int j = 0; j < 7; j++) m[j] = gn_fold<false>(n[j], n[j + 7]);            // lanes 0..31: n[j] over both halves; lanes 32..63: n[j + 7]
#pragma unroll
    for (int j = 0; j < 7; j++) {
        double v = m[j];
        v = gn_dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
        v = gn_dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
        v = gn_dpp_add<0x141>(v);      // row_half_mirror
        v = gn_dpp_add<0x140>(v);      // row_mirror
        m[j] = v;
    }
    const int lane = threadIdx.x & 63;
    if ((lane & 15) == 0) {
        const int row = lane >> 4, base = (row & 1) * 14 + (row >> 1) * 7;       // row 0: sums 0..6, row 1: 14..20, row 2: 7..13, row 3: 21..27
#pragma unroll
        for (int j = 0; j < 7; j++) out[base + j] = m[j];
    }
}

// Solve of the 6x6 normal equations by the square-root-free Cholesky form H = L D L^T (unit lower L), every index static
// (registers, no scratch).  This runs on ONE thread between two barriers of every iteration, so what counts is the length of
// its dependent chain: one division per column, where L L^T needs a square root and a division (~30 dependent instructions more
// per column).  Same pivots as the L L^T form (d_j = l_jj^2), same positivity test; the solution agrees to rounding, which is
// what stage 5's tolerance is stated in.  Returns false when a pivot is not safely positive; the caller then takes the general
// path (solve_sym6).
__device__ __forceinline__ bool chol6(const double* H, const double* g, double dmax, double* x)
{
    double L[6][6], d[6], inv[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double w[6];                                                        // w[k] = L[j][k] * d[k]
        double s = H[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) { w[k] = L[j][k] * d[k]; s = __builtin_fma(-L[j][k], w[k], s); }
        ok = ok && (s > 1e-13 * dmax);
        d[j] = s;
        inv[j] = gn_rcp(s);
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double t = H[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) t = __builtin_fma(-L[i][k], w[k], t);
            L[i][j] = t * inv[j];
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { double t = g[i];
#pragma unroll
        for (int k = 0; k < i; k++) t = __builtin_fma(-L[i][k], y[k], t);
        y[i] = t; }
#pragma unroll
    for (int i = 5; i >= 0; i--) { double t = y[i] * inv[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) t = __builtin_fma(-L[k][i], x[k], t);
        x[i] = t; }
    return ok;
}

// one m_evalRGN (S5:275-390) with the rotation taken from sh.R / sh.delta (prepared by thread 0).  On return every
// thread sees sh.ok, sh.cost, sh.step, and sh.delta / sh.R already advanced by the step (S5:576-577).
template <int GN_NT>
__device__ void eval_rgn(const GNParams& P, const svo_stereo_camera& cam, int T, const unsigned char* mask,
                         const double* lmk, const float* obs, double* residual, GnShared& sh, int dbg = 0)
{
    // Contraction ON inside the iteration (the file is compiled with -ffp-contract=off for the kernels that are compared bit for
    // bit): stage 5 is held to the oracle by a tolerance, and a lone wave pays ~5 cycles per instruction whatever it is, so
    // a * b + c as one v_fma_f64 instead of two instructions is a third of this function's arithmetic.
#pragma clang fp contract(fast)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const double b2 = P.use_robust_kernel ? P.kernel_param * P.kernel_param : 0;
    const double b2_1 = P.use_robust_kernel ? 1. / b2 : 0;
    const double t1 = sh.delta[3], t2 = sh.delta[4], t3 = sh.delta[5];
    const bool small_angle = sh.R.small_angle != 0;
    double acc[GN_NSUM];
#pragma unroll
    for (int i = 0; i < GN_NSUM; i++) acc[i] = 0;
    for (int m = tid; m < T; m += blockDim.x) {
        if (!mask[m] || dbg == 60) continue;
        const double X1p = lmk[3 * m], Y1p = lmk[3 * m + 1], Z1p = lmk[3 * m + 2];
        const double* r = sh.R.r;
        const double X1c = r[0] * X1p + r[1] * Y1p + r[2] * Z1p + t1;
        const double Y1c = r[3] * X1p + r[4] * Y1p + r[5] * Z1p + t2;
        const double Z1c = r[6] * X1p + r[7] * Y1p + r[8] * Z1p + t3;
        const double X2c = X1c - cam.baseline;
        // S5:189-193 and 251-254 with the 28 divisions by Z1c / Z1c^2 folded into two reciprocals
        const double iz = gn_rcp(Z1c), iz2 = iz * iz;
        const float pl_x = (float)(cam.l_fx * X1c * iz + cam.l_cx), pl_y = (float)(cam.l_fy * Y1c * iz + cam.l_cy);
        const float pr_x = (float)(cam.r_fx * X2c * iz + cam.r_cx), pr_y = (float)(cam.r_fy * Y1c * iz + cam.r_cy);
        double J[4][6];
        bool good = true;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double X1cd, Y1cd, Z1cd;
            if (j < 3) {
                if (small_angle) {
                    if (j == 0) { X1cd = 0; Y1cd = -Z1p; Z1cd = Y1p; }
                    else if (j == 1) { X1cd = Z1p; Y1cd = 0; Z1cd = -X1p; }
                    else { X1cd = -Y1p; Y1cd = X1p; Z1cd = 0; }
                } else {
                    const double* d = sh.R.dr[j];
                    X1cd = d[0] * X1p + d[1] * Y1p + d[2] * Z1p;
                    Y1cd = d[3] * X1p + d[4] * Y1p + d[5] * Z1p;
                    Z1cd = d[6] * X1p + d[7] * Y1p + d[8] * Z1p;
                }
            } else { X1cd = j == 3; Y1cd = j == 4; Z1cd = j == 5; }
            const double ju = (X1cd * Z1c - X1c * Z1cd) * iz2, jv = (Y1cd * Z1c - Y1c * Z1cd) * iz2, jur = (X1cd * Z1c - X2c * Z1cd) * iz2;
            J[0][j] = cam.l_fx * ju; J[1][j] = cam.l_fy * jv; J[2][j] = cam.r_fx * jur; J[3][j] = cam.r_fy * jv;
            if (isnan(ju) || isinf(ju) || isnan(jv) || isinf(jv) || isnan(jur) || isinf(jur)) good = false;
        }
        if (!good) continue;                                                   // S5:322
        const float* o = obs + 8 * (long long)m;
        const double ri[4] = { (double)(o[4] - pl_x), (double)(o[5] - pl_y), (double)(o[6] - pr_x), (double)(o[7] - pr_y) };   // float subtraction, S5:335-338
        const double s = ri[0] * ri[0] + ri[1] * ri[1] + ri[2] * ri[2] + ri[3] * ri[3];
        residual[m] = s;                                                        // S5:345
        double rho_p = 1, fi;
        if (P.use_robust_kernel) { const double q = 1 + (s * b2_1); rho_p = gn_rsqrt(q); const double nn = q * rho_p; fi = b2 * (nn - 1); }
        else fi = 0.5 * s;
        acc[27] += fi;
        int h = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double jr = J[0][a] * ri[0] + J[1][a] * ri[1] + J[2][a] * ri[2] + J[3][a] * ri[3];
            acc[21 + a] += rho_p * jr;                                          // gradient weighted ...
#pragma unroll
            for (int b = a; b < 6; b++) { acc[h] += J[0][a] * J[0][b] + J[1][a] * J[1][b] + J[2][a] * J[2][b] + J[3][a] * J[3][b]; h++; }   // ... Hessian NOT (S5:364-369)
        }
    }
    // block reduction of the 28 sums: inside every wave on the permlane / DPP network (no LDS), then GN_NT / 64 x 28 doubles through
    // LDS (1.3 KB; rounds 1-4 staged all 28 x GN_NT values there: 86 KB, and one more barrier).  Wave 0 adds the wave sums and its
    // first thread goes on to the solve: LDS operations of one wave are ordered, so no block barrier in between.
    gn_wave_sums(acc, sh.part[wid]);
    __syncthreads();
    if (wid == 0) {
        if (lane < GN_NSUM) {
            double sum = sh.part[0][lane];
#pragma unroll
            for (int w = 1; w < GN_NT / 64; w++) sum += sh.part[w][lane];
            sh.tot[lane] = sum;
        }
        wave_lds_sync();
    }
    if (tid == 0) {
        double H[36], g[6], x[6] = { 0, 0, 0, 0, 0, 0 };
        {
            int h = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int b = a; b < 6; b++) { const double v = sh.tot[h]; H[a * 6 + b] = v; H[b * 6 + a] = v; h++; }
            }
        }
        double dmax = 0; bool bad = false;
#pragma unroll
        for (int a = 0; a < 6; a++) { g[a] = sh.tot[21 + a]; bad = bad || isnan(g[a]); dmax = fmax(dmax, fabs(H[a * 7])); }
#pragma unroll
        for (int i = 0; i < 36; i++) bad = bad || isnan(H[i]) || isinf(H[i]);
        int ok;
        if (bad) ok = 0;
        else if (dbg == 61 || dbg == 63) ok = 1;
        else if (dmax > 0 && chol6(H, g, dmax, x)) ok = 1;
        else { ok = solve_sym6_from_sums(sh.tot, sh.step);                      // rank-deficient: pseudo-inverse path (out of line)
#pragma unroll
            for (int a = 0; a < 6; a++) x[a] = sh.step[a]; }
        sh.ok = ok;
        sh.cost = sh.tot[27];
#pragma unroll
        for (int a = 0; a < 6; a++) { sh.step[a] = x[a]; sh.delta[a] += x[a]; }
        if (ok && dbg != 62 && dbg != 63) { double d6[6]; for (int a = 0; a < 6; a++) d6[a] = sh.delta[a]; rodrigues_with_derivs(d6, sh.R); }
    }
    __syncthreads();
}

template <int GN_NT>
__global__ void __launch_bounds__(GN_NT) k_gauss_newton(DevCtx c, GNParams P, uint8_t* scratch)
{
    SVO_TL_SCOPE(c, TL_GN, 0);
    SVO_LATENCY_CHAIN(c);
    // dynamic LDS: keys[LC] u64 | hkey[2 LC] | hval[2 LC] | cellxy[LC] | state[LC] u8 | mask[LC] u8 | scan[40] | GnShared, LC = min(pmax, GN_LCAP).
    // A lane that tracks more than LC pairs (`big`) keeps the same arrays, sized pmax, in its region of c.gn_scratch.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PM = P.pmax, LC = PM < GN_LCAP ? PM : GN_LCAP;
    const int lane_id = blockIdx.x, tid = threadIdx.x;
    int* scan = (int*)(smem + (size_t)LC * 30);
    GnShared& sh = *(GnShared*)(scan + 40);
    LaneState& ls = c.lane[lane_id];
    svo_result& res = c.results[lane_id];
    if (!P.standalone && (!ls.has_prev || ls.m_error == SVO_VOEC_BAD_TRACKING)) return;      // P:305, P:332
    const svo_stereo_camera cam = c.cams[lane_id];
    // tracked pairs of all octaves, concatenated octave by octave (S5:419-461); toff[o] = first flat index of octave o
    const int cur = 1 - ls.prev_slot, prev = ls.prev_slot;
    int toff[SVO_MAX_OCTAVES + 1];
    toff[0] = 0;
    for (int o = 0; o < SVO_MAX_OCTAVES; o++) toff[o + 1] = toff[o] + (o < c.n_oct ? c.n_tracked[lane_id * c.oct_cap + o] : 0);
    int T = toff[SVO_MAX_OCTAVES];
    if (T > P.pmax) { T = P.pmax; if (tid == 0) { atomicOr(&c.status[lane_id], SVO_ST_KPS_OVERFLOW); atomicOr(&res.status, (int)SVO_ST_KPS_OVERFLOW); } }
    auto oct_of = [&](int i) { int o = 0; for (int q = 1; q < SVO_MAX_OCTAVES; q++) if (i >= toff[q] && q < c.n_oct) o = q; return o; };
    float* obs = c.gn_obs + (long long)lane_id * c.max_kps * 8;
    double* lmk = c.gn_lmk + (long long)lane_id * c.max_kps * 3;
    double* residual = c.residual + (long long)lane_id * c.max_kps;
    int* outl = c.outliers + (long long)lane_id * c.max_kps;
    int* cur_idx = c.trk_kq + (long long)lane_id * c.oct_cap * c.max_kps;      // tracked[..].second per point (the kept list of stage 4 is dead by now)

    const bool big = T > LC;                                                     // block-uniform
    const int AN = big ? PM : LC;
    unsigned char* abase = big ? scratch + (size_t)lane_id * ((size_t)PM * 30) : smem;
    unsigned long long* keys = (unsigned long long*)abase;
    uint32_t* hkey = (uint32_t*)(keys + AN);
    uint32_t* hval = hkey + 2 * AN;
    uint32_t* cellxy = hval + 2 * AN;
    unsigned char* state = (unsigned char*)(cellxy + AN);
    unsigned char* mask = state + AN;
    int Pn = 64; while (Pn < T) Pn <<= 1;
    // ---- gather the four keypoint lists (S5:419-461, single octave) and the NMS sort keys ----
    for (int i = tid; i < Pn; i += blockDim.x) keys[i] = 0;
    for (int i = tid; i < T; i += blockDim.x) mask[i] = 0;
    __syncthreads();
    for (int i = tid; i < T; i += blockDim.x) {
        const int o = oct_of(i), vl = lane_id * c.oct_cap + o;
        const svo_index_pair tp = c.tracked[(long long)vl * c.max_kps + (i - toff[o])];
        const svo_dmatch a = c.matches[match_base(c, vl, prev) + tp.first], b = c.matches[match_base(c, vl, cur) + tp.second];
        const svo_keypoint l1 = c.kps[feat_base(c, vl, prev, 0) + a.queryIdx], r1 = c.kps[feat_base(c, vl, prev, 1) + a.trainIdx];
        const svo_keypoint l2 = c.kps[feat_base(c, vl, cur, 0) + b.queryIdx], r2 = c.kps[feat_base(c, vl, cur, 1) + b.trainIdx];
        const float sn = (float)(size_t)(c.n_oct > 1 ? (1 << o) : 1);            // scale_norm, S5:422 (applied only when nOctaves > 1)
        float* ob = obs + 8 * (long long)i;
        if (c.n_oct > 1) { ob[0] = l1.x * sn; ob[1] = l1.y * sn; ob[2] = r1.x * sn; ob[3] = r1.y * sn; ob[4] = l2.x * sn; ob[5] = l2.y * sn; ob[6] = r2.x * sn; ob[7] = r2.y * sn; }
        else { ob[0] = l1.x; ob[1] = l1.y; ob[2] = r1.x; ob[3] = r1.y; ob[4] = l2.x; ob[5] = l2.y; ob[6] = r2.x; ob[7] = r2.y; }
        cur_idx[i] = tp.second;
        keys[i] = ((unsigned long long)ord32(l1.response) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
    }
    __threadfence_block();
    if (c.debug_mode == 64) return;                                            // timing probes 64-67: the set-up, phase by phase (tests/dev/gn_breakdown.py)
    // ---- m_non_max_sup mask overload on the previous-left coordinates (S5:465-474 -> S2:225-283); cap = T ----
    if (!big) {
        // at most GN_LCAP keys, a few hundred as a rule: every key counts the keys above it (the keys are unique, so the counts are the descending ranks).
        // All lanes read the same LDS word at a time (a broadcast): T reads per key and two barriers, where the bitonic network
        // needs 45 compare-exchange stages at 512 keys (~10 us of the ~25 us this kernel spends before its first iteration).
        unsigned long long* sorted = (unsigned long long*)hkey;             // 2 PM u32 = PM u64, free until the NMS builds its hash
        __syncthreads();
        // (round 6: this loop was 26 of the kernel's ~44 us of set-up at 380 tracks -- one dependent ds_read_b64 per key, ~70 cycles each.
        // Four keys per step through two 16-byte reads, four steps unrolled: sixteen keys in flight.  The pad keys[T .. Pn) is 0: never above.)
        for (int i = tid; i < T; i += blockDim.x) {
            const unsigned long long k = keys[i];
            int above = 0;
            const ulonglong2* k2 = (const ulonglong2*)keys;
#pragma unroll 4
            for (int j = 0; j < (T + 3) / 4; j++) {
                const ulonglong2 a = k2[2 * j], b = k2[2 * j + 1];
                above += (a.x > k ? 1 : 0) + (a.y > k ? 1 : 0) + (b.x > k ? 1 : 0) + (b.y > k ? 1 : 0);
            }
            sorted[above] = k;
        }
        __syncthreads();
        for (int i = tid; i < T; i += blockDim.x) keys[i] = sorted[i];
        __syncthreads();
    } else bitonic_sort_lds<true>(keys, Pn);
    if (c.debug_mode == 65) return;
    {
        const unsigned cell = (unsigned)((double)P.min_distance / 2.0);
        const float inv = 1.0f / (float)cell;
        const unsigned glx = (unsigned)(1 + (float)P.img_w * inv), gly = (unsigned)(1 + (float)P.img_h * inv);
        for (int i = tid; i < T; i += blockDim.x) {
            const int m = (int)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull));
            const size_t ux = (size_t)(obs[8 * (long long)m] * inv), uy = (size_t)(obs[8 * (long long)m + 1] * inv);
            cellxy[i] = (ux < glx && uy < gly) ? (((uint32_t)ux << 16) | (uint32_t)uy) : 0xFFFFFFFFu;
        }
        __syncthreads();
        int hsz = 128; while (hsz < 2 * T) hsz <<= 1;
        grid_nms_block<0>(T, gly, cellxy, hkey, hval, hsz, state, scan + 32);
        int cnt = 0;
        for (int i = tid; i < T; i += blockDim.x) if (state[i] == 1) { mask[(int)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull))] = 1; cnt++; }
        int tot; block_exclusive_scan(cnt, scan, &tot);
        if (tid == 0) sh.n_non_masked = tot;
        __syncthreads();
    }
    int n_non_masked = sh.n_non_masked;
    if (c.debug_mode == 66) return;
    if (n_non_masked < 8) { if (tid == 0) { res.valid = 0; res.n_residual = 0; res.n_outliers = 0; } return; }       // S5:521-526
    // ---- triangulation (S5:529-544) ----
    const double cul = cam.l_cx, cvl = cam.l_cy, fl = cam.l_fx, cur_ = cam.r_cx, fr = cam.r_fx, baseline = cam.baseline;
    auto triangulate = [&]() {
        for (int m = tid; m < T; m += blockDim.x) {
            if (!mask[m]) continue;
            const double ul = obs[8 * (long long)m], vl = obs[8 * (long long)m + 1], ur = obs[8 * (long long)m + 2];
            const double b_d = baseline / (fl * (cur_ - ur) + fr * (ul - cul));
            lmk[3 * m] = b_d * fr * (ul - cul); lmk[3 * m + 1] = b_d * fr * (vl - cvl); lmk[3 * m + 2] = b_d * fl * fr;
        }
    };
    triangulate();
    if (c.debug_mode == 67) return;
    if (tid == 0) {
        double d6[6] = { 0, 0, 0, 0, 0, 0 };
        if (P.use_custom_initial_pose) { for (int k = 0; k < 6; k++) d6[k] = P.init[k]; }                            // S5:504-505
        else if (P.use_previous_pose_as_initial) { for (int k = 0; k < 6; k++) d6[k] = ls.last_pose[k]; }            // S5:506-507
        for (int k = 0; k < 6; k++) sh.delta[k] = d6[k];
        rodrigues_with_derivs(d6, sh.R);
    }
    __threadfence_block();
    __syncthreads();
    if (c.debug_mode == 10) return;
    double pCost = 0, cCost = 0; bool done = false, abort_ = false;
    unsigned timesInc = 0; int num_it = 0, num_it_final = 0, err_code = res.error_code;
    bool first = true;
    // Both phases run through ONE copy of the iteration body (a second inlined copy of eval_rgn made the kernel ~100 KB of code, more
    // than the instruction cache holds: every iteration then fetched its code again).  phase 0 = S5:549-598, phase 1 = S5:650-700;
    // timesInc, pCost, cCost carry over from one to the other.
    const bool cap1 = (c.debug_mode == 11) || (c.debug_mode >= 60 && c.debug_mode <= 63);
    int n_res = 0, n_out = 0;
    for (int phase = 0; phase < 2; phase++) {
        const int limit = phase ? P.max_iters : P.initial_max_iters;
        int it = 0;
        done = false; abort_ = false;
        while (it < limit && !done && !abort_ && !(cap1 && it >= 1)) {
            pCost = cCost;
            if (first) { for (int m = tid; m < T; m += blockDim.x) residual[m] = DBL_MAX; first = false; }            // S5:296
            eval_rgn<GN_NT>(P, cam, T, mask, lmk, obs, residual, sh, c.debug_mode);
            if (phase == 0) err_code = SVO_VOEC_NONE;                                                                // S5:299
            cCost = sh.cost;
            if (!sh.ok) {
                if (tid == 0) {
                    ls.m_error = SVO_VOEC_BAD_COND_NUMBER; res.valid = 0; res.num_it = phase ? num_it : it;
                    if (phase == 0) { res.error_code = SVO_VOEC_BAD_COND_NUMBER; res.n_residual = 0; res.n_outliers = 0; }      // S5:380-386, 569-573
                    else { res.num_it_final = it; res.error_code = err_code; res.n_residual = T; res.n_outliers = n_out; }      // S5:670-675 (result.error_code untouched)
                }
                return;
            }
            double m2 = 0;
            for (int k = 0; k < 6; k++) m2 += sh.step[k] * sh.step[k];
            if (it > 0) {
                done = sqrt(m2) < P.min_mod_out_vector;
                if (pCost < cCost) { if (++timesInc > (unsigned)P.max_incr_cost) { err_code = phase ? SVO_VOEC_INCR_FUNC_COST_STG2 : SVO_VOEC_INCR_FUNC_COST_STG1; abort_ = true; } }
            }
            it++;
            // no barrier here: eval_rgn ends with one, and thread 0 rewrites sh.* only behind the next call's two barriers
        }
        if (phase) { num_it_final = it; break; }
        num_it = it;
        // ---- keep only the inliers (S5:601-611); "outliers" receives the INLIER cur-match indices ----
        n_res = first ? 0 : T;
        for (int base = 0; base < n_res; base += blockDim.x) {
            const int i = base + tid;
            int keep = 0;
            if (i < n_res) { if (residual[i] > P.residual_threshold) mask[i] = 0; else keep = 1; }
            int tot;
            const int off = block_exclusive_scan(keep, scan, &tot);
            if (keep) outl[n_out + off] = cur_idx[i];
            n_out += tot;
            __syncthreads();
        }
        {
            int cnt = 0;
            for (int m = tid; m < T; m += blockDim.x) cnt += mask[m];
            int tot; block_exclusive_scan(cnt, scan, &tot);
            n_non_masked = tot;
            __syncthreads();
        }
        if (n_non_masked < 8) {                                                                                      // S5:616-621
            if (tid == 0) { res.valid = 0; res.num_it = num_it; res.error_code = err_code; res.n_residual = n_res; res.n_outliers = n_out; }
            return;
        }
        triangulate();                                                                                               // S5:623-638
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) {
        double pose[6], delta[6];
        for (int k = 0; k < 6; k++) delta[k] = sh.delta[k];
        delta_to_pose(delta, pose);                                                                                  // S5:717-718
        for (int k = 0; k < 6; k++) { res.outPose[k] = pose[k]; res.delta[k] = delta[k]; }
        if (!P.use_custom_initial_pose && P.use_previous_pose_as_initial) for (int k = 0; k < 6; k++) ls.last_pose[k] = delta[k];   // S5:720-721
        res.tracked_feats_from_last_frame = T;                                                                       // S5:724
        res.tracked_feats_from_last_KF = ls.num_tracked_last_kf;                                                     // S5:725
        res.num_it = num_it; res.num_it_final = num_it_final; res.error_code = err_code;
        res.n_residual = n_res > T ? n_res : T; res.n_outliers = n_out;
        res.valid = !abort_;                                                                                         // S5:727
    }
}

static size_t gn_smem(int pmax)
{
    const int lc = pmax < GN_LCAP ? pmax : GN_LCAP;
    return (size_t)lc * 30 + sizeof(int) * 40 + sizeof(GnShared) + 16;
}
size_t gn_scratch_bytes_per_lane(int pmax) { return (size_t)pmax * 30; }

hipError_t svo_raise_dyn_smem(const void* kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> cur;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = cur[std::make_pair(dev, kernel)];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

hipError_t configure_gauss_newton(int pmax)
{
    const void* fn[3] = { (const void*)k_gauss_newton<256>, (const void*)k_gauss_newton<384>, (const void*)k_gauss_newton<512> };
    const int nt[3] = { 256, 384, 512 };
    for (int i = 0; i < 3; i++) {
        const hipError_t e = svo_raise_dyn_smem(fn[i], gn_smem(pmax)); (void)nt;
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void launch_gauss_newton(const DevCtx& c, const GNParams& P, hipStream_t st)
{
    // threads per lane: one pass of the per-track loop covers that many tracks per iteration (config 2 tracks ~290: 256 threads walk
    // the loop twice); SVO_GN_NT = 256 / 384 / 512 overrides the default for an A/B
    static int nt = 0;
    if (!nt) { const char* e = getenv("SVO_GN_NT"); const int v = e ? atoi(e) : 0; nt = (v == 256 || v == 384 || v == 512) ? v : 384; }
    if (nt == 512) hipLaunchKernelGGL(k_gauss_newton<512>, dim3(c.n_lanes), dim3(512), gn_smem(P.pmax), st, c, P, c.gn_scratch);
    else if (nt == 384) hipLaunchKernelGGL(k_gauss_newton<384>, dim3(c.n_lanes), dim3(384), gn_smem(P.pmax), st, c, P, c.gn_scratch);
    else hipLaunchKernelGGL(k_gauss_newton<256>, dim3(c.n_lanes), dim3(256), gn_smem(P.pmax), st, c, P, c.gn_scratch);
}

// ------------------------------------------------------------------------------------------------------------
// getProjectedCoords (common.cpp:415-466): landmarks of the untracked previous pairings triangulated as in stage 5
// (C:436-455) and projected after the change in pose through m_pinhole_stereo_projection (C:464, S5:180-195).
// One thread per point; same expressions, in the same order, as oracle/svo_oracle.c (svo_oracle_projected_coords).
// ------------------------------------------------------------------------------------------------------------
struct Delta6 { double v[6]; };

__global__ void __launch_bounds__(256) k_project_points(const float* uvu, int n, svo_stereo_camera cam, Delta6 dp, float* pix)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rot R; rodrigues_with_derivs<false>(dp.v, R);
    const double ul = (double)uvu[3 * i], vl = (double)uvu[3 * i + 1], ur = (double)uvu[3 * i + 2];
    const double cul = cam.l_cx, cvl = cam.l_cy, fl = cam.l_fx, cur = cam.r_cx, fr = cam.r_fx;
    const double disparity = fl * (cur - ur) + fr * (ul - cul);
    const double b_d = cam.baseline / disparity;
    const double X1p = b_d * fr * (ul - cul), Y1p = b_d * fr * (vl - cvl), Z1p = b_d * fl * fr;
    const double* r = R.r;
    const double X1c = r[0] * X1p + r[1] * Y1p + r[2] * Z1p + dp.v[3];
    const double Y1c = r[3] * X1p + r[4] * Y1p + r[5] * Z1p + dp.v[4];
    const double Z1c = r[6] * X1p + r[7] * Y1p + r[8] * Z1p + dp.v[5];
    const double X2c = X1c - cam.baseline;
    pix[4 * i + 0] = (float)(cam.l_fx * X1c / Z1c + cam.l_cx);
    pix[4 * i + 1] = (float)(cam.l_fy * Y1c / Z1c + cam.l_cy);
    pix[4 * i + 2] = (float)(cam.r_fx * X2c / Z1c + cam.r_cx);
    pix[4 * i + 3] = (float)(cam.r_fy * Y1c / Z1c + cam.r_cy);
}

void launch_project_points(const float* uvu, int n, const svo_stereo_camera& cam, const double* delta6, float* pix, hipStream_t st)
{
    Delta6 d; for (int k = 0; k < 6; k++) d.v[k] = delta6[k];
    if (n > 0) hipLaunchKernelGGL(k_project_points, dim3((n + 255) / 256), dim3(256), 0, st, uvu, n, cam, d, pix);
}
