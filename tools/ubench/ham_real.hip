// Experiment harness: the product's k_hamming launch geometry on synthetic descriptors, with switches, to find what
// keeps it at ~2x the VALU floor.  Variants are selected by template flags; prints ms and cycles per wave pair-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int QPT, int UNROLL, bool ATOMIC, bool LDS>
__global__ void __launch_bounds__(256) k_ham(const uint8_t* __restrict__ desc_q, const uint8_t* __restrict__ desc_t, const int* __restrict__ counts,
                                             unsigned* out, int max_kps, int nsplit)
{
    __shared__ uint4 tile[LDS ? 512 : 1];
    const int lane = blockIdx.y, split = blockIdx.z;
    const int nq = counts[lane * 2], nt = counts[lane * 2 + 1];
    const int qbase = blockIdx.x * (256 * QPT) + threadIdx.x;
    if ((int)(blockIdx.x * 256 * QPT) >= nq || nt <= 0) return;
    uint32_t qw[QPT][8];
#pragma unroll
    for (int i = 0; i < QPT; i++) {
        const int q = min(qbase + 256 * i, nq - 1);
        const uint4* p = (const uint4*)(desc_q + ((long long)lane * max_kps + q) * 32);
        const uint4 a = p[0], b = p[1];
        qw[i][0] = a.x; qw[i][1] = a.y; qw[i][2] = a.z; qw[i][3] = a.w; qw[i][4] = b.x; qw[i][5] = b.y; qw[i][6] = b.z; qw[i][7] = b.w;
    }
    const int per = (nt + nsplit - 1) / nsplit;
    const int j_begin = split * per, j_end = min(nt, j_begin + per);
    unsigned best[QPT];
#pragma unroll
    for (int i = 0; i < QPT; i++) best[i] = 0xFFFFFFFFu;
    const uint32_t* __restrict__ tw = (const uint32_t*)(desc_t + (long long)lane * max_kps * 32);
    if (!LDS) {
#pragma unroll UNROLL
        for (int j = j_begin; j < j_end; j++) {
            const uint32_t* __restrict__ t = tw + (long long)j * 8;
            uint32_t tv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) tv[k] = t[k];
#pragma unroll
            for (int i = 0; i < QPT; i++) {
                unsigned d = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) d += __popc(qw[i][k] ^ tv[k]);
                best[i] = min(best[i], (d << 16) | (unsigned)j);
            }
        }
    } else {
        for (int j0 = j_begin; j0 < j_end; j0 += 256) {
            __syncthreads();
            const int jj = min(j0 + (int)threadIdx.x, nt - 1);
            tile[threadIdx.x * 2] = ((const uint4*)tw)[jj * 2]; tile[threadIdx.x * 2 + 1] = ((const uint4*)tw)[jj * 2 + 1];
            __syncthreads();
            const int jn = min(256, j_end - j0);
#pragma unroll UNROLL
            for (int j = 0; j < jn; j++) {
                const uint4 a = tile[j * 2], b = tile[j * 2 + 1];
                const uint32_t tv[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
                for (int i = 0; i < QPT; i++) {
                    unsigned d = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) d += __popc(qw[i][k] ^ tv[k]);
                    best[i] = min(best[i], (d << 16) | (unsigned)(j0 + j));
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QPT; i++) {
        const int q = qbase + 256 * i;
        if (q < nq) {
            unsigned* o = out + (long long)lane * max_kps + q;
            if (ATOMIC) atomicMin(o, best[i]); else if (split == 0) *o = best[i];
        }
    }
}

template <int QPT, int UNROLL, bool ATOMIC, bool LDS>
static void run(const char* name, const uint8_t* dq, const uint8_t* dt, const int* cnt, unsigned* out, int lanes, int max_kps, int nsplit, int n, bool compact_grid)
{
    const int gx = compact_grid ? (n + 256 * QPT - 1) / (256 * QPT) : (max_kps + 256 * QPT - 1) / (256 * QPT);
    float best_ms = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipMemsetAsync(out, 0xFF, (size_t)lanes * max_kps * 4, 0);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        k_ham<QPT, UNROLL, ATOMIC, LDS><<<dim3(gx, lanes, nsplit), 256>>>(dq, dt, cnt, out, max_kps, nsplit);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best_ms) best_ms = ms;
    }
    const double wave_steps = (double)lanes * ((n + 63) / 64) * n / QPT * QPT;         // one step = one train against QPT queries
    const double cyc = best_ms * 1e-3 * 2.4e9 * 1024.0 / ((double)lanes * ((n + 63) / 64) * (double)n);
    printf("%-44s nsplit %2d grid_x %2d: %.3f ms  (%.1f cycles per wave per (train x 64 queries) per SIMD; floor ~80)\n", name, nsplit, gx, best_ms, cyc);
    (void)wave_steps;
}

int main()
{
    const int lanes = 64, max_kps = 4096, n = 1930;
    uint8_t* dq, *dt; int* cnt; unsigned* out;
    hipMalloc(&dq, (size_t)lanes * max_kps * 32); hipMalloc(&dt, (size_t)lanes * max_kps * 32); hipMalloc(&cnt, lanes * 8); hipMalloc(&out, (size_t)lanes * max_kps * 4);
    std::vector<uint32_t> h((size_t)lanes * max_kps * 8);
    uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    hipMemcpy(dq, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    hipMemcpy(dt, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> c(lanes * 2, n); hipMemcpy(cnt, c.data(), lanes * 8, hipMemcpyHostToDevice);
    for (int nsplit : { 4, 8, 16 }) {
        run<1, 4, true, false>("smem qpt1 unroll4 compact", dq, dt, cnt, out, lanes, max_kps, nsplit, n, true);
        run<1, 8, true, false>("smem qpt1 unroll8 compact", dq, dt, cnt, out, lanes, max_kps, nsplit, n, true);
        run<2, 4, true, false>("smem qpt2 unroll4 compact", dq, dt, cnt, out, lanes, max_kps, nsplit, n, true);
        run<1, 4, true, true>("lds  qpt1 unroll4 compact", dq, dt, cnt, out, lanes, max_kps, nsplit, n, true);
        run<2, 4, true, true>("lds  qpt2 unroll4 compact", dq, dt, cnt, out, lanes, max_kps, nsplit, n, true);
        run<4, 2, true, true>("lds  qpt4 unroll2 compact", dq, dt, cnt, out, lanes, max_kps, nsplit, n, true);
        run<4, 4, true, true>("lds  qpt4 unroll4 compact", dq, dt, cnt, out, lanes, max_kps, nsplit, n, true);
    }
    return 0;
}
