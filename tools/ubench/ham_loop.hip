// Micro-benchmark: the Hamming inner loop with the train words already in SGPRs (no memory at all) vs from LDS vs from
// scalar loads: separates the VALU cost of a 256-bit popcount distance from the cost of feeding it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) k_regs(const uint32_t* __restrict__ q, uint32_t* out, int n, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4, uint32_t s5, uint32_t s6, uint32_t s7)
{
    uint32_t qw[8];
    for (int k = 0; k < 8; k++) qw[k] = q[threadIdx.x * 8 + k];
    unsigned best = 0xFFFFFFFFu;
#pragma unroll 4
    for (int j = 0; j < n; j++) {
        const uint32_t t[8] = { s0 + j, s1 ^ j, s2 + j, s3 ^ j, s4 + j, s5 ^ j, s6 + j, s7 ^ j };      // SALU
        unsigned d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __popc(qw[k] ^ t[k]);
        best = min(best, (d << 16) | (unsigned)j);
    }
    out[blockIdx.x * 256 + threadIdx.x] = best;
}

__global__ void __launch_bounds__(256) k_smem(const uint32_t* __restrict__ q, const uint32_t* __restrict__ tw, uint32_t* out, int n)
{
    uint32_t qw[8];
    for (int k = 0; k < 8; k++) qw[k] = q[threadIdx.x * 8 + k];
    unsigned best = 0xFFFFFFFFu;
    const uint32_t* __restrict__ base = tw + (blockIdx.x & 63) * 2048 * 8;
#pragma unroll 4
    for (int j = 0; j < n; j++) {
        const uint32_t* __restrict__ t = base + j * 8;
        unsigned d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __popc(qw[k] ^ t[k]);
        best = min(best, (d << 16) | (unsigned)j);
    }
    out[blockIdx.x * 256 + threadIdx.x] = best;
}

__global__ void __launch_bounds__(256) k_lds(const uint32_t* __restrict__ q, const uint32_t* __restrict__ tw, uint32_t* out, int n)
{
    __shared__ uint4 tile[512 * 2];
    uint32_t qw[8];
    for (int k = 0; k < 8; k++) qw[k] = q[threadIdx.x * 8 + k];
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = ((const uint4*)tw)[i];
    __syncthreads();
    unsigned best = 0xFFFFFFFFu;
#pragma unroll 4
    for (int j = 0; j < n; j++) {
        const uint4 a = tile[(j & 511) * 2], b = tile[(j & 511) * 2 + 1];
        const unsigned d = __popc(qw[0] ^ a.x) + __popc(qw[1] ^ a.y) + __popc(qw[2] ^ a.z) + __popc(qw[3] ^ a.w) + __popc(qw[4] ^ b.x) + __popc(qw[5] ^ b.y) + __popc(qw[6] ^ b.z) + __popc(qw[7] ^ b.w);
        best = min(best, (d << 16) | (unsigned)j);
    }
    out[blockIdx.x * 256 + threadIdx.x] = best;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount, blocks = n_cu * 8, n = 2048;
    uint32_t* q, *t, *out; hipMalloc(&q, 256 * 32); hipMalloc(&t, 64 * 2048 * 32); hipMalloc(&out, blocks * 256 * 4);
    hipMemset(q, 0x5A, 256 * 32); hipMemset(t, 0x3C, 64 * 2048 * 32);
    for (int variant = 0; variant < 3; variant++) {
        float best_ms = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            if (variant == 0) k_regs<<<blocks, 256>>>(q, out, n, 1, 2, 3, 4, 5, 6, 7, 8);
            if (variant == 1) k_smem<<<blocks, 256>>>(q, t, out, n);
            if (variant == 2) k_lds<<<blocks, 256>>>(q, t, out, n);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best_ms) best_ms = ms;
        }
        const double cyc = best_ms * 1e-3 * 2.4e9 / (8.0 * n);          // 8 waves per SIMD, n pair-steps each
        printf("%s: %.3f ms, %.1f cycles per wave pair-step per SIMD (18.5 instr)\n", variant == 0 ? "sgpr-const" : variant == 1 ? "scalar-load" : "lds-broadcast", best_ms, cyc);
    }
    return 0;
}
