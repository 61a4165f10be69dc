// Does "block b runs on XCD b % 8" survive CONCURRENT dispatches from several HIP streams?  k_resize / k_fast / k_describe map their tiles to
// XCDs through blockIdx & 7 (each XCD has its own L2); the schedule runs kernels of several contexts at once.  Every block records the XCD it
// runs on (HW_REG_XCC_ID) and then holds its CU for a while; the grid is launched alone, then on two and three streams at the same time.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/xcc_map.hip -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(int* out, int spin)
{
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = (int)(x & 15u);
    }
    const long long t0 = clock64();
    while (clock64() - t0 < spin) { }
}

static double match(const std::vector<int>& v) { long m = 0; for (size_t b = 0; b < v.size(); b++) m += v[b] == (int)(b & 7); return (double)m / v.size(); }
// longest run structure: fraction of blocks whose XCD equals (b + k) % 8 for the best constant shift k (a shifted round-robin keeps locality of
// CHUNKED mappings only if the shift is constant over the grid)
static double best_shift(const std::vector<int>& v, int* kbest) { double best = 0; for (int k = 0; k < 8; k++) { long m = 0; for (size_t b = 0; b < v.size(); b++) m += v[b] == (int)((b + k) & 7); if ((double)m / v.size() > best) { best = (double)m / v.size(); *kbest = k; } } return best; }

int main()
{
    const int NS = 3, grids[3] = { 2048, 8192, 30000 }, threads[2] = { 128, 256 };
    hipStream_t st[NS];
    for (int i = 0; i < NS; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    for (int g : grids) for (int nt : threads) for (int spin : { 2000, 20000 }) {
        int* d[NS]; std::vector<int> h[NS];
        for (int i = 0; i < NS; i++) { CK(hipMalloc(&d[i], sizeof(int) * g)); h[i].resize(g); }
        for (int conc = 1; conc <= NS; conc++) {
            for (int rep = 0; rep < 3; rep++) {
                for (int i = 0; i < conc; i++) CK(hipMemsetAsync(d[i], 0xFF, sizeof(int) * g, st[i]));
                CK(hipDeviceSynchronize());
                for (int i = 0; i < conc; i++) hipLaunchKernelGGL(probe, dim3(g), dim3(nt), 0, st[i], d[i], spin);
                CK(hipDeviceSynchronize());
                for (int i = 0; i < conc; i++) CK(hipMemcpy(h[i].data(), d[i], sizeof(int) * g, hipMemcpyDeviceToHost));
                printf("grid %5d x %3d threads, spin %5d, %d concurrent, rep %d:", g, nt, spin, conc, rep);
                for (int i = 0; i < conc; i++) { int k = 0; const double bs = best_shift(h[i], &k); printf("  stream %d: b%%8 match %.4f, best constant shift %d: %.4f", i, match(h[i]), k, bs); }
                printf("\n");
            }
        }
        for (int i = 0; i < NS; i++) CK(hipFree(d[i]));
    }
    return 0;
}
