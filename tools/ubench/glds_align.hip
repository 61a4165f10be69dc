// glds_align.hip -- what gfx950's LDS-DMA (global_load_lds_dwordx4) tolerates, pinned on the hardware before k_fast /
// k_describe / k_resize stage their windows with it:
//   (a) per-lane SOURCE addresses at 0 / 4 / 8 / 12 (and 1) bytes off a 16-byte boundary;
//   (b) the wave-uniform LDS DESTINATION base at 0 / 4 / 8 bytes off a 16-byte boundary;
//   (c) inactive lanes (exec mask): is their 16-byte slot skipped, and do the active ones keep slot = lane * 16;
//   (d) the 4-byte form with per-lane sources 1 .. 3 bytes off a dword.
// Build: hipcc --offload-arch=gfx950 -O2 -o glds_align glds_align.hip;  prints one line per case: ok / WRONG (+ first difference).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>

__global__ void k_case(const uint8_t* src, uint8_t* out, int src_off, int lds_off, unsigned long long mask, int size)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 16 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < (int)sizeof(lds); i += 64) lds[i] = 0xEE;
    __syncthreads();
    if ((mask >> lane) & 1ull) {
        // lane L reads 16 (or 4) bytes from a scattered place: row L of a 256-byte-pitch image, plus the offset under test
        const uint8_t* g = src + (size_t)lane * 256 + src_off;
        if (size == 16) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)(lds + lds_off), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)(lds + lds_off), 4, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();
    for (int i = lane; i < (int)sizeof(lds); i += 64) out[i] = lds[i];
}

int main()
{
    const size_t SRC = 64 * 256 + 64;
    std::vector<uint8_t> h(SRC);
    for (size_t i = 0; i < SRC; i++) h[i] = (uint8_t)((i * 131u + (i >> 8) * 7u) & 0xFF);
    uint8_t* d_src; uint8_t* d_out;
    hipMalloc(&d_src, SRC); hipMalloc(&d_out, 64 * 16 + 64);
    hipMemcpy(d_src, h.data(), SRC, hipMemcpyHostToDevice);
    struct Case { int src_off, lds_off; unsigned long long mask; int size; const char* what; };
    const Case cases[] = {
        { 0, 0, ~0ull, 16, "x4 src%16=0  lds%16=0  all lanes" },
        { 4, 0, ~0ull, 16, "x4 src%16=4" }, { 8, 0, ~0ull, 16, "x4 src%16=8" }, { 12, 0, ~0ull, 16, "x4 src%16=12" },
        { 1, 0, ~0ull, 16, "x4 src%16=1 (byte-misaligned)" }, { 2, 0, ~0ull, 16, "x4 src%16=2" },
        { 0, 4, ~0ull, 16, "x4 lds%16=4" }, { 0, 8, ~0ull, 16, "x4 lds%16=8" }, { 8, 8, ~0ull, 16, "x4 src%16=8 lds%16=8" },
        { 0, 0, 0x00FFFFFFFFFFFFFFull, 16, "x4 lanes 0..55 only" }, { 0, 0, 0xAAAAAAAAAAAAAAAAull, 16, "x4 odd lanes only" },
        { 8, 0, 0x000FFFFFFFFFFFFFull, 16, "x4 src%16=8 lanes 0..51" },
        { 0, 0, ~0ull, 4, "x1 src%4=0" }, { 1, 0, ~0ull, 4, "x1 src%4=1" }, { 2, 0, ~0ull, 4, "x1 src%4=2" }, { 3, 0, ~0ull, 4, "x1 src%4=3" },
    };
    int bad = 0;
    for (const Case& c : cases) {
        hipMemset(d_out, 0, 64 * 16 + 64);
        hipLaunchKernelGGL(k_case, dim3(1), dim3(64), 0, 0, d_src, d_out, c.src_off, c.lds_off, c.mask, c.size);
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%-40s HIP ERROR %s\n", c.what, hipGetErrorString(e)); bad++; break; }
        uint8_t o[64 * 16 + 64]; hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
        uint8_t want[64 * 16 + 64]; memset(want, 0xEE, sizeof(want));
        for (int l = 0; l < 64; l++)
            if ((c.mask >> l) & 1ull) memcpy(want + c.lds_off + l * c.size, h.data() + (size_t)l * 256 + c.src_off, c.size);
        int first = -1;
        for (int i = 0; i < (int)sizeof(o); i++) if (o[i] != want[i]) { first = i; break; }
        if (first < 0) printf("%-40s ok\n", c.what);
        else {
            bad++;
            printf("%-40s WRONG at lds byte %d (lane slot %d): got", c.what, first, (first - c.lds_off) / c.size);
            for (int i = first; i < first + 8; i++) printf(" %02x", o[i]);
            printf(" want");
            for (int i = first; i < first + 8; i++) printf(" %02x", want[i]);
            printf("\n");
        }
    }
    printf("%d case(s) differ from 'dest = base + lane * size, bytes as addressed'\n", bad);
    return 0;
}
