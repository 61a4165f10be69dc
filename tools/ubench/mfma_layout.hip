#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(int* out)
{
    const int l = threadIdx.x;
    v4i a = {0,0,0,0}, b = {0,0,0,0};
    if (l < 32) { a.x = (l + 1); b.x = 1; }          // byte 0 of k-block 0: A[row l][k0] = l+1, B[k0][col l] = 1
    v16i c; for (int i = 0; i < 16; i++) c[i] = 1000 * (i + 1);
    v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; i++) out[l * 16 + i] = d[i];
}
int main()
{
    int* d; hipMalloc(&d, 64 * 16 * 4); k<<<1, 64>>>(d); int h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) { printf("lane %2d:", l); for (int v = 0; v < 16; v++) printf(" %5d", h[l * 16 + v]); printf("\n"); }
    return 0;
}
