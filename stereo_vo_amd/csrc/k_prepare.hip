// k_prepare.hip -- stage 1 on the device: grey conversion and rectification of the level-0 images
// (libstereo-odometry/src/stage1_rectify.cpp:47-85; cvCvtColor BGR2GRAY at S1:50-51, CStereoRectifyMap::rectify ->
// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) at S1:66-72).  Arithmetic frozen in oracle/svo_oracle.c (svo_oracle_prepare):
//   grey  = (4899 R + 9617 G + 1868 B + 8192) >> 14
//   remap : 1/32-pixel fixed-point coordinates, weights (32-fx)(32-fy)*32 ..., out = (sum p*w + 16384) >> 15,
//           taps outside the image read 0.
// The maps arrive already in fixed point (svo_set_rectify_map converts the caller's float maps once):
//   .x = (sx + 1) | (sy + 1) << 16   (0xFFFFFFFF: the sample lies wholly outside)      .y = fx | fy << 8
// HBM-bound by design: 8 map bytes + ~1 source byte (x channels) read and 1 byte written per pixel.
#include "svo_device.h"
#include "svo_kernels.h"

__device__ __forceinline__ int grey_at(const uint8_t* p, int channels)
{
    if (channels == 1) return p[0];
    return (4899 * (int)p[2] + 9617 * (int)p[1] + 1868 * (int)p[0] + 8192) >> 14;
}

// grid = (ceil(w / 256), ceil(h / 4), image); one thread = 4 adjacent output pixels = one dword store
__global__ void __launch_bounds__(256) k_prepare(PrepArgs a)
{
    const int img = blockIdx.z;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= a.w || y >= a.h) return;
    const uint8_t* src = a.src[img];
    const uint2* map = a.maps ? a.maps[img] : nullptr;
    const int ch = a.channels;
    uint32_t out = 0;
    if (!map) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = min(x4 + k, a.w - 1);
            out |= (uint32_t)grey_at(src + (long long)y * a.src_stride + (long long)x * ch, ch) << (8 * k);
        }
    } else {
        uint2 m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) m[k] = map[(long long)y * a.w + min(x4 + k, a.w - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int v = 0;
            if (m[k].x != 0xFFFFFFFFu) {
                const int sx = (int)(m[k].x & 0xFFFFu) - 1, sy = (int)(m[k].x >> 16) - 1, fx = (int)(m[k].y & 0xFFu), fy = (int)(m[k].y >> 8);
                int p[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; dy++)
#pragma unroll
                    for (int dx = 0; dx < 2; dx++) {
                        const int xx = sx + dx, yy = sy + dy;
                        p[dy][dx] = (xx >= 0 && xx < a.w && yy >= 0 && yy < a.h) ? grey_at(src + (long long)yy * a.src_stride + (long long)xx * ch, ch) : 0;
                    }
                const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
                v = (p[0][0] * w00 + p[0][1] * w01 + p[1][0] * w10 + p[1][1] * w11 + 16384) >> 15;
            }
            out |= (uint32_t)v << (8 * k);
        }
    }
    *(uint32_t*)(a.dst + ((long long)img * a.dst_img_stride + (long long)y * a.dst_pitch + x4)) = out;   // pitch % 64 == 0: the row tail is padding
}

void launch_prepare(const PrepArgs& a, int n_img, hipStream_t st)
{
    hipLaunchKernelGGL(k_prepare, dim3((a.w + 255) / 256, (a.h + 3) / 4, n_img), dim3(256), 0, st, a);
}
