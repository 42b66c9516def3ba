// lowres.hip -- plane production on the device.  Lookahead planes: frame_init_lowres_core (reference common/pixel.cpp:595-618)
// fused with extendPicBorder (pixel.cpp:893-911; row margins from common/ipfilter.cpp:56-74).
//
// One thread produces four horizontally adjacent samples of all four planes at one padded position.  A
// padded position outside the picture takes the sample of the nearest picture position (that is what the
// border extension copies there), so the margins are computed rather than copied and one launch writes
// every byte of the four planes that the reference writes.  The kernel is HBM-bound: per lowres sample it
// reads 4 source bytes and writes 4; the three source rows of an output row are shared with the next row
// through L2.
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

__device__ __forceinline__ uint32_t lowres_filter(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    // ((((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1), "slower than naive bilinear, but matches asm"
    return (((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1;
}

__global__ __launch_bounds__(256) void lowres_init_kernel(LowresArgs a)
{
    const int px4 = blockIdx.x * 256 + threadIdx.x; // group of four padded columns
    const int py = blockIdx.y;                      // padded row
    const int padw = a.width + 2 * a.margin_x;
    if (4 * px4 >= padw)
        return;
    const int y = min(max(py - a.margin_y, 0), a.lines - 1);
    const gptr_t s0 = as_global(a.src) + (long)(2 * y) * a.src_stride;
    const gptr_t s1 = s0 + a.src_stride, s2 = s1 + a.src_stride;
    uint32_t o[4] = { 0, 0, 0, 0 };
    const int x0 = 4 * px4 - a.margin_x;
    if (x0 >= 0 && x0 + 3 < a.width)
    {
        // inside the picture: source columns 2 x0 .. 2 x0 + 8 of three rows as dwords
        uint32_t r[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
            r[0][k] = ld4g(s0 + 2 * x0 + 4 * k);
            r[1][k] = ld4g(s1 + 2 * x0 + 4 * k);
            r[2][k] = ld4g(s2 + 2 * x0 + 4 * k);
        }
        auto px = [&](int row, int col) { return (r[row][col >> 2] >> (8 * (col & 3))) & 0xffu; };
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int c = 2 * i;
            o[0] |= lowres_filter(px(0, c), px(1, c), px(0, c + 1), px(1, c + 1)) << (8 * i);
            o[1] |= lowres_filter(px(0, c + 1), px(1, c + 1), px(0, c + 2), px(1, c + 2)) << (8 * i);
            o[2] |= lowres_filter(px(1, c), px(2, c), px(1, c + 1), px(2, c + 1)) << (8 * i);
            o[3] |= lowres_filter(px(1, c + 1), px(2, c + 1), px(1, c + 2), px(2, c + 2)) << (8 * i);
        }
    }
    else
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int x = min(max(4 * px4 + i - a.margin_x, 0), a.width - 1);
        const uint32_t r0[3] = { s0[2 * x], s0[2 * x + 1], s0[2 * x + 2] };
        const uint32_t r1[3] = { s1[2 * x], s1[2 * x + 1], s1[2 * x + 2] };
        const uint32_t r2[3] = { s2[2 * x], s2[2 * x + 1], s2[2 * x + 2] };
        o[0] |= lowres_filter(r0[0], r1[0], r0[1], r1[1]) << (8 * i);
        o[1] |= lowres_filter(r0[1], r1[1], r0[2], r1[2]) << (8 * i);
        o[2] |= lowres_filter(r1[0], r2[0], r1[1], r2[1]) << (8 * i);
        o[3] |= lowres_filter(r1[1], r2[1], r1[2], r2[2]) << (8 * i);
    }
    const long off = (long)py * a.dst_stride + 4 * px4; // dst pointers address the padded buffer's first byte
#pragma unroll
    for (int k = 0; k < 4; k++)
        *(uint32_t *)(a.dst[k] + off) = o[k];
}

// extendPicBorder (pixel.cpp:893-911), optionally behind weight_pp (pixel.cpp:510-529): every padded
// position takes the (weighted) sample of the nearest picture position.  One thread per four columns.
// In place (src == dst, no weighting) only positions outside the picture are written.
__global__ __launch_bounds__(256) void plane_border_kernel(PlaneBorderArgs a)
{
    const int px4 = blockIdx.x * 256 + threadIdx.x;
    const int py = a.py0 + blockIdx.y;
    const int padw = a.width + 2 * a.margin_x;
    if (4 * px4 >= padw)
        return;
    const int y = min(max(py - a.margin_y, 0), a.height - 1);
    const int x0 = 4 * px4 - a.margin_x;
    const bool inside = x0 >= 0 && x0 + 3 < a.width;
    if (inside && !a.weighted && py == y + a.margin_y)
        return; // picture samples stay as they are
    const gptr_t s = as_global(a.src) + (long)y * a.stride;
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int x = min(max(x0 + i, 0), a.width - 1);
        int v = s[x];
        if (a.weighted)
        {
            v = ((a.w0 * (v << 6) + a.round) >> a.shift) + a.offset; // val = src << (14 - 8), weight_pp_c
            v = min(max(v, 0), 255);
        }
        o |= (uint32_t)v << (8 * i);
    }
    // picture origin - margins; 4-byte aligned when stride, origin and margin_x are (checked by the host)
    *(uint32_t *)(a.dst + ((long)(py - a.margin_y) * a.stride + x0)) = o;
}

void launch_plane_border(const PlaneBorderArgs &a, hipStream_t st)
{
    const int padw = a.width + 2 * a.margin_x;
    if (a.pyn > 0)
        hipLaunchKernelGGL(plane_border_kernel, dim3(((padw + 3) / 4 + 255) / 256, a.pyn), dim3(256), 0, st, a);
}

void launch_lowres_init(const LowresArgs &a, hipStream_t st)
{
    const int padw = a.width + 2 * a.margin_x, padh = a.lines + 2 * a.margin_y;
    hipLaunchKernelGGL(lowres_init_kernel, dim3(((padw + 3) / 4 + 255) / 256, padh), dim3(256), 0, st, a);
}

} // namespace x265hip
