// prim_kernels.hip -- single primitive calls on raw device pointers (gfx950): the kernels behind the
// function-pointer table (sad / sad_x3 / sad_x4 / satd / sa8d_inter / sse_pp, luma_hpp / vpp / hps / vsp, ...).
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

// ---------------------------------------------------------------------------------------------
// primitive-level kernels: one call of sad / sad_x3 / sad_x4 / satd on raw device pointers

__global__ __launch_bounds__(64) void prim_cmp_kernel(PrimCmpArgs a, int32_t *out)
{
    __shared__ int acc[4];
    const int lane = threadIdx.x;
    if (lane < 4)
        acc[lane] = 0;
    wave_sync();
    const int ntx = a.w >> 2, nt = ntx * (a.h >> 2);
    // sa8d_inter (pixel.cpp:1117-1141): 16x16 units, 8x8 units, or satd where the block has neither
    const int kind = a.metric != X265_HIP_SA8D ? 0 : ((a.w | a.h) & 15) == 0 ? 2 : ((a.w | a.h) & 7) == 0 ? 1 : 0;
    if (kind)
    {
        __shared__ int units[16];
        if (lane < 16)
            units[lane] = 0;
        wave_sync();
        const int nux = a.w >> (2 + kind), nunits = nux * (a.h >> (2 + kind));
        for (int i = lane; i < ntx * (a.h >> 3); i += 64) // even: lanes 2j, 2j + 1 hold the halves of an 8x8 block
        {
            const int ty = i / ntx, tx = i - ty * ntx;
            uint32_t pix[8][2];
            ref_strip<2>(a.fref[0] + (long)(8 * ty) * a.rstride + 4 * tx, a.rstride, 0, 0, pix);
            atomicAdd(&units[(ty >> (kind - 1)) * nux + (tx >> kind)],
                      sa8d_strip<true>(a.fenc + (long)(8 * ty) * a.fstride + 4 * tx, a.fstride, pix, tx & 1));
        }
        wave_sync();
        if (lane == 0)
        {
            int c = 0;
            for (int u = 0; u < nunits; u++)
                c += (units[u] + 2) >> 2;
            out[0] = c;
        }
        return;
    }
    const int total = a.nref * nt;
    for (int i = lane; i < total; i += 64)
    {
        int c = i / nt, t = i - c * nt;
        int ty = t / ntx, tx = t - ty * ntx;
        uint32_t pix[4][2];
        ref_tile(a.fref[c] + (long)(4 * ty) * a.rstride + 4 * tx, a.rstride, 0, 0, pix);
        const uint8_t *f = a.fenc + (long)(4 * ty) * a.fstride + 4 * tx;
        int v = a.metric == X265_HIP_SSE ? sse_strip<4, true>(f, a.fstride, pix) : tile_metric<true>(f, a.fstride, pix, a.metric != X265_HIP_SAD);
        atomicAdd(&acc[c], v);
    }
    wave_sync();
    if (lane < a.nref)
        out[lane] = acc[lane];
}

// luma_hpp / luma_vpp / luma_hps / luma_vsp on raw device pointers, one thread per output sample
__global__ __launch_bounds__(256) void prim_filter_kernel(PrimFilterArgs a)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.w * a.rows)
        return;
    const int y = idx / a.w, x = idx - y * a.w;
    int c[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        c[k] = k_taps[a.coeff][k];
    int acc = 0;
    if (a.kind == PRIM_HPP || a.kind == PRIM_HPS)
    {
        const uint8_t *p = (const uint8_t *)a.src + (long)y * a.sstride + x - 3;
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc += p[k] * c[k];
        if (a.kind == PRIM_HPP)
            ((uint8_t *)a.dst)[(long)y * a.dstride + x] = (uint8_t)clip_pixel((acc + 32) >> 6);
        else
            ((int16_t *)a.dst)[(long)y * a.dstride + x] = (int16_t)(acc - 8192);
    }
    else if (a.kind == PRIM_VPP)
    {
        const uint8_t *p = (const uint8_t *)a.src + (long)(y - 3) * a.sstride + x;
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc += p[(long)k * a.sstride] * c[k];
        ((uint8_t *)a.dst)[(long)y * a.dstride + x] = (uint8_t)clip_pixel((acc + 32) >> 6);
    }
    else if (a.kind == PRIM_VPS)
    {
        // luma_vps (ipfilter.cpp:203-239): shift 0, offset -8192
        const uint8_t *p = (const uint8_t *)a.src + (long)(y - 3) * a.sstride + x;
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc += p[(long)k * a.sstride] * c[k];
        ((int16_t *)a.dst)[(long)y * a.dstride + x] = (int16_t)(acc - 8192);
    }
    else if (a.kind == PRIM_VSS)
    {
        // luma_vss (ipfilter.cpp:284-317): shift 6, no offset, narrowed to int16 as the reference does
        const int16_t *p = (const int16_t *)a.src + (long)(y - 3) * a.sstride + x;
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc += p[(long)k * a.sstride] * c[k];
        ((int16_t *)a.dst)[(long)y * a.dstride + x] = (int16_t)(acc >> 6);
    }
    else if (a.kind == PRIM_P2S)
    {
        // luma_p2s (ipfilter.cpp:37-54)
        ((int16_t *)a.dst)[(long)y * a.dstride + x] = (int16_t)((((const uint8_t *)a.src)[(long)y * a.sstride + x] << 6) - 8192);
    }
    else // PRIM_VSP
    {
        const int16_t *p = (const int16_t *)a.src + (long)(y - 3) * a.sstride + x;
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc += p[(long)k * a.sstride] * c[k];
        ((uint8_t *)a.dst)[(long)y * a.dstride + x] = (uint8_t)clip_pixel((acc + 2048 + (8192 << 6)) >> 12);
    }
}

void launch_prim_cmp(const PrimCmpArgs &a, int32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(prim_cmp_kernel, dim3(1), dim3(64), 0, st, a, out);
}

// pixelavg_pp on raw device pointers (reference common/pixel.cpp:546-560), one thread per sample
__global__ __launch_bounds__(256) void prim_avg_kernel(PrimAvgArgs a)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.w * a.h)
        return;
    const int y = idx / a.w, x = idx - y * a.w;
    a.dst[(long)y * a.dstride + x] = (uint8_t)((a.src0[(long)y * a.sstride0 + x] + a.src1[(long)y * a.sstride1 + x] + 1) >> 1);
}

// luma_addAvg on raw device pointers (reference common/pixel.cpp:819-839), one thread per sample
__global__ __launch_bounds__(256) void prim_addavg_kernel(PrimAddAvgArgs a)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.w * a.h)
        return;
    const int y = idx / a.w, x = idx - y * a.w;
    a.dst[(long)y * a.dstride + x] = (uint8_t)clip_pixel((a.src0[(long)y * a.sstride0 + x] + a.src1[(long)y * a.sstride1 + x] + 64 + 2 * 8192) >> 7);
}

void launch_prim_addavg(const PrimAddAvgArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(prim_addavg_kernel, dim3((a.w * a.h + 255) / 256), dim3(256), 0, st, a);
}

void launch_prim_avg(const PrimAvgArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(prim_avg_kernel, dim3((a.w * a.h + 255) / 256), dim3(256), 0, st, a);
}

void launch_prim_filter(const PrimFilterArgs &a, hipStream_t st)
{
    const int total = a.w * a.rows;
    hipLaunchKernelGGL(prim_filter_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
}

} // namespace x265hip
