// me_device.h -- device-side building blocks of the ME cost path (gfx950, wave64).
//
// Everything here is written for one 64-lane wavefront cooperating on one unit of work (one PU
// search, one cost request); cross-lane traffic goes through a small per-wave LDS record and LDS
// atomics, never through block-wide barriers, so waves never wait on each other.
//
// Reference behaviour reproduced (paths under /root/reference/source):
//   SAD / SATD            common/pixel.cpp:92-156, 181-288
//   8-tap luma filters    common/ipfilter.cpp:76-160, 162-201, 241-282; taps TComRom.cpp:335-341
//   mvcost                encoder/bitcost.h:42-45
//   subpelCompare         encoder/motion.cpp:1126-1164
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/x265_hip.h"

namespace x265hip {

struct PlaneDesc
{
    const uint8_t *origin; // device address of picture sample (0,0)
    int stride;
    int pad;
};

// lives in device memory, one per context; kernels receive a pointer to it
struct DeviceTables
{
    PlaneDesc planes[X265_HIP_MAX_PLANES];
    const uint16_t *costtab[X265_HIP_MAX_QP]; // 65536 entries each, centre at +32768
};

__device__ __constant__ const int8_t k_taps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};

// ---- small helpers -------------------------------------------------------------------------

// 4 bytes from any byte address (gfx950 global/LDS loads accept unaligned dword addresses)
__device__ __forceinline__ uint32_t ld4u(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ int byte_of(uint32_t v, int i)
{
    return (int)((v >> (8 * i)) & 0xffu);
}

__device__ __forceinline__ int uniform(int v)
{
    return __builtin_amdgcn_readfirstlane(v);
}

// narrowing to int16 before the clamp is what the reference does (ipfilter.cpp:105-109)
__device__ __forceinline__ int clip_pixel(int v)
{
    int s = (int)(int16_t)v;
    return s < 0 ? 0 : (s > 255 ? 255 : s);
}

// make LDS traffic of this wave visible to its other lanes (single-wave scope)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- one 4x4 tile ---------------------------------------------------------------------------

// Fetch (or interpolate) the 4x4 reference tile whose top-left full-pel sample is `r`, for the
// quarter-pel fraction (xf, yf).  Output pix[row][col] in 0..255.  All work stays in registers.
__device__ __forceinline__ void ref_tile(const uint8_t *r_, int stride, int xf, int yf, int pix[4][4])
{
    if ((xf | yf) == 0)
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            uint32_t d = ld4u(r_ + y * stride);
#pragma unroll
            for (int x = 0; x < 4; x++)
                pix[y][x] = byte_of(d, x);
        }
    }
    else if (yf == 0)
    {
        // luma_hpp: dst = clip((sum_k src[x-3+k]*c[k] + 32) >> 6)
        int c[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            c[k] = k_taps[xf][k];
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            const uint8_t *p = r_ + y * stride - 3;
            uint32_t d0 = ld4u(p), d1 = ld4u(p + 4), d2 = ld4u(p + 8);
            int s[11];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                s[i] = byte_of(d0, i);
                s[4 + i] = byte_of(d1, i);
            }
            s[8] = byte_of(d2, 0);
            s[9] = byte_of(d2, 1);
            s[10] = byte_of(d2, 2);
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int acc = 0;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    acc += s[x + k] * c[k];
                pix[y][x] = clip_pixel((acc + 32) >> 6);
            }
        }
    }
    else if (xf == 0)
    {
        // luma_vpp: same rounding, taps run down the column.  Source rows stream through once and
        // are scattered into the 16 running sums (row r feeds output row y with tap r - y).
        int c[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            c[k] = k_taps[yf][k];
        int acc[4][4] = {};
#pragma unroll
        for (int r = 0; r < 11; r++)
        {
            uint32_t d = ld4u(r_ + (r - 3) * stride);
#pragma unroll
            for (int y = 0; y < 4; y++)
                if (r - y >= 0 && r - y < 8)
#pragma unroll
                    for (int x = 0; x < 4; x++)
                        acc[y][x] += byte_of(d, x) * c[r - y];
        }
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                pix[y][x] = clip_pixel((acc[y][x] + 32) >> 6);
    }
    else
    {
        // luma_hps (isRowExt = 1: rows -3 .. +7, value = sum - 8192 kept as int16) followed by
        // luma_vsp: clip((sum_k im[y-3+k]*c[k] + 2048 + (8192 << 6)) >> 12), streamed row by row
        int cx[8], cy[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            cx[k] = k_taps[xf][k];
            cy[k] = k_taps[yf][k];
        }
        int acc[4][4] = {};
#pragma unroll
        for (int r = 0; r < 11; r++)
        {
            const uint8_t *p = r_ + (r - 3) * stride - 3;
            uint32_t d0 = ld4u(p), d1 = ld4u(p + 4), d2 = ld4u(p + 8);
            int s[11];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                s[i] = byte_of(d0, i);
                s[4 + i] = byte_of(d1, i);
            }
            s[8] = byte_of(d2, 0);
            s[9] = byte_of(d2, 1);
            s[10] = byte_of(d2, 2);
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int h = 0;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    h += s[x + k] * cx[k];
                h = (int)(int16_t)(h - 8192);
#pragma unroll
                for (int y = 0; y < 4; y++)
                    if (r - y >= 0 && r - y < 8)
                        acc[y][x] += h * cy[r - y];
            }
        }
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                pix[y][x] = clip_pixel((acc[y][x] + 2048 + (8192 << 6)) >> 12);
    }
}

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// SAD or SATD of one 4x4 tile.  `f` walks the source block with `fstride`.
// SATD: sum |H4 * D * H4| / 2 per tile; the per-tile sum is always even, so halving per tile is
// exactly what the reference's 4x4 / 8x4 grouping yields (pixel.cpp:201-252).
__device__ __forceinline__ int tile_metric(const uint8_t *f, int fstride, const int pix[4][4], int metric)
{
    int d[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        uint32_t fv = ld4u(f + y * fstride);
#pragma unroll
        for (int x = 0; x < 4; x++)
            d[y][x] = byte_of(fv, x) - pix[y][x];
    }
    int acc = 0;
    if (metric == X265_HIP_SAD)
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                acc += iabs(d[y][x]);
        return acc;
    }
    int m[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int s01 = d[y][0] + d[y][1], t01 = d[y][0] - d[y][1];
        int s23 = d[y][2] + d[y][3], t23 = d[y][2] - d[y][3];
        m[y][0] = s01 + s23;
        m[y][1] = t01 + t23;
        m[y][2] = s01 - s23;
        m[y][3] = t01 - t23;
    }
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        int s01 = m[0][x] + m[1][x], t01 = m[0][x] - m[1][x];
        int s23 = m[2][x] + m[3][x], t23 = m[2][x] - m[3][x];
        acc += iabs(s01 + s23) + iabs(s01 - s23) + iabs(t01 + t23) + iabs(t01 - t23);
    }
    return acc >> 1;
}

// ---- mvcost -----------------------------------------------------------------------------------

// m_cost[v - mvp]; indices wrap inside the 65536-entry row (the reference reads outside its row
// there, which is undefined; in-range motion vectors never wrap)
__device__ __forceinline__ int tab_at(const uint16_t *tab, int mvp, int v)
{
    return tab[(uint16_t)(v - mvp + 32768)];
}

// BitCost::mvcost: the sum of the two entries truncated to uint16 (bitcost.h:45)
__device__ __forceinline__ int mvcost_q(const uint16_t *tab, int mvpx, int mvpy, int qx, int qy)
{
    return (uint16_t)(tab_at(tab, mvpx, qx) + tab_at(tab, mvpy, qy));
}

} // namespace x265hip
