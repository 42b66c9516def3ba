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
    // the bytes of a plane's buffer relative to its origin, [first, end): what me_route_kernel's record guard checks a
    // job's reach against (api.hip's check_reach / check_inside on the device)
    long span_first[X265_HIP_MAX_PLANES], span_end[X265_HIP_MAX_PLANES];
    const float *bitsizes; // BitCost::s_bitsizes, 32769 entries (bitcost.cpp:68-80): inter_decide.hip
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

// the same from plane memory: a global_load instead of a flat_load (which also queues on the LDS path)
typedef const __attribute__((address_space(1))) uint8_t *gptr_t;
__device__ __forceinline__ gptr_t as_global(const uint8_t *p) { return (gptr_t)p; }
__device__ __forceinline__ uint32_t ld4g(gptr_t p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// 12 bytes from any byte address as three dwords, as one unaligned global_load_dwordx3.  (Round 1 fetched the
// enclosing aligned 16 bytes and shifted them into place with three v_alignbyte: one pass through the texture-address
// unit instead of two, but these kernels are short of VALU issue slots, not of load bandwidth: me_kernel 3 % faster
// with the plain load, cost_kernel unchanged.)
__device__ __forceinline__ void ld12g(gptr_t p, uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    u32x3 w;
    __builtin_memcpy(&w, (const void *)p, 12);
    d0 = w.x, d1 = w.y, d2 = w.z;
}

// 16 bytes from any byte address, one unaligned global_load_dwordx4
__device__ __forceinline__ uint4 ld16g(gptr_t p)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w;
    __builtin_memcpy(&w, (const void *)p, 16);
    return make_uint4(w.x, w.y, w.z, w.w);
}

// The same loads from LDS (a window of a reference plane staged there), for the tile routines below, which take the
// pointer type as a template parameter.  The LDS of gfx950 does accept ds_read_b32 / b96 at any byte address
// (tools/ubench/lds_unaligned.hip checks it on the box), but slowly: with the rows of me_team_kernel's windows read that
// way SQ_LDS_UNALIGNED_STALL was 90 % of the LDS's active cycles and the kernel LDS-bound (profiles/r03_team_pmc.txt).
// So a read at byte address a fetches the aligned dwords around it and shifts (v_alignbyte_b32).
struct ldsa_t
{
    uint32_t a; // byte address in LDS
    __device__ __forceinline__ ldsa_t() : a(0) {}
    __device__ __forceinline__ explicit ldsa_t(const uint8_t *generic)
        : a((uint32_t)(size_t)(const __attribute__((address_space(3))) uint8_t *)generic)
    {
    }
    __device__ __forceinline__ ldsa_t operator+(int o) const
    {
        ldsa_t r;
        r.a = a + (uint32_t)o;
        return r;
    }
    __device__ __forceinline__ ldsa_t operator-(int o) const
    {
        ldsa_t r;
        r.a = a - (uint32_t)o;
        return r;
    }
    __device__ __forceinline__ ldsa_t &operator+=(int o)
    {
        a += (uint32_t)o;
        return *this;
    }
};
__device__ __forceinline__ uint32_t lds_dword(uint32_t aligned_addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(3))) uint32_t *)aligned_addr;
#else
    return aligned_addr; // (the host pass only parses this)
#endif
}
__device__ __forceinline__ uint32_t ld4g(ldsa_t p)
{
    const uint32_t al = p.a & ~3u;
    return __builtin_amdgcn_alignbyte(lds_dword(al + 4), lds_dword(al), p.a & 3u);
}
__device__ __forceinline__ void ld12g(ldsa_t p, uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
    const uint32_t al = p.a & ~3u, sh = p.a & 3u;
    const uint32_t w0 = lds_dword(al), w1 = lds_dword(al + 4), w2 = lds_dword(al + 8), w3 = lds_dword(al + 12);
    d0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
    d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
    d2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
}

__device__ __forceinline__ int byte_of(uint32_t v, int i)
{
    return (int)((v >> (8 * i)) & 0xffu);
}

__device__ __forceinline__ int uniform(int v)
{
    return __builtin_amdgcn_readfirstlane(v);
}

// The reference narrows to int16 before the clamp (ipfilter.cpp:105-109).  For 8-bit input the value is
// always inside int16 (|sum| <= 255 * 88 + offset before the shift of the pp filters, at most
// (14248 * 88 + 526336) >> 12 for vsp), so the narrowing never changes it and a plain clamp is exact.
__device__ __forceinline__ int clip_pixel(int v)
{
    return min(max(v, 0), 255); // v_med3_i32
}

// Blocks are dealt round-robin to the eight XCDs, each with an L2 of its own, while neighbours in a work list
// (the PUs of one CTU, the same PU against another candidate) mostly read the same pixels.  This maps block b of
// n to a list position so that runs of `chunk` consecutive positions land on one XCD; the mapping is a
// permutation of [0, n) (identity on the ragged end).
__device__ __forceinline__ int xcd_chunked(int b, int n, int chunk)
{
    const int span = 8 * chunk;
    if (b >= n - n % span)
        return b;
    const int xcd = b & 7, local = b >> 3;
    return ((local / chunk) * 8 + xcd) * chunk + local % chunk;
}

// make LDS traffic of this wave visible to its other lanes (single-wave scope)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- one 4x4 tile ---------------------------------------------------------------------------
//
// Pixels travel as packed pairs: P[row][0] = (x0, x1), P[row][1] = (x2, x3), 16 bits each, values 0..255.
// The filters run on v_dot4_i32_i8: pixels are biased to signed bytes (p ^ 0x80 = p - 128), so a dot
// product with the signed taps gives sum - 128 * 64 = sum - 8192 (every tap row sums to 64), which is
// exactly luma_hps's intermediate; the pp filters fold + 8192 and the rounding into the accumulator seed.

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }

#define X265HIP_PK8(a, b, c, d) \
    ((uint32_t)((a) & 0xff) | ((uint32_t)((b) & 0xff) << 8) | ((uint32_t)((c) & 0xff) << 16) | ((uint32_t)((d) & 0xff) << 24))
__device__ __constant__ const uint32_t k_taps_s8[4][2] = {
    { X265HIP_PK8(0, 0, 0, 64), X265HIP_PK8(0, 0, 0, 0) },
    { X265HIP_PK8(-1, 4, -10, 58), X265HIP_PK8(17, -5, 1, 0) },
    { X265HIP_PK8(-1, 4, -11, 40), X265HIP_PK8(40, -11, 4, -1) },
    { X265HIP_PK8(0, 1, -5, 17), X265HIP_PK8(58, -10, 4, -1) },
};

// the same vertical taps as 16-bit pairs for v_dot2_i32_i16 over row pairs (2j, 2j + 1): entries 0..3 serve
// output rows that start on an even row, (c0,c1) (c2,c3) (c4,c5) (c6,c7); entries 4..8 serve rows that start
// on an odd row, (0,c0) (c1,c2) (c3,c4) (c5,c6) (c7,0)
#define X265HIP_PK16(a, b) ((uint32_t)((a) & 0xffff) | ((uint32_t)((b) & 0xffff) << 16))
#define X265HIP_VTAPS(c0, c1, c2, c3, c4, c5, c6, c7)                                                        \
    { X265HIP_PK16(c0, c1), X265HIP_PK16(c2, c3), X265HIP_PK16(c4, c5), X265HIP_PK16(c6, c7), X265HIP_PK16(0, c0), \
      X265HIP_PK16(c1, c2), X265HIP_PK16(c3, c4), X265HIP_PK16(c5, c6), X265HIP_PK16(c7, 0), 0, 0, 0 }
__device__ __constant__ __attribute__((aligned(16))) const uint32_t k_vtaps_s16[4][12] = {
    X265HIP_VTAPS(0, 0, 0, 64, 0, 0, 0, 0),
    X265HIP_VTAPS(-1, 4, -10, 58, 17, -5, 1, 0),
    X265HIP_VTAPS(-1, 4, -11, 40, 40, -11, 4, -1),
    X265HIP_VTAPS(0, 1, -5, 17, 58, -10, 4, -1),
};

// first link of a dot-product chain with the seed as a direct operand (the compiler's accumulate-in-place
// form would spend a v_mov per chain on it)
template <int SEED>
__device__ __forceinline__ int sdot4_first(uint32_t a, uint32_t b)
{
    int r;
    if (SEED == 0)
        asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(SEED));
    return r;
}

template <int SEED>
__device__ __forceinline__ int sdot2_first(uint32_t a, uint32_t b)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(SEED));
    return r;
}

__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(as_s2(a), as_s2(b), c, false);
}

// out[x] = SEED + sum_k (byte[x + k] of the 12 signed bytes d0:d1:d2) * tap[k], x = 0..3
template <int SEED>
__device__ __forceinline__ void tap8x4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t tlo, uint32_t thi, int out[4])
{
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        const uint32_t lo = x ? __builtin_amdgcn_alignbyte(d1, d0, x) : d0;
        const uint32_t hi = x ? __builtin_amdgcn_alignbyte(d2, d1, x) : d1;
        out[x] = __builtin_amdgcn_sdot4((int)lo, (int)tlo, sdot4_first<SEED>(hi, thi), false);
    }
}

__device__ __forceinline__ uint32_t pack2(int lo, int hi) { return (uint32_t)lo | ((uint32_t)hi << 16); } // v_lshl_or_b32

// Fetch (or interpolate) a strip of NT vertically stacked 4x4 reference tiles whose top-left full-pel
// sample is `ref`, for the quarter-pel fraction (xf, yf).  All work stays in registers.  A strip of two tiles
// filters 15 source rows instead of 2 x 11 (vertical taps overlap), which is what NT = 2 is for.
// `diag` sends a half-fractional position through the diagonal path as well: with the tap row {0,0,0,64,...}
// for the integer direction that path gives (64 S + 2048) >> 12 = (S + 32) >> 6, the hpp / vpp result, bit for
// bit.  A wave whose lanes hold positions of different kinds then runs one path instead of three.
//
// SH = true produces the 14-bit intermediates of bi-prediction instead of pixels, as packed int16 pairs
// (TComPrediction::xPredInterLumaBlk to a ShortYuv, TComPrediction.cpp:436-473): luma_p2s = (p << 6) - 8192,
// luma_hps = sum - 8192, luma_vps = sum - 8192, luma_hps + luma_vss = (sum of intermediates * taps) >> 6
// (ipfilter.cpp:38-54, 117-160, 204-239, 285-317); `diag` is not used with it.
__device__ __forceinline__ uint32_t pack_s16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100); }

// AH > 0: the rows of the horizontal and the diagonal path are fetched AH ahead of their use (left alone the compiler,
// which orders for few registers, waits for every row load right behind it: R or R + 7 memory round trips in a row).
// For kernels with registers to spare (me_big_kernel); in me_kernel and cost_kernel it cost more in spills than it gave.
template <int NT, bool SH = false, int AH = 0, typename PT = gptr_t>
__device__ __forceinline__ void ref_strip(const uint8_t *ref, int stride, int xf, int yf, uint32_t P[4 * NT][2],
                                          bool diag = false)
{
    constexpr int R = 4 * NT;
    const uint32_t bias = 0x80808080u;
    const PT r_ = (PT)ref;
    if ((xf | yf) == 0)
    {
#pragma unroll
        for (int y = 0; y < R; y++)
        {
            const uint32_t d = ld4g(r_ + y * stride);
            P[y][0] = __builtin_amdgcn_perm(0, d, 0x0c010c00);
            P[y][1] = __builtin_amdgcn_perm(0, d, 0x0c030c02);
            if (SH)
            {
                const s16x2 offs = { 8192, 8192 };
                P[y][0] = as_u32(as_s2(P[y][0] << 6) - offs);
                P[y][1] = as_u32(as_s2(P[y][1] << 6) - offs);
            }
        }
    }
    else if (yf == 0 && !diag)
    {
        // luma_hpp: dst = clip((sum_k src[x-3+k]*c[k] + 32) >> 6)
        const uint32_t tlo = k_taps_s8[xf][0], thi = k_taps_s8[xf][1];
        PT p = r_ - 3;
        uint32_t ah[AH ? AH : 1][3];
        if (AH)
#pragma unroll
            for (int y = 0; y < AH && y < R; y++)
            {
                ld12g(p, ah[y][0], ah[y][1], ah[y][2]);
                p += stride;
            }
#pragma unroll
        for (int y = 0; y < R; y++)
        {
            int h[4];
            uint32_t d0, d1, d2;
            if (AH)
            {
                d0 = ah[y % (AH ? AH : 1)][0], d1 = ah[y % (AH ? AH : 1)][1], d2 = ah[y % (AH ? AH : 1)][2];
                if (y + AH < R)
                {
                    ld12g(p, ah[y % (AH ? AH : 1)][0], ah[y % (AH ? AH : 1)][1], ah[y % (AH ? AH : 1)][2]);
                    p += stride;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            else
            {
                ld12g(p, d0, d1, d2);
                p += stride;
            }
            if (SH)
            {
                tap8x4<0>(d0 ^ bias, d1 ^ bias, d2 ^ bias, tlo, thi, h);
                P[y][0] = pack_s16(h[0], h[1]);
                P[y][1] = pack_s16(h[2], h[3]);
                continue;
            }
            tap8x4<8192 + 32>(d0 ^ bias, d1 ^ bias, d2 ^ bias, tlo, thi, h);
#pragma unroll
            for (int x = 0; x < 4; x++)
                h[x] = clip_pixel(h[x] >> 6);
            P[y][0] = pack2(h[0], h[1]);
            P[y][1] = pack2(h[2], h[3]);
        }
    }
    else if (xf == 0 && !diag)
    {
        // luma_vpp: same rounding, taps run down the column.  The R + 7 source rows are transposed
        // (v_perm_b32) into four column vectors of 4 (NT + 2) bytes; output row y takes bytes y .. y + 7.
        constexpr int NB = NT + 2;
        const uint32_t tlo = k_taps_s8[yf][0], thi = k_taps_s8[yf][1];
        uint32_t c[4][NB];
        PT p = r_ - 3 * stride;
#pragma unroll
        for (int b = 0; b < NB; b++)
        {
            uint32_t d[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                d[i] = 0u;
                if (4 * b + i < R + 7)
                {
                    d[i] = ld4g(p) ^ bias;
                    p += stride;
                }
            }
            const uint32_t t0 = __builtin_amdgcn_perm(d[1], d[0], 0x05010400), t1 = __builtin_amdgcn_perm(d[1], d[0], 0x07030602);
            const uint32_t t2 = __builtin_amdgcn_perm(d[3], d[2], 0x05010400), t3 = __builtin_amdgcn_perm(d[3], d[2], 0x07030602);
            c[0][b] = __builtin_amdgcn_perm(t2, t0, 0x05040100);
            c[1][b] = __builtin_amdgcn_perm(t2, t0, 0x07060302);
            c[2][b] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
            c[3][b] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
        {
            int v[4][4]; // [x][y]
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                if (SH)
                    tap8x4<0>(c[x][t], c[x][t + 1], c[x][t + 2], tlo, thi, v[x]);
                else
                {
                    tap8x4<8192 + 32>(c[x][t], c[x][t + 1], c[x][t + 2], tlo, thi, v[x]);
#pragma unroll
                    for (int y = 0; y < 4; y++)
                        v[x][y] = clip_pixel(v[x][y] >> 6);
                }
            }
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                P[4 * t + y][0] = SH ? pack_s16(v[0][y], v[1][y]) : pack2(v[0][y], v[1][y]);
                P[4 * t + y][1] = SH ? pack_s16(v[2][y], v[3][y]) : pack2(v[2][y], v[3][y]);
            }
        }
    }
    else
    {
        // luma_hps (isRowExt = 1: rows -3 .. R + 3, value = sum - 8192, always inside int16) followed by
        // luma_vsp: clip((sum_k im[y-3+k]*c[k] + 2048 + (8192 << 6)) >> 12).  The R + 7 intermediate rows are
        // packed in pairs (2j, 2j + 1) and the vertical taps run on v_dot2_i32_i16.
        constexpr int NP = R / 2 + 4; // row pairs; the last one holds a single row
        const uint32_t tlo = k_taps_s8[xf][0], thi = k_taps_s8[xf][1];
        const uint4 ve = *(const uint4 *)&k_vtaps_s16[yf][0], vo = *(const uint4 *)&k_vtaps_s16[yf][4];
        const uint32_t vo4 = k_vtaps_s16[yf][8];
        PT p = r_ - 3 * stride - 3;
        constexpr int seed = SH ? 0 : 2048 + (8192 << 6);
        uint32_t q[NP][4];
        constexpr int ROWS = R + 7, A = AH ? AH : 1;
        uint32_t ah[A][3];
        if (AH)
#pragma unroll
            for (int y = 0; y < AH && y < ROWS; y++)
            {
                ld12g(p, ah[y][0], ah[y][1], ah[y][2]);
                p += stride;
            }
        // pairs are produced in order and a pair of output rows is emitted as soon as its five pairs exist,
        // so only a window of five pairs is live at a time
#pragma unroll
        for (int j = 0; j < NP; j++)
        {
            int a[4], b[4] = { 0, 0, 0, 0 };
            uint32_t d0, d1, d2, e0 = 0, e1 = 0, e2 = 0;
            if (AH)
            {
                d0 = ah[(2 * j) % A][0], d1 = ah[(2 * j) % A][1], d2 = ah[(2 * j) % A][2];
                e0 = ah[(2 * j + 1) % A][0], e1 = ah[(2 * j + 1) % A][1], e2 = ah[(2 * j + 1) % A][2];
                if (2 * j + AH < ROWS)
                {
                    ld12g(p, ah[(2 * j) % A][0], ah[(2 * j) % A][1], ah[(2 * j) % A][2]);
                    p += stride;
                }
                if (2 * j + 1 + AH < ROWS)
                {
                    ld12g(p, ah[(2 * j + 1) % A][0], ah[(2 * j + 1) % A][1], ah[(2 * j + 1) % A][2]);
                    p += stride;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            else
            {
                ld12g(p, d0, d1, d2);
                p += stride;
            }
            tap8x4<0>(d0 ^ bias, d1 ^ bias, d2 ^ bias, tlo, thi, a);
            if (j < NP - 1)
            {
                if (!AH)
                {
                    ld12g(p, e0, e1, e2);
                    p += stride;
                }
                tap8x4<0>(e0 ^ bias, e1 ^ bias, e2 ^ bias, tlo, thi, b);
            }
#pragma unroll
            for (int x = 0; x < 4; x++)
                q[j][x] = j < NP - 1 ? __builtin_amdgcn_perm((uint32_t)b[x], (uint32_t)a[x], 0x05040100) : ((uint32_t)a[x] & 0xffffu);
            if (j >= 4)
            {
                const int k = j - 4, y = 2 * k; // output row y starts on pair k, row y + 1 in the middle of it
                int e[4], o[4];
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    e[x] = sdot2(q[k + 3][x], ve.w, sdot2(q[k + 2][x], ve.z, sdot2(q[k + 1][x], ve.y, sdot2_first<seed>(q[k][x], ve.x))));
                    o[x] = sdot2(q[k + 4][x], vo4, sdot2(q[k + 3][x], vo.w, sdot2(q[k + 2][x], vo.z, sdot2(q[k + 1][x], vo.y, sdot2_first<seed>(q[k][x], vo.x)))));
                }
                if (SH)
                {
                    P[y][0] = pack_s16(e[0] >> 6, e[1] >> 6);
                    P[y][1] = pack_s16(e[2] >> 6, e[3] >> 6);
                    P[y + 1][0] = pack_s16(o[0] >> 6, o[1] >> 6);
                    P[y + 1][1] = pack_s16(o[2] >> 6, o[3] >> 6);
                    continue;
                }
                P[y][0] = pack2(clip_pixel(e[0] >> 12), clip_pixel(e[1] >> 12));
                P[y][1] = pack2(clip_pixel(e[2] >> 12), clip_pixel(e[3] >> 12));
                P[y + 1][0] = pack2(clip_pixel(o[0] >> 12), clip_pixel(o[1] >> 12));
                P[y + 1][1] = pack2(clip_pixel(o[2] >> 12), clip_pixel(o[3] >> 12));
            }
        }
    }
}

// luma_addAvg (reference common/pixel.cpp:819-839) of two packed pairs of 14-bit intermediates: two pixels,
// clip((a + b + 64 + 2 * 8192) >> 7), as a packed pair of pixels
__device__ __forceinline__ uint32_t add_avg_pair(uint32_t a, uint32_t b)
{
    const uint32_t sel_lo = 0x00000001u, sel_hi = 0x00010000u; // (1, 0) and (0, 1) as int16 pairs
    const int lo = sdot2(b, sel_lo, sdot2_first<64 + 2 * 8192>(a, sel_lo));
    const int hi = sdot2(b, sel_hi, sdot2_first<64 + 2 * 8192>(a, sel_hi));
    return pack2(clip_pixel(lo >> 7), clip_pixel(hi >> 7));
}

// one 4x4 tile
__device__ __forceinline__ void ref_tile(const uint8_t *ref, int stride, int xf, int yf, uint32_t P[4][2], bool diag = false)
{
    ref_strip<1>(ref, stride, xf, yf, P, diag);
}

// 4x4 bytes held as one dword per row -> one dword per column
__device__ __forceinline__ void transpose4x4_u8(uint32_t d[4])
{
    const uint32_t t0 = __builtin_amdgcn_perm(d[1], d[0], 0x05010400), t1 = __builtin_amdgcn_perm(d[1], d[0], 0x07030602);
    const uint32_t t2 = __builtin_amdgcn_perm(d[3], d[2], 0x05010400), t3 = __builtin_amdgcn_perm(d[3], d[2], 0x07030602);
    d[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100);
    d[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302);
    d[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
    d[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
}

// One 4x4 tile at a position with exactly one fractional component, luma_hpp and luma_vpp through ONE filter
// routine: lanes of a wave that hold horizontal and vertical half-/quarter-pel candidates (every sub-pel diamond has
// both) then run it once instead of one path after the other.  A vertical position is filtered as a horizontal one
// on the transposed rows and comes back transposed (P[x] = column x of the prediction); SAD and the 4x4 Hadamard
// SATD are invariant under transposing both operands, so the caller transposes its source tile for those lanes
// (xf == 0) and compares as usual.
template <typename PT = gptr_t>
__device__ __forceinline__ void ref_tile_1d(const uint8_t *ref, int stride, int xf, int yf, uint32_t P[4][2])
{
    const uint32_t bias = 0x80808080u;
    const bool vert = xf == 0;
    const int fr = vert ? yf : xf;
    const uint32_t tlo = k_taps_s8[fr][0], thi = k_taps_s8[fr][1];
    const PT r_ = (PT)ref;
    uint32_t c[4][3];
    if (vert)
    {
        PT p = r_ - 3 * stride;
#pragma unroll
        for (int b = 0; b < 3; b++)
        {
            uint32_t d[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                d[i] = 0u;
                if (4 * b + i < 11)
                {
                    d[i] = ld4g(p) ^ bias;
                    p += stride;
                }
            }
            transpose4x4_u8(d);
#pragma unroll
            for (int x = 0; x < 4; x++)
                c[x][b] = d[x];
        }
    }
    else
    {
        PT p = r_ - 3;
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            ld12g(p, c[y][0], c[y][1], c[y][2]);
            p += stride;
#pragma unroll
            for (int k = 0; k < 3; k++)
                c[y][k] ^= bias;
        }
    }
#pragma unroll
    for (int v = 0; v < 4; v++)
    {
        int o[4];
        tap8x4<8192 + 32>(c[v][0], c[v][1], c[v][2], tlo, thi, o);
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = clip_pixel(o[k] >> 6);
        P[v][0] = pack2(o[0], o[1]);
        P[v][1] = pack2(o[2], o[3]);
    }
}

// SAD or SATD of one 4x4 tile.  `f` walks the source block with `fstride`.
// SATD: sum |H4 * D * H4| / 2 per tile; the per-tile sum is always even, so halving per tile is
// exactly what the reference's 4x4 / 8x4 grouping yields (pixel.cpp:201-252).  All of it runs on packed
// 16-bit pairs (|coefficient| <= 16 * 255); the last butterfly stage is folded into
// |a + b| + |a - b| = 2 * max(|a|, |b|), which also supplies the division by two.
// 4x4 Hadamard SATD of differences held as 16-bit pairs: a[y] = columns (0, 1), b[y] = columns (2, 3) of row y
__device__ __forceinline__ int satd4x4_pairs(const s16x2 a[4], const s16x2 b[4])
{
    // columns: 4-point Hadamard down the rows, two columns per register
    s16x2 ra[4], rb[4];
    {
        const s16x2 s01 = a[0] + a[1], t01 = a[0] - a[1], s23 = a[2] + a[3], t23 = a[2] - a[3];
        ra[0] = s01 + s23;
        ra[1] = t01 + t23;
        ra[2] = s01 - s23;
        ra[3] = t01 - t23;
    }
    {
        const s16x2 s01 = b[0] + b[1], t01 = b[0] - b[1], s23 = b[2] + b[3], t23 = b[2] - b[3];
        rb[0] = s01 + s23;
        rb[1] = t01 + t23;
        rb[2] = s01 - s23;
        rb[3] = t01 - t23;
    }
    // rows: (v0, v1), (v2, v3) -> s = (v0 + v2, v1 + v3), t = (v0 - v2, v1 - v3); the outputs are
    // s.lo +- s.hi and t.lo +- t.hi, whose absolute sums are 2 max(|s.lo|, |s.hi|) + 2 max(|t.lo|, |t.hi|)
    s16x2 sum = { 0, 0 };
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        const uint32_t s = as_u32(ra[y] + rb[y]), t = as_u32(ra[y] - rb[y]);
        const s16x2 u = as_s2(__builtin_amdgcn_perm(t, s, 0x05040100)); // (s.lo, t.lo)
        const s16x2 v = as_s2(__builtin_amdgcn_perm(t, s, 0x07060302)); // (s.hi, t.hi)
        const s16x2 mx = __builtin_elementwise_max(u, v), mn = __builtin_elementwise_min(u, v);
        const s16x2 zero = { 0, 0 };
        sum += __builtin_elementwise_max(mx, zero - mn);
    }
    const uint32_t su = as_u32(sum);
    return (int)((su & 0xffffu) + (su >> 16));
}

// the source tile already in registers (one dword per row)
__device__ __forceinline__ int tile_metric_rows(const uint32_t frow[4], const uint32_t P[4][2], int metric);

template <bool F_GLOBAL = false> // source block in plane memory (true) or wherever `f` points, e.g. LDS (false)
__device__ __forceinline__ int tile_metric(const uint8_t *f, int fstride, const uint32_t P[4][2], int metric)
{
    uint32_t frow[4];
#pragma unroll
    for (int y = 0; y < 4; y++)
        frow[y] = F_GLOBAL ? ld4g(as_global(f) + y * fstride) : ld4u(f + y * fstride);
    return tile_metric_rows(frow, P, metric);
}

__device__ __forceinline__ int tile_metric_rows(const uint32_t frow[4], const uint32_t P[4][2], int metric)
{
    auto ldf = [&](int y) { return frow[y]; };
    if (metric == X265_HIP_SAD)
    {
        uint32_t acc = 0;
#pragma unroll
        for (int y = 0; y < 4; y++)
            acc = __builtin_amdgcn_sad_u8(ldf(y), __builtin_amdgcn_perm(P[y][1], P[y][0], 0x06040200), acc);
        return (int)acc;
    }
    s16x2 a[4], b[4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        const uint32_t fv = ldf(y);
        a[y] = as_s2(__builtin_amdgcn_perm(0, fv, 0x0c010c00)) - as_s2(P[y][0]);
        b[y] = as_s2(__builtin_amdgcn_perm(0, fv, 0x0c030c02)) - as_s2(P[y][1]);
    }
    return satd4x4_pairs(a, b);
}

// the same with the source tile already unpacked to 16-bit pairs (fa[y] = (x0, x1), fb[y] = (x2, x3)): for callers that
// compare one source tile with several predictions
__device__ __forceinline__ int tile_metric_pairs(const uint32_t fa[4], const uint32_t fb[4], const uint32_t P[4][2], int metric)
{
    if (metric == X265_HIP_SAD)
    {
        uint32_t acc = 0;
#pragma unroll
        for (int y = 0; y < 4; y++)
            acc = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(fb[y], fa[y], 0x06040200), __builtin_amdgcn_perm(P[y][1], P[y][0], 0x06040200), acc);
        return (int)acc;
    }
    s16x2 a[4], b[4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        a[y] = as_s2(fa[y]) - as_s2(P[y][0]);
        b[y] = as_s2(fb[y]) - as_s2(P[y][1]);
    }
    return satd4x4_pairs(a, b);
}

// ---- mode-decision distortions of an 8-row strip ---------------------------------------------------
//
// sa8d (reference common/pixel.cpp:290-325): sum of the absolute coefficients of the 8x8 Hadamard transform of
// the differences.  Lanes 2j and 2j + 1 hold the left and right 4-column halves of one 8x8 block; each runs the
// 8-point transform down its columns and the 4-point transform along its rows, the last butterfly across the
// halves is folded into |a + b| + |a - b| = 2 max(|a|, |b|), the even lane taking vertical frequencies 0..3 and
// the odd lane 4..7.  The return value is this lane's share of _sa8d_8x8's sum; the caller adds the shares of a
// block (or of the four blocks of a 16x16) before the reference's rounding.  Packed 16-bit throughout:
// coefficients before the last stage are at most 32 * 255, and a lane's sixteen maxima sum to half of 32
// coefficients' absolute sum, at most sqrt(32) * 64 * 255 / 2 < 65536.
// Both lanes of a pair must be active.
template <bool F_GLOBAL>
__device__ __forceinline__ int sa8d_strip(const uint8_t *f, int fstride, const uint32_t P[8][2], bool odd)
{
    auto ldf = [&](int y) { return F_GLOBAL ? ld4g(as_global(f) + y * fstride) : ld4u(f + y * fstride); };
    s16x2 a[8], b[8];
#pragma unroll
    for (int y = 0; y < 8; y++)
    {
        const uint32_t fv = ldf(y);
        a[y] = as_s2(__builtin_amdgcn_perm(0, fv, 0x0c010c00)) - as_s2(P[y][0]);
        b[y] = as_s2(__builtin_amdgcn_perm(0, fv, 0x0c030c02)) - as_s2(P[y][1]);
    }
#pragma unroll
    for (int s = 1; s < 8; s <<= 1)
#pragma unroll
        for (int y = 0; y < 8; y++)
            if (!(y & s))
            {
                const s16x2 u = a[y], v = a[y | s], p = b[y], q = b[y | s];
                a[y] = u + v;
                a[y | s] = u - v;
                b[y] = p + q;
                b[y | s] = p - q;
            }
    // rows: (c0, c1), (c2, c3) -> the four 4-point outputs as (s.lo + s.hi, t.lo + t.hi), (s.lo - s.hi, t.lo - t.hi)
#pragma unroll
    for (int y = 0; y < 8; y++)
    {
        const uint32_t sm = as_u32(a[y] + b[y]), df = as_u32(a[y] - b[y]);
        const s16x2 u = as_s2(__builtin_amdgcn_perm(df, sm, 0x05040100)), v = as_s2(__builtin_amdgcn_perm(df, sm, 0x07060302));
        a[y] = u + v;
        b[y] = u - v;
    }
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 acc = { 0, 0 };
    const s16x2 zero = { 0, 0 };
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        const s16x2 keep_a = odd ? a[y + 4] : a[y], send_a = odd ? a[y] : a[y + 4];
        const s16x2 keep_b = odd ? b[y + 4] : b[y], send_b = odd ? b[y] : b[y + 4];
        const s16x2 oth_a = as_s2((uint32_t)__builtin_amdgcn_update_dpp(0, (int)as_u32(send_a), 0xB1, 0xf, 0xf, true)); // quad_perm 1,0,3,2
        const s16x2 oth_b = as_s2((uint32_t)__builtin_amdgcn_update_dpp(0, (int)as_u32(send_b), 0xB1, 0xf, 0xf, true));
        const s16x2 ma = __builtin_elementwise_max(__builtin_elementwise_max(keep_a, zero - keep_a), __builtin_elementwise_max(oth_a, zero - oth_a));
        const s16x2 mb = __builtin_elementwise_max(__builtin_elementwise_max(keep_b, zero - keep_b), __builtin_elementwise_max(oth_b, zero - oth_b));
        acc += __builtin_bit_cast(u16x2, ma) + __builtin_bit_cast(u16x2, mb);
    }
    const uint32_t su = __builtin_bit_cast(uint32_t, acc);
    return (int)(2 * ((su & 0xffffu) + (su >> 16)));
}

// sse_pp (reference common/pixel.cpp:158-177) of NR rows x 4 columns
template <int NR, bool F_GLOBAL>
__device__ __forceinline__ int sse_strip(const uint8_t *f, int fstride, const uint32_t P[NR][2])
{
    int acc = 0;
#pragma unroll
    for (int y = 0; y < NR; y++)
    {
        const uint32_t fv = F_GLOBAL ? ld4g(as_global(f) + y * fstride) : ld4u(f + y * fstride);
        const s16x2 d0 = as_s2(__builtin_amdgcn_perm(0, fv, 0x0c010c00)) - as_s2(P[y][0]);
        const s16x2 d1 = as_s2(__builtin_amdgcn_perm(0, fv, 0x0c030c02)) - as_s2(P[y][1]);
        acc = __builtin_amdgcn_sdot2(d0, d0, acc, false);
        acc = __builtin_amdgcn_sdot2(d1, d1, acc, false);
    }
    return acc;
}

// ---- the caller's window arithmetic (host and device) ------------------------------------------------------
// TComDataCU::clipMv (reference Lib/TLibCommon/TComDataCU.cpp:2502-2514): quarter-pel in and out, relative to the CU
// at (cu_x, cu_y) of a pic_w x pic_h picture with 64x64 CTUs
__host__ __device__ inline void clip_mv(int &qx, int &qy, int cu_x, int cu_y, int pic_w, int pic_h)
{
    const int xmax = (pic_w + 8 - cu_x - 1) << 2, xmin = (-64 - 8 - cu_x + 1) << 2;
    const int ymax = (pic_h + 8 - cu_y - 1) << 2, ymin = (-64 - 8 - cu_y + 1) << 2;
    qx = qx > xmax ? xmax : qx < xmin ? xmin : qx;
    qy = qy > ymax ? ymax : qy < ymin ? ymin : qy;
}

// TEncSearch::xMergeEstimation's candidates (reference Lib/TLibEncoder/TEncSearch.cpp:2194-2210).  Whether candidate c
// of a unit is measured: present, with a reference, and neither list's vector at the frame threads' limit.
__host__ __device__ inline bool merge_cand_used(const x265_hip_merge_pu &p, int c)
{
    if (c >= p.num_cand)
        return false;
    const x265_hip_merge_cand &m = p.cand[c];
    const bool v0 = m.ref_plane[0] >= 0, v1 = m.ref_plane[1] >= 0 && p.weighted != 1;
    return (v0 || v1) && m.mv[1] < p.mvy_limit && m.mv[3] < p.mvy_limit;
}

// The prediction motionCompensation(cu, .., REF_PIC_LIST_X, ..) makes for the candidate (TComPrediction.cpp:252-297), as
// a prediction request against the source picture (x265_hip_mc_satd_batch).  A candidate that is not measured gives a
// request any list may hold (the unit's 4x4 corner predicted from the source picture itself) and false.
__host__ __device__ inline bool merge_make_req(const x265_hip_merge_pu &p, int c, int pic_w, int pic_h, x265_hip_pred_req &r)
{
    r = x265_hip_pred_req{};
    r.dst_plane = r.ref0_plane = p.fenc_plane;
    r.ref1_plane = -1;
    r.pu_x = p.pu_x, r.pu_y = p.pu_y;
    r.pu_w = r.pu_h = 4;
    if (!merge_cand_used(p, c))
        return false;
    r.pu_w = p.pu_w, r.pu_h = p.pu_h;
    const x265_hip_merge_cand &m = p.cand[c];
    bool v0 = m.ref_plane[0] >= 0, v1 = m.ref_plane[1] >= 0 && p.weighted != 1; // a P slice predicts from list 0 (:261-262)
    // xCheckIdenticalMotion (:232-250; never in a slice with weighted bi-prediction): list 0 alone, unweighted
    const bool same = v0 && v1 && p.weighted != 2 && m.ref_plane[0] == m.ref_plane[1] && m.mv[0] == m.mv[2] && m.mv[1] == m.mv[3];
    if (same)
        v1 = false;
    int x0 = m.mv[0], y0 = m.mv[1], x1 = m.mv[2], y1 = m.mv[3];
    clip_mv(x0, y0, p.cu_x, p.cu_y, pic_w, pic_h);
    clip_mv(x1, y1, p.cu_x, p.cu_y, pic_w, pic_h);
    if (v0 && v1)
    {
        r.ref0_plane = m.ref_plane[0], r.ref1_plane = m.ref_plane[1];
        r.qmv0_x = (int16_t)x0, r.qmv0_y = (int16_t)y0, r.qmv1_x = (int16_t)x1, r.qmv1_y = (int16_t)y1;
        if (p.weighted == 2) // xPredInterBi :347-350
        {
            r.weighted_bi = 1;
            r.wp_w0 = m.wp_w[0], r.wp_o0 = m.wp_o[0], r.wp_w1 = m.wp_w[1], r.wp_o1 = m.wp_o[1];
            r.log2_denom = m.log2_denom[0];
        }
    }
    else
    {
        const int l = v0 ? 0 : 1;
        r.ref0_plane = m.ref_plane[l];
        r.qmv0_x = (int16_t)(l ? x1 : x0), r.qmv0_y = (int16_t)(l ? y1 : y0);
        if (p.weighted) // :265-279 (P) / xPredInterBi :356-368 (B): the intermediate and addWeightUni
        {
            r.ref1_plane = X265_HIP_PRED_WEIGHTED;
            r.qmv1_x = m.wp_w[l], r.qmv1_y = m.wp_o[l];
            r.log2_denom = m.log2_denom[l];
        }
    }
    return true;
}

// TEncSearch::xSetSearchRange (reference Lib/TLibEncoder/TEncSearch.cpp:2600-2617): full-pel window around the
// clipped predictor; MV arithmetic is int16 in the reference (mv.h)
__host__ __device__ inline void search_window(int mvpx, int mvpy, int merange, int ref_lag, int cu_x, int cu_y, int pic_w, int pic_h,
                                              int &lox, int &loy, int &hix, int &hiy)
{
    clip_mv(mvpx, mvpy, cu_x, cu_y, pic_w, pic_h);
    const int d = (int16_t)(merange << 2);
    lox = (int16_t)(mvpx - d);
    loy = (int16_t)(mvpy - d);
    hix = (int16_t)(mvpx + d);
    hiy = (int16_t)(mvpy + d);
    clip_mv(lox, loy, cu_x, cu_y, pic_w, pic_h);
    clip_mv(hix, hiy, cu_x, cu_y, pic_w, pic_h);
    lox >>= 2;
    loy >>= 2;
    hix >>= 2;
    hiy >>= 2;
    loy = loy < ref_lag ? loy : ref_lag;
    hiy = hiy < ref_lag ? hiy : ref_lag;
}

// ---- mvcost -----------------------------------------------------------------------------------

// m_cost[v - mvp]; indices wrap inside the 65536-entry row (the reference reads outside its row
// there, which is undefined; in-range motion vectors never wrap)
__device__ __forceinline__ int tab_at(const uint16_t *tab, int mvp, int v)
{
    // the table is device memory: a global load (a generic pointer makes it a flat load, which also queues on the LDS path)
    return ((const __attribute__((address_space(1))) uint16_t *)tab)[(uint16_t)(v - mvp + 32768)];
}

// BitCost::mvcost: the sum of the two entries truncated to uint16 (bitcost.h:45)
__device__ __forceinline__ int mvcost_q(const uint16_t *tab, int mvpx, int mvpy, int qx, int qy)
{
    return (uint16_t)(tab_at(tab, mvpx, qx) + tab_at(tab, mvpy, qy));
}

// ---- record guards (x265_hip_ctx_check_records mode 2): api.hip's checks against the plane directory in device memory
__device__ __forceinline__ bool guard_partition(int w, int h) // x265_hip_partition_from_sizes >= 0 (primitives.h:38-46)
{
    const int a = w > h ? w : h, b = w > h ? h : w; // the table is symmetric in (w, h)
    switch (a)
    {
    case 4: return b == 4;
    case 8: return b == 8 || b == 4;
    case 16: return b == 16 || b == 8 || b == 12 || b == 4;
    case 32: return b == 32 || b == 16 || b == 24 || b == 8;
    case 64: return b == 64 || b == 32 || b == 48 || b == 16;
    default: return false;
    }
}


// check_block: both planes uploaded, the size a luma partition
__device__ __forceinline__ bool guard_block(const DeviceTables *tabs, int fp, int rp, int w, int h)
{
    return (unsigned)fp < X265_HIP_MAX_PLANES && (unsigned)rp < X265_HIP_MAX_PLANES && tabs->planes[fp].origin && tabs->planes[rp].origin &&
           guard_partition(w, h);
}

// check_inside: the block itself, exactly
__device__ __forceinline__ bool guard_inside(const DeviceTables *tabs, int id, int x, int y, int w, int h)
{
    const long stride = tabs->planes[id].stride;
    const long first = (long)y * stride + x, last = (long)(y + h - 1) * stride + x + w;
    return first >= tabs->span_first[id] && last <= tabs->span_end[id] && x + w <= stride;
}

// check_reach: picture samples [x0, x1) x [y0, y1) with the filters' 3 before / 4 behind and the loads' overhang
__device__ __forceinline__ bool guard_reach(const DeviceTables *tabs, int id, long x0, long y0, long x1, long y1)
{
    const long stride = tabs->planes[id].stride;
    const long first = (y0 - 3) * stride + (x0 - 3 - 3), last = (y1 + 4 - 1) * stride + (x1 + 4 + 12);
    return first >= tabs->span_first[id] && last <= tabs->span_end[id] + 16;
}

// check_displaced: the block displaced by a quarter-pel vector
__device__ __forceinline__ bool guard_displaced(const DeviceTables *tabs, int id, int x, int y, int w, int h, int qx, int qy)
{
    const long bx = x + (qx >> 2), by = y + (qy >> 2);
    return guard_reach(tabs, id, bx, by, bx + w, by + h);
}

} // namespace x265hip
