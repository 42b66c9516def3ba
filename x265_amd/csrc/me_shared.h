// me_shared.h -- what the motionEstimate kernels (me_kernel.hip, me_team.hip) share: the search pattern table, the
// workload[] table, and the layout of the routing lists that split a job list between the kernels.
#pragma once

#include "kernels.h"
#include "me_device.h"

namespace x265hip {

__device__ __forceinline__ int s16(int v) { return (int)(int16_t)v; }

#define PAT_HEX 0
#define PAT_SQ 8
#define PAT_STAR8 17  // small star ring in visiting order, units of dist / 2
#define PAT_STAR16 25 // large star ring (dist > 8) in visiting order, units of dist / 4
#define PAT_HEX4 41   // UMH hexagon grid, units of the ring index
#define PAT_N 57

#define PAT(dx, dy) (int)(((uint32_t)(uint16_t)(dy) << 16) | (uint16_t)(dx))
// hex2 (motion.cpp:68-70) and square1 (motion.cpp:71-73)
__device__ __constant__ const int k_pat[PAT_N] = {
    PAT(-1, -2), PAT(-2, 0), PAT(-1, 2), PAT(1, 2), PAT(2, 0), PAT(1, -2), PAT(-1, -2), PAT(-2, 0),
    PAT(0, 0), PAT(0, -1), PAT(0, 1), PAT(-1, 0), PAT(1, 0), PAT(-1, -1), PAT(-1, 1), PAT(1, -1), PAT(1, 1),
    // StarPatternSearch, dist <= 8 (motion.cpp:300-420): points 2,1,3,4,5,6,8,7; the diagonal ones sit at dist / 2
    PAT(0, -2), PAT(-1, -1), PAT(1, -1), PAT(-2, 0), PAT(2, 0), PAT(-1, 1), PAT(1, 1), PAT(0, 2),
    // dist > 8 (motion.cpp:421-520): top, left, right, bottom, then three steps along each edge of the diamond
    PAT(0, -4), PAT(-4, 0), PAT(4, 0), PAT(0, 4),
    PAT(-1, -3), PAT(1, -3), PAT(-1, 3), PAT(1, 3), PAT(-2, -2), PAT(2, -2), PAT(-2, 2), PAT(2, 2),
    PAT(-3, -1), PAT(3, -1), PAT(-3, 1), PAT(3, 1),
    // hex4 (motion.cpp:132-138)
    PAT(0, -4), PAT(0, 4), PAT(-2, -3), PAT(2, -3), PAT(-4, -2), PAT(4, -2), PAT(-4, -1), PAT(4, -1),
    PAT(-4, 0), PAT(4, 0), PAT(-4, 1), PAT(4, 1), PAT(-4, 2), PAT(4, 2), PAT(-2, 3), PAT(2, 3),
};
__device__ __forceinline__ int pat_dx(int p) { return (int)(int16_t)p; }
__device__ __forceinline__ int pat_dy(int p) { return p >> 16; }

// {hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd} per subme level, motion.cpp:47-57:
// column c of that table, entry i in bits 4 i .. 4 i + 3
__device__ __forceinline__ constexpr uint32_t workload_column(int c)
{
    constexpr uint8_t w[8][5] = { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
                                  { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };
    uint32_t v = 0;
    for (int i = 0; i < 8; i++)
        v |= (uint32_t)w[i][c] << (4 * i);
    return v;
}

// ---- routing --------------------------------------------------------------------------------------------------------
// A motionEstimate launch splits its job list by kind (me_route_kernel), and each kind has a kernel of its own:
//   MC_BIG   blocks of at least ME_BIG_TILES 4x4 tiles whose width is a multiple of 16, full-resolution reference:
//            me_big_kernel (one wave per call, strips of 16 x 4 pixels / stacked tiles, 4 waves per SIMD)
//   MC_T4    diamond / hexagon searches of blocks up to 8x8, full-resolution reference: me_team_kernel, sixteen calls
//            per wave in lockstep (a team of 4 lanes per call, one lane per 4x4 tile)
//   MC_T8    the same for 16x16, 16x8, 8x16, 12x16, 8x8 ...: four calls per wave, a team of 16 lanes, one lane per tile
//   MC_T32   the same for 32x32, 32x16, 32x24, 24x32, 32x8: four calls per wave, a team of 16 lanes, one lane per 8x8 cell
//   MC_REST  everything else (UMH / STAR / FULL, Lowres references, odd sizes): me_rest_kernel, one wave per call
// A list per kind is kept in ME_PARTS parts (one counter for a whole grid would serialise: an atomic on one address
// takes ~12 ns): the jobs of one wave of me_route_kernel (64 consecutive jobs) go to one part, consecutive waves to
// consecutive parts, and part p is served by the waves p, p + ME_PARTS, ... of the kind's kernel, which all run on one
// XCD: neighbouring jobs, which mostly read the same pixels, meet in one L2.  Per (kind, part) one append counter
// (me_route_kernel) and one hand-out counter (the consumer), 64 bytes apart.  Two sets of counters alternate between
// launches (`parity`): a launch's me_route_kernel clears the set of the next launch.
#define ME_PARTS 64
enum { MC_REST = 0, MC_BIG = 1, MC_T4 = 2, MC_T8 = 3, MC_T32 = 4, ME_CLASSES = 5 };
#define ME_ROUTE_SET (2 * ME_CLASSES * ME_PARTS * 16) // words per counter set: append counters, then hand-out counters
#define ME_ROUTE_HEAD (2 * ME_ROUTE_SET)              // the lists start behind the two sets
#ifndef ME_BIG_TILES
#define ME_BIG_TILES 128
#endif

__host__ __device__ inline int me_route_cap(int n) // entries a part of a list can receive
{
    const int waves = (n + 63) / 64;
    return ((waves + ME_PARTS - 1) / ME_PARTS) * 64;
}
__device__ __forceinline__ uint32_t *route_append(uint32_t *route, int parity, int cls, int p)
{
    return route + parity * ME_ROUTE_SET + (cls * ME_PARTS + p) * 16;
}
__device__ __forceinline__ uint32_t *route_handout(uint32_t *route, int parity, int cls, int p)
{
    return route + parity * ME_ROUTE_SET + ((ME_CLASSES + cls) * ME_PARTS + p) * 16;
}
__device__ __forceinline__ const uint32_t *route_list(const uint32_t *route, int cap, int cls, int p)
{
    return route + ME_ROUTE_HEAD + (size_t)(cls * ME_PARTS + p) * cap;
}

__host__ __device__ inline bool big_block(int w, int h) { return (w & 15) == 0 && (w >> 2) * (h >> 2) >= ME_BIG_TILES; }

__host__ __device__ inline int me_job_class(int w, int h, int method)
{
    if (method & X265_HIP_LOWRES_REF)
        return MC_REST;
    if (big_block(w, h))
        return MC_BIG;
    // (the region a job may touch is w + 26 samples wide at least and must hold a row of the class's window, w + 16 for
    // the class's widest block: hence the lower bounds on w)
    if (method <= X265_HIP_HEX)
    {
        if (w <= 8 && h <= 8)
            return MC_T4;
        if (w >= 8 && w <= 16 && h <= 16 && (h & 7) == 0)
            return MC_T8;
        if (w >= 24 && w <= 32 && h <= 32 && (w & 7) == 0 && (h & 7) == 0)
            return MC_T32;
    }
    return MC_REST;
}

// me_team.hip
void launch_me_team(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, uint32_t *route, int parity, int cap, x265_hip_me_result *out,
                    int out_stride, hipStream_t st, bool with_t32);

} // namespace x265hip
