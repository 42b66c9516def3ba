// me_kernel.hip -- MotionEstimate::motionEstimate replayed on gfx950, one wavefront per call.
//
// Reference: encoder/motion.cpp:526-1164 (non-lowres branch): all five search methods, sub-pel
// levels 0..7, with the reference's visiting order, strict-less tie-breaks and quirks.
//
// Shape of the kernel.  A search is a chain of small candidate groups, each depending on the
// winner of the previous one.  The wave runs a state machine:
//
//     loop:  evaluate the queued group (if any)      <- the only place pixels are touched
//            apply the group's results in push order  <- COPYn_IF_LT semantics
//            switch (state): decide, queue the next group, pick the next state
//
// The evaluation is written once and inlined once, so the whole kernel is a single function whose
// register budget the compiler sees (no calls, no stack).  All search state is wave-uniform; the
// queue (<= 16 candidates: position, mv-rate mode, tag) lives in LDS; after an evaluation lane k
// holds candidate k's cost, and the scalar side reads it with v_readlane.
//
// Evaluation: work items are (candidate, 4x4 tile) pairs dealt round-robin to the 64 lanes; a lane
// fetches or interpolates its reference tile in registers (me_device.h: full-pel, hpp, vpp, or
// hps -> vsp over an 11x11 footprint), takes SAD or the 4x4 Hadamard against the PU's source tile
// (LDS) and adds into the candidate's LDS slot.  Lane c meanwhile fetches candidate c's mv rate.
#include <stddef.h>
#include <type_traits>

#include "kernels.h"
#include "me_device.h"
#include "me_shared.h"

namespace x265hip {

#define QMAX 16
#ifndef ME_ONE_D
#define ME_ONE_D 1 // hpp and vpp tiles of q_eval through one routine (ref_tile_1d); 2: only in the diamond / hexagon instantiations
#endif
#define SPEC_SQ_TILES 4    // speculate the square refine for PUs of at most this many 4x4 tiles
#define SPEC_QPEL_TILES 16 // same for the quarter-pel ring
#ifndef ME_DIRECT_STATES
#define ME_DIRECT_STATES 1
#endif
#ifndef ME_SPEC_LOW
#define ME_SPEC_LOW 1 // Lowres references: SATD re-measure + quarter-pel ring together with the half-pel ring
#endif

struct __attribute__((aligned(16))) WaveLds
{
    uint8_t fenc[64 * 64]; // PU source pixels at stride 64 (setSourcePU, motion.cpp:110-126)
    int qx[QMAX];          // candidate position, quarter-pel
    int qy[QMAX];
    int cost[QMAX];
    int aux[QMAX];         // (tag << 2) | mv-rate mode
    int pat[PAT_N];        // search patterns, (dy << 16) | (dx & 0xffff): hex2[0..7], square1[0..8], star rings, hex4
    int sel[2][QMAX];      // q_eval_big: the queue split into full-pel SAD candidates and the others
};

// how the evaluation seeds a candidate's cost with its motion-vector rate
enum { MV_NONE = 0,  // distortion only
       MV_COST = 1,  // BitCost::mvcost(): sum of the two entries truncated to uint16 (bitcost.h:45)
       MV_RAW = 2 }; // plain int sum of the two entries (UMH hexagon grid, motion.cpp:832)

// aux bit 30: this candidate is measured with SATD whatever the group's metric (Lowres references: the SATD re-measure
// and the quarter-pel ring evaluated together with the SAD half-pel ring)
#define AUX_SATD (1 << 30)

// how a group's results are applied before the state code runs
enum { FL_NONE = 0,  // the state reads the costs itself
       FL_BEST = 1,  // strict-less into (bcost, bmv), push order (COST_MV / COST_MV_X4)
       FL_STAR = 2 }; // same, also recording point and distance (COST_MV_PT_DIST*)

struct Wave
{
    WaveLds *s;
    const uint8_t *ref; // reference plane at the PU origin (fref of motion.cpp:537)
    int stride;
    int w, h, ntx, nt;  // PU size, 4x4 tiles per row, tiles in the PU
    const uint16_t *tab;
    int mvpx, mvpy;
    int lane;
    int nq;
    int vc, vx, vy, va; // after an evaluation: lane k holds candidate k's cost / position / aux word
    // reference is a Lowres (ReferencePlanes::isLowres): ref is lowresPlane[0], these are [1..3], all at the
    // PU origin and of one stride
    bool lowres;
    const uint8_t *lref1, *lref2, *lref3;
};

__device__ __forceinline__ void wave_init(Wave &wv, WaveLds *s, int w, int h)
{
    wv.s = s;
    wv.w = w;
    wv.h = h;
    wv.ntx = w >> 2;
    wv.nt = (w >> 2) * (h >> 2);
    wv.lane = threadIdx.x & 63;
    wv.nq = 0;
    wv.tab = nullptr;
    wv.mvpx = wv.mvpy = 0;
    wv.vc = wv.vx = wv.vy = wv.va = 0;
    wv.lowres = false;
    wv.lref1 = wv.lref2 = wv.lref3 = nullptr;
}

// setSourcePU: the block's pixels into LDS at stride 64, and the pattern table beside them.  Four loads per lane are in
// flight at a time (a 64x64 block was sixteen load-then-store round trips in a row), the table's load among them.
__device__ __forceinline__ void load_fenc(Wave &wv, const uint8_t *src, int sstride)
{
    const int ndw = wv.ntx * wv.h;
    const float inv_ntx = __builtin_amdgcn_rcpf((float)wv.ntx); // i / ntx below is exact: i < 1024, ntx <= 16
    const int pat = wv.lane < PAT_N ? k_pat[wv.lane] : 0;
    for (int base = wv.lane; base < ndw; base += 256)
    {
        uint32_t v[4];
        int at[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int i = base + 64 * k;
            const int y = (int)(((float)i + 0.5f) * inv_ntx), x = i - y * wv.ntx;
            at[k] = y * 64 + 4 * x;
            v[k] = i < ndw ? ld4g(as_global(src) + (y * sstride + 4 * x)) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + 64 * k < ndw)
                *(uint32_t *)(wv.s->fenc + at[k]) = v[k];
    }
    if (wv.lane < PAT_N)
        wv.s->pat[wv.lane] = pat;
    wave_sync();
}

// queue values are wave-uniform: every lane stores the same word
__device__ __forceinline__ void q_push(Wave &wv, int qx, int qy, int mvmode, int tag = 0)
{
    wv.s->qx[wv.nq] = qx;
    wv.s->qy[wv.nq] = qy;
    wv.s->aux[wv.nq] = (tag << 2) | mvmode;
    wv.nq++;
}

// ReferencePlanes::lowresQPelCost's pixels for one 4x4 tile (common/lowres.h:73-93): the half-pel plane the
// quarter-pel vector selects, or the rounded average of two of them (pixelavg_pp, common/pixel.cpp:546-560)
// when a component is odd.  `toff` is the tile's byte offset inside the block.
__device__ __forceinline__ void lowres_tile(const uint8_t *p0, const uint8_t *p1, const uint8_t *p2, const uint8_t *p3,
                                            int stride, int qx, int qy, int toff, uint32_t P[4][2])
{
    // the planes arrive by value: selecting between members of a struct behind a reference makes the compiler
    // index the struct in memory (scratch)
    auto plane = [&](int hpel) {
        const uint8_t *lo = (hpel & 1) ? p1 : p0, *hi = (hpel & 1) ? p3 : p2;
        return as_global((hpel & 2) ? hi : lo);
    };
    const gptr_t a = plane((qy & 2) | ((qx & 2) >> 1)) + (qx >> 2) + (qy >> 2) * stride + toff;
    uint32_t d[4];
    if ((qx | qy) & 1)
    {
        const int bx = s16(qx + (qx & 1) * 2), by = s16(qy + (qy & 1) * 2);
        const gptr_t b = plane((by & 2) | ((bx & 2) >> 1)) + (bx >> 2) + (by >> 2) * stride + toff;
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            const uint32_t u = ld4g(a + y * stride), v = ld4g(b + y * stride);
            d[y] = (u | v) - (((u ^ v) >> 1) & 0x7f7f7f7fu); // per byte (u + v + 1) >> 1
        }
    }
    else
#pragma unroll
        for (int y = 0; y < 4; y++)
            d[y] = ld4g(a + y * stride);
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        P[y][0] = __builtin_amdgcn_perm(0, d[y], 0x0c010c00);
        P[y][1] = __builtin_amdgcn_perm(0, d[y], 0x0c030c02);
    }
}

// Full-pel SAD of the candidates 0 .. nq-1 (all full-pel) of a block whose width is a multiple of 16, added into
// cost[]: a work item is a strip of 16 x 4 pixels (per row one 16-byte load from the plane, one from LDS, four
// v_sad_u8: a quarter of the instructions per pixel of the 4x4 tile form)
__device__ __forceinline__ void wide_sad(Wave &wv, int nq)
{
    WaveLds *s = wv.s;
    const int ns = wv.nt >> 2, total = nq * ns, wx = wv.w >> 4;
    // exact: i < 1024, ns <= 64
    const float inv_ns = __builtin_amdgcn_rcpf((float)ns), inv_wx = __builtin_amdgcn_rcpf((float)wx);
    for (int i = wv.lane; i < total; i += 64)
    {
        const int c = (int)(((float)i + 0.5f) * inv_ns), t = i - c * ns;
        const int sy = (int)(((float)t + 0.5f) * inv_wx), sx = t - sy * wx;
        gptr_t r = as_global(wv.ref) + (((s->qy[c] >> 2) + 4 * sy) * wv.stride + (s->qx[c] >> 2) + 16 * sx);
        const uint8_t *f = s->fenc + sy * 4 * 64 + sx * 16;
        uint32_t a = 0;
        // two rows in flight at a time: the kernel has 72 registers per lane
#pragma unroll
        for (int y = 0; y < 4; y += 2)
        {
            const uint4 r0 = ld16g(r), r1 = ld16g(r + wv.stride);
            r += 2 * wv.stride;
            const uint4 f0 = *(const uint4 *)(f + y * 64);
            a = __builtin_amdgcn_sad_u8(f0.x, r0.x, a);
            a = __builtin_amdgcn_sad_u8(f0.y, r0.y, a);
            a = __builtin_amdgcn_sad_u8(f0.z, r0.z, a);
            a = __builtin_amdgcn_sad_u8(f0.w, r0.w, a);
            const uint4 f1 = *(const uint4 *)(f + y * 64 + 64);
            a = __builtin_amdgcn_sad_u8(f1.x, r1.x, a);
            a = __builtin_amdgcn_sad_u8(f1.y, r1.y, a);
            a = __builtin_amdgcn_sad_u8(f1.z, r1.z, a);
            a = __builtin_amdgcn_sad_u8(f1.w, r1.w, a);
            __builtin_amdgcn_sched_barrier(0);
        }
        atomicAdd(&s->cost[c], (int)a);
    }
}

// cost[c] = mv rate (per its mode) + SAD or SATD of the PU against the reference at candidate c's
// quarter-pel position (subpelCompare + cmp, motion.cpp:1126-1164)
// LOW: 1 the reference is a Lowres, 0 it is not, -1 as wv.lowres says
// WIDE (the caller vouches for a full-resolution reference, at least 64 tiles and a width that is a multiple of 16:
// 32x32, 64x16, 16x64): groups of full-pel SAD candidates go through wide_sad
template <int LOW = -1, bool WIDE = false, bool ONE_D = false>
__device__ __forceinline__ void q_eval(Wave &wv, int group_metric)
{
    static_assert(X265_HIP_SAD == 0 && X265_HIP_SATD == 1, "AUX_SATD turns SAD into SATD by or-ing a bit");
    const bool is_low = LOW < 0 ? wv.lowres : LOW != 0;
    WaveLds *s = wv.s;
    // opaque: otherwise the LDS address of this lane's queue slot is computed once per kernel, spilled to scratch for
    // want of registers and reloaded at the start of every evaluation (a memory round trip for one v_add)
    int lane = wv.lane;
    asm volatile("" : "+v"(lane));
    const int nq = wv.nq;
    // the two cost-table entries of this lane's candidate are asked for here and first used behind the item loop, so
    // that their round trip runs beside the loop's pixel loads instead of in front of them
    int ta = 0, tb = 0, px = 0, py = 0, pa = 0;
    if (lane < nq)
    {
        px = s->qx[lane];
        py = s->qy[lane];
        pa = s->aux[lane];
        if ((pa & 3) != MV_NONE)
        {
            ta = tab_at(wv.tab, wv.mvpx, px);
            tb = tab_at(wv.tab, wv.mvpy, py);
        }
    }
    if (lane < QMAX)
        s->cost[lane] = 0;
    wave_sync();
    // a group of full-pel SAD candidates (every round of the integer stage) of a block of at least 64 tiles whose width
    // is a multiple of 16 (32x32, 64x16, 16x64; 16x16 was 4 % slower this way): strips of 16 x 4 pixels instead of tiles
    if (WIDE && group_metric == X265_HIP_SAD && __ballot(lane < nq && ((px | py) & 3)) == 0)
        wide_sad(wv, nq);
    else
    {
    const int total = nq * wv.nt;
    // i / nt and t / ntx by reciprocal: exact for i < 4096, nt <= 256 (error 4e-4 << 0.5 / nt)
    const float inv_nt = __builtin_amdgcn_rcpf((float)wv.nt), inv_ntx = __builtin_amdgcn_rcpf((float)wv.ntx);
    // blocks whose sides are powers of two (all but the 12 / 24 / 48 ones): shifts and masks instead
    const bool pow2 = ((wv.nt & (wv.nt - 1)) | (wv.ntx & (wv.ntx - 1))) == 0;
    const int sh_nt = __builtin_ctz(wv.nt), sh_ntx = __builtin_ctz(wv.ntx);
    for (int i = lane; i < total; i += 64)
    {
        int c, ty, tx;
        if (pow2)
        {
            c = i >> sh_nt;
            const int t = i & (wv.nt - 1);
            ty = t >> sh_ntx;
            tx = t & (wv.ntx - 1);
        }
        else
        {
            c = (int)(((float)i + 0.5f) * inv_nt);
            const int t = i - c * wv.nt;
            ty = (int)(((float)t + 0.5f) * inv_ntx);
            tx = t - ty * wv.ntx;
        }
        const int qx = s->qx[c], qy = s->qy[c];
        const uint8_t *r = wv.ref + (((qy >> 2) + 4 * ty) * wv.stride + (qx >> 2) + 4 * tx); // one 32-bit offset, one 64-bit add
        const uint8_t *f = s->fenc + ty * 4 * 64 + tx * 4;
        int v;
        const int metric = LOW == 1 ? (group_metric | ((s->aux[c] >> 30) & 1)) : group_metric;
        if (metric == X265_HIP_SAD && ((qx | qy) & 3) == 0)
        {
            // full-pel SAD, the bulk of the integer stage: four v_sad_u8 on the packed rows
            uint32_t a = 0;
#pragma unroll
            for (int y = 0; y < 4; y++)
                a = __builtin_amdgcn_sad_u8(*(const uint32_t *)(f + y * 64), ld4g(as_global(r) + y * wv.stride), a);
            v = (int)a;
        }
        else
        {
            uint32_t pix[4][2], frow[4];
#pragma unroll
            for (int y = 0; y < 4; y++)
                frow[y] = *(const uint32_t *)(f + y * 64);
            if (is_low)
                lowres_tile(wv.ref, wv.lref1, wv.lref2, wv.lref3, wv.stride, qx, qy, 4 * ty * wv.stride + 4 * tx, pix);
            else
            {
                // lanes of this pass that need the diagonal filter pull the half-fractional ones along; without any,
                // the horizontal and the vertical ones share one routine (ref_tile_1d)
                const int xf = qx & 3, yf = qy & 3;
                const bool any_diag = __ballot(xf && yf) != 0;
                if (ONE_D && !any_diag && (xf | yf))
                {
                    ref_tile_1d(r, wv.stride, xf, yf, pix);
                    if (xf == 0)
                        transpose4x4_u8(frow);
                }
                else
                    ref_tile(r, wv.stride, xf, yf, pix, any_diag);
            }
            v = tile_metric_rows(frow, pix, metric);
        }
        atomicAdd(&s->cost[c], v);
    }
    }
    wave_sync();
    const int seed = (pa & 3) == MV_COST ? (int)(uint16_t)(ta + tb) : ta + tb; // MV_NONE: 0 + 0
    wv.vc = lane < nq ? s->cost[lane & (QMAX - 1)] + seed : 0x7fffffff;
    wv.vx = px;
    wv.vy = py;
    wv.va = pa;
}

// The same for blocks of at least 64 tiles whose width is a multiple of 16 (32x32 and up, 64x16, 16x64): the 4x4 tile
// is the wrong unit for them.  A full-pel SAD candidate is dealt out as strips of 16 x 4 pixels (per row one 16-byte
// load from the plane, one from LDS, four v_sad_u8: a quarter of the instructions per pixel); every other candidate
// as strips of NT stacked tiles, NT = 4 for 64x64 (ref_strip<4>: 23 filtered rows for four tiles instead of 4 x 11),
// 2 for 128 or 192 tiles, 1 for 64 tiles, so that a candidate still fills a pass of the wave.  Either way a candidate
// is a multiple of 16 items and a DPP row of lanes belongs to one candidate: the partial sums meet inside the row
// first and one lane of it adds them to the candidate's cost.
#ifndef ME_WIDE
#define ME_WIDE 1
#endif
#ifndef ME_BIG_AHEAD
#define ME_BIG_AHEAD 4 // rows of a strip in flight in me_big_kernel's filters
#endif

__device__ __forceinline__ int row_sum16(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);  // quad_perm 1,0,3,2
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);  // quad_perm 2,3,0,1
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false); // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false); // row_mirror
    return v;
}

// candidates sel[0 .. count) as strips of NT tiles
template <int NT>
__device__ __forceinline__ void big_strips(Wave &wv, const int *sel, int count, int metric)
{
    WaveLds *s = wv.s;
    const int ns = wv.nt / NT, total = count * ns;
    const float inv_ns = __builtin_amdgcn_rcpf((float)ns), inv_ntx = __builtin_amdgcn_rcpf((float)wv.ntx);
    for (int i = wv.lane; i < total; i += 64)
    {
        const int k = (int)(((float)i + 0.5f) * inv_ns), t = i - k * ns, c = sel[k];
        const int ty = (int)(((float)t + 0.5f) * inv_ntx), tx = t - ty * wv.ntx;
        const int qx = s->qx[c], qy = s->qy[c];
        const uint8_t *r = wv.ref + (((qy >> 2) + 4 * NT * ty) * wv.stride + (qx >> 2) + 4 * tx);
        const uint8_t *f = s->fenc + ty * 4 * NT * 64 + tx * 4;
        const bool any_diag = __ballot((qx & 3) && (qy & 3)) != 0;
        uint32_t pix[4 * NT][2];
        ref_strip<NT, false, ME_BIG_AHEAD>(r, wv.stride, qx & 3, qy & 3, pix, any_diag);
        int v = 0;
#pragma unroll
        for (int u = 0; u < NT; u++)
            v += tile_metric(f + u * 4 * 64, 64, &pix[4 * u], metric);
        v = row_sum16(v);
        if ((wv.lane & 15) == 0)
            atomicAdd(&s->cost[c], v);
    }
}

__device__ __forceinline__ void q_eval_big(Wave &wv, int metric)
{
    WaveLds *s = wv.s;
    const int lane = wv.lane, nq = wv.nq;
    int ta = 0, tb = 0, px = 0, py = 0, pa = 0; // as in q_eval
    bool wide = false;
    if (lane < nq)
    {
        px = s->qx[lane];
        py = s->qy[lane];
        pa = s->aux[lane];
        if ((pa & 3) != MV_NONE)
        {
            ta = tab_at(wv.tab, wv.mvpx, px);
            tb = tab_at(wv.tab, wv.mvpy, py);
        }
        wide = metric == X265_HIP_SAD && ((px | py) & 3) == 0;
    }
    // the two kinds of candidate, each list in queue order
    const uint32_t mw = (uint32_t)__ballot(wide), mt = (uint32_t)__ballot(lane < nq && !wide);
    if (lane < nq)
        s->sel[wide ? 0 : 1][__builtin_amdgcn_mbcnt_lo(wide ? mw : mt, 0u)] = lane;
    if (lane < QMAX)
        s->cost[lane] = 0;
    wave_sync();
    const int nwide = __popc(mw), ntall = __popc(mt);
    if (nwide)
    {
        const int ns = wv.nt >> 2, total = nwide * ns, wx = wv.w >> 4;
        // exact as in q_eval: i < 1024, ns <= 64
        const float inv_ns = __builtin_amdgcn_rcpf((float)ns), inv_wx = __builtin_amdgcn_rcpf((float)wx);
        for (int i = lane; i < total; i += 64)
        {
            const int k = (int)(((float)i + 0.5f) * inv_ns), t = i - k * ns, c = s->sel[0][k];
            const int sy = (int)(((float)t + 0.5f) * inv_wx), sx = t - sy * wx;
            gptr_t r = as_global(wv.ref) + (((s->qy[c] >> 2) + 4 * sy) * wv.stride + (s->qx[c] >> 2) + 16 * sx);
            const uint8_t *f = s->fenc + sy * 4 * 64 + sx * 16;
            uint32_t a = 0;
            // the four row loads first (left to itself the compiler reuses one register quad and waits for each)
            uint4 rv[4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                rv[y] = ld16g(r);
                r += wv.stride;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                const uint4 fv = *(const uint4 *)(f + y * 64);
                a = __builtin_amdgcn_sad_u8(fv.x, rv[y].x, a);
                a = __builtin_amdgcn_sad_u8(fv.y, rv[y].y, a);
                a = __builtin_amdgcn_sad_u8(fv.z, rv[y].z, a);
                a = __builtin_amdgcn_sad_u8(fv.w, rv[y].w, a);
            }
            const int v = row_sum16((int)a);
            if ((lane & 15) == 0)
                atomicAdd(&s->cost[c], v);
        }
    }
    if (ntall)
    {
        if (wv.nt == 256)
            big_strips<4>(wv, s->sel[1], ntall, metric);
        else if (wv.nt >= 128)
            big_strips<2>(wv, s->sel[1], ntall, metric);
        else
            big_strips<1>(wv, s->sel[1], ntall, metric);
    }
    wave_sync();
    const int seed = (pa & 3) == MV_COST ? (int)(uint16_t)(ta + tb) : ta + tb; // MV_NONE: 0 + 0
    wv.vc = lane < nq ? s->cost[lane & (QMAX - 1)] + seed : 0x7fffffff;
    wv.vx = px;
    wv.vy = py;
    wv.va = pa;
}

// k is wave-uniform at every call site
__device__ __forceinline__ int q_cost(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.vc, k); }
__device__ __forceinline__ int q_x(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.vx, k); }
__device__ __forceinline__ int q_y(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.vy, k); }
__device__ __forceinline__ int q_tag(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.va, k) >> 2; }


// queue a full-pel point with mvcost(mv << 2); the shift is an int16 operation in the reference
// (mv.h:63).  tag carries (point << 16) | dist for the STAR pattern.
__device__ __forceinline__ void push_f(Wave &wv, int mx, int my, int point = 0, int dist = 0)
{
    q_push(wv, s16(s16(mx) << 2), s16(s16(my) << 2), MV_COST, (point << 16) | dist);
}

// Queue `count` pattern points in one go: lane k computes centre + pat[first + k] * mul, in quarter-pel
// after << sh (int16 wrap as in the reference's MV arithmetic), and writes its own queue slot.
__device__ __forceinline__ void push_pat(Wave &wv, int cx, int cy, int first, int count, int mul, int sh, int mvmode)
{
    if (wv.lane < count)
    {
        const int p = wv.s->pat[first + wv.lane];
        const int slot = wv.nq + wv.lane;
        wv.s->qx[slot] = s16(s16(cx + pat_dx(p) * mul) << sh);
        wv.s->qy[slot] = s16(s16(cy + pat_dy(p) * mul) << sh);
        wv.s->aux[slot] = mvmode;
    }
    wv.nq += count;
}

// Queue the points lanes 0 .. count-1 offer, skipping the invalid ones and keeping lane order (= the
// reference's visiting order): lane k's slot is the number of valid lanes below it.
__device__ __forceinline__ void push_lanes(Wave &wv, int count, int qx, int qy, int aux, bool valid)
{
    const bool mine = valid && wv.lane < count;
    const unsigned long long m = __ballot(mine);
    if (mine)
    {
        const int slot = wv.nq + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        wv.s->qx[slot] = qx;
        wv.s->qy[slot] = qy;
        wv.s->aux[slot] = aux;
    }
    wv.nq += __popcll(m);
}

// pattern entry k as a wave-uniform value (k uniform)
// Pattern entry k as a wave-uniform value (k uniform).  IN_REG: from `patv`, a register copy of the table with one lane
// per entry (pat_vector()).  The searches against Lowres references use that: they are latency-bound chains and the
// lookahead kernel has room for it.  In me_kernel the register was the allocator's first choice for a spill and every
// hexagon step waited for a scratch reload, so the other searches read the LDS copy.
__device__ __forceinline__ int pat_vector() { return (int)(threadIdx.x & 63) < PAT_N ? k_pat[threadIdx.x & 63] : 0; }
template <bool IN_REG>
__device__ __forceinline__ int pat_at(const Wave &wv, int patv, int k)
{
    return IN_REG ? __builtin_amdgcn_readlane(patv, k) : uniform(wv.s->pat[k]);
}

// First minimum among the evaluated candidates [off, off + count) (count >= 1, all inside lanes 0..15):
// returns (cost << 4) | (k - off).  Sequential strict-less updates in push order pick exactly this one.
__device__ __forceinline__ int q_argmin(const Wave &wv, int off, int count)
{
    const int k = wv.lane - off;
    int key = (k >= 0 && k < count) ? (wv.vc << 4) | k : 0x7fffffff;
    key = min(key, __builtin_amdgcn_update_dpp(0x7fffffff, key, 0xB1, 0xf, 0xf, false));  // quad_perm 1,0,3,2
    key = min(key, __builtin_amdgcn_update_dpp(0x7fffffff, key, 0x4E, 0xf, 0xf, false));  // quad_perm 2,3,0,1
    key = min(key, __builtin_amdgcn_update_dpp(0x7fffffff, key, 0x141, 0xf, 0xf, false)); // row_half_mirror
    key = min(key, __builtin_amdgcn_update_dpp(0x7fffffff, key, 0x140, 0xf, 0xf, false)); // row_mirror
    return __builtin_amdgcn_readlane(key, 0);
}

__device__ __constant__ const int8_t k_twopoint[16][2] = { { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 },
                                                          { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 } };
__device__ __constant__ const int8_t k_umh_oct[8][2] = { { 0, -2 }, { -1, -1 }, { 1, -1 }, { -2, 0 }, { 2, 0 }, { -1, 1 }, { 1, 1 }, { 0, 2 } };
__device__ __constant__ const int8_t k_umh_hex[8][2] = { { -1, -2 }, { 1, -2 }, { -2, -1 }, { 2, -1 }, { -2, 1 }, { 2, 1 }, { -1, 2 }, { 1, 2 } };
__device__ __constant__ const uint8_t k_range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };

// One position per lane: SAD of the whole PU at full-pel (mx, my), source pixels broadcast from LDS
__device__ __forceinline__ int lane_sad(const Wave &wv, int mx, int my)
{
    const uint8_t *r = wv.ref + my * wv.stride + mx;
    uint32_t acc = 0;
    for (int y = 0; y < wv.h; y++)
    {
        const uint8_t *rr = r + y * wv.stride;
        const uint8_t *ff = wv.s->fenc + y * 64;
        for (int x = 0; x < wv.ntx; x++)
            acc = __builtin_amdgcn_sad_u8(*(const uint32_t *)(ff + 4 * x), ld4g(as_global(rr) + 4 * x), acc);
    }
    return (int)acc;
}

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = min(v, __shfl_xor(v, off, 64));
    return v;
}

enum State
{
    ST_PRE, ST_INT_BEGIN, ST_DIA, ST_HEX_BEGIN, ST_HEX1, ST_HEXN, ST_SQ,
    ST_U0, ST_U1, ST_U2, ST_CROSS, ST_U3, ST_U3B, ST_U4, ST_U5, ST_U5B, ST_URING,
    ST_STAR_BEGIN, ST_STAR_RING, ST_STAR_END, ST_STAR_TPA, ST_STAR_AFTER_A, ST_STAR_LOOP, ST_STAR_TPB,
    ST_RASTER, ST_POST, ST_HPEL, ST_REM, ST_QPEL_BEGIN, ST_QPEL, ST_LOWRES_HPEL, ST_LOWRES_QPEL, ST_DONE
};

// (cost, vector) come back wave-uniform in rcost / rmx / rmy
// SMALL: the diamond and hexagon searches only (the state of the other methods, and their code, drop out of the
// instantiation: the hexagon search of the medium preset then keeps more of its scalar state in registers)
// LOWRES: the reference is a Lowres (the lookahead's searches): the other sub-pel stage drops out, and vice versa
// BIG: big_block() sizes, evaluated by q_eval_big (never against a Lowres: the lookahead's blocks are 8x8)
// WIDE: the block has at least 64 tiles and a width that is a multiple of 16 (q_eval's wide_sad path)
// MSET (instantiations that are not SMALL): 1 the UMH search only, 2 STAR and FULL only, 3 all of them.  Like SMALL: the
// states, variables and code of the methods a call cannot use drop out of its instantiation.
template <bool SMALL, bool LOWRES, bool BIG = false, bool WIDE = false, int MSET = 3>
__device__ __forceinline__ void motion_estimate_uv(Wave &wv, const x265_hip_me_job &j, int &rcost, int &rmx, int &rmy, int patv = 0)
{
    static_assert(!SMALL || MSET == 3, "MSET selects among the methods SMALL leaves out");
    constexpr bool DIRECT = LOWRES && ME_DIRECT_STATES;
    // integer-stage state, all wave-uniform
    const int lox = j.mvmin_x, loy = j.mvmin_y, hix = j.mvmax_x, hiy = j.mvmax_y; // full-pel, inclusive
    int bcost = 0, bx = 0, by = 0; // best so far: full-pel during the integer stage, quarter-pel after ST_POST
    int point = 0, dist = 0;       // STAR bookkeeping
    const int qlox = s16(lox << 2), qloy = s16(loy << 2), qhix = s16(hix << 2), qhiy = s16(hiy << 2);
    int merange = j.merange;
    const int method = j.method & ~X265_HIP_LOWRES_REF;
    const int nmvc = j.num_mvc;
    const int subme = j.subme;
    // workload[subme] (motion.cpp:47-57) out of nibble-packed constants, one per column: scalar arithmetic instead of
    // five loads from a table in memory whose round trip stood at the head of every call
    const int wsh = 4 * (subme & 7);
    const int hpel_iters = (int)(workload_column(0) >> wsh) & 15, hpel_dirs = (int)(workload_column(1) >> wsh) & 15;
    const int qpel_iters = (int)(workload_column(2) >> wsh) & 15, qpel_dirs = (int)(workload_column(3) >> wsh) & 15;
    const int hsatd = (int)(workload_column(4) >> wsh) & 15;

#define IN_RANGE(x, y) ((x) >= lox && (x) <= hix && (y) >= loy && (y) <= hiy)

    // clipped quarter-pel predictor (motion.cpp:552; MV::clipped, mv.h:89-94)
    int pmx = min((int)j.qmvp_x, qhix), pmy = min((int)j.qmvp_y, qhiy);
    pmx = max(pmx, qlox);
    pmy = max(pmy, qloy);
    int bpx = pmx, bpy = pmy, bprecost = 0;                              // bestpre
    const int fpx = s16(s16(pmx + 2) >> 2), fpy = s16(s16(pmy + 2) >> 2); // pmv.roundToFPel(), mv.h:76
    const bool sub = ((pmx | pmy) & 3) != 0;
    const bool nz = (pmx | pmy) != 0;

    // first group: predictor (SAD only, no mv rate), its full-pel rounding, the zero vector and the
    // candidates (motion.cpp:551-598); consumed in the reference's order in ST_PRE
    wv.nq = 0;
    q_push(wv, pmx, pmy, MV_NONE);
    if (sub)
        q_push(wv, s16(fpx << 2), s16(fpy << 2), MV_COST);
    if (nz)
        q_push(wv, 0, 0, MV_COST);
    const int first_mvc = wv.nq;
    {
        // lane i takes candidate i (one load for all of them; a loop over the list was one memory round trip per
        // candidate at the head of the call); the valid ones are queued in list order
        int mx = 0, my = 0;
        bool valid = false;
        if (wv.lane < nmvc)
        {
            int pair;
            __builtin_memcpy(&pair, &j.mvc[2 * wv.lane], 4);
            mx = max(min((int)(int16_t)pair, qhix), qlox);
            my = max(min(pair >> 16, qhiy), qloy);
            // zero and pmv are skipped here; "m != bestpre" (motion.cpp:584) is resolved when consuming
            valid = (mx | my) && !(mx == pmx && my == pmy);
        }
        push_lanes(wv, nmvc, mx, my, MV_COST, valid);
    }

    int st = ST_PRE, metric = X265_HIP_SAD, flush = FL_NONE;
    // per-method scratch (uniform)
    int cnt = 0, dir = 0;                             // DIA / HEX iteration count, hexagon direction
    int ox = 0, oy = 0;                               // pattern origin
    int ucost1 = 0, ucost2 = 0, cross_start = 1, urange = 0, ring = 1;
    bool ring_raw = false;
    int cr_i = 0, cr_phase = 4, cr_xmax = 0, cr_ymax = 0, cr_start = 0, cr_next = ST_POST; // CROSS generator
    int saved = 0, rounds = 0, early = 3, sd = 1, star_phase = 0;                          // STAR
    int ras_step = 1, ras_next = ST_POST;
    bool ras_quirk = false;
    int sub_it = 0;
    // Speculation: a group may carry candidates that are only needed if the step before them does not
    // move the centre (the square refine behind a hexagon step, the quarter-pel ring behind the half-pel
    // ring).  They are evaluated in the same pass when that costs no or little extra time, consumed in
    // the reference's order if the centre stayed, and dropped otherwise, so results are unchanged.
    const bool spec_sq = wv.nt <= SPEC_SQ_TILES;
    const bool spec_low = LOWRES && ME_SPEC_LOW && (1 + hpel_dirs + qpel_dirs) * wv.nt <= 64 && 1 + hpel_dirs + qpel_dirs <= QMAX;
    const bool spec_qp = hsatd && hpel_iters == 1 && qpel_iters > 0 && wv.nt <= SPEC_QPEL_TILES &&
                         1 + hpel_dirs + qpel_dirs <= QMAX;

    // square refine around (bx, by) from the costs at queue offset `off` (motion.cpp:688-699)
    // ring of `count` square1 points (x mul) around (bx, by) whose costs sit at queue offset `off`:
    // strict-less in push order, i.e. the first minimum if it beats bcost; moves (bx, by) there
    auto take_ring = [&](int off, int count, int mul) {
        const int key = q_argmin(wv, off, count);
        if ((key >> 4) < bcost)
        {
            bcost = key >> 4;
            const int p = pat_at<LOWRES>(wv, patv, PAT_SQ + 1 + (key & 15));
            bx = s16(bx + pat_dx(p) * mul);
            by = s16(by + pat_dy(p) * mul);
            return true;
        }
        return false;
    };
    // square refine (motion.cpp:688-699)
    auto take_square = [&](int off) { take_ring(off, 8, 1); };
    auto push_square = [&]() { push_pat(wv, bx, by, PAT_SQ + 1, 8, 1, 2, MV_COST); };

    for (;;)
    {
        metric = uniform(metric); // keeps it in a scalar register across the states (it was a spilled vector register)
        if (wv.nq)
        {
            if constexpr (BIG)
                q_eval_big(wv, metric);
            else
                q_eval<LOWRES ? 1 : 0, WIDE, ME_ONE_D == 1 || (ME_ONE_D == 2 && SMALL)>(wv, metric);
        }
        if (flush != FL_NONE)
        {
            if (wv.nq)
            {
                const int key = q_argmin(wv, 0, wv.nq);
                if ((key >> 4) < bcost)
                {
                    const int k = key & 15;
                    bcost = key >> 4;
                    bx = q_x(wv, k) >> 2;
                    by = q_y(wv, k) >> 2;
                    if (flush == FL_STAR)
                    {
                        const int a = q_tag(wv, k);
                        point = a >> 16;
                        dist = a & 0xffff;
                    }
                }
            }
            wv.nq = 0;
        }
        flush = FL_NONE;
        if (st == ST_DONE)
            break;

        // the states of UMH, STAR and the raster scans: a switch of their own, absent from the SMALL instantiation
        bool other = true;
        if constexpr (!SMALL && (MSET & 1) != 0)
        {
            other = false;
            switch (st)
            {
        // ---- UMH (motion.cpp:702-885)
#define THRESH(v) (bcost < (((v) >> 4) * ((wv.h * wv.h) >> 4))) /* SAD_THRESH, size_scale (motion.cpp:60-66) */
        case ST_U0:
            ucost2 = bcost;
            if ((bx | by) && !(bx == fpx && by == fpy))
            {
                push_f(wv, bx, by - 1);
                push_f(wv, bx, by + 1);
                push_f(wv, bx - 1, by);
                push_f(wv, bx + 1, by);
                flush = FL_BEST;
            }
            st = ST_U1;
            break;

        case ST_U1:
            if (bcost == ucost2)
                cross_start = 3;
            ox = bx;
            oy = by;
            if (bcost == ucost2 && THRESH(2000))
            {
                for (int k = 0; k < 8; k++)
                    push_f(wv, ox + k_umh_oct[k][0], oy + k_umh_oct[k][1]);
                flush = FL_BEST;
                st = ST_U2;
            }
            else
                st = ST_U4;
            break;

        case ST_U2:
            if (bcost == ucost1 && THRESH(500))
                st = ST_POST;
            else if (bcost == ucost2)
            {
                urange = s16((merange >> 1) | 1);
                cr_start = 3;
                cr_xmax = cr_ymax = urange;
                cr_i = s16(cr_start);
                cr_phase = 0;
                cr_next = ST_U3;
                st = ST_CROSS;
            }
            else
                st = ST_U4;
            break;

        // CROSS macro as a resumable generator around (ox, oy) (motion.cpp:258-282): phase 0 paired
        // x4 steps in x (only when the arm fits the range), 1 guarded steps in x, 2 / 3 the same in y
        case ST_CROSS:
        {
            while (cr_phase < 4 && wv.nq <= QMAX - 4)
            {
                if (cr_phase == 0)
                {
                    if (cr_xmax <= min(hix - ox, ox - lox) && cr_i < cr_xmax - 2)
                    {
                        push_f(wv, ox + cr_i, oy);
                        push_f(wv, ox - cr_i, oy);
                        push_f(wv, ox + cr_i + 2, oy);
                        push_f(wv, ox - cr_i - 2, oy);
                        cr_i = s16(cr_i + 4);
                    }
                    else
                        cr_phase = 1;
                }
                else if (cr_phase == 1)
                {
                    if (cr_i < cr_xmax)
                    {
                        if (ox + cr_i <= hix)
                            push_f(wv, ox + cr_i, oy);
                        if (ox - cr_i >= lox)
                            push_f(wv, ox - cr_i, oy);
                        cr_i = s16(cr_i + 2);
                    }
                    else
                    {
                        cr_i = s16(cr_start);
                        cr_phase = 2;
                    }
                }
                else if (cr_phase == 2)
                {
                    if (cr_ymax <= min(hiy - oy, oy - loy) && cr_i < cr_ymax - 2)
                    {
                        push_f(wv, ox, oy + cr_i);
                        push_f(wv, ox, oy - cr_i);
                        push_f(wv, ox, oy + cr_i + 2);
                        push_f(wv, ox, oy - cr_i - 2);
                        cr_i = s16(cr_i + 4);
                    }
                    else
                        cr_phase = 3;
                }
                else
                {
                    if (cr_i < cr_ymax)
                    {
                        if (oy + cr_i <= hiy)
                            push_f(wv, ox, oy + cr_i);
                        if (oy - cr_i >= loy)
                            push_f(wv, ox, oy - cr_i);
                        cr_i = s16(cr_i + 2);
                    }
                    else
                        cr_phase = 4;
                }
            }
            flush = FL_BEST;
            if (cr_phase == 4)
                st = cr_next;
            break;
        }

        case ST_U3:
        {
            for (int k = 0; k < 8; k++)
                push_f(wv, ox + k_umh_hex[k][0], oy + k_umh_hex[k][1]);
            flush = FL_BEST;
            st = ST_U3B;
            break;
        }

        case ST_U3B:
            if (bcost == ucost2)
                st = ST_POST;
            else
            {
                cross_start = urange + 2;
                st = ST_U4;
            }
            break;

        case ST_U4:
        {
            // adaptive search range from predictor agreement and the current SAD (motion.cpp:743-795)
            if (nmvc)
            {
                int mvd, denom = 1;
                const bool is64 = (wv.w == 64 && wv.h == 64);
                const int d0 = abs(j.qmvp_x - j.mvc[0]) + abs(j.qmvp_y - j.mvc[1]);
                if (nmvc == 1)
                    mvd = is64 ? 25 : d0;
                else
                {
                    denom = nmvc - 1;
                    mvd = 0;
                    if (!is64)
                    {
                        mvd = d0;
                        denom++;
                    }
                    for (int i = 0; i < nmvc - 1; i++)
                        mvd += abs(j.mvc[2 * i] - j.mvc[2 * i + 2]) + abs(j.mvc[2 * i + 1] - j.mvc[2 * i + 3]);
                }
                const int sad_ctx = THRESH(1000) ? 0 : THRESH(2000) ? 1 : THRESH(4000) ? 2 : 3;
                const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
                merange = (merange * k_range_mul[mvd_ctx][sad_ctx]) >> 2;
            }
            cr_start = cross_start;
            cr_xmax = merange;
            cr_ymax = merange >> 1;
            cr_i = s16(cr_start);
            cr_phase = 0;
            cr_next = ST_U5;
            st = ST_CROSS;
            break;
        }
#undef THRESH

        case ST_U5:
            push_f(wv, ox - 2, oy - 2);
            push_f(wv, ox - 2, oy + 2);
            push_f(wv, ox + 2, oy - 2);
            push_f(wv, ox + 2, oy + 2);
            flush = FL_BEST;
            st = ST_U5B;
            break;

        case ST_U5B:
        case ST_URING:
        {
            // hexagon-grid rings i = 1 .. merange/4 around the new centre (motion.cpp:802-881).  Inside
            // the window the reference adds the two table entries as ints (no uint16 truncation) and
            // moves once per ring; near the border it falls back to guarded scalar candidates.
            if (st == ST_U5B)
            {
                ox = bx;
                oy = by;
                ring = 1;
            }
            else
            {
                if (ring_raw)
                {
                    // strict-less updates in table order: the first minimum, if it beats bcost
                    const int key = q_argmin(wv, 0, 16);
                    wv.nq = 0;
                    if ((key >> 4) < bcost)
                    {
                        const int p = pat_at<LOWRES>(wv, patv, PAT_HEX4 + (key & 15));
                        bcost = key >> 4;
                        bx = s16(ox + ring * pat_dx(p));
                        by = s16(oy + ring * pat_dy(p));
                    }
                }
                ring = (ring + 1) & 0xffff; // uint16_t counter in the reference
                if (!(ring <= merange >> 2))
                {
                    st = IN_RANGE(bx, by) ? ST_HEX_BEGIN : ST_POST; // "goto me_hex2" (:882-883)
                    break;
                }
            }
            const int room = min(min(hix - ox, ox - lox), min(hiy - oy, oy - loy));
            ring_raw = !(4 * ring > room);
            {
                const int p = wv.s->pat[PAT_HEX4 + (wv.lane & 15)];
                if (ring_raw)
                    push_lanes(wv, 16, (ox + pat_dx(p) * ring) * 4, (oy + pat_dy(p) * ring) * 4, MV_RAW, true);
                else
                {
                    const int mx = s16(ox + s16(pat_dx(p) * s16(ring))), my = s16(oy + s16(pat_dy(p) * s16(ring)));
                    push_lanes(wv, 16, s16(s16(mx) << 2), s16(s16(my) << 2), MV_COST, IN_RANGE(mx, my));
                    flush = FL_BEST;
                }
            }
            st = ST_URING;
            break;
        }
            default:
                other = true;
                break;
            }
        }
        if constexpr (!SMALL && (MSET & 2) != 0)
        {
            if (other)
            {
            other = false;
            switch (st)
            {
        // ---- STAR (motion.cpp:887-995; StarPatternSearch :284-524).  The reference's sad_x4 fast path
        // and its guarded scalar path visit the same points in the same order: the guarded form is used.
        case ST_STAR_BEGIN:
        case ST_STAR_RING:
        {
            if (st == ST_STAR_BEGIN)
            {
                ox = bx;
                oy = by;
                rounds = 0;
                sd = 1;
            }
            else
            {
                bool cont = true;
                if (bcost < saved)
                    rounds = 0;
                else if (++rounds >= early)
                    cont = false;
                if (cont)
                {
                    sd <<= 1;
                    if (sd >= 16 && sd > s16(merange))
                        cont = false;
                }
                if (!cont)
                {
                    st = ST_STAR_END;
                    break;
                }
            }
            saved = bcost;
            {
                // lane k offers the k-th point of the ring in the reference's visiting order.  A point is
                // guarded only on the sides it is displaced to, as in the reference (e.g. "if (top >= mvmin.y)").
                const bool small = sd <= 8;
                const int k = wv.lane & 15;
                const int p = wv.s->pat[(small ? PAT_STAR8 : PAT_STAR16) + (small ? (k & 7) : k)];
                const int ux = pat_dx(p), uy = pat_dy(p);
                // small ring: units of dist / 2 (the diagonal points, 1 3 6 8, exist only for dist >= 2 and carry
                // dist / 2 as their distance); large ring: units of dist / 4, no point number
                const bool diag = small && ((0x66 >> (k & 7)) & 1);
                const int dx = small ? (ux * sd) >> 1 : ux * (sd >> 2), dy = small ? (uy * sd) >> 1 : uy * (sd >> 2);
                const int mx = s16(ox + dx), my = s16(oy + dy);
                const int pnt = small ? (0x78654312 >> (4 * (k & 7))) & 15 : 0;
                const int dst = diag ? sd >> 1 : sd;
                const bool ok = !(diag && sd == 1) && (ux < 0 ? mx >= lox : (ux > 0 ? mx <= hix : true)) &&
                                (uy < 0 ? my >= loy : (uy > 0 ? my <= hiy : true));
                push_lanes(wv, small ? 8 : 16, s16(s16(mx) << 2), s16(s16(my) << 2), ((((pnt << 16) | dst)) << 2) | MV_COST, ok);
            }
            flush = FL_STAR;
            st = ST_STAR_RING;
            break;
        }

        case ST_STAR_END:
        {
            const bool two = dist == 1 && point;
            if (two)
            {
                // nearest two outer points of the winning direction (offsets[], motion.cpp:139-149)
                const int x1 = s16(bx + k_twopoint[(point - 1) * 2][0]), y1 = s16(by + k_twopoint[(point - 1) * 2][1]);
                const int x2 = s16(bx + k_twopoint[(point - 1) * 2 + 1][0]), y2 = s16(by + k_twopoint[(point - 1) * 2 + 1][1]);
                if (IN_RANGE(x1, y1))
                    push_f(wv, x1, y1);
                if (IN_RANGE(x2, y2))
                    push_f(wv, x2, y2);
                flush = FL_BEST; // COST_MV: point / distance are not touched
                saved = bcost;
            }
            if (star_phase == 0)
                st = two ? ST_STAR_TPA : (dist == 1 ? ST_POST : ST_STAR_AFTER_A);
            else
                st = two ? ST_STAR_TPB : (dist == 1 ? ST_POST : ST_STAR_LOOP);
            break;
        }

        case ST_STAR_TPA:
            st = bcost == saved ? ST_POST : ST_STAR_AFTER_A;
            break;

        case ST_STAR_AFTER_A:
            if (dist > 5) // raster refinement if the first pass went far (RasterDistance = 5)
            {
                ras_step = 5;
                ras_quirk = true;
                ras_next = ST_STAR_LOOP;
                st = ST_RASTER;
            }
            else
                st = ST_STAR_LOOP;
            break;

        case ST_STAR_LOOP:
            if (dist > 0)
            {
                dist = 0;
                point = 0;
                star_phase = 1;
                early = 32;
                st = ST_STAR_BEGIN;
            }
            else
                st = ST_POST;
            break;

        case ST_STAR_TPB:
            st = ST_POST;
            break;

        // ---- raster scan of (lox + step*ix, loy + step*iy), row-major, first minimum wins: FULL
        // (motion.cpp:997-1032, step 1) and the STAR raster (:924-958, step 5).  One position per lane.
        // In the STAR form every fourth point of a complete group of four is charged mvcost(mv << 3),
        // a quirk of the reference (:951) that is reproduced.
        case ST_RASTER:
        {
            st = ras_next;
            if (hix < lox || hiy < loy)
                break;
            const int nx = (hix - lox) / ras_step + 1, ny = (hiy - loy) / ras_step + 1;
            const int groups = (hix - lox - 3 * ras_step >= 0) ? (hix - lox - 3 * ras_step) / (4 * ras_step) + 1 : 0;
            const int total = nx * ny;
            for (int base = 0; base < total; base += 64)
            {
                const int idx = base + wv.lane;
                int cost = 0x7fffffff;
                if (idx < total)
                {
                    const int iy = idx / nx, ix = idx - iy * nx;
                    const int mx = lox + ix * ras_step, my = loy + iy * ras_step;
                    const int sh = (ras_quirk && ix < 4 * groups && (ix & 3) == 3) ? 3 : 2;
                    cost = lane_sad(wv, mx, my) + mvcost_q(wv.tab, wv.mvpx, wv.mvpy, s16(mx << sh), s16(my << sh));
                }
                const int m = uniform(wave_min_i32(cost));
                if (m < bcost)
                {
                    const unsigned long long hit = __ballot(cost == m);
                    const int first = base + (int)__builtin_ctzll(hit);
                    const int iy = first / nx, ix = first - iy * nx;
                    bcost = m;
                    bx = lox + ix * ras_step;
                    by = loy + iy * ras_step;
                }
            }
            break;
        }

            default:
                other = true;
                break;
            }
            }
        }
        if (other)
        switch (st)
        {
        case ST_PRE:
        {
            int k = 0;
            bprecost = q_cost(wv, k++);
            bx = fpx;
            by = fpy;
            bcost = bprecost;
            if (sub)
                bcost = q_cost(wv, k++); // full-pel start re-measured with its mv rate (:562-567)
            if (nz)
            {
                const int cost = q_cost(wv, k++); // zero vector (:570-578)
                if (cost < bcost)
                {
                    bcost = cost;
                    bx = by = 0;
                }
            }
            for (k = first_mvc; k < wv.nq; k++)
            {
                const int mx = q_x(wv, k), my = q_y(wv, k);
                if (mx == bpx && my == bpy)
                    continue;
                const int cost = q_cost(wv, k);
                if (cost < bprecost) // candidates compete with bprecost only (:592)
                {
                    bprecost = cost;
                    bpx = mx;
                    bpy = my;
                }
            }
            wv.nq = 0;
            st = ST_INT_BEGIN;
        }
        // In the searches against Lowres references (latency-bound chains) states that queue nothing go straight on to
        // the next one's code (fall through or goto) instead of round the loop and through the dispatch again:
        // measured in the lookahead kernel, the state code between evaluations was 37 % of a search's time, a fifth
        // of it such empty trips.  (In me_kernel's other instantiations the same jumps cost 14 more spilled
        // registers and 4 % of the bench list, so they stay with the loop.)
        if (!DIRECT)
            break;
        [[fallthrough]];

        case ST_INT_BEGIN:
            if (method == X265_HIP_DIA)
            {
                cnt = merange;
                push_pat(wv, bx, by, PAT_SQ + 1, 4, 1, 2, MV_COST); // (0,-1) (0,1) (-1,0) (1,0)
                st = ST_DIA;
            }
            else if (SMALL || method == X265_HIP_HEX)
            {
                st = ST_HEX_BEGIN;
                if (DIRECT)
                    goto hex_begin;
            }
            else if ((MSET & 1) != 0 && (MSET == 1 || method == X265_HIP_UMH))
            {
                // refine predictors: DIA1_ITER(pmv), DIA1_ITER(0) (motion.cpp:707-712)
                ucost1 = bcost;
                push_f(wv, fpx, fpy - 1);
                push_f(wv, fpx, fpy + 1);
                push_f(wv, fpx - 1, fpy);
                push_f(wv, fpx + 1, fpy);
                if (fpx | fpy)
                {
                    push_f(wv, 0, -1);
                    push_f(wv, 0, 1);
                    push_f(wv, -1, 0);
                    push_f(wv, 1, 0);
                }
                flush = FL_BEST;
                st = ST_U0;
            }
            else if ((MSET & 2) != 0 && method == X265_HIP_STAR)
            {
                point = dist = 0;
                star_phase = 0;
                early = 3;
                st = ST_STAR_BEGIN;
            }
            else if ((MSET & 2) != 0)
            {
                ras_step = 1;
                ras_quirk = false;
                ras_next = ST_POST;
                st = ST_RASTER;
            }
            break;

        // ---- DIA: radius-1 diamond, packed (cost << 4) + code, centre wins ties (motion.cpp:605-626)
        case ST_DIA:
        {
            // the reference packs (cost << 4) + code with codes 1, 3, 4, 12 in visiting order and takes the
            // minimum against bcost << 4: the first minimum, if strictly better than the centre
            wv.nq = 0;
            if (!take_ring(0, 4, 1))
            {
                st = ST_POST;
                if (DIRECT)
                    goto post_stage;
                break;
            }
            if (--cnt && IN_RANGE(bx, by))
                push_pat(wv, bx, by, PAT_SQ + 1, 4, 1, 2, MV_COST);
            else
            {
                st = ST_POST;
                if (DIRECT)
                    goto post_stage;
            }
            break;
        }

        // ---- HEX: radius-2 hexagon then square refine (motion.cpp:628-700)
        case ST_HEX_BEGIN:
        hex_begin:
            push_pat(wv, bx, by, PAT_HEX + 1, 6, 1, 2, MV_COST); // hex2[1..6]: directions 0..5
            if (spec_sq)
                push_square();
            st = ST_HEX1;
            break;

        case ST_HEX1:
        case ST_HEXN:
        {
            bool go_on = false, moved = false;
            const int nstep = st == ST_HEX1 ? 6 : 3; // the speculated square sits behind the step's points
            // the reference's packed minimum ((cost << 3) + code against bcost << 3): first minimum of the
            // step if strictly better than the centre
            const int key = q_argmin(wv, 0, nstep);
            wv.nq = 0;
            if ((key >> 4) < bcost)
            {
                moved = true;
                bcost = key >> 4;
                if (st == ST_HEX1)
                {
                    dir = key & 15;
                    cnt = (merange >> 1) - 1;
                }
                else
                {
                    dir += (key & 15) - 1;
                    dir = dir < 0 ? dir + 6 : (dir >= 6 ? dir - 6 : dir); // mod6m1
                    cnt--;
                }
                const int p = pat_at<LOWRES>(wv, patv, PAT_HEX + dir + 1);
                bx = s16(bx + pat_dx(p));
                by = s16(by + pat_dy(p));
                go_on = cnt > 0 && IN_RANGE(bx, by);
            }
            if (go_on)
            {
                // half hexagon, not overlapping the previous one
                push_pat(wv, bx, by, PAT_HEX + dir, 3, 1, 2, MV_COST);
                if (spec_sq)
                    push_square();
                st = ST_HEXN;
            }
            else
            {
                if (spec_sq && !moved)
                {
                    take_square(nstep); // the centre stayed: the square was evaluated with this step
                    st = ST_POST;
                    if (DIRECT)
                        goto post_stage;
                }
                else
                {
                    push_square();
                    st = ST_SQ;
                }
            }
            break;
        }

        case ST_SQ:
            take_square(0);
            wv.nq = 0;
            st = ST_POST;
            if (!DIRECT)
                break;
            [[fallthrough]];

        // ---- after the integer stage (motion.cpp:1039-1119)
        case ST_POST:
        post_stage:
            if (bprecost < bcost) // pre-test winner bypasses the integer result
            {
                bx = bpx;
                by = bpy;
                bcost = bprecost;
            }
            else
            {
                bx = s16(bx << 2);
                by = s16(by << 2);
            }
            if (!bcost)
            {
                bcost = uniform(mvcost_q(wv.tab, wv.mvpx, wv.mvpy, bx, by)); // zero residual: rate only (:1049-1054)
                st = ST_DONE;
                break;
            }
            if (LOWRES)
            {
                // motion.cpp:1055-1080: one half-pel ring with SAD, then the SATD re-measure of the winner
                // together with its quarter-pel ring; the iteration counts of workload[] are not used
                metric = X265_HIP_SAD;
                push_pat(wv, bx, by, PAT_SQ + 1, hpel_dirs, 2, 0, MV_COST);
                if (spec_low)
                {
                    // what follows if the half-pel ring does not move the centre, in the same pass (idle lanes)
                    q_push(wv, bx, by, MV_COST, AUX_SATD >> 2);
                    push_pat(wv, bx, by, PAT_SQ + 1, qpel_dirs, 1, 0, MV_COST | AUX_SATD);
                }
                st = ST_LOWRES_HPEL;
                break;
            }
            sub_it = 0;
            metric = hsatd;
            // with hpel_satd the SATD re-measure of the centre does not depend on the ring: same group
            if (hsatd)
                q_push(wv, bx, by, MV_COST);
            push_pat(wv, bx, by, PAT_SQ + 1, hpel_dirs, 2, 0, MV_COST);
            if (spec_qp)
                push_pat(wv, bx, by, PAT_SQ + 1, qpel_dirs, 1, 0, MV_COST);
            st = ST_HPEL;
            break;

        case ST_HPEL:
        {
            if (LOWRES) // not a state of this instantiation
            {
                st = ST_DONE;
                break;
            }
            int k = 0;
            if (hsatd && sub_it == 0)
                bcost = q_cost(wv, k++);
            const bool moved = take_ring(k, hpel_dirs, 2);
            k += hpel_dirs;
            wv.nq = 0;
            if (++sub_it < hpel_iters)
                push_pat(wv, bx, by, PAT_SQ + 1, hpel_dirs, 2, 0, MV_COST);
            else if (!hsatd)
            {
                metric = X265_HIP_SATD; // half-pel used SAD: re-measure with SATD before quarter-pel (:1103-1105)
                q_push(wv, bx, by, MV_COST);
                st = ST_REM;
            }
            else if (spec_qp && !moved)
            {
                // the centre stayed: the first quarter-pel ring was evaluated with the half-pel ring
                take_ring(k, qpel_dirs, 1);
                sub_it = 1;
                if (sub_it < qpel_iters)
                {
                    push_pat(wv, bx, by, PAT_SQ + 1, qpel_dirs, 1, 0, MV_COST);
                    st = ST_QPEL;
                }
                else
                    st = ST_DONE;
            }
            else
                st = ST_QPEL_BEGIN;
            break;
        }

        case ST_LOWRES_HPEL:
            if (!LOWRES) // not a state of this instantiation
            {
                st = ST_DONE;
                break;
            }
            if (!take_ring(0, hpel_dirs, 2) && spec_low)
            {
                bcost = q_cost(wv, hpel_dirs);
                take_ring(hpel_dirs + 1, qpel_dirs, 1);
                wv.nq = 0;
                st = ST_DONE;
                break;
            }
            wv.nq = 0;
            metric = X265_HIP_SATD;
            q_push(wv, bx, by, MV_COST);
            push_pat(wv, bx, by, PAT_SQ + 1, qpel_dirs, 1, 0, MV_COST);
            st = ST_LOWRES_QPEL;
            break;

        case ST_LOWRES_QPEL:
            if (!LOWRES) // not a state of this instantiation
            {
                st = ST_DONE;
                break;
            }
            bcost = q_cost(wv, 0);
            take_ring(1, qpel_dirs, 1);
            wv.nq = 0;
            st = ST_DONE;
            break;

        case ST_REM:
            if (LOWRES) // not a state of this instantiation
            {
                st = ST_DONE;
                break;
            }
            bcost = q_cost(wv, 0);
            wv.nq = 0;
            st = ST_QPEL_BEGIN;
            break;

        case ST_QPEL_BEGIN:
            if (LOWRES) // not a state of this instantiation
            {
                st = ST_DONE;
                break;
            }
            sub_it = 0;
            metric = X265_HIP_SATD;
            if (qpel_iters > 0)
            {
                push_pat(wv, bx, by, PAT_SQ + 1, qpel_dirs, 1, 0, MV_COST);
                st = ST_QPEL;
            }
            else
                st = ST_DONE;
            break;

        case ST_QPEL:
            if (LOWRES) // not a state of this instantiation
            {
                st = ST_DONE;
                break;
            }
            take_ring(0, qpel_dirs, 1);
            wv.nq = 0;
            if (++sub_it < qpel_iters)
                push_pat(wv, bx, by, PAT_SQ + 1, qpel_dirs, 1, 0, MV_COST);
            else
                st = ST_DONE;
            break;

        default:
            st = ST_DONE;
            break;
        }
    }
#undef IN_RANGE
    rcost = bcost;
    rmx = bx;
    rmy = by;
}

template <bool BIG>
__device__ __forceinline__ void motion_estimate(Wave &wv, const x265_hip_me_job &j, x265_hip_me_result *out)
{
    int bcost, bx, by;
    const bool small = (j.method & ~X265_HIP_LOWRES_REF) <= X265_HIP_HEX;
    if constexpr (BIG)
    {
        if (small)
            motion_estimate_uv<true, false, true>(wv, j, bcost, bx, by);
        else
            motion_estimate_uv<false, false, true>(wv, j, bcost, bx, by);
    }
    else if (wv.lowres)
    {
        if (small)
            motion_estimate_uv<true, true>(wv, j, bcost, bx, by, pat_vector());
        else
            motion_estimate_uv<false, true>(wv, j, bcost, bx, by, pat_vector());
    }
    else if (small)
    {
        if (ME_WIDE && (wv.w & 15) == 0 && wv.nt >= 64)
            motion_estimate_uv<true, false, false, true>(wv, j, bcost, bx, by);
        else
            motion_estimate_uv<true, false>(wv, j, bcost, bx, by);
    }
    else if ((j.method & ~X265_HIP_LOWRES_REF) == X265_HIP_UMH)
        motion_estimate_uv<false, false, false, false, 1>(wv, j, bcost, bx, by);
    else
        motion_estimate_uv<false, false, false, false, 2>(wv, j, bcost, bx, by);
    if (wv.lane == 0)
    {
        out->cost = bcost;
        out->mvx = (int16_t)bx;
        out->mvy = (int16_t)by;
    }
}

#ifndef ME_XCD_CHUNK
#define ME_XCD_CHUNK 1024 // about what one XCD holds at a time (768 waves): measured 16 .. 1024, larger is better
#endif
#ifndef ME_BIG_WAVES
#define ME_BIG_WAVES 4
#endif
// One motionEstimate call per wave (me_kernel, me_big_kernel).  A launch that routes (me_shared.h) splits the list by
// kind first: me_big_kernel takes the MC_BIG jobs (more registers per lane, fewer waves per SIMD), me_team_kernel
// (me_team.hip) the small diamond / hexagon searches, several per wave, and me_kernel in its list mode the rest.  Without
// routing (host-array callers whose list holds nothing but MC_REST jobs) me_kernel runs one workgroup per job.
template <bool BIG>
__device__ __forceinline__ void me_job(WaveLds &lds, const DeviceTables *tabs, const x265_hip_me_job &j, x265_hip_me_result *out)
{
    Wave wv;
    wave_init(wv, &lds, j.pu_w, j.pu_h);
    const PlaneDesc fp = tabs->planes[j.fenc_plane], rp = tabs->planes[j.ref_plane];
    wv.stride = rp.stride;
    wv.ref = rp.origin + (long)j.pu_y * rp.stride + j.pu_x;
    wv.tab = tabs->costtab[j.qp];
    wv.mvpx = j.qmvp_x;
    wv.mvpy = j.qmvp_y;
    if (!BIG && (j.method & X265_HIP_LOWRES_REF))
    {
        wv.lowres = true;
        const long off = (long)j.pu_y * rp.stride + j.pu_x;
        wv.lref1 = tabs->planes[j.ref_plane + 1].origin + off;
        wv.lref2 = tabs->planes[j.ref_plane + 2].origin + off;
        wv.lref3 = tabs->planes[j.ref_plane + 3].origin + off;
    }
    load_fenc(wv, fp.origin + (long)j.pu_y * fp.stride + j.pu_x, fp.stride);
    motion_estimate<BIG>(wv, j, out);
}

// The waves of a list kernel: gridDim.x is a multiple of ME_PARTS; wave b serves part b mod ME_PARTS, starts with the
// list entry of its own rank within the part and takes further ones from the part's hand-out counter; it leaves when
// that has passed the end of the part's list (it only grows), so every wave reaches its exit.
template <bool BIG>
__device__ __forceinline__ void me_list_loop(WaveLds &lds, const DeviceTables *tabs, const x265_hip_me_job *jobs, uint32_t *route, int parity, int cap,
                                             x265_hip_me_result *out, int out_stride)
{
    constexpr int cls = BIG ? MC_BIG : MC_REST;
    const int p = blockIdx.x & (ME_PARTS - 1), waves = gridDim.x / ME_PARTS;
    const uint32_t count = min(*route_append(route, parity, cls, p), (uint32_t)cap);
    const uint32_t *const list = route_list(route, cap, cls, p);
    uint32_t k = blockIdx.x / ME_PARTS;
    while (k < count)
    {
        const uint32_t id = list[k];
        me_job<BIG>(lds, tabs, jobs[id], (x265_hip_me_result *)((char *)out + (size_t)id * out_stride));
        wave_sync();
        k = 0;
        if ((threadIdx.x & 63) == 0)
            k = (uint32_t)waves + atomicAdd(route_handout(route, parity, cls, p), 1u);
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    }
}

// results are out_stride bytes apart: 8 for x265_hip_me_result arrays, more when they head a larger record
#ifndef ME_WAVES
#define ME_WAVES 7
#endif
// one workgroup per job of the list (launches that do not route)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ME_WAVES))) void me_kernel(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n,
                                                x265_hip_me_result *out, int out_stride)
{
    __shared__ WaveLds lds;
    if ((int)blockIdx.x >= n)
        return;
    const int id = xcd_chunked(blockIdx.x, n, ME_XCD_CHUNK);
    me_job<false>(lds, tabs, jobs[id], (x265_hip_me_result *)((char *)out + (size_t)id * out_stride));
}

// the MC_REST jobs of a routed launch.  Six waves per SIMD: with the list loop's state around the search, seven (72
// registers) spilled 23 of them
#ifndef ME_REST_WAVES
#define ME_REST_WAVES 6
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ME_REST_WAVES))) void me_rest_kernel(const DeviceTables *tabs, const x265_hip_me_job *jobs,
                                                uint32_t *route, int parity, int cap, x265_hip_me_result *out, int out_stride)
{
    __shared__ WaveLds lds;
    me_list_loop<false>(lds, tabs, jobs, route, parity, cap, out, out_stride);
}

// The lists of a routed launch (me_shared.h).  A wave's 64 consecutive jobs go to one part, each kind with one atomic
// per wave; list order inside a part follows job order inside a wave.  Block 0 also clears the counter set of the next
// launch: that set was last used by the launch before this one, which is over (launches of a context are ordered).
//
// The record guard (`guard` not null; x265_hip_ctx_check_records mode 2): a thread puts its record through the checks
// the host-array entry point makes of it (check_job in api.hip: plane ids, partition, method / subme / candidate count,
// cost table, and that every sample the search may touch, filter margins included, lies inside the planes' buffers).  A
// record that fails is put on no list (no kernel ever reads it), gets the result (cost -1, mv 0) and is counted in
// guard[0]; guard[1] keeps the lowest index of one.  Nothing is copied back to the host.
__device__ bool guard_me_job(const DeviceTables *tabs, const x265_hip_me_job &j)
{
    const int method = j.method & ~X265_HIP_LOWRES_REF;
    if (j.fenc_plane >= X265_HIP_MAX_PLANES || j.ref_plane >= X265_HIP_MAX_PLANES)
        return false;
    const PlaneDesc fp = tabs->planes[j.fenc_plane], rp = tabs->planes[j.ref_plane];
    if (!fp.origin || !rp.origin || !guard_partition(j.pu_w, j.pu_h))
        return false;
    if (method > X265_HIP_FULL || j.subme > 7 || j.num_mvc > X265_HIP_MAX_MVC)
        return false;
    if (j.method & X265_HIP_LOWRES_REF)
    {
        if (j.pu_w != 8 || j.pu_h != 8 || j.ref_plane + 3 >= X265_HIP_MAX_PLANES)
            return false;
        for (int k = 1; k < 4; k++)
            if (!tabs->planes[j.ref_plane + k].origin || tabs->planes[j.ref_plane + k].stride != rp.stride ||
                tabs->span_first[j.ref_plane + k] != tabs->span_first[j.ref_plane] || tabs->span_end[j.ref_plane + k] != tabs->span_end[j.ref_plane])
                return false;
    }
    if (j.qp >= X265_HIP_MAX_QP || !tabs->costtab[j.qp])
        return false;
    if (j.mvmin_x > j.mvmax_x || j.mvmin_y > j.mvmax_y)
        return false;
    // reach, as check_job: the window and the zero vector, the steps a search may take outside them, the sub-pel
    // stage's moves; then the filters' 3 before / 4 behind and the loads' overhang (check_reach)
    const int int_over = method == X265_HIP_DIA ? 1 : method == X265_HIP_HEX ? 3 : method == X265_HIP_UMH ? 4 : 0;
    const int qsteps = (j.method & X265_HIP_LOWRES_REF) ? 3 : (int)((0x65365332u >> (4 * j.subme)) & 15); // {2,3,3,5,6,3,5,6}
    const long over = int_over + (qsteps + 3) / 4;
    const long lox = min((long)j.mvmin_x, 0L) - over, hix = max((long)j.mvmax_x, 0L) + over;
    const long loy = min((long)j.mvmin_y, 0L) - over, hiy = max((long)j.mvmax_y, 0L) + over;
    {
        const long first = (long)j.pu_y * fp.stride + j.pu_x, last = (long)(j.pu_y + j.pu_h - 1) * fp.stride + j.pu_x + j.pu_w;
        if (first < tabs->span_first[j.fenc_plane] || last > tabs->span_end[j.fenc_plane] || j.pu_x + j.pu_w > fp.stride)
            return false;
    }
    const long x0 = j.pu_x + lox, y0 = j.pu_y + loy, x1 = j.pu_x + hix + j.pu_w, y1 = j.pu_y + hiy + j.pu_h;
    const long first = (y0 - 3) * rp.stride + (x0 - 3 - 3), last = (y1 + 4 - 1) * rp.stride + (x1 + 4 + 12);
    return first >= tabs->span_first[j.ref_plane] && last <= tabs->span_end[j.ref_plane] + 16;
}

__global__ __launch_bounds__(256) void me_route_kernel(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, uint32_t *route, int parity, int cap,
                                                       int team_on, uint32_t *guard, x265_hip_me_result *out, int out_stride)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < ME_ROUTE_SET; k += 256)
            route[(parity ^ 1) * ME_ROUTE_SET + k] = 0u;
    int cls = -1;
    if (i < n)
    {
        if (guard && !guard_me_job(tabs, jobs[i]))
        {
            atomicAdd(&guard[0], 1u);
            atomicMin(&guard[1], (uint32_t)i);
            x265_hip_me_result *o = (x265_hip_me_result *)((char *)out + (size_t)i * out_stride);
            o->cost = -1;
            o->mvx = o->mvy = 0;
        }
        else
        {
            cls = me_job_class(jobs[i].pu_w, jobs[i].pu_h, jobs[i].method);
            if ((!team_on && cls >= MC_T4) || (team_on == 2 && cls == MC_T32))
                cls = MC_REST;
        }
    }
    const int p = (i >> 6) & (ME_PARTS - 1);
#pragma unroll
    for (int k = 0; k < ME_CLASSES; k++)
    {
        const unsigned long long m = __ballot(cls == k);
        if (!m)
            continue;
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0)
            base = atomicAdd(route_append(route, parity, k, p), (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (cls == k)
        {
            const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (slot < (uint32_t)cap) // always, with counters that start at zero; never write outside the list
                const_cast<uint32_t *>(route_list(route, cap, k, p))[slot] = (uint32_t)i;
        }
    }
}

#ifndef ME_BIG_WAVES
#define ME_BIG_WAVES 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ME_BIG_WAVES))) void me_big_kernel(const DeviceTables *tabs, const x265_hip_me_job *jobs,
                                                uint32_t *route, int parity, int cap, x265_hip_me_result *out, int out_stride)
{
    __shared__ WaveLds lds;
    me_list_loop<true>(lds, tabs, jobs, route, parity, cap, out, out_stride);
}

// The steps of predInterSearch in front of motionEstimate (reference Lib/TLibEncoder/TEncSearch.cpp:2315-2339): the AMVP
// candidates are clipped and the block predicted at each of them from the reconstructed reference is compared with the
// source block by SAD (the pixels of xPredInterLumaBlk are subpelCompare's); the first minimum is the predictor;
// xSetSearchRange gives the window.  Three launches: a thread per job writes the two candidates as cost requests,
// cost_kernel measures them (its items are dealt over the lanes whatever the blocks' sizes; round 2 spent a whole wave on
// the two candidates of a job: 0.55 ms for a 4K picture's 516 k iterations, more than half of what the searches take),
// a thread per job picks, computes the window and writes the motionEstimate job and the front of the result record.
__global__ __launch_bounds__(256) void inter_requests_kernel(const x265_hip_inter_job *jobs, int n, int pic_w, int pic_h, x265_hip_cost_req *reqs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const x265_hip_inter_job &j = jobs[i];
#pragma unroll
    for (int k = 0; k < 2; k++)
    {
        const int kk = k < (int)j.num_amvp ? k : 0; // a job with one candidate measures it twice (the second cost is not used)
        int qx = j.amvp[2 * kk], qy = j.amvp[2 * kk + 1];
        clip_mv(qx, qy, j.cu_x, j.cu_y, pic_w, pic_h);
        x265_hip_cost_req r;
        r.fenc_plane = j.fenc_plane;
        r.ref_plane = j.pred_plane;
        r.pu_x = j.pu_x;
        r.pu_y = j.pu_y;
        r.pu_w = j.pu_w;
        r.pu_h = j.pu_h;
        r.metric = X265_HIP_SAD;
        r.pad = 0;
        r.qmv_x = (int16_t)qx;
        r.qmv_y = (int16_t)qy;
        reqs[2 * i + k] = r;
    }
}

__global__ __launch_bounds__(256) void inter_finish_kernel(const x265_hip_inter_job *jobs, int n, int pic_w, int pic_h, const int32_t *sad,
                                                           x265_hip_me_job *mejobs, x265_hip_inter_result *out)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n)
        return;
    const x265_hip_inter_job &j = jobs[id];
    const int c0 = sad[2 * id], c1 = j.num_amvp > 1 ? sad[2 * id + 1] : 0x7fffffff;
    const int pick = c1 < c0 ? 1 : 0; // "if (bestCost > cost)": the first minimum
    const int mvpx = j.amvp[2 * pick], mvpy = j.amvp[2 * pick + 1];
    int lox, loy, hix, hiy;
    search_window(mvpx, mvpy, j.merange, j.ref_lag, j.cu_x, j.cu_y, pic_w, pic_h, lox, loy, hix, hiy);
    x265_hip_me_job m;
    m.fenc_plane = j.fenc_plane;
    m.ref_plane = j.ref_plane;
    m.pu_x = j.pu_x;
    m.pu_y = j.pu_y;
    m.pu_w = j.pu_w;
    m.pu_h = j.pu_h;
    m.method = j.method;
    m.subme = j.subme;
    m.mvmin_x = (int16_t)lox;
    m.mvmin_y = (int16_t)loy;
    m.mvmax_x = (int16_t)hix;
    m.mvmax_y = (int16_t)hiy;
    m.qmvp_x = (int16_t)mvpx;
    m.qmvp_y = (int16_t)mvpy;
    m.merange = j.merange;
    m.qp = j.qp;
    m.num_mvc = j.num_mvc;
    for (int k = 0; k < 2 * X265_HIP_MAX_MVC; k++)
        m.mvc[k] = j.mvc[k];
    mejobs[id] = m;
    x265_hip_inter_result &r = out[id];
    r.amvp_sad[0] = c0;
    r.amvp_sad[1] = c1;
    r.mvmin_x = (int16_t)lox;
    r.mvmin_y = (int16_t)loy;
    r.mvmax_x = (int16_t)hix;
    r.mvmax_y = (int16_t)hiy;
    r.mvp_x = (int16_t)mvpx;
    r.mvp_y = (int16_t)mvpy;
    r.mvp_idx = (uint8_t)pick;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
}

// ---- whole-frame lookahead search (x265_hip_lookahead_batch) ----------------------------------------------------------
// CostEstimate::estimateFrameCost's motion search of one (picture, reference) pair: every 8x8 CU of the half-resolution
// picture, in reverse raster order, with the predictor and the candidates taken from the CUs already estimated
// (EstimateRow::estimateCUCost, reference encoder/slicetype.cpp:1540-1582).  CU (x, y) needs (x + 1, y), (x, y + 1),
// (x - 1, y + 1) and (x + 1, y + 1): a row is a chain from right to left, and a row can follow the row below it two
// CUs behind -- the reference runs one pool job per row that way (CostEstimate::processRow, :1458-1499).  Here one
// wave takes a row.  Its own previous vector and the two vectors of the row below that it has already seen stay in
// registers; the only thing it waits for is the (cost, vector) word of CU (x - 1, y + 1), which the wave of the row
// below publishes with one 8-byte store.  That word is asked for one CU ahead (the load flies during the search), as
// is the next CU's source block, so a step of the chain is the latency of one search.
// Result words start as all ones (the launch clears them so): a cost of -1 reads "not there yet".
// Rows are handed out through one counter per search, bottom row first, so a wave only ever waits for waves that
// took their rows before it did and are therefore running or done: every wave reaches its exit.  A wave that polls
// LOOK_SPIN_LIMIT times in vain (never seen) raises status[0], fills the rest of its row with INT32_MAX and leaves.
#define LOOK_SPIN_LIMIT (1 << 22)

typedef unsigned long long look_word; // x265_hip_me_result as one word: cost in the low half, then mvx, mvy

__device__ __forceinline__ look_word look_load(const look_word *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int look_median(int a, int b, int c) // median(), slicetype.cpp:40-49
{
    return max(min(a, b), min(max(a, b), c));
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void lookahead_kernel(const DeviceTables *tabs, const x265_hip_lookahead_job *jobs,
                                                                                               int njobs, int W, int H, uint32_t *tickets,
                                                                                               look_word *results, int *status)
{
    __shared__ WaveLds lds;
    __shared__ x265_hip_me_job jl;
    const int job = blockIdx.x % njobs;
    const int lane = threadIdx.x & 63;
    int ticket = 0;
    if (lane == 0)
        ticket = (int)atomicAdd(&tickets[job * 16], 1u);
    ticket = uniform(ticket);
    if (ticket >= H)
        return;
    const int cuy = H - 1 - ticket;
    const x265_hip_lookahead_job &lj = jobs[job];
    Wave wv;
    wave_init(wv, &lds, 8, 8);
    const int patv = pat_vector();
    if (lane < PAT_N)
        lds.pat[lane] = patv; // (load_fenc's part; visible behind the wave_sync in front of the first search)
    const PlaneDesc fp = tabs->planes[lj.fenc_plane], rp = tabs->planes[lj.ref_plane];
    const uint8_t *const r1 = tabs->planes[lj.ref_plane + 1].origin, *const r2 = tabs->planes[lj.ref_plane + 2].origin,
                  *const r3 = tabs->planes[lj.ref_plane + 3].origin;
    wv.stride = rp.stride;
    wv.tab = tabs->costtab[lj.qp];
    wv.lowres = true;
    if (lane == 0)
    {
        jl.fenc_plane = lj.fenc_plane;
        jl.ref_plane = lj.ref_plane;
        jl.pu_w = jl.pu_h = 8;
        jl.method = lj.method | X265_HIP_LOWRES_REF;
        jl.subme = lj.subme;
        jl.merange = lj.merange;
        jl.qp = lj.qp;
        jl.pu_y = (int16_t)(8 * cuy);
        jl.mvmin_y = (int16_t)(-8 * cuy - 8);
        jl.mvmax_y = (int16_t)((H - cuy - 1) * 8 + 8);
    }
    look_word *const mine = results + ((size_t)job * H + cuy) * W;
    const look_word *const below = mine + W; // only read when cuy < H - 1
    const bool has_below = cuy < H - 1;
    int spins = 0;
    bool dead = false;
    // the vector of a CU of the row below, x | y << 16, once it is there
    auto wait_mv = [&](int cux, look_word v) {
        while ((int)(uint32_t)v == -1)
        {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > LOOK_SPIN_LIMIT || (!(spins & 1023) && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            {
                dead = true;
                return 0;
            }
            v = look_load(&below[cux]);
        }
        return (int)(uint32_t)(v >> 32);
    };
    // source block of CU cux: lanes 0..15 hold one dword each (row lane / 2, half lane & 1)
    const uint8_t *const frow = fp.origin + (long)(8 * cuy) * fp.stride;
    auto fetch_fenc = [&](int cux) {
        return lane < 16 ? ld4g(as_global(frow) + (lane >> 1) * fp.stride + 8 * cux + 4 * (lane & 1)) : 0u;
    };
    uint32_t fnext = fetch_fenc(W - 1);
    int right = 0, b_c = 0, b_r = 0, b_l = 0; // (x + 1, y), (x, y + 1), (x + 1, y + 1), (x - 1, y + 1)
    if (has_below)
    {
        b_c = wait_mv(W - 1, look_load(&below[W - 1]));
        if (W > 1 && !dead)
            b_l = wait_mv(W - 2, look_load(&below[W - 2]));
    }
    for (int cux = W - 1; cux >= 0 && !dead; cux--)
    {
        // next step's word of the row below and source block: in flight during this search
        const bool want_next = has_below && cux >= 2;
        look_word vnext = ~0ull;
        if (want_next)
            vnext = look_load(&below[cux - 2]);
        if (lane < 16)
            *(uint32_t *)(lds.fenc + (lane >> 1) * 64 + 4 * (lane & 1)) = fnext;
        if (cux > 0)
            fnext = fetch_fenc(cux - 1);
        // Reverse-order MV prediction (slicetype.cpp:1556-1580): mvc[0] and mvc[2] start as zero
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0, numc = 0;
        auto push = [&](int v) {
            if (numc == 0)
                c0 = v;
            else if (numc == 1)
                c1 = v;
            else if (numc == 2)
                c2 = v;
            else
                c3 = v;
            numc++;
        };
        if (cux < W - 1)
            push(right);
        if (has_below)
        {
            push(b_c);
            if (cux > 0)
                push(b_l);
            if (cux < W - 1)
                push(b_r);
        }
        int mvpx = (int16_t)c0, mvpy = c0 >> 16;
        if (numc > 1)
        {
            mvpx = look_median((int16_t)c0, (int16_t)c1, (int16_t)c2);
            mvpy = look_median(c0 >> 16, c1 >> 16, c2 >> 16);
        }
        if (lane == 0)
        {
            jl.pu_x = (int16_t)(8 * cux);
            jl.mvmin_x = (int16_t)(-8 * cux - 8);
            jl.mvmax_x = (int16_t)((W - cux - 1) * 8 + 8);
            jl.qmvp_x = (int16_t)mvpx;
            jl.qmvp_y = (int16_t)mvpy;
            jl.num_mvc = (uint8_t)numc;
            jl.mvc[0] = (int16_t)c0, jl.mvc[1] = (int16_t)(c0 >> 16);
            jl.mvc[2] = (int16_t)c1, jl.mvc[3] = (int16_t)(c1 >> 16);
            jl.mvc[4] = (int16_t)c2, jl.mvc[5] = (int16_t)(c2 >> 16);
            jl.mvc[6] = (int16_t)c3, jl.mvc[7] = (int16_t)(c3 >> 16);
        }
        const long off = (long)(8 * cuy) * rp.stride + 8 * cux;
        wv.ref = rp.origin + off;
        wv.lref1 = r1 + off;
        wv.lref2 = r2 + off;
        wv.lref3 = r3 + off;
        wv.mvpx = mvpx;
        wv.mvpy = mvpy;
        wave_sync();
        int cost, mx, my;
        if (lj.method <= X265_HIP_HEX)
            motion_estimate_uv<true, true>(wv, jl, cost, mx, my, patv);
        else
            motion_estimate_uv<false, true>(wv, jl, cost, mx, my, patv);
        right = (int)((uint32_t)(mx & 0xffff) | ((uint32_t)my << 16));
        if (lane == 0)
            __hip_atomic_store(&mine[cux], (look_word)(uint32_t)cost | ((look_word)(uint32_t)right << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b_r = b_c;
        b_c = b_l;
        if (want_next)
            b_l = wait_mv(cux - 2, vnext);
        wave_sync(); // the search's last LDS reads are over before the next block is written
    }
    if (dead && lane == 0)
    {
        __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int x = W - 1; x >= 0; x--)
            if ((int)(uint32_t)look_load(&mine[x]) == -1)
                __hip_atomic_store(&mine[x], (look_word)0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Cost requests in batches of `rb` (at most 32, see batch_requests) per wave.  Batches differ sevenfold in
// work when block sizes are mixed, so a fixed stride leaves SIMDs idle at the end; instead the batches are dealt in chunks of COST_CHUNK, round-robin,
// to up to COST_PARTS parts (so a list sorted by size loads every part alike), part p is served by the blocks
// p, p + parts, ..., and their waves take the part's batches in order from its counter in device memory (one
// counter for all waves would serialise: about 12 ns per atomic on one address).  The blocks of a part run on
// one XCD (blocks go round-robin over the eight, and parts is a multiple of eight whenever it matters), so the
// neighbouring requests of a chunk, which mostly read the same pixels, meet in one L2.  A wave fetches
// the next batch's request records while it computes the current one (the plane directory sits in LDS), so
// the only memory latency in front of a batch's arithmetic is that of its pixel loads.  Two sets of counters
// alternate between launches: a launch counts in set `parity` and clears the other one for its successor.
// Every wave leaves the loop as soon as its part's counter passes the part's last batch.
// Work items are (request, strip) pairs dealt round-robin to the 64 lanes, a strip being 4 pixels wide and 8
// tall (4 tall for block heights that are not multiples of 8), so requests of different sizes in one batch do
// not wait for each other: a prefix sum of the item counts sits in LDS and a lane finds its request by binary
// search (batches whose requests all have at most two items skip the search: lane = 2 * request + item).
// Lanes read source and reference pixels straight from the planes (a request is a single candidate, nothing
// is reused); partial sums meet in LDS atomics.
#define COST_BATCH 32
#ifndef COST_DIAG_UNIFY
#define COST_DIAG_UNIFY 1
#endif
#define COST_MAX_BLOCKS (256 * 6)
#define COST_PARTS 64 // counters are 64 bytes apart
#define COST_CHUNK 8

struct CostLds
{
    int pre[COST_BATCH + 1]; // exclusive prefix of item counts
    int cost[COST_BATCH];
    // per request, resolved once by lane k: block origins, strides,
    // ntx | xf << 8 | yf << 10 | metric << 12 | (8-row strips) << 14 | second fraction << 15 | sa8d kind << 19 |
    // units per row << 21 | INFO_LOWRES | INFO_MCAVG | INFO_PRED_BI | INFO_PRED_W
    const uint8_t *fenc[COST_BATCH];
    const uint8_t *ref[COST_BATCH];
    int fstride[COST_BATCH], rstride[COST_BATCH], info[COST_BATCH];
    // bidirectional requests: the second prediction
    const uint8_t *ref1[COST_BATCH];
    int rstride1[COST_BATCH];
};

// rounded average of two predictions, packed pairs: pixelavg_pp (reference common/pixel.cpp:546-560)
__device__ __forceinline__ uint32_t avg_pairs(uint32_t a, uint32_t b)
{
    return ((a + b + 0x00010001u) >> 1) & 0x01ff01ffu; // halves are <= 255 + 255 + 1: no carry across them
}

__device__ __forceinline__ int req_ref0(const x265_hip_cost_req &r) { return r.ref_plane; }
__device__ __forceinline__ int req_qx0(const x265_hip_cost_req &r) { return r.qmv_x; }
__device__ __forceinline__ int req_qy0(const x265_hip_cost_req &r) { return r.qmv_y; }
__device__ __forceinline__ int req_ref1(const x265_hip_cost_req &) { return 0; }
__device__ __forceinline__ int req_qx1(const x265_hip_cost_req &) { return 0; }
__device__ __forceinline__ int req_qy1(const x265_hip_cost_req &) { return 0; }
__device__ __forceinline__ int req_fenc(const x265_hip_cost_req &r) { return r.fenc_plane; }
__device__ __forceinline__ int req_fenc(const x265_hip_bidir_req &r) { return r.fenc_plane; }
__device__ __forceinline__ int req_metric(const x265_hip_cost_req &r) { return r.metric; }
__device__ __forceinline__ int req_metric(const x265_hip_bidir_req &r) { return r.metric; }
// prediction requests: the "source" plane is the destination, there is no metric
__device__ __forceinline__ int req_fenc(const x265_hip_pred_req &r) { return r.dst_plane; }
__device__ __forceinline__ int req_metric(const x265_hip_pred_req &) { return 0; }
__device__ __forceinline__ int req_ref0(const x265_hip_pred_req &r) { return r.ref0_plane; }
__device__ __forceinline__ int req_qx0(const x265_hip_pred_req &r) { return r.qmv0_x; }
__device__ __forceinline__ int req_qy0(const x265_hip_pred_req &r) { return r.qmv0_y; }
__device__ __forceinline__ int req_ref1(const x265_hip_pred_req &r) { return r.ref1_plane; }
__device__ __forceinline__ int req_qx1(const x265_hip_pred_req &r) { return r.qmv1_x; }
__device__ __forceinline__ int req_qy1(const x265_hip_pred_req &r) { return r.qmv1_y; }
// weighted uni-directional prediction: w0 | offset << 16 | log2 denominator << 24
__device__ __forceinline__ int req_weight(const x265_hip_pred_req &r)
{
    return (r.qmv1_x & 0xffff) | ((r.qmv1_y & 0xff) << 16) | ((r.log2_denom & 7) << 24);
}
__device__ __forceinline__ int req_weight(const x265_hip_cost_req &) { return 0; }
__device__ __forceinline__ int req_weight(const x265_hip_bidir_req &) { return 0; }
// weighted bi-prediction: w0 | w1 << 16 and (o0 + o1) | log2 denominator << 16
__device__ __forceinline__ bool req_weighted_bi(const x265_hip_pred_req &r) { return r.weighted_bi != 0; }
__device__ __forceinline__ int req_bi_weights(const x265_hip_pred_req &r) { return (r.wp_w0 & 0xffff) | ((int)r.wp_w1 << 16); }
__device__ __forceinline__ int req_bi_offset(const x265_hip_pred_req &r) { return ((r.wp_o0 + r.wp_o1) & 0xffff) | ((r.log2_denom & 7) << 16); }
__device__ __forceinline__ bool req_weighted_bi(const x265_hip_cost_req &) { return false; }
__device__ __forceinline__ bool req_weighted_bi(const x265_hip_bidir_req &) { return false; }
__device__ __forceinline__ int req_bi_weights(const x265_hip_cost_req &) { return 0; }
__device__ __forceinline__ int req_bi_weights(const x265_hip_bidir_req &) { return 0; }
__device__ __forceinline__ int req_bi_offset(const x265_hip_cost_req &) { return 0; }
__device__ __forceinline__ int req_bi_offset(const x265_hip_bidir_req &) { return 0; }

// weight_sp on a packed pair of 14-bit intermediates (reference common/pixel.cpp:492-508): the pair of pixels
__device__ __forceinline__ uint32_t weight_pair(uint32_t p, int wpack)
{
    const int w0 = (int16_t)(wpack & 0xffff), offset = (int8_t)((wpack >> 16) & 0xff), shift = ((wpack >> 24) & 7) + 6;
    const int round = 1 << (shift - 1);
    int lo = ((((int16_t)(p & 0xffff) + 8192) * w0 + round) >> shift) + offset;
    int hi = (((((int)p >> 16) + 8192) * w0 + round) >> shift) + offset;
    lo = min(max(lo, 0), 255);
    hi = min(max(hi, 0), 255);
    return (uint32_t)lo | ((uint32_t)hi << 16);
}
// TComWeightPrediction::addWeightBi on packed pairs of the two lists' 14-bit intermediates (reference
// TComWeightPrediction.cpp:46-49, 225-249): clip((w0 (P0 + 8192) + w1 (P1 + 8192) + round + (offset << (shift - 1))) >> shift)
__device__ __forceinline__ uint32_t weight_bi_pair(uint32_t a, uint32_t b, int wpack, int opack)
{
    const int w0 = (int16_t)(wpack & 0xffff), w1 = wpack >> 16, offset = (int16_t)(opack & 0xffff), shift = (opack >> 16) + 7;
    const int add = (1 + offset) * (1 << (shift - 1)) + 8192 * (w0 + w1);
    int lo = (w0 * (int16_t)(a & 0xffff) + w1 * (int16_t)(b & 0xffff) + add) >> shift;
    int hi = (w0 * ((int)a >> 16) + w1 * ((int)b >> 16) + add) >> shift;
    lo = min(max(lo, 0), 255);
    hi = min(max(hi, 0), 255);
    return (uint32_t)lo | ((uint32_t)hi << 16);
}
__device__ __forceinline__ int req_ref0(const x265_hip_bidir_req &r) { return r.ref0_plane; }
__device__ __forceinline__ int req_qx0(const x265_hip_bidir_req &r) { return r.qmv0_x; }
__device__ __forceinline__ int req_qy0(const x265_hip_bidir_req &r) { return r.qmv0_y; }
__device__ __forceinline__ int req_ref1(const x265_hip_bidir_req &r) { return r.ref1_plane; }
__device__ __forceinline__ int req_qx1(const x265_hip_bidir_req &r) { return r.qmv1_x; }
__device__ __forceinline__ int req_qy1(const x265_hip_bidir_req &r) { return r.qmv1_y; }

// MODE_BIDIR: requests are x265_hip_bidir_req; the block is compared with the average of two predictions.
// MODE_DIST: the metric may also be sa8d_inter or sse_pp (reference common/pixel.cpp:158-177, 290-376,
// 1117-1141).  sa8d rounds once per 8x8 or once per 16x16 block, so those blocks' raw sums meet in a per-request
// array of LDS accumulators (`units`) and are rounded when the batch is written out; a request's items start on
// an even slot so that the two halves of an 8x8 block sit in lanes 2j and 2j + 1 (sa8d_strip).
// MODE_PRED: requests are x265_hip_pred_req; nothing is compared, the predicted strips are stored into the
// destination plane (uni-directional: the pixels of the other modes; bi-directional: luma_addAvg of the two
// lists' 14-bit intermediates, INFO_PRED_BI, or with INFO_PRED_WBI their weighted sum, parameters in wbi[];
// weighted uni-directional: weight_sp of the 14-bit intermediate, INFO_PRED_W, parameters in rstride1[]).
// The modes are bits: bidir_cost_kernel is MODE_BIDIR | MODE_DIST (all four comparisons).  With INFO_MCAVG a
// bidirectional request combines its two predictions the way motion compensation does (14-bit intermediates +
// luma_addAvg, xPredInterBi) instead of pixelavg_pp: what xMergeEstimation measures on a bi-directional merge
// candidate (TEncSearch.cpp:2209-2210) and mode decision on a bi-predicted block.
// MODE_PRED | MODE_PRED_SATD: the predicted strips are not stored but compared (satd) with the block of the
// "destination" plane, which is the source picture here (x265_hip_mc_satd_batch: xMergeEstimation's measurement).
enum { MODE_COST = 0, MODE_BIDIR = 1, MODE_DIST = 2, MODE_PRED = 4, MODE_PRED_SATD = 8 };
enum { INFO_LOWRES = 1 << 24, INFO_MCAVG = 1 << 25, INFO_PRED_BI = 1 << 26, INFO_PRED_W = 1 << 27, INFO_PRED_WBI = 1 << 28 };
#define DIST_UNITS 16 // 64x64: sixteen 16x16 blocks; 32x24 / 24x32: twelve 8x8 blocks

template <int MODE, typename Req>
__device__ __forceinline__ void cost_body(const DeviceTables *tabs, const Req *reqs, int n, int32_t *out, uint32_t *counters, int parity, int rb)
{
    constexpr bool BIDIR = (MODE & MODE_BIDIR) != 0, DIST = (MODE & MODE_DIST) != 0, PRED = (MODE & MODE_PRED) != 0;
    constexpr bool PRED_SATD = (MODE & MODE_PRED_SATD) != 0;
    // A batch of 32 64x64 blocks is 64 passes of the item loop, and a list of such blocks has fewer batches than the
    // chip has waves: the work per batch is bounded instead.  Every wave looks at the same 64 requests spread over
    // the list and halves rb until a batch holds about 512 items (eight passes) by their mean.  The positions are
    // multiples of the golden ratio (mod 1): equally spaced ones alias with the period of a list built picture by
    // picture (measured: sixteen of them fell on 64x64 blocks of a mixed list).  The loads are issued first, so
    // that they fly while the plane directory is copied.
    int sample_items = 0;
    if (!PRED && rb > 1)
    {
        const int sl = threadIdx.x & 63;
        const int si = n >= 64 ? (int)(((unsigned long long)((unsigned)sl * 0x9E3779B1u) * (unsigned long long)n) >> 32) : sl;
        if (si < n)
        {
            const int w = reqs[si].pu_w, h = reqs[si].pu_h;
            sample_items = (w >> 2) * (h >> (2 + ((h & 7) == 0)));
        }
    }
    __shared__ CostLds lds[4];
    __shared__ int count_lds[4][COST_BATCH];                              // items of a request (pre[] holds the padded count)
    __shared__ int units_lds[DIST ? 4 : 1][COST_BATCH * DIST_UNITS];
    // Lowres references of bidirectional requests: half-pel planes 1..3 of both references at the displaced block
    __shared__ const uint8_t *low_lds[BIDIR ? 4 : 1][6][BIDIR ? COST_BATCH : 1];
    __shared__ int wbi_lds[PRED ? 4 : 1][2][PRED ? COST_BATCH : 1];      // weighted bi-prediction: weights, offset and shift
    __shared__ PlaneDesc planes[X265_HIP_MAX_PLANES];
    static_assert(sizeof(PlaneDesc) == 16, "plane directory copy below: one 16-byte entry per thread and round");
    for (int i = threadIdx.x; i < X265_HIP_MAX_PLANES; i += 256)
        ((uint4 *)planes)[i] = ((const uint4 *)tabs->planes)[i];
    __syncthreads();

    CostLds &L = lds[threadIdx.x >> 6];
    int *const count = count_lds[threadIdx.x >> 6];
    int *const units = units_lds[DIST ? threadIdx.x >> 6 : 0];
    auto &low = low_lds[BIDIR ? threadIdx.x >> 6 : 0];
    auto &wbi = wbi_lds[PRED ? threadIdx.x >> 6 : 0];
    const int lane = threadIdx.x & 63;
    if (!PRED && rb > 1)
    {
        const int cnt = n < 64 ? n : 64;
        // one block size throughout (by the sample): batches can be cut to exactly one pass; otherwise a batch
        // of a pass and a half leaves half of the second pass idle, so four passes are kept
        const bool one_size = __ballot(lane < cnt && sample_items != __shfl(sample_items, 0)) == 0;
        for (int o = 32; o; o >>= 1)
            sample_items += __shfl_xor(sample_items, o);
        while (rb > 1 && sample_items * rb > 512 * cnt)
            rb >>= 1;
        // and while the list has fewer than eight batches per wave slot of the chip, as long as a batch still fills
        // its passes of the item loop: 95 k 64x64 blocks in batches of 4 are 4.6 batches per wave slot, five rounds
        // of which the last is 60 % full
        while (rb > 1 && (n + rb - 1) / rb < 8 * 5120 && sample_items * (rb >> 1) >= (one_size ? 64 : 256) * cnt)
            rb >>= 1;
        rb = uniform(rb);
    }
    const int nbatch = (n + rb - 1) / rb;
    const int parts = min((int)gridDim.x, COST_PARTS), part = blockIdx.x % parts;
    const int nwaves = 4 * (((int)gridDim.x - part + parts - 1) / parts); // waves serving this part
    // batch number j of this part is batch_of(j); hi = the first j whose batch lies past the list for sure
    auto batch_of = [&](int j) { return ((j / COST_CHUNK) * parts + part) * COST_CHUNK + j % COST_CHUNK; };
    const int nchunk = (nbatch + COST_CHUNK - 1) / COST_CHUNK;
    const int hi = ((nchunk - part + parts - 1) / parts) * COST_CHUNK;
    uint32_t *const counter = counters + (parity * COST_PARTS + part) * 16;
    if (blockIdx.x == 0 && threadIdx.x < COST_PARTS)
        counters[((parity ^ 1) * COST_PARTS + threadIdx.x) * 16] = 0;
    int b = (blockIdx.x / parts) * 4 + (threadIdx.x >> 6);
    Req cur = {};
    if (lane < rb && b < hi && (long)batch_of(b) * rb + lane < n)
        cur = reqs[batch_of(b) * rb + lane];
    while (b < hi)
    {
        int nb = 0;
        if (lane == 0)
            nb = (int)atomicAdd(counter, 1u);
        nb = nwaves + uniform(nb);
        const int base = batch_of(b) * rb;
        const bool have = lane < rb && base + lane < n;
        int ni = 0, nunits = 0;
        if (have)
        {
            const PlaneDesc fp = planes[req_fenc(cur)], rp = planes[req_ref0(cur)];
            const int tall = (cur.pu_h & 7) == 0;
            const int qx = req_qx0(cur), qy = req_qy0(cur);
            ni = (cur.pu_w >> 2) * (cur.pu_h >> (2 + tall));
            L.fenc[lane] = fp.origin + (long)cur.pu_y * fp.stride + cur.pu_x;
            L.ref[lane] = rp.origin + (long)(cur.pu_y + (qy >> 2)) * rp.stride + cur.pu_x + (qx >> 2);
            L.fstride[lane] = fp.stride;
            L.rstride[lane] = rp.stride;
            const int rmetric = req_metric(cur) & 3;
            int info = (cur.pu_w >> 2) | ((qx & 3) << 8) | ((qy & 3) << 10) | (rmetric << 12) | (tall << 14);
            if (DIST && rmetric == X265_HIP_SA8D)
            {
                // the table of pixel.cpp:1117-1141: sa8d16<> / sa8d8<> / satd by the block's divisibility
                const int kind = ((cur.pu_w | cur.pu_h) & 15) == 0 ? 2 : ((cur.pu_w | cur.pu_h) & 7) == 0 ? 1 : 0;
                if (kind)
                {
                    const int nux = cur.pu_w >> (2 + kind);
                    nunits = nux * (cur.pu_h >> (2 + kind));
                    info |= (kind << 19) | (nux << 21);
                }
            }
            if (PRED && req_ref1(cur) == X265_HIP_PRED_WEIGHTED)
            {
                info |= INFO_PRED_W;
                L.rstride1[lane] = req_weight(cur);
            }
            if (BIDIR || (PRED && req_ref1(cur) >= 0))
            {
                if (PRED)
                    info |= INFO_PRED_BI;
                if (PRED && req_weighted_bi(cur))
                {
                    info |= INFO_PRED_WBI;
                    wbi[0][lane] = req_bi_weights(cur);
                    wbi[1][lane] = req_bi_offset(cur);
                }
                if (BIDIR && (req_metric(cur) & X265_HIP_MC_AVG))
                    info |= INFO_MCAVG;
                const PlaneDesc rp1 = planes[req_ref1(cur)];
                const int qx1 = req_qx1(cur), qy1 = req_qy1(cur);
                L.ref1[lane] = rp1.origin + (long)(cur.pu_y + (qy1 >> 2)) * rp1.stride + cur.pu_x + (qx1 >> 2);
                L.rstride1[lane] = rp1.stride;
                info |= ((qx1 & 3) << 15) | ((qy1 & 3) << 17);
                if (BIDIR && (req_metric(cur) & X265_HIP_LOWRES_REF))
                {
                    // the lookahead's bidirectional cost (slicetype.cpp:1585-1595): both references are Lowres
                    // (planes ref .. ref + 3, one stride each)
                    info |= INFO_LOWRES;
                    const long off0 = (long)(cur.pu_y + (qy >> 2)) * rp.stride + cur.pu_x + (qx >> 2);
                    const long off1 = (long)(cur.pu_y + (qy1 >> 2)) * rp1.stride + cur.pu_x + (qx1 >> 2);
                    for (int k = 1; k < 4; k++)
                    {
                        low[k - 1][lane] = planes[req_ref0(cur) + k].origin + off0;
                        low[k + 2][lane] = planes[req_ref1(cur) + k].origin + off1;
                    }
                }
            }
            L.info[lane] = info;
        }
        if (lane < COST_BATCH)
            count[lane] = ni;
        if (DIST)
        {
            ni = (ni + 1) & ~1;
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 z = { 0, 0, 0, 0 };
            static_assert(COST_BATCH * DIST_UNITS == 64 * 8, "two 16-byte stores per lane clear the accumulators");
            ((i32x4 *)units)[2 * lane] = z;
            ((i32x4 *)units)[2 * lane + 1] = z;
        }
        const bool small = __ballot(ni > 2) == 0;
        if (lane < COST_BATCH)
        {
            // inclusive scan: row_shr 1, 2, 4, 8 inside each 16-lane row (lanes shifted in read 0), then the
            // first row's total onto the second row
            int inc = ni;
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
            const int row0 = __builtin_amdgcn_readlane(inc, 15);
            if (lane >= 16)
                inc += row0;
            L.pre[lane + 1] = inc;
            L.cost[lane] = 0;
            if (lane == 0)
                L.pre[0] = 0;
        }
        // next batch's records: in flight during this batch's arithmetic
        {
            const int nid = batch_of(nb) * rb + lane;
            if (lane < rb && nb < hi && nid < n)
                cur = reqs[nid];
        }
        wave_sync();
        const int total = small ? 2 * rb : L.pre[COST_BATCH];
        for (int i = lane; i < total; i += 64)
        {
            int k, t;
            if (small)
            {
                k = i >> 1;
                t = i & 1;
            }
            else
            {
                k = 0;
#pragma unroll
                for (int step = 16; step > 0; step >>= 1)
                    if (L.pre[k + step] <= i)
                        k += step;
                t = i - L.pre[k];
            }
            if (t >= count[k]) // also skips requests past n (no items)
                continue;
            const int info = L.info[k], ntx = info & 255, fstride = L.fstride[k], rstride = L.rstride[k];
            const int xf = (info >> 8) & 3, yf = (info >> 10) & 3, metric = (info >> 12) & (DIST ? 3 : 1);
            const int ty = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)ntx)), tx = t - ty * ntx; // exact: t < 256, ntx <= 16
            if (PRED)
            {
                // store a strip of the prediction: NT tiles of 4 rows
                auto emit = [&](auto nt_tag) {
                    constexpr int NT = decltype(nt_tag)::value, R = 4 * NT;
                    uint8_t *dst = const_cast<uint8_t *>(L.fenc[k]) + R * ty * fstride + 4 * tx;
                    const uint8_t *r0 = L.ref[k] + R * ty * rstride + 4 * tx;
                    uint32_t pix[R][2];
                    if (info & INFO_PRED_BI)
                    {
                        const int rs1 = L.rstride1[k];
                        uint32_t pix1[R][2];
                        ref_strip<NT, true>(r0, rstride, xf, yf, pix);
                        ref_strip<NT, true>(L.ref1[k] + R * ty * rs1 + 4 * tx, rs1, (info >> 15) & 3, (info >> 17) & 3, pix1);
                        if (info & INFO_PRED_WBI)
                        {
                            const int wpack = wbi[0][k], opack = wbi[1][k];
#pragma unroll
                            for (int y = 0; y < R; y++)
                            {
                                pix[y][0] = weight_bi_pair(pix[y][0], pix1[y][0], wpack, opack);
                                pix[y][1] = weight_bi_pair(pix[y][1], pix1[y][1], wpack, opack);
                            }
                        }
                        else
                        {
#pragma unroll
                            for (int y = 0; y < R; y++)
                            {
                                pix[y][0] = add_avg_pair(pix[y][0], pix1[y][0]);
                                pix[y][1] = add_avg_pair(pix[y][1], pix1[y][1]);
                            }
                        }
                    }
                    else if (info & INFO_PRED_W)
                    {
                        const int wpack = L.rstride1[k];
                        ref_strip<NT, true>(r0, rstride, xf, yf, pix);
#pragma unroll
                        for (int y = 0; y < R; y++)
                        {
                            pix[y][0] = weight_pair(pix[y][0], wpack);
                            pix[y][1] = weight_pair(pix[y][1], wpack);
                        }
                    }
                    else
                        ref_strip<NT>(r0, rstride, xf, yf, pix);
                    if (PRED_SATD)
                    {
                        int v = 0;
#pragma unroll
                        for (int half = 0; half < NT; half++)
                            v += tile_metric<true>(dst + 4 * half * fstride, fstride, pix + 4 * half, 1);
                        atomicAdd(&L.cost[k], v);
                        return;
                    }
#pragma unroll
                    for (int y = 0; y < R; y++)
                    {
                        const uint32_t four = __builtin_amdgcn_perm(pix[y][1], pix[y][0], 0x06040200);
                        __builtin_memcpy(dst + y * fstride, &four, 4);
                    }
                };
                if ((info >> 14) & 1)
                    emit(std::integral_constant<int, 2>());
                else
                    emit(std::integral_constant<int, 1>());
                continue;
            }
            int v;
            if ((info >> 14) & 1)
            {
                const uint8_t *f = L.fenc[k] + 8 * ty * fstride + 4 * tx;
                uint32_t pix[8][2];
                if (BIDIR && (info & INFO_LOWRES))
                {
                    // lowresMC of both lists (lowres.h:53-73), two tiles each
                    const int rs1 = L.rstride1[k], xf1 = (info >> 15) & 3, yf1 = (info >> 17) & 3;
                    uint32_t pix1[8][2];
#pragma unroll
                    for (int half = 0; half < 2; half++)
                    {
                        lowres_tile(L.ref[k], low[0][k], low[1][k], low[2][k], rstride, xf, yf, (8 * ty + 4 * half) * rstride + 4 * tx,
                                    pix + 4 * half);
                        lowres_tile(L.ref1[k], low[3][k], low[4][k], low[5][k], rs1, xf1, yf1, (8 * ty + 4 * half) * rs1 + 4 * tx,
                                    pix1 + 4 * half);
                    }
#pragma unroll
                    for (int y = 0; y < 8; y++)
                    {
                        pix[y][0] = avg_pairs(pix[y][0], pix1[y][0]);
                        pix[y][1] = avg_pairs(pix[y][1], pix1[y][1]);
                    }
                }
                else if (BIDIR && (info & INFO_MCAVG))
                {
                    const int rs1 = L.rstride1[k];
                    uint32_t pix1[8][2];
                    ref_strip<2, true>(L.ref[k] + 8 * ty * rstride + 4 * tx, rstride, xf, yf, pix);
                    ref_strip<2, true>(L.ref1[k] + 8 * ty * rs1 + 4 * tx, rs1, (info >> 15) & 3, (info >> 17) & 3, pix1);
#pragma unroll
                    for (int y = 0; y < 8; y++)
                    {
                        pix[y][0] = add_avg_pair(pix[y][0], pix1[y][0]);
                        pix[y][1] = add_avg_pair(pix[y][1], pix1[y][1]);
                    }
                }
                else
                    // lanes of this pass that need the diagonal filter pull the half-fractional ones along (one path, not three)
                    ref_strip<2>(L.ref[k] + 8 * ty * rstride + 4 * tx, rstride, xf, yf, pix, COST_DIAG_UNIFY && __ballot(xf && yf) != 0);
                if (BIDIR && !(info & (INFO_LOWRES | INFO_MCAVG)))
                {
                    const int rs1 = L.rstride1[k];
                    uint32_t pix1[8][2];
                    ref_strip<2>(L.ref1[k] + 8 * ty * rs1 + 4 * tx, rs1, (info >> 15) & 3, (info >> 17) & 3, pix1);
#pragma unroll
                    for (int y = 0; y < 8; y++)
                    {
                        pix[y][0] = avg_pairs(pix[y][0], pix1[y][0]);
                        pix[y][1] = avg_pairs(pix[y][1], pix1[y][1]);
                    }
                }
                const int kind = DIST ? (info >> 19) & 3 : 0;
                if (DIST && kind)
                {
                    // unit = the 8x8 (kind 1) or 16x16 (kind 2) block this strip lies in
                    const int u = (ty >> (kind - 1)) * ((info >> 21) & 7) + (tx >> kind);
                    atomicAdd(&units[k * DIST_UNITS + u], sa8d_strip<true>(f, fstride, pix, tx & 1));
                    continue;
                }
                if (DIST && metric == X265_HIP_SSE)
                    v = sse_strip<8, true>(f, fstride, pix);
                else
                {
                    const int m = DIST ? metric != X265_HIP_SAD : metric; // sa8d_inter of the remaining blocks is satd
                    v = tile_metric<true>(f, fstride, pix, m) + tile_metric<true>(f + 4 * fstride, fstride, pix + 4, m);
                }
            }
            else
            {
                uint32_t pix[4][2];
                if (BIDIR && (info & INFO_MCAVG))
                {
                    const int rs1 = L.rstride1[k];
                    uint32_t pix1[4][2];
                    ref_strip<1, true>(L.ref[k] + 4 * ty * rstride + 4 * tx, rstride, xf, yf, pix);
                    ref_strip<1, true>(L.ref1[k] + 4 * ty * rs1 + 4 * tx, rs1, (info >> 15) & 3, (info >> 17) & 3, pix1);
#pragma unroll
                    for (int y = 0; y < 4; y++)
                    {
                        pix[y][0] = add_avg_pair(pix[y][0], pix1[y][0]);
                        pix[y][1] = add_avg_pair(pix[y][1], pix1[y][1]);
                    }
                }
                else
                    ref_strip<1>(L.ref[k] + 4 * ty * rstride + 4 * tx, rstride, xf, yf, pix, COST_DIAG_UNIFY && __ballot(xf && yf) != 0);
                if (BIDIR && !(info & INFO_MCAVG))
                {
                    const int rs1 = L.rstride1[k];
                    uint32_t pix1[4][2];
                    ref_strip<1>(L.ref1[k] + 4 * ty * rs1 + 4 * tx, rs1, (info >> 15) & 3, (info >> 17) & 3, pix1);
#pragma unroll
                    for (int y = 0; y < 4; y++)
                    {
                        pix[y][0] = avg_pairs(pix[y][0], pix1[y][0]);
                        pix[y][1] = avg_pairs(pix[y][1], pix1[y][1]);
                    }
                }
                const uint8_t *f = L.fenc[k] + 4 * ty * fstride + 4 * tx;
                if (DIST && metric == X265_HIP_SSE)
                    v = sse_strip<4, true>(f, fstride, pix);
                else
                    v = tile_metric<true>(f, fstride, pix, DIST ? metric != X265_HIP_SAD : metric);
            }
            atomicAdd(&L.cost[k], v);
        }
        wave_sync();
        if (have && (!PRED || PRED_SATD))
        {
            int c = L.cost[lane];
            if (DIST)
                for (int u = 0; u < nunits; u++)
                    c += (units[lane * DIST_UNITS + u] + 2) >> 2;
            out[base + lane] = c;
        }
        wave_sync(); // the next batch rewrites this wave's LDS record
        b = nb;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void cost_kernel(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n,
                                                                                           int32_t *out, uint32_t *counters, int parity, int rb)
{
    cost_body<MODE_COST>(tabs, reqs, n, out, counters, parity, rb);
}

__global__ __launch_bounds__(256) void bidir_cost_kernel(const DeviceTables *tabs, const x265_hip_bidir_req *reqs, int n, int32_t *out,
                                                         uint32_t *counters, int parity, int rb)
{
    cost_body<MODE_BIDIR | MODE_DIST>(tabs, reqs, n, out, counters, parity, rb);
}

// motion-compensated prediction into a destination plane (x265_hip_predict_batch)
__global__ __launch_bounds__(256) void predict_kernel(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, uint32_t *counters, int parity, int rb)
{
    cost_body<MODE_PRED>(tabs, reqs, n, nullptr, counters, parity, rb);
}

// satd of the source block against the motion-compensated prediction (x265_hip_mc_satd_batch)
__global__ __launch_bounds__(256) void mc_satd_kernel(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, int32_t *out,
                                                      uint32_t *counters, int parity, int rb)
{
    cost_body<MODE_PRED | MODE_PRED_SATD>(tabs, reqs, n, out, counters, parity, rb);
}

// cost requests whose metric may be any of sad, satd, sa8d_inter, sse_pp (x265_hip_dist_batch)
__global__ __launch_bounds__(256) void dist_kernel(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int32_t *out,
                                                   uint32_t *counters, int parity, int rb)
{
    cost_body<MODE_DIST>(tabs, reqs, n, out, counters, parity, rb);
}

// Requests per batch: 32 for long lists; short lists get smaller batches so that there are still a couple of
// thousand of them (a 1080p picture tiled by 64x64 blocks is 510 requests: sixteen waves' worth at 32 per batch,
// while a batch of one such block is already two full passes of a wave).
static int batch_requests(int n)
{
    int rb = COST_BATCH;
    while (rb > 1 && (n + rb - 1) / rb < 2048)
        rb >>= 1;
    return rb;
}

// workgroups for a list: the kernel may lower rb down to 1 for lists of large blocks (cost_body), so enough waves
// for that (the launch caps the number at COST_MAX_BLOCKS)
static int cost_blocks(int n, int)
{
    return (n + 3) / 4;
}

size_t me_route_words(int n) { return (size_t)ME_ROUTE_HEAD + (size_t)ME_CLASSES * ME_PARTS * me_route_cap(n > 0 ? n : 0); }
size_t me_route_counter_words() { return ME_ROUTE_HEAD; }
bool me_job_is_routed(int w, int h, int method) { return me_job_class(w, h, method) != MC_REST; }

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// waves per part for a list kernel that can hold `resident` waves on the chip, for a launch of n jobs (a list of a
// routed launch is at most the whole list)
static int list_grid(int n, int per_wave, int resident)
{
    const int want = (n + ME_PARTS * per_wave - 1) / (ME_PARTS * per_wave);
    const int most = resident / ME_PARTS;
    return ME_PARTS * (want < 1 ? 1 : want > most ? most : want);
}

#ifndef ME_T32_MIN_JOBS
#define ME_T32_MIN_JOBS 262144
#endif

static void me_launch_both(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, x265_hip_me_result *out, int out_stride, const MeRoute &r)
{
    if (n <= 0)
        return;
    if (!r.route) // the caller knows that the list holds nothing but MC_REST jobs
    {
        hipLaunchKernelGGL(me_kernel, dim3(n), dim3(64), 0, r.stream, tabs, jobs, n, out, out_stride);
        return;
    }
    const int cap = me_route_cap(n), parity = r.parity & 1;
    // experiments: 0 sends the small searches to me_rest_kernel, 2 those of MC_T32, 2 those of MC_T32 whatever the length of the list
    static const int team_env = env_int("X265_HIP_ME_TEAM", 1);
    // Four 32x32 calls share a wave in me_team_kernel and take as long as the longest of them: worth it when there are
    // waves enough to fill the chip all the same (a 4K picture's list: 0.956 against 0.971 ms), not for a short list
    // (a 1080p picture's 128,700 calls: 0.299 against 0.283 ms; profiles/r03_team_order.txt)
    const int team_on = team_env == 1 && n < ME_T32_MIN_JOBS ? 2 : team_env;
    // The lists first (made beside the searches on a side stream the kernel had to find room among their workgroups and
    // sometimes ended with them: nothing ran side by side), then the big jobs on the side stream and the other jobs on
    // the main one: the small jobs fill what the big ones leave idle (their last round, or most of the chip when
    // there are few of them).
    hipLaunchKernelGGL(me_route_kernel, dim3((n + 255) / 256), dim3(256), 0, r.stream, tabs, jobs, n, r.route, parity, cap, team_on, r.guard, out, out_stride);
    hipStream_t sb = r.side ? r.side : r.stream;
    if (r.side)
    {
        (void)hipEventRecord(r.fork, r.stream);
        (void)hipStreamWaitEvent(r.side, r.fork, 0);
    }
    static const int big_grid = [] {
        const int g = env_int("X265_HIP_ME_BIG_GRID", 0); // experiments; a multiple of ME_PARTS
        return g >= ME_PARTS ? g / ME_PARTS * ME_PARTS : 256 * 4 * ME_BIG_WAVES;
    }();
    const int bg = list_grid(n, 1, big_grid);
    hipLaunchKernelGGL(me_big_kernel, dim3(bg), dim3(64), 0, sb, tabs, jobs, r.route, parity, cap, out, out_stride);
    if (team_on)
        launch_me_team(tabs, jobs, n, r.route, parity, cap, out, out_stride, r.stream, team_on != 2);
    hipLaunchKernelGGL(me_rest_kernel, dim3(list_grid(n, 1, 256 * 4 * ME_REST_WAVES)), dim3(64), 0, r.stream, tabs, jobs, r.route, parity, cap, out, out_stride);
    if (r.side)
    {
        (void)hipEventRecord(r.join, r.side);
        (void)hipStreamWaitEvent(r.stream, r.join, 0);
    }
}

void launch_me(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, x265_hip_me_result *out, const MeRoute &r)
{
    me_launch_both(tabs, jobs, n, out, (int)sizeof(x265_hip_me_result), r);
}

void launch_cost(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st);

void launch_inter_search(const DeviceTables *tabs, const x265_hip_inter_job *jobs, int n, int pic_w, int pic_h, x265_hip_me_job *mejobs,
                         x265_hip_inter_result *out, const MeRoute &r, uint32_t *counters, int parity)
{
    hipStream_t st = r.stream;
    if (n <= 0)
        return;
    // behind the n motionEstimate jobs `mejobs` has room for the candidates' 2 n cost requests and their 2 n costs
    x265_hip_cost_req *reqs = (x265_hip_cost_req *)(mejobs + n);
    int32_t *sad = (int32_t *)(reqs + 2 * (size_t)n);
    hipLaunchKernelGGL(inter_requests_kernel, dim3((n + 255) / 256), dim3(256), 0, st, jobs, n, pic_w, pic_h, reqs);
    launch_cost(tabs, reqs, 2 * n, sad, counters, parity, st);
    hipLaunchKernelGGL(inter_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, jobs, n, pic_w, pic_h, sad, mejobs, out);
    // (cost, mv) head the result record
    static_assert(offsetof(x265_hip_inter_result, cost) == 0 && offsetof(x265_hip_inter_result, mvx) == 4, "x265_hip_me_result heads the record");
    me_launch_both(tabs, mejobs, n, (x265_hip_me_result *)out, (int)sizeof(x265_hip_inter_result), r);
}

void launch_lookahead(const DeviceTables *tabs, const x265_hip_lookahead_job *jobs, int njobs, int w_cu, int h_cu, uint32_t *tickets,
                      x265_hip_me_result *results, int *status, hipStream_t st)
{
    static_assert(sizeof(x265_hip_me_result) == sizeof(look_word) && offsetof(x265_hip_me_result, cost) == 0 &&
                      offsetof(x265_hip_me_result, mvx) == 4 && offsetof(x265_hip_me_result, mvy) == 6,
                  "a result is published as one 8-byte word");
    if (njobs > 0)
        hipLaunchKernelGGL(lookahead_kernel, dim3(njobs * h_cu), dim3(64), 0, st, tabs, jobs, njobs, w_cu, h_cu, tickets, (look_word *)results, status);
}

void launch_bidir_cost(const DeviceTables *tabs, const x265_hip_bidir_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st)
{
    if (n > 0)
    {
        const int rb = batch_requests(n), blocks = cost_blocks(n, rb);
        hipLaunchKernelGGL(bidir_cost_kernel, dim3(blocks < COST_MAX_BLOCKS ? blocks : COST_MAX_BLOCKS), dim3(256), 0, st, tabs, reqs, n, out, counters, parity, rb);
    }
}

void launch_predict(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, uint32_t *counters, int parity, hipStream_t st)
{
    if (n > 0)
    {
        const int rb = batch_requests(n), blocks = ((n + rb - 1) / rb + 3) / 4;
        hipLaunchKernelGGL(predict_kernel, dim3(blocks < COST_MAX_BLOCKS ? blocks : COST_MAX_BLOCKS), dim3(256), 0, st, tabs, reqs, n, counters, parity, rb);
    }
}

void launch_mc_satd(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st)
{
    if (n > 0)
    {
        const int rb = batch_requests(n), blocks = ((n + rb - 1) / rb + 3) / 4;
        hipLaunchKernelGGL(mc_satd_kernel, dim3(blocks < COST_MAX_BLOCKS ? blocks : COST_MAX_BLOCKS), dim3(256), 0, st, tabs, reqs, n, out, counters, parity, rb);
    }
}

void launch_dist(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st)
{
    if (n > 0)
    {
        const int rb = batch_requests(n), blocks = cost_blocks(n, rb);
        hipLaunchKernelGGL(dist_kernel, dim3(blocks < COST_MAX_BLOCKS ? blocks : COST_MAX_BLOCKS), dim3(256), 0, st, tabs, reqs, n, out, counters, parity, rb);
    }
}

void launch_cost(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st)
{
    if (n > 0)
    {
        const int rb = batch_requests(n), blocks = cost_blocks(n, rb);
        hipLaunchKernelGGL(cost_kernel, dim3(blocks < COST_MAX_BLOCKS ? blocks : COST_MAX_BLOCKS), dim3(256), 0, st, tabs, reqs, n, out, counters, parity, rb);
    }
}

} // namespace x265hip
