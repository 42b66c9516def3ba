// kernels.hip -- HIP kernels of the ME cost path for gfx950 (wave64).
//
//   cost_kernel        one wave per (block, reference, quarter-pel MV, metric) request
//   me_kernel          one wave per MotionEstimate::motionEstimate call (all five search methods,
//                      sub-pel levels 0..7)            reference: encoder/motion.cpp:526-1164
//   fullsearch_kernel  one workgroup per 64x64 CTU: exhaustive integer search of its 85 square PUs
//   prim_* kernels     single primitive calls on raw device pointers (the function-pointer table)
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

// ---------------------------------------------------------------------------------------------
// per-wave LDS record: the PU's source pixels at stride 64 (setSourcePU, motion.cpp:110-126) and
// a queue of up to QMAX candidates that are evaluated together

#define QMAX 16

struct WaveLds
{
    uint8_t fenc[64 * 64];
    int qx[QMAX]; // quarter-pel position
    int qy[QMAX];
    int cost[QMAX];
    int aux[QMAX];
};

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
    // after q_eval: lane k holds candidate k's cost / position / aux word
    int vc, vx, vy, va;
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
}

__device__ __forceinline__ void load_fenc(Wave &wv, const uint8_t *src, int sstride)
{
    const int ndw = wv.ntx * wv.h;
    for (int i = wv.lane; i < ndw; i += 64)
    {
        int y = i / wv.ntx, x = i - y * wv.ntx;
        *(uint32_t *)(wv.s->fenc + y * 64 + 4 * x) = ld4u(src + y * sstride + 4 * x);
    }
    wave_sync();
}

// how q_eval seeds a candidate's cost with its motion-vector rate
enum { MV_NONE = 0,  // distortion only
       MV_COST = 1,  // BitCost::mvcost(): sum of the two entries truncated to uint16 (bitcost.h:45)
       MV_RAW = 2 }; // plain int sum of the two entries (UMH hexagon grid, motion.cpp:832)

// queue values are wave-uniform: every lane stores the same word
__device__ __forceinline__ void q_push(Wave &wv, int qx, int qy, int mvmode, int aux = 0)
{
    wv.s->qx[wv.nq] = qx;
    wv.s->qy[wv.nq] = qy;
    wv.s->aux[wv.nq] = (aux << 2) | mvmode;
    wv.nq++;
}

// Evaluate every queued candidate: cost[c] = mv rate (per its mode) + SAD or SATD of the PU against
// the reference at the candidate's quarter-pel position (subpelCompare + cmp, motion.cpp:1126-1164).
// Work items are (candidate, 4x4 tile) pairs dealt round-robin to the lanes; partial sums meet in
// LDS.  Out of line and by value on purpose: there are ~25 call sites, and arguments by value keep
// the caller's search state in registers.
__device__ __noinline__ int q_eval_core(WaveLds *s, const uint8_t *ref, int stride, int ntx, int nt, const uint16_t *tab,
                                        int mvpx, int mvpy, int nq, int metric)
{
    const int lane = threadIdx.x & 63;
    // lane c fetches candidate c's mv rate; the table reads stay in flight under the pixel work
    int seed = 0;
    if (lane < nq)
    {
        const int mode = s->aux[lane] & 3;
        if (mode != MV_NONE)
        {
            const int a = tab_at(tab, mvpx, s->qx[lane]), b = tab_at(tab, mvpy, s->qy[lane]);
            seed = mode == MV_COST ? (int)(uint16_t)(a + b) : a + b;
        }
    }
    if (lane < QMAX)
        s->cost[lane] = 0;
    wave_sync();
    const int total = nq * nt;
    // i / nt and t / ntx by reciprocal: exact for i < 4096, nt <= 256 (error 4e-4 << 0.5 / nt)
    const float inv_nt = 1.0f / (float)nt, inv_ntx = 1.0f / (float)ntx;
    for (int i = lane; i < total; i += 64)
    {
        const int c = (int)(((float)i + 0.5f) * inv_nt), t = i - c * nt;
        const int ty = (int)(((float)t + 0.5f) * inv_ntx), tx = t - ty * ntx;
        const int qx = s->qx[c], qy = s->qy[c];
        int pix[4][4];
        ref_tile(ref + ((qy >> 2) + 4 * ty) * stride + (qx >> 2) + 4 * tx, stride, qx & 3, qy & 3, pix);
        const int v = tile_metric(s->fenc + ty * 4 * 64 + tx * 4, 64, pix, metric);
        atomicAdd(&s->cost[c], v);
    }
    wave_sync();
    return lane < nq ? s->cost[lane] + seed : 0x7fffffff;
}

__device__ __forceinline__ void q_eval(Wave &wv, int metric)
{
    wv.vc = q_eval_core(wv.s, wv.ref, wv.stride, wv.ntx, wv.nt, wv.tab, wv.mvpx, wv.mvpy, wv.nq, metric);
    const int l = wv.lane & (QMAX - 1);
    wv.vx = wv.s->qx[l];
    wv.vy = wv.s->qy[l];
    wv.va = wv.s->aux[l];
}

// k is wave-uniform at every call site
__device__ __forceinline__ int q_cost(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.vc, k); }
__device__ __forceinline__ int q_x(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.vx, k); }
__device__ __forceinline__ int q_y(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.vy, k); }
__device__ __forceinline__ int q_aux(const Wave &wv, int k) { return __builtin_amdgcn_readlane(wv.va, k) >> 2; }

__device__ __forceinline__ int s16(int v) { return (int)(int16_t)v; }

__device__ __forceinline__ int mvc_q(const Wave &wv, int qx, int qy)
{
    return uniform(mvcost_q(wv.tab, wv.mvpx, wv.mvpy, qx, qy));
}

// mvcost of a full-pel vector: the reference shifts an int16 pair left by two (mv.h:63)
__device__ __forceinline__ int mvc_f(const Wave &wv, int mx, int my)
{
    return mvc_q(wv, s16(mx << 2), s16(my << 2));
}

// ---------------------------------------------------------------------------------------------
// integer-stage search state (all wave-uniform)

struct Srch
{
    int lox, loy, hix, hiy; // full-pel range, inclusive
    int bcost, bx, by;      // best so far, full-pel
    int point, dist;        // STAR bookkeeping
};

__device__ __forceinline__ bool in_range(const Srch &s, int x, int y)
{
    return x >= s.lox && x <= s.hix && y >= s.loy && y <= s.hiy;
}

// queue a full-pel point; aux carries (point << 16) | dist for the STAR pattern
__device__ __forceinline__ void push_f(Wave &wv, int mx, int my, int point = 0, int dist = 0)
{
    // mvcost(mv << 2): the shift is an int16 operation in the reference (mv.h:63)
    q_push(wv, s16(s16(mx) << 2), s16(s16(my) << 2), MV_COST, (point << 16) | dist);
}

// Evaluate the queued full-pel points with SAD + mvcost and apply the strict-less updates in push
// order (COST_MV / COST_MV_X4 / COST_MV_PT_DIST*, motion.cpp:165-235).  Empties the queue.
template <bool STAR>
__device__ __forceinline__ void flush_f(Wave &wv, Srch &s)
{
    if (wv.nq)
        q_eval(wv, X265_HIP_SAD);
    for (int k = 0; k < wv.nq; k++)
    {
        int mx = q_x(wv, k) >> 2, my = q_y(wv, k) >> 2;
        int cost = q_cost(wv, k);
        if (cost < s.bcost)
        {
            s.bcost = cost;
            s.bx = mx;
            s.by = my;
            if (STAR)
            {
                int a = q_aux(wv, k);
                s.point = a >> 16;
                s.dist = a & 0xffff;
            }
        }
    }
    wv.nq = 0;
}

// queue one full-pel point; callers flush before the queue (QMAX entries) can overflow
template <bool STAR>
__device__ __forceinline__ void try_f(Wave &wv, Srch &s, int mx, int my, int point = 0, int dist = 0)
{
    (void)s;
    push_f(wv, mx, my, point, dist);
}

__device__ __constant__ const int8_t k_hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
__device__ __constant__ const uint8_t k_mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
__device__ __constant__ const int8_t k_square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
__device__ __constant__ const int8_t k_hex4[16][2] = { { 0, -4 }, { 0, 4 }, { -2, -3 }, { 2, -3 }, { -4, -2 }, { 4, -2 }, { -4, -1 }, { 4, -1 },
                                                      { -4, 0 }, { 4, 0 }, { -4, 1 }, { 4, 1 }, { -4, 2 }, { 4, 2 }, { -2, 3 }, { 2, 3 } };
__device__ __constant__ const int8_t k_twopoint[16][2] = { { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 },
                                                          { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 } };
// {hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd}, motion.cpp:47-57
__device__ __constant__ const uint8_t k_workload[8][5] = { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
                                                          { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };
__device__ __constant__ const uint8_t k_range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };

// StarPatternSearch, motion.cpp:284-524.  The reference's sad_x4 fast path and its guarded scalar
// path visit the same points in the same order, so the guarded form covers both.
__device__ __forceinline__ void star_pattern(Wave &wv, Srch &s, int early_exit_iters, int merange)
{
    const int ox = s.bx, oy = s.by;
    int saved = s.bcost;
    int rounds = 0;

    for (int dist = 1; dist <= 8; dist <<= 1)
    {
        const int half = dist >> 1;
        const int top = s16(oy - dist), bottom = s16(oy + dist), left = s16(ox - dist), right = s16(ox + dist);
        const int top2 = s16(oy - half), bottom2 = s16(oy + half), left2 = s16(ox - half), right2 = s16(ox + half);
        saved = s.bcost;
        if (top >= s.loy)
            try_f<true>(wv, s, ox, top, 2, dist);
        if (dist > 1 && top2 >= s.loy)
        {
            if (left2 >= s.lox)
                try_f<true>(wv, s, left2, top2, 1, half);
            if (right2 <= s.hix)
                try_f<true>(wv, s, right2, top2, 3, half);
        }
        if (left >= s.lox)
            try_f<true>(wv, s, left, oy, 4, dist);
        if (right <= s.hix)
            try_f<true>(wv, s, right, oy, 5, dist);
        if (dist > 1 && bottom2 <= s.hiy)
        {
            if (left2 >= s.lox)
                try_f<true>(wv, s, left2, bottom2, 6, half);
            if (right2 <= s.hix)
                try_f<true>(wv, s, right2, bottom2, 8, half);
        }
        if (bottom <= s.hiy)
            try_f<true>(wv, s, ox, bottom, 7, dist);
        flush_f<true>(wv, s);
        if (s.bcost < saved)
            rounds = 0;
        else if (++rounds >= early_exit_iters)
            return;
    }

    for (int dist = 16; dist <= s16(merange); dist <<= 1)
    {
        const int q = dist >> 2;
        const int top = s16(oy - dist), bottom = s16(oy + dist), left = s16(ox - dist), right = s16(ox + dist);
        saved = s.bcost;
        if (top >= s.loy)
            try_f<true>(wv, s, ox, top, 0, dist);
        if (left >= s.lox)
            try_f<true>(wv, s, left, oy, 0, dist);
        if (right <= s.hix)
            try_f<true>(wv, s, right, oy, 0, dist);
        if (bottom <= s.hiy)
            try_f<true>(wv, s, ox, bottom, 0, dist);
        for (int idx = 1; idx < 4; idx++)
        {
            const int yt = s16(top + q * idx), yb = s16(bottom - q * idx);
            const int xl = s16(ox - q * idx), xr = s16(ox + q * idx);
            if (yt >= s.loy)
            {
                if (xl >= s.lox)
                    try_f<true>(wv, s, xl, yt, 0, dist);
                if (xr <= s.hix)
                    try_f<true>(wv, s, xr, yt, 0, dist);
            }
            if (yb <= s.hiy)
            {
                if (xl >= s.lox)
                    try_f<true>(wv, s, xl, yb, 0, dist);
                if (xr <= s.hix)
                    try_f<true>(wv, s, xr, yb, 0, dist);
            }
        }
        flush_f<true>(wv, s);
        if (s.bcost < saved)
            rounds = 0;
        else if (++rounds >= early_exit_iters)
            return;
    }
}

// CROSS macro of the UMH search (motion.cpp:258-282) around (ox, oy)
__device__ __forceinline__ void umh_cross(Wave &wv, Srch &s, int ox, int oy, int start, int x_max, int y_max)
{
    int i = s16(start);
    if (x_max <= min(s.hix - ox, ox - s.lox))
        for (; i < x_max - 2; i = s16(i + 4))
        {
            if (wv.nq > QMAX - 4)
                flush_f<false>(wv, s);
            try_f<false>(wv, s, ox + i, oy);
            try_f<false>(wv, s, ox - i, oy);
            try_f<false>(wv, s, ox + i + 2, oy);
            try_f<false>(wv, s, ox - i - 2, oy);
        }
    for (; i < x_max; i = s16(i + 2))
    {
        if (wv.nq > QMAX - 2)
            flush_f<false>(wv, s);
        if (ox + i <= s.hix)
            try_f<false>(wv, s, ox + i, oy);
        if (ox - i >= s.lox)
            try_f<false>(wv, s, ox - i, oy);
    }
    i = s16(start);
    if (y_max <= min(s.hiy - oy, oy - s.loy))
        for (; i < y_max - 2; i = s16(i + 4))
        {
            if (wv.nq > QMAX - 4)
                flush_f<false>(wv, s);
            try_f<false>(wv, s, ox, oy + i);
            try_f<false>(wv, s, ox, oy - i);
            try_f<false>(wv, s, ox, oy + i + 2);
            try_f<false>(wv, s, ox, oy - i - 2);
        }
    for (; i < y_max; i = s16(i + 2))
    {
        if (wv.nq > QMAX - 2)
            flush_f<false>(wv, s);
        if (oy + i <= s.hiy)
            try_f<false>(wv, s, ox, oy + i);
        if (oy - i >= s.loy)
            try_f<false>(wv, s, ox, oy - i);
    }
    flush_f<false>(wv, s);
}

__device__ __forceinline__ void try4(Wave &wv, Srch &s, int ox, int oy, int ax, int ay, int bx, int by, int cx, int cy,
                                     int dx, int dy)
{
    try_f<false>(wv, s, ox + ax, oy + ay);
    try_f<false>(wv, s, ox + bx, oy + by);
    try_f<false>(wv, s, ox + cx, oy + cy);
    try_f<false>(wv, s, ox + dx, oy + dy);
}

// One position per lane: SAD of the whole PU at full-pel (mx, my), sources broadcast from LDS
__device__ __forceinline__ int lane_sad(const Wave &wv, int mx, int my)
{
    const uint8_t *r = wv.ref + my * wv.stride + mx;
    uint32_t acc = 0;
    for (int y = 0; y < wv.h; y++)
    {
        const uint8_t *rr = r + y * wv.stride;
        const uint8_t *ff = wv.s->fenc + y * 64;
        for (int x = 0; x < wv.ntx; x++)
            acc = __builtin_amdgcn_sad_u8(*(const uint32_t *)(ff + 4 * x), ld4u(rr + 4 * x), acc);
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

// Raster scan of the grid (x0 + step*ix, y0 + step*iy), row-major, first minimum wins: the FULL
// search (motion.cpp:997-1032, step 1) and the STAR raster refinement (motion.cpp:924-958, step 5).
// In the STAR form every fourth point of a complete group of four is charged mvcost(mv << 3), a
// quirk of the reference (motion.cpp:951) that is reproduced.
__device__ __forceinline__ void raster_scan(Wave &wv, Srch &s, int step, bool star_quirk)
{
    if (s.hix < s.lox || s.hiy < s.loy)
        return;
    const int nx = (s.hix - s.lox) / step + 1, ny = (s.hiy - s.loy) / step + 1;
    const int groups = (s.hix - s.lox - 3 * step >= 0) ? (s.hix - s.lox - 3 * step) / (4 * step) + 1 : 0;
    const int total = nx * ny;
    for (int base = 0; base < total; base += 64)
    {
        int idx = base + wv.lane;
        int cost = 0x7fffffff;
        if (idx < total)
        {
            int iy = idx / nx, ix = idx - iy * nx;
            int mx = s.lox + ix * step, my = s.loy + iy * step;
            int sh = (star_quirk && ix < 4 * groups && (ix & 3) == 3) ? 3 : 2;
            cost = lane_sad(wv, mx, my) + mvcost_q(wv.tab, wv.mvpx, wv.mvpy, s16(mx << sh), s16(my << sh));
        }
        int m = uniform(wave_min_i32(cost));
        if (m < s.bcost)
        {
            unsigned long long hit = __ballot(cost == m);
            int first = base + (int)__builtin_ctzll(hit);
            int iy = first / nx, ix = first - iy * nx;
            s.bcost = m;
            s.bx = s.lox + ix * step;
            s.by = s.loy + iy * step;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MotionEstimate::motionEstimate, motion.cpp:526-1124 (non-lowres branch), one wave per call

__device__ __forceinline__ void motion_estimate(Wave &wv, const x265_hip_me_job &j, x265_hip_me_result *out)
{
    Srch s;
    s.lox = j.mvmin_x;
    s.loy = j.mvmin_y;
    s.hix = j.mvmax_x;
    s.hiy = j.mvmax_y;
    s.point = s.dist = 0;
    const int qlox = s16(s.lox << 2), qloy = s16(s.loy << 2), qhix = s16(s.hix << 2), qhiy = s16(s.hiy << 2);
    int merange = j.merange;
    const int method = j.method;
    const int nmvc = j.num_mvc;

    // clipped quarter-pel predictor: SAD only, no mvcost (motion.cpp:552-559; MV::clipped mv.h:89-94)
    int pmx = min((int)j.qmvp_x, qhix), pmy = min((int)j.qmvp_y, qhiy);
    pmx = max(pmx, qlox);
    pmy = max(pmy, qloy);
    int bpx = pmx, bpy = pmy; // bestpre

    // first evaluation group: predictor, its full-pel rounding, the zero vector and the candidates
    // (motion.cpp:551-598).  The order in which results are consumed is the reference's.
    const int rfx = s16(s16(pmx + 2) >> 2), rfy = s16(s16(pmy + 2) >> 2); // roundToFPel, mv.h:76
    const bool sub = ((pmx | pmy) & 3) != 0;
    const bool nz = (pmx | pmy) != 0;
    wv.nq = 0;
    q_push(wv, pmx, pmy, MV_NONE);
    if (sub)
        q_push(wv, s16(rfx << 2), s16(rfy << 2), MV_COST);
    if (nz)
        q_push(wv, 0, 0, MV_COST);
    int first_mvc = wv.nq;
    for (int i = 0; i < nmvc; i++)
    {
        int mx = min((int)j.mvc[2 * i], qhix), my = min((int)j.mvc[2 * i + 1], qhiy);
        mx = max(mx, qlox);
        my = max(my, qloy);
        // zero and pmv are skipped here; equality with a changing bestpre is resolved below
        if ((mx | my) && !(mx == pmx && my == pmy))
            q_push(wv, mx, my, MV_COST);
    }
    q_eval(wv, X265_HIP_SAD);
    int bprecost;
    {
        int k = 0;
        bprecost = q_cost(wv, k++);
        s.bx = rfx;
        s.by = rfy;
        s.bcost = bprecost;
        if (sub)
            s.bcost = q_cost(wv, k++);
        if (nz)
        {
            int cost = q_cost(wv, k++);
            if (cost < s.bcost)
            {
                s.bcost = cost;
                s.bx = s.by = 0;
            }
        }
        for (k = first_mvc; k < wv.nq; k++)
        {
            int mx = q_x(wv, k), my = q_y(wv, k);
            if (mx == bpx && my == bpy) // "m != bestpre", motion.cpp:584
                continue;
            int cost = q_cost(wv, k);
            if (cost < bprecost)
            {
                bprecost = cost;
                bpx = mx;
                bpy = my;
            }
        }
        wv.nq = 0;
    }

    const int fpx = rfx, fpy = rfy; // pmv.roundToFPel(), motion.cpp:600
    int ox = s.bx, oy = s.by;

    bool do_hex = (method == X265_HIP_HEX);

    if (method == X265_HIP_DIA)
    {
        // radius-1 diamond; packed (cost << 4) + code, centre wins ties (motion.cpp:605-626)
        int i = merange;
        do
        {
            push_f(wv, s.bx, s.by - 1);
            push_f(wv, s.bx, s.by + 1);
            push_f(wv, s.bx - 1, s.by);
            push_f(wv, s.bx + 1, s.by);
            q_eval(wv, X265_HIP_SAD);
            int best = s.bcost << 4;
            int c0 = (q_cost(wv, 0) << 4) + 1;
            int c1 = (q_cost(wv, 1) << 4) + 3;
            int c2 = (q_cost(wv, 2) << 4) + 4;
            int c3 = (q_cost(wv, 3) << 4) + 12;
            wv.nq = 0;
            best = min(best, c0);
            best = min(best, c1);
            best = min(best, c2);
            best = min(best, c3);
            if (!(best & 15))
                break;
            s.bx = s16(s.bx - (((int)((unsigned)best << 28)) >> 30));
            s.by = s16(s.by - (((int)((unsigned)best << 30)) >> 30));
            s.bcost = best >> 4;
        }
        while (--i && in_range(s, s.bx, s.by));
    }
    else if (method == X265_HIP_UMH)
    {
        // motion.cpp:702-885
        int cross_start = 1;
        bool leave = false;
        const int sscale = (wv.h * wv.h) >> 4; // size_scale, motion.cpp:60-66
#define THRESH(v) (s.bcost < (((v) >> 4) * sscale))
        const int ucost1 = s.bcost;
        try4(wv, s, fpx, fpy, 0, -1, 0, 1, -1, 0, 1, 0);
        if (fpx | fpy)
            try4(wv, s, 0, 0, 0, -1, 0, 1, -1, 0, 1, 0);
        flush_f<false>(wv, s);
        const int ucost2 = s.bcost;
        if ((s.bx | s.by) && !(s.bx == fpx && s.by == fpy))
        {
            try4(wv, s, s.bx, s.by, 0, -1, 0, 1, -1, 0, 1, 0);
            flush_f<false>(wv, s);
        }
        if (s.bcost == ucost2)
            cross_start = 3;

        ox = s.bx;
        oy = s.by;
        if (s.bcost == ucost2 && THRESH(2000))
        {
            try4(wv, s, ox, oy, 0, -2, -1, -1, 1, -1, -2, 0);
            try4(wv, s, ox, oy, 2, 0, -1, 1, 1, 1, 0, 2);
            flush_f<false>(wv, s);
            if (s.bcost == ucost1 && THRESH(500))
                leave = true;
            else if (s.bcost == ucost2)
            {
                const int range = s16((merange >> 1) | 1);
                umh_cross(wv, s, ox, oy, 3, range, range);
                try4(wv, s, ox, oy, -1, -2, 1, -2, -2, -1, 2, -1);
                try4(wv, s, ox, oy, -2, 1, 2, 1, -1, 2, 1, 2);
                flush_f<false>(wv, s);
                if (s.bcost == ucost2)
                    leave = true;
                else
                    cross_start = range + 2;
            }
        }
        if (!leave)
        {
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
                int sad_ctx = THRESH(1000) ? 0 : THRESH(2000) ? 1 : THRESH(4000) ? 2 : 3;
                int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
                merange = (merange * k_range_mul[mvd_ctx][sad_ctx]) >> 2;
            }
#undef THRESH
            umh_cross(wv, s, ox, oy, cross_start, merange, merange >> 1);
            try4(wv, s, ox, oy, -2, -2, -2, 2, 2, -2, 2, 2);
            flush_f<false>(wv, s);

            // hexagon-grid rings around the new centre (motion.cpp:802-881).  Inside the window the
            // reference adds the two table entries as ints, without the uint16 truncation of mvcost()
            ox = s.bx;
            oy = s.by;
            int i = 1;
            do
            {
                const int room = min(min(s.hix - ox, ox - s.lox), min(s.hiy - oy, oy - s.loy));
                if (4 * i > room)
                {
                    for (int k = 0; k < 16; k++)
                    {
                        int mx = s16(ox + s16(k_hex4[k][0] * s16(i))), my = s16(oy + s16(k_hex4[k][1] * s16(i)));
                        if (in_range(s, mx, my))
                            try_f<false>(wv, s, mx, my);
                    }
                    flush_f<false>(wv, s);
                }
                else
                {
                    for (int k = 0; k < 16; k++)
                        q_push(wv, (ox + k_hex4[k][0] * i) * 4, (oy + k_hex4[k][1] * i) * 4, MV_RAW);
                    q_eval(wv, X265_HIP_SAD);
                    int dirx = 0, diry = 0;
                    bool moved = false;
                    for (int k = 0; k < 16; k++)
                    {
                        int cost = q_cost(wv, k);
                        if (cost < s.bcost)
                        {
                            s.bcost = cost;
                            dirx = k_hex4[k][0];
                            diry = k_hex4[k][1];
                            moved = true;
                        }
                    }
                    wv.nq = 0;
                    if (moved)
                    {
                        s.bx = s16(ox + i * dirx);
                        s.by = s16(oy + i * diry);
                    }
                }
                i = (i + 1) & 0xffff; // uint16_t counter in the reference
            }
            while (i <= merange >> 2);
            do_hex = in_range(s, s.bx, s.by); // "goto me_hex2", motion.cpp:882-883
        }
    }
    else if (method == X265_HIP_STAR)
    {
        // motion.cpp:887-995
        s.point = s.dist = 0;
        star_pattern(wv, s, 3, merange);
        bool done = false;
        if (s.dist == 1)
        {
            if (s.point)
            {
                const int saved = s.bcost;
                const int x1 = s16(s.bx + k_twopoint[(s.point - 1) * 2][0]), y1 = s16(s.by + k_twopoint[(s.point - 1) * 2][1]);
                const int x2 = s16(s.bx + k_twopoint[(s.point - 1) * 2 + 1][0]), y2 = s16(s.by + k_twopoint[(s.point - 1) * 2 + 1][1]);
                if (in_range(s, x1, y1))
                    try_f<false>(wv, s, x1, y1);
                if (in_range(s, x2, y2))
                    try_f<false>(wv, s, x2, y2);
                flush_f<false>(wv, s);
                if (s.bcost == saved)
                    done = true;
            }
            else
                done = true;
        }
        if (!done)
        {
            if (s.dist > 5)
                raster_scan(wv, s, 5, true);
            while (s.dist > 0)
            {
                s.dist = 0;
                s.point = 0;
                star_pattern(wv, s, 32, merange);
                if (s.dist == 1)
                {
                    if (!s.point)
                        break;
                    const int x1 = s16(s.bx + k_twopoint[(s.point - 1) * 2][0]), y1 = s16(s.by + k_twopoint[(s.point - 1) * 2][1]);
                    const int x2 = s16(s.bx + k_twopoint[(s.point - 1) * 2 + 1][0]), y2 = s16(s.by + k_twopoint[(s.point - 1) * 2 + 1][1]);
                    if (in_range(s, x1, y1))
                        try_f<false>(wv, s, x1, y1);
                    if (in_range(s, x2, y2))
                        try_f<false>(wv, s, x2, y2);
                    flush_f<false>(wv, s);
                    break;
                }
            }
        }
    }
    else if (method == X265_HIP_FULL)
    {
        raster_scan(wv, s, 1, false);
    }

    if (do_hex)
    {
        // hexagon radius 2 then square refine (motion.cpp:628-700).  First hexagon: codes 2..7 in the
        // order (-2,0)(-1,2)(1,2)(2,0)(1,-2)(-1,-2); smaller packed value wins, centre wins ties.
        for (int k = 1; k <= 6; k++) // hex2[1..6] is exactly that order
            push_f(wv, s.bx + k_hex2[k][0], s.by + k_hex2[k][1]);
        q_eval(wv, X265_HIP_SAD);
        int best = s.bcost << 3;
        for (int k = 0; k < 6; k++)
        {
            int v = (q_cost(wv, k) << 3) + (k + 2);
            best = min(best, v);
        }
        wv.nq = 0;
        if (best & 7)
        {
            int dir = (best & 7) - 2;
            s.bx = s16(s.bx + k_hex2[dir + 1][0]);
            s.by = s16(s.by + k_hex2[dir + 1][1]);
            for (int i = (merange >> 1) - 1; i > 0 && in_range(s, s.bx, s.by); i--)
            {
                for (int k = 0; k < 3; k++)
                    push_f(wv, s.bx + k_hex2[dir + k][0], s.by + k_hex2[dir + k][1]);
                q_eval(wv, X265_HIP_SAD);
                best &= ~7;
                for (int k = 0; k < 3; k++)
                {
                    int v = (q_cost(wv, k) << 3) + (k + 1);
                    best = min(best, v);
                }
                wv.nq = 0;
                if (!(best & 7))
                    break;
                dir += (best & 7) - 2;
                dir = k_mod6m1[dir + 1];
                s.bx = s16(s.bx + k_hex2[dir + 1][0]);
                s.by = s16(s.by + k_hex2[dir + 1][1]);
            }
        }
        s.bcost = best >> 3;

        for (int k = 1; k <= 8; k++)
            push_f(wv, s.bx + k_square1[k][0], s.by + k_square1[k][1]);
        q_eval(wv, X265_HIP_SAD);
        int dir = 0;
        for (int k = 1; k <= 8; k++)
        {
            int cost = q_cost(wv, k - 1);
            if (cost < s.bcost)
            {
                s.bcost = cost;
                dir = k;
            }
        }
        wv.nq = 0;
        s.bx = s16(s.bx + k_square1[dir][0]);
        s.by = s16(s.by + k_square1[dir][1]);
    }

    // pre-test winner bypasses the integer result (motion.cpp:1039-1045)
    int bcost = s.bcost, bx, by;
    if (bprecost < bcost)
    {
        bx = bpx;
        by = bpy;
        bcost = bprecost;
    }
    else
    {
        bx = s16(s.bx << 2);
        by = s16(s.by << 2);
    }

    if (!bcost)
    {
        bcost = mvc_q(wv, bx, by); // motion.cpp:1049-1054
    }
    else
    {
        const int subme = j.subme;
        const int hpel_iters = k_workload[subme][0], hpel_dirs = k_workload[subme][1];
        const int qpel_iters = k_workload[subme][2], qpel_dirs = k_workload[subme][3];
        const int hsatd = k_workload[subme][4];
        // The centre re-measure (SATD) does not depend on the half-pel results when hpel_satd is
        // set, so it rides in the same evaluation group as the first half-pel ring.
        for (int it = 0; it < hpel_iters; it++)
        {
            const bool centre = hsatd && it == 0;
            if (centre)
                q_push(wv, bx, by, MV_COST);
            for (int i = 1; i <= hpel_dirs; i++)
                q_push(wv, s16(bx + k_square1[i][0] * 2), s16(by + k_square1[i][1] * 2), MV_COST);
            q_eval(wv, hsatd);
            int k = 0;
            if (centre)
                bcost = q_cost(wv, k++);
            int bdir = 0;
            for (int i = 1; i <= hpel_dirs; i++, k++)
            {
                int cost = q_cost(wv, k);
                if (cost < bcost)
                {
                    bcost = cost;
                    bdir = i;
                }
            }
            wv.nq = 0;
            bx = s16(bx + k_square1[bdir][0] * 2);
            by = s16(by + k_square1[bdir][1] * 2);
        }
        if (!hsatd)
        {
            q_push(wv, bx, by, MV_COST);
            q_eval(wv, X265_HIP_SATD);
            bcost = q_cost(wv, 0);
            wv.nq = 0;
        }
        for (int it = 0; it < qpel_iters; it++)
        {
            for (int i = 1; i <= qpel_dirs; i++)
                q_push(wv, s16(bx + k_square1[i][0]), s16(by + k_square1[i][1]), MV_COST);
            q_eval(wv, X265_HIP_SATD);
            int bdir = 0;
            for (int i = 1; i <= qpel_dirs; i++)
            {
                int cost = q_cost(wv, i - 1);
                if (cost < bcost)
                {
                    bcost = cost;
                    bdir = i;
                }
            }
            wv.nq = 0;
            bx = s16(bx + k_square1[bdir][0]);
            by = s16(by + k_square1[bdir][1]);
        }
    }
    if (wv.lane == 0)
    {
        out->cost = bcost;
        out->mvx = (int16_t)bx;
        out->mvy = (int16_t)by;
    }
}

__global__ __launch_bounds__(64) void me_kernel(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n,
                                                x265_hip_me_result *out)
{
    __shared__ WaveLds lds;
    const int id = blockIdx.x;
    if (id >= n)
        return;
    const x265_hip_me_job &j = jobs[id];
    Wave wv;
    wave_init(wv, &lds, j.pu_w, j.pu_h);
    const PlaneDesc fp = tabs->planes[j.fenc_plane], rp = tabs->planes[j.ref_plane];
    wv.stride = rp.stride;
    wv.ref = rp.origin + (long)j.pu_y * rp.stride + j.pu_x;
    wv.tab = tabs->costtab[j.qp];
    wv.mvpx = j.qmvp_x;
    wv.mvpy = j.qmvp_y;
    load_fenc(wv, fp.origin + (long)j.pu_y * fp.stride + j.pu_x, fp.stride);
    motion_estimate(wv, j, &out[id]);
}

__global__ __launch_bounds__(64) void cost_kernel(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n,
                                                  int32_t *out)
{
    __shared__ WaveLds lds;
    const int id = blockIdx.x;
    if (id >= n)
        return;
    const x265_hip_cost_req &r = reqs[id];
    Wave wv;
    wave_init(wv, &lds, r.pu_w, r.pu_h);
    const PlaneDesc fp = tabs->planes[r.fenc_plane], rp = tabs->planes[r.ref_plane];
    wv.stride = rp.stride;
    wv.ref = rp.origin + (long)r.pu_y * rp.stride + r.pu_x;
    load_fenc(wv, fp.origin + (long)r.pu_y * fp.stride + r.pu_x, fp.stride);
    q_push(wv, r.qmv_x, r.qmv_y, MV_NONE);
    q_eval(wv, r.metric);
    if (wv.lane == 0)
        out[id] = wv.vc;
}

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
    const int total = a.nref * nt;
    for (int i = lane; i < total; i += 64)
    {
        int c = i / nt, t = i - c * nt;
        int ty = t / ntx, tx = t - ty * ntx;
        int pix[4][4];
        ref_tile(a.fref[c] + (long)(4 * ty) * a.rstride + 4 * tx, a.rstride, 0, 0, pix);
        int v = tile_metric(a.fenc + (long)(4 * ty) * a.fstride + 4 * tx, a.fstride, pix, a.metric);
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
    else // PRIM_VSP
    {
        const int16_t *p = (const int16_t *)a.src + (long)(y - 3) * a.sstride + x;
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc += p[(long)k * a.sstride] * c[k];
        ((uint8_t *)a.dst)[(long)y * a.dstride + x] = (uint8_t)clip_pixel((acc + 2048 + (8192 << 6)) >> 12);
    }
}

// ---------------------------------------------------------------------------------------------
// fullsearch_kernel (first form): one 256-thread workgroup per CTU.  The (64+nx-1) x (64+ny-1)
// reference window and the CTU's source pixels sit in LDS; each thread takes one search position
// per pass, forms the 64 8x8 SADs with v_sad_u8 and builds the 16x16 / 32x32 / 64x64 costs from them.
// Per pass the 85 PU costs are packed with the thread id ((cost << 8) | tid) and reduced to the
// workgroup minimum; strict-less against the running best keeps the first minimum in raster order.

#define FS_MAXPOS 132
#define FS_WSTRIDE 200 // >= 64 + 132 - 1 + 3, multiple of 4

struct FsLds
{
    uint8_t fenc[64 * 64];
    uint8_t win[(64 + FS_MAXPOS - 1) * FS_WSTRIDE + 16];
    unsigned pass_best[X265_HIP_PUS_PER_CTU];
    int best_cost[X265_HIP_PUS_PER_CTU];
    int best_idx[X265_HIP_PUS_PER_CTU];
};

__device__ __forceinline__ unsigned dpp_min_u32(unsigned v)
{
    // 64-lane minimum: quads, half rows, rows by DPP, then the four row results by readlane
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0xffffffff, (int)v, 0xB1, 0xf, 0xf, false)); // quad_perm 1,0,3,2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0xffffffff, (int)v, 0x4E, 0xf, 0xf, false)); // quad_perm 2,3,0,1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0xffffffff, (int)v, 0x141, 0xf, 0xf, false)); // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0xffffffff, (int)v, 0x140, 0xf, 0xf, false)); // row_mirror
    unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
    unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

__global__ __launch_bounds__(256) void fullsearch_kernel(const DeviceTables *tabs, const x265_hip_ctu_job *jobs, int n,
                                                         x265_hip_me_result *out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    FsLds &L = *(FsLds *)smem;
    const int id = blockIdx.x;
    if (id >= n)
        return;
    const x265_hip_ctu_job j = jobs[id];
    const int tid = threadIdx.x;
    const PlaneDesc fp = tabs->planes[j.fenc_plane], rp = tabs->planes[j.ref_plane];
    const uint16_t *tab = tabs->costtab[j.qp];
    const int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    const int ww = 64 + nx - 1, wh = 64 + ny - 1;

    // stage source CTU and reference window (dword granules, unaligned sources)
    const uint8_t *fsrc = fp.origin + (long)j.ctu_y * fp.stride + j.ctu_x;
    for (int i = tid; i < 64 * 16; i += 256)
    {
        int y = i >> 4, x = i & 15;
        *(uint32_t *)(L.fenc + y * 64 + 4 * x) = ld4u(fsrc + (long)y * fp.stride + 4 * x);
    }
    const uint8_t *rsrc = rp.origin + (long)(j.ctu_y + j.mvmin_y) * rp.stride + j.ctu_x + j.mvmin_x;
    const int wdw = (ww + 3) >> 2;
    for (int i = tid; i < wh * wdw; i += 256)
    {
        int y = i / wdw, x = i - y * wdw;
        *(uint32_t *)(L.win + y * FS_WSTRIDE + 4 * x) = ld4u(rsrc + (long)y * rp.stride + 4 * x);
    }
    if (tid < X265_HIP_PUS_PER_CTU)
    {
        L.best_cost[tid] = 0x7fffffff;
        L.best_idx[tid] = 0;
        L.pass_best[tid] = 0xffffffffu;
    }
    __syncthreads();

    const int total = nx * ny;
    for (int base = 0; base < total; base += 256)
    {
        const int idx = base + tid;
        const bool live = idx < total;
        const int py = live ? idx / nx : 0, px = live ? idx - py * nx : 0;
        int mvc = 0;
        if (live)
            mvc = mvcost_q(tab, j.qmvp_x, j.qmvp_y, s16((j.mvmin_x + px) << 2), s16((j.mvmin_y + py) << 2));
        const uint8_t *wbase = L.win + py * FS_WSTRIDE + px;

        unsigned s64 = 0;
        for (int q32 = 0; q32 < 4; q32++)
        {
            unsigned s32 = 0;
            for (int q16 = 0; q16 < 4; q16++)
            {
                unsigned s16sum = 0;
                const int x16 = (q32 & 1) * 32 + (q16 & 1) * 16, y16 = (q32 >> 1) * 32 + (q16 >> 1) * 16;
#pragma unroll
                for (int q8 = 0; q8 < 4; q8++)
                {
                    const int x8 = x16 + (q8 & 1) * 8, y8 = y16 + (q8 >> 1) * 8;
                    unsigned s8 = 0;
#pragma unroll
                    for (int r = 0; r < 8; r++)
                    {
                        const uint2 f = *(const uint2 *)(L.fenc + (y8 + r) * 64 + x8);
                        const uint8_t *w = wbase + (y8 + r) * FS_WSTRIDE + x8;
                        s8 = __builtin_amdgcn_sad_u8(f.x, ld4u(w), s8);
                        s8 = __builtin_amdgcn_sad_u8(f.y, ld4u(w + 4), s8);
                    }
                    s16sum += s8;
                    const int pu = 21 + (y8 >> 3) * 8 + (x8 >> 3);
                    unsigned key = live ? ((s8 + mvc) << 8) | tid : 0xffffffffu;
                    key = dpp_min_u32(key);
                    if ((tid & 63) == 0)
                        atomicMin(&L.pass_best[pu], key);
                }
                s32 += s16sum;
                const int pu = 5 + (y16 >> 4) * 4 + (x16 >> 4);
                unsigned key = live ? ((s16sum + mvc) << 8) | tid : 0xffffffffu;
                key = dpp_min_u32(key);
                if ((tid & 63) == 0)
                    atomicMin(&L.pass_best[pu], key);
            }
            s64 += s32;
            unsigned key = live ? ((s32 + mvc) << 8) | tid : 0xffffffffu;
            key = dpp_min_u32(key);
            if ((tid & 63) == 0)
                atomicMin(&L.pass_best[1 + q32], key);
        }
        {
            unsigned key = live ? ((s64 + mvc) << 8) | tid : 0xffffffffu;
            key = dpp_min_u32(key);
            if ((tid & 63) == 0)
                atomicMin(&L.pass_best[0], key);
        }
        __syncthreads();
        if (tid < X265_HIP_PUS_PER_CTU)
        {
            unsigned k = L.pass_best[tid];
            int cost = (int)(k >> 8);
            if (k != 0xffffffffu && cost < L.best_cost[tid])
            {
                L.best_cost[tid] = cost;
                L.best_idx[tid] = base + (int)(k & 255);
            }
            L.pass_best[tid] = 0xffffffffu;
        }
        __syncthreads();
    }
    if (tid < X265_HIP_PUS_PER_CTU)
    {
        const int bi = L.best_idx[tid];
        const int py = bi / nx, px = bi - py * nx;
        x265_hip_me_result r;
        r.cost = L.best_cost[tid];
        r.mvx = (int16_t)((j.mvmin_x + px) << 2);
        r.mvy = (int16_t)((j.mvmin_y + py) << 2);
        out[(long)id * X265_HIP_PUS_PER_CTU + tid] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers

void launch_me(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, x265_hip_me_result *out, hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(me_kernel, dim3(n), dim3(64), 0, st, tabs, jobs, n, out);
}

void launch_cost(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int32_t *out, hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(cost_kernel, dim3(n), dim3(64), 0, st, tabs, reqs, n, out);
}

void launch_prim_cmp(const PrimCmpArgs &a, int32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(prim_cmp_kernel, dim3(1), dim3(64), 0, st, a, out);
}

void launch_prim_filter(const PrimFilterArgs &a, hipStream_t st)
{
    const int total = a.w * a.rows;
    hipLaunchKernelGGL(prim_filter_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
}

size_t fullsearch_lds_bytes() { return sizeof(FsLds); }

int fullsearch_setup()
{
    return (int)hipFuncSetAttribute((const void *)fullsearch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sizeof(FsLds));
}

void launch_fullsearch(const DeviceTables *tabs, const x265_hip_ctu_job *jobs, int n, x265_hip_me_result *out,
                       hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(fullsearch_kernel, dim3(n), dim3(256), sizeof(FsLds), st, tabs, jobs, n, out);
}

} // namespace x265hip
