// refine.hip -- one sub-pel refinement ring per request, with the interpolation shared between its candidates.
//
// Reference: the sub-pel stage of MotionEstimate::motionEstimate (encoder/motion.cpp:1047-1119): each iteration
// measures subpelCompare(ref, bmv + square1[i] * step, cmp) (:1126-1164) for i = 1 .. 4 or 8 around one centre, step 2
// (half-pel ring) or 1 (quarter-pel ring), plus (first half-pel iteration with SATD) the centre itself.  The nine
// positions are a 3 x 3 grid: three horizontal positions k x three vertical positions j.  luma_hps + luma_vsp
// (common/ipfilter.cpp:117-160, 241-282) is separable and its first stage is exact (no rounding at 8 bits), so
//   * one horizontal pass per horizontal position gives int16 rows I_k = sum_t c[fx_k][t] * (p - 128) that the three
//     vertical positions of that column of the grid share;
//   * with the tap row {0,0,0,64,0,0,0,0} a stage is the identity up to scale, and the two-stage formula with it
//     reproduces luma_hpp / luma_vpp / the full-pel copy bit for bit ((64 S + 2048) >> 12 == (S + 32) >> 6), so all
//     nine positions go through the same code whatever their fractions are.
//
// Shape of the kernel (the batch hand-out is cost_body's, me_kernel.hip).  Work items are (request, k, strip of 4 x 8
// pixels), dealt k-major over a batch so that the lanes of a pass share k.  A lane
//   1. filters the 16 reference rows its three vertical positions need (rows iy(j = -1) - 3 .. + 15) horizontally:
//      16-byte load, 8 v_dot4_i32_i8 per row, kept as eight registers x 4 pixels of vertical int16 pairs;
//   2. per vertical position j: 5 v_dot2_i32_i16 per pixel over those pairs with a 9-tap window (the 8 taps at offset
//      delta_j = iy(j) - iy(-1), so one static register pattern serves every j; taps pre-multiplied by 16: the upper
//      half of the sum is the >> 12), clamp on packed pairs, SAD or the packed 4x4 Hadamard of me_device.h against the
//      source block, LDS atomic into the candidate's sum.
// A first version that staged the horizontal passes in LDS (cells of 16 x 16, both vertical pairings, groups of cells
// per wave) issued 1.65 x the VALU instructions of cost_kernel for the same candidates (profiles/r02_refine_lds_pmc.txt:
// the staging pass and the cell bookkeeping cost as much as they saved) and was dropped for this register form.
#include <stdlib.h>

#include "kernels.h"
#include "me_device.h"

namespace x265hip {

#define RF_BATCH 32 // requests per batch (at most)
#define RF_PARTS 64
#ifndef RF_CHUNK
#define RF_CHUNK 8
#endif
#ifndef RF_WAVES
#define RF_WAVES 4 // waves per SIMD the kernel is compiled for (128 VGPRs)
#endif
#ifndef RF_SHIFTED_TAPS
#define RF_SHIFTED_TAPS 1
#endif
#define RF_MAX_BLOCKS (256 * RF_WAVES)
#define RF_AHEAD 4 // reference rows in flight per lane during the horizontal pass

// 9-tap windows of the vertical filter x 16 as int16 pairs, per (fraction, delta): entries 0..4 serve even output
// rows (C0,C1) (C2,C3) (C4,C5) (C6,C7) (C8,0), entries 5..9 odd ones (0,C0) (C1,C2) (C3,C4) (C5,C6) (C7,C8), where
// C = (c0..c7, 0) for delta 0 and (0, c0..c7) for delta 1
#define RF_P(a, b) X265HIP_PK16(16 * (a), 16 * (b))
#define RF_WIN(C0, C1, C2, C3, C4, C5, C6, C7, C8) \
    { RF_P(C0, C1), RF_P(C2, C3), RF_P(C4, C5), RF_P(C6, C7), RF_P(C8, 0), RF_P(0, C0), RF_P(C1, C2), RF_P(C3, C4), RF_P(C5, C6), RF_P(C7, C8), 0, 0 }
#define RF_TAPS(c0, c1, c2, c3, c4, c5, c6, c7) RF_WIN(c0, c1, c2, c3, c4, c5, c6, c7, 0), RF_WIN(0, c0, c1, c2, c3, c4, c5, c6, c7)
__device__ __constant__ __attribute__((aligned(16))) const uint32_t k_ring_taps[8][12] = {
    RF_TAPS(0, 0, 0, 64, 0, 0, 0, 0),
    RF_TAPS(-1, 4, -10, 58, 17, -5, 1, 0),
    RF_TAPS(-1, 4, -11, 40, 40, -11, 4, -1),
    RF_TAPS(0, 1, -5, 17, 58, -10, 4, -1),
};
// index into square1 (motion.cpp:71-73) of grid position (k, j) = (dx + 1, dy + 1)
__device__ __constant__ const uint8_t k_sq_of[3][3] = { { 5, 3, 6 }, { 1, 0, 2 }, { 7, 4, 8 } };

// A list made of up to four runs of requests of one PU size each (the sizes an encoder works through: 64, 32, 16, 8):
// run s holds requests req0[s] .. req0[s + 1] - 1 in batches of rb[s], its first batch is number bat0[s] of the launch.
// One launch for the four runs instead of four launches: the short batches of the small sizes fill the end of the
// launch, where a launch of large blocks alone leaves waves idle for up to one long batch.
struct RingSections
{
    int nsec; // 0: no runs given, the kernel sizes its batches from a sample of the list
    int req0[5], bat0[5], rb[4];
};

struct RingLds
{
    int pre[RF_BATCH + 1]; // exclusive prefix of the requests' strip counts
    int prea[RF_BATCH + 1]; // ... counting only the requests whose side columns are horizontal-only (ring_strip's hz_only)
    int cost[RF_BATCH * X265_HIP_RING_COSTS];
    const uint8_t *fenc[RF_BATCH]; // PU origin in the source plane
    const uint8_t *ref[RF_BATCH];  // PU origin in the reference plane
    int fstride[RF_BATCH], rstride[RF_BATCH];
    int info[RF_BATCH]; // ntx | tall << 5 | metric << 6 | step << 7 | eight directions << 9 | centre << 10
    int qmv[RF_BATCH];  // (cy << 16) | (cx & 0xffff)
};

__device__ __forceinline__ int rf_s16(int v) { return (int)(int16_t)v; }

// One strip of NT tiles for one horizontal position: `src` is the reference at row iy(-1) - 3, column ix - 3 of
// the strip; jneed says which vertical positions this lane wants, tj[j] their tap windows (fraction * 2 + delta).
// Adds the lane's costs into cost[sq[j]].
// hz_only (the same for every lane of the pass): each lane wants the middle vertical position alone and that position
// is a full row (the side columns of a four-point ring around a centre whose y is whole, e.g. the half-pel ring around
// an integer winner): the strip's own 4 NT rows filtered horizontally are the prediction, luma_hpp's (S + 32) >> 6
// (ipfilter.cpp:76-115) -- no rows above and below, no vertical pass.  `src` is then the strip's row -3 all the same.
template <int NT>
__device__ __forceinline__ void ring_strip(gptr_t src, int stride, int fx, int jneed, const int tj[3], const uint32_t *taps_lds,
                                           const uint8_t *f, int fstride, int metric, int *cost, const int sq[3], bool hz_only)
{
    constexpr int NP = 2 * NT + 4; // row pairs: rows 0 .. 4 NT + 7
    uint32_t q[NP][4];
    // the source strip, fetched once for the three vertical positions and early, so that no comparison waits for it
    uint32_t frow[4 * NT];
    {
        gptr_t fp = as_global(f);
        const long lf = fstride;
#pragma unroll
        for (int y = 0; y < 4 * NT; y++, fp += lf)
            frow[y] = ld4g(fp);
    }
    if (hz_only)
    {
        const uint32_t tlo = k_taps_s8[fx][0], thi = k_taps_s8[fx][1];
        const long lstride = stride;
        gptr_t rp = src + 3 * lstride;
        uint32_t w[4 * NT][3];
#pragma unroll
        for (int y = 0; y < 4 * NT; y++, rp += lstride)
            ld12g(rp, w[y][0], w[y][1], w[y][2]);
        uint32_t P[4 * NT][2];
#pragma unroll
        for (int y = 0; y < 4 * NT; y++)
        {
            int o[4];
            // the taps run over p - 128: S + 32 = o + 64 * 128 + 32; (S + 32) >> 6, clamped to 0 .. 255, as int16 pairs
            tap8x4<8192 + 32>(w[y][0] ^ 0x80808080u, w[y][1] ^ 0x80808080u, w[y][2] ^ 0x80808080u, tlo, thi, o);
            const s16x2 lo = { 0, 0 }, hi255 = { 255, 255 };
            const s16x2 p0 = as_s2(pack_s16(o[0] >> 6, o[1] >> 6)), p1 = as_s2(pack_s16(o[2] >> 6, o[3] >> 6));
            P[y][0] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(p0, lo), hi255));
            P[y][1] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(p1, lo), hi255));
        }
        int v = 0;
#pragma unroll
        for (int t = 0; t < NT; t++)
        {
            uint32_t fa[4], fb[4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                fa[y] = __builtin_amdgcn_perm(0, frow[4 * t + y], 0x0c010c00);
                fb[y] = __builtin_amdgcn_perm(0, frow[4 * t + y], 0x0c030c02);
            }
            v += tile_metric_pairs(fa, fb, &P[4 * t], metric);
        }
        atomicAdd(&cost[sq[1]], v);
        return;
    }
    // ---- horizontal pass: rows 2p, 2p + 1 as vertical pairs ---------------------------------------------------------
    if (__ballot(fx != 0) == 0)
    {
        // no lane of this pass filters horizontally: I = 64 (p - 128), the luma_p2s form
        gptr_t rp = src + 3;
        const long lstride = stride;
#pragma unroll
        for (int p = 0; p < NP; p++, rp += 2 * lstride)
        {
            const uint32_t a = ld4g(rp) ^ 0x80808080u, b = ld4g(rp + lstride) ^ 0x80808080u;
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                const int va = __builtin_amdgcn_sbfe((int)a, 8 * x, 8), vb = __builtin_amdgcn_sbfe((int)b, 8 * x, 8);
                q[p][x] = as_u32(as_s2(pack_s16(va, vb)) << 6); // v_pk_lshlrev_b16
            }
        }
    }
    else
    {
        // rows are fetched RF_AHEAD ahead of their use as aligned 16-byte pieces (the compiler, left alone, waits for
        // every load right where it issues it: sixteen exposed latencies per item)
        const uint32_t tlo = k_taps_s8[fx][0], thi = k_taps_s8[fx][1];
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        constexpr int ROWS = 2 * NP;
        // a row's 12 bytes straight from their byte address (three dwords; the aligned 16-byte form needs three
        // v_alignbyte per row to undo the alignment, and this kernel is short of VALU slots, not of load bandwidth)
        gptr_t next = src;
        const long lstride = stride;
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        auto fetch = [&](int) {
            u32x3 w3;
            __builtin_memcpy(&w3, (const void *)next, 12);
            next += lstride;
            const u32x4 w = { w3.x, w3.y, w3.z, 0 };
            return w;
        };
#if RF_SHIFTED_TAPS
        // the taps of pixel x laid over the row's 12 bytes (bytes x .. x + 7): eleven dot products per row and no
        // byte shifts of the data, against eight dot products and six shifts
        uint32_t T[4][3];
#pragma unroll
        for (int x = 1; x < 4; x++)
        {
            T[x][0] = tlo << (8 * x);
            T[x][1] = __builtin_amdgcn_alignbyte(thi, tlo, 4 - x);
            T[x][2] = thi >> (32 - 8 * x);
        }
        auto hrow = [&](int, const u32x4 w, int o[4]) {
            const uint32_t d0 = w.x ^ 0x80808080u, d1 = w.y ^ 0x80808080u, d2 = w.z ^ 0x80808080u;
            o[0] = __builtin_amdgcn_sdot4((int)d0, (int)tlo, sdot4_first<0>(d1, thi), false);
#pragma unroll
            for (int x = 1; x < 4; x++)
                o[x] = __builtin_amdgcn_sdot4((int)d0, (int)T[x][0], __builtin_amdgcn_sdot4((int)d1, (int)T[x][1], sdot4_first<0>(d2, T[x][2]), false), false);
        };
#else
        auto hrow = [&](int, const u32x4 w, int o[4]) {
            tap8x4<0>(w.x ^ 0x80808080u, w.y ^ 0x80808080u, w.z ^ 0x80808080u, tlo, thi, o);
        };
#endif
        u32x4 ahead[RF_AHEAD];
#pragma unroll
        for (int r = 0; r < RF_AHEAD; r++)
            ahead[r] = fetch(r);
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            int a[4], b[4];
            const u32x4 wa = ahead[(2 * p) % RF_AHEAD], wb = ahead[(2 * p + 1) % RF_AHEAD];
            if (2 * p + RF_AHEAD < ROWS)
                ahead[(2 * p) % RF_AHEAD] = fetch(2 * p + RF_AHEAD);
            if (2 * p + 1 + RF_AHEAD < ROWS)
                ahead[(2 * p + 1) % RF_AHEAD] = fetch(2 * p + 1 + RF_AHEAD);
            __builtin_amdgcn_sched_barrier(0);
            hrow(2 * p, wa, a);
            hrow(2 * p + 1, wb, b);
#pragma unroll
            for (int x = 0; x < 4; x++)
                q[p][x] = pack_s16(a[x], b[x]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- vertical pass + comparison per vertical position -----------------------------------------------------------
    constexpr int seed = 16 * (2048 + (8192 << 6));
    // the source strip as 16-bit pairs, once for the (up to) three comparisons
    uint32_t fa[4 * NT], fb[4 * NT];
#pragma unroll
    for (int y = 0; y < 4 * NT; y++)
    {
        fa[y] = __builtin_amdgcn_perm(0, frow[y], 0x0c010c00);
        fb[y] = __builtin_amdgcn_perm(0, frow[y], 0x0c030c02);
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
    {
        if (__ballot((jneed >> j) & 1) == 0)
            continue;
        const uint32_t *tp = taps_lds + tj[j] * 12;
        const uint4 t03 = *(const uint4 *)tp, t47 = *(const uint4 *)(tp + 4);
        const uint2 t89 = *(const uint2 *)(tp + 8);
        const uint32_t te[5] = { t03.x, t03.y, t03.z, t03.w, t47.x }, to[5] = { t47.y, t47.z, t47.w, t89.x, t89.y };
        int v = 0;
#pragma unroll
        for (int t = 0; t < NT; t++)
        {
            uint32_t P[4][2];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                const int u = 2 * t + (y >> 1);
                const uint32_t *w = (y & 1) ? to : te;
                int s[4];
#pragma unroll
                for (int x = 0; x < 4; x++)
                    s[x] = sdot2(q[u + 4][x], w[4], sdot2(q[u + 3][x], w[3], sdot2(q[u + 2][x], w[2], sdot2(q[u + 1][x], w[1], sdot2_first<seed>(q[u][x], w[0])))));
                // upper halves = sums >> 12 (taps were x 16), as int16 pairs; then the clamp to 0..255
                const s16x2 lo = { 0, 0 }, hi255 = { 255, 255 };
                s16x2 p0 = as_s2(__builtin_amdgcn_perm((uint32_t)s[1], (uint32_t)s[0], 0x07060302));
                s16x2 p1 = as_s2(__builtin_amdgcn_perm((uint32_t)s[3], (uint32_t)s[2], 0x07060302));
                P[y][0] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(p0, lo), hi255));
                P[y][1] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(p1, lo), hi255));
            }
            v += tile_metric_pairs(fa + 4 * t, fb + 4 * t, P, metric);
        }
        if ((jneed >> j) & 1)
            atomicAdd(&cost[sq[j]], v);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RF_WAVES))) void refine_kernel(const DeviceTables *tabs, const x265_hip_refine_req *reqs,
                                                                                           int n, int32_t *out, uint32_t *counters, int parity, int rb, int max_items, int max_items_one, RingSections secs)
{
    __shared__ RingLds lds[4];
    __shared__ __attribute__((aligned(16))) uint32_t taps_lds[8 * 12];
    if (threadIdx.x < 8 * 12)
        taps_lds[threadIdx.x] = (&k_ring_taps[0][0])[threadIdx.x];
    __syncthreads();
    RingLds &L = lds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;

    // Work per batch is bounded: every wave looks at the same 64 requests spread over the list (multiples of the
    // golden ratio, see cost_body) and halves rb until a batch holds about max_items items by their mean: 1024 for a
    // list of mixed PU sizes, 256 (three or four passes of the item loop) when the sample shows one size only -- a list
    // of one size in batches of a thousand items is little more than one batch per wave of the chip, and the launch
    // then ends with most waves idle (per-size lists at 4K: 0.205 ms -> 0.185 ms); mixed lists need the longer
    // batches (at 256 their idle lanes cost a third).  (A list whose sizes come in long runs, e.g. sorted by PU
    // size, defeats a size taken from a sample of the whole list: callers submit such lists one size at a time.
    // Deciding the length of a run where it is taken, from the request it starts with, was tried with a
    // compare-and-swap hand-out and serialised the waves of a part: 3 us per run.)
    const bool sectioned = secs.nsec > 0;
    if (!sectioned && rb > 1)
    {
        const int si = n >= 64 ? (int)(((unsigned long long)((unsigned)lane * 0x9E3779B1u) * (unsigned long long)n) >> 32) : lane;
        const int cnt = n < 64 ? n : 64;
        int items = 0;
        if (si < n)
            items = 3 * (reqs[si].pu_w >> 2) * ((reqs[si].pu_h + 7) >> 3);
        const bool one_size = __ballot(lane < cnt && items != __shfl(items, 0)) == 0;
        for (int o = 32; o; o >>= 1)
            items += __shfl_xor(items, o);
        const int bound = one_size ? max_items_one : max_items;
        while (rb > 1 && items * rb > bound * cnt)
            rb >>= 1;
        rb = uniform(rb);
    }
    const int nbatch = sectioned ? secs.bat0[secs.nsec] : (n + rb - 1) / rb;
    // batch g of the launch: its first request, the end of its run and the run's batch size (all wave-uniform)
    auto locate = [&](int g, int &first, int &limit, int &size) {
        first = g * rb, limit = n, size = rb;
        if (sectioned)
        {
            first = secs.req0[0] + g * secs.rb[0], limit = secs.req0[1], size = secs.rb[0];
#pragma unroll
            for (int q = 1; q < 4; q++)
                if (q < secs.nsec && g >= secs.bat0[q])
                    first = secs.req0[q] + (g - secs.bat0[q]) * secs.rb[q], limit = secs.req0[q + 1], size = secs.rb[q];
        }
    };
    const int parts = min((int)gridDim.x, RF_PARTS), part = blockIdx.x % parts;
    const int nwaves = 4 * (((int)gridDim.x - part + parts - 1) / parts);
    auto batch_of = [&](int j) { return ((j / RF_CHUNK) * parts + part) * RF_CHUNK + j % RF_CHUNK; };
    const int nchunk = (nbatch + RF_CHUNK - 1) / RF_CHUNK;
    const int hi = ((nchunk - part + parts - 1) / parts) * RF_CHUNK;
    uint32_t *const counter = counters + (parity * RF_PARTS + part) * 16;
    if (blockIdx.x == 0 && threadIdx.x < RF_PARTS)
        counters[((parity ^ 1) * RF_PARTS + threadIdx.x) * 16] = 0;
    int b = (blockIdx.x / parts) * 4 + (threadIdx.x >> 6);
    x265_hip_refine_req cur = {};
    if (b < hi && batch_of(b) < nbatch)
    {
        int first, limit, size;
        locate(batch_of(b), first, limit, size);
        if (lane < size && first + lane < limit)
            cur = reqs[first + lane];
    }

    while (b < hi)
    {
        int nb = 0;
        if (lane == 0)
            nb = (int)atomicAdd(counter, 1u);
        nb = nwaves + uniform(nb);
        int base = 0, limit = 0, bsize = 0;
        if (batch_of(b) < nbatch)
            locate(batch_of(b), base, limit, bsize);
        const bool have = lane < bsize && base + lane < limit;
        int ns = 0; // strips of this lane's request
        const bool cur_eight = (cur.pattern & X265_HIP_RING_8) != 0;
        const int cur_cy = cur.qmv_y;
        if (have)
        {
            const PlaneDesc fp = tabs->planes[cur.fenc_plane], rp = tabs->planes[cur.ref_plane];
            const int tall = (cur.pu_h & 7) == 0;
            ns = (cur.pu_w >> 2) * (cur.pu_h >> (2 + tall));
            L.fenc[lane] = fp.origin + (long)cur.pu_y * fp.stride + cur.pu_x;
            L.ref[lane] = rp.origin + (long)cur.pu_y * rp.stride + cur.pu_x;
            L.fstride[lane] = fp.stride;
            L.rstride[lane] = rp.stride;
            L.info[lane] = (cur.pu_w >> 2) | (tall << 5) | ((cur.metric & 1) << 6) | (((cur.pattern & X265_HIP_RING_HPEL) ? 2 : 1) << 7) |
                           (((cur.pattern & X265_HIP_RING_8) ? 1 : 0) << 9) | (((cur.pattern & X265_HIP_RING_CENTRE) ? 1 : 0) << 10);
            L.qmv[lane] = (int)(((uint32_t)(uint16_t)cur.qmv_y << 16) | (uint16_t)cur.qmv_x);
        }
        {
            // inclusive scans over lanes 0..31 (lanes 32..63 do the same for the horizontal-only requests): row_shr 1, 2, 4,
            // 8 inside each 16-lane row, then the row's total to the row behind it
            const int mine = __shfl(ns, lane & 31), hz = __shfl((int)(have && !(cur_eight) && (cur_cy & 3) == 0), lane & 31);
            int inc = lane < 32 ? mine : (hz ? mine : 0);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
            const int row0 = __builtin_amdgcn_readlane(inc, 15), row2 = __builtin_amdgcn_readlane(inc, 47);
            if ((lane & 31) >= 16)
                inc += lane < 32 ? row0 : row2;
            (lane < 32 ? L.pre : L.prea)[(lane & 31) + 1] = inc;
            if ((lane & 31) == 0)
                (lane < 32 ? L.pre : L.prea)[0] = 0;
        }
        for (int i = lane; i < RF_BATCH * X265_HIP_RING_COSTS; i += 64)
            L.cost[i] = 0;
        // next batch's records: in flight during this batch's arithmetic
        if (nb < hi && batch_of(nb) < nbatch)
        {
            int first, nlimit, size;
            locate(batch_of(nb), first, nlimit, size);
            if (lane < size && first + lane < nlimit)
                cur = reqs[first + lane];
        }
        wave_sync();
        // strips of the batch; items are (k, strip): the side columns (k = 0, then k = 2) of the horizontal-only requests,
        // the side columns of the others, then the middle column of all, so that the lanes of a pass share a kind
        const int S = L.pre[RF_BATCH], SA = L.prea[RF_BATCH], SB = S - SA;
        for (int i = lane; i < 3 * S; i += 64)
        {
            int k, si, r = 0, t;
            bool side_hz = false;
            if (i < 2 * SA)
            {
                k = i >= SA ? 2 : 0, si = i - (k >> 1) * SA, side_hz = true;
#pragma unroll
                for (int s = 16; s > 0; s >>= 1)
                    if (L.prea[r + s] <= si)
                        r += s;
                t = si - L.prea[r];
            }
            else if (i < 2 * S)
            {
                k = i - 2 * SA >= SB ? 2 : 0, si = i - 2 * SA - (k >> 1) * SB;
#pragma unroll
                for (int s = 16; s > 0; s >>= 1)
                    if (L.pre[r + s] - L.prea[r + s] <= si)
                        r += s;
                t = si - (L.pre[r] - L.prea[r]);
            }
            else
            {
                k = 1, si = i - 2 * S;
#pragma unroll
                for (int s = 16; s > 0; s >>= 1)
                    if (L.pre[r + s] <= si)
                        r += s;
                t = si - L.pre[r];
            }
            const int info = L.info[r], ntx = info & 31, tall = (info >> 5) & 1, metric = (info >> 6) & 1, step = (info >> 7) & 3;
            const int eight = (info >> 9) & 1, centre = (info >> 10) & 1;
            const int ty = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)ntx)), tx = t - ty * ntx; // exact: t < 256, ntx <= 16
            const int qm = L.qmv[r], cx = (int)(int16_t)qm, cy = qm >> 16;
            const int qx = rf_s16(cx + (k - 1) * step);
            const int iy0 = rf_s16(cy - step) >> 2; // integer row of the topmost vertical position
            // which of the three vertical positions the pattern has in this column of the grid
            const int jneed = eight ? (k == 1 && !centre ? 5 : 7) : (k == 1 ? (centre ? 7 : 5) : 2);
            int tj[3], sq[3];
#pragma unroll
            for (int j = 0; j < 3; j++)
            {
                const int qy = rf_s16(cy + (j - 1) * step);
                tj[j] = (qy & 3) * 2 + ((qy >> 2) - iy0);
                sq[j] = r * X265_HIP_RING_COSTS + k_sq_of[k][j];
            }
            const int rstride = L.rstride[r], fstride = L.fstride[r];
            const int rows = 4 << tall;
            const gptr_t src = as_global(L.ref[r]) + (long)(rows * ty + iy0 - 3) * rstride + 4 * tx + (qx >> 2) - 3;
            const uint8_t *f = L.fenc[r] + (long)(rows * ty) * fstride + 4 * tx;
            // (every lane of the pass, idle ones aside: a pass at the border of two kinds takes the general path)
            const bool hz_only = __ballot(!side_hz) == 0;
            const gptr_t src_hz = as_global(L.ref[r]) + (long)(rows * ty + (cy >> 2) - 3) * rstride + 4 * tx + (qx >> 2) - 3;
            if (tall)
                ring_strip<2>(hz_only ? src_hz : src, rstride, qx & 3, jneed, tj, taps_lds, f, fstride, metric, L.cost, sq, hz_only);
            else
                ring_strip<1>(hz_only ? src_hz : src, rstride, qx & 3, jneed, tj, taps_lds, f, fstride, metric, L.cost, sq, hz_only);
        }
        wave_sync();
        // ---- results: nine per request, INT_MAX where the pattern has no candidate ---------------------------------
        for (int i = lane; i < bsize * X265_HIP_RING_COSTS; i += 64)
        {
            const int r = i / X265_HIP_RING_COSTS, k = i - r * X265_HIP_RING_COSTS;
            if (base + r < limit)
            {
                const int info = L.info[r], ndirs = ((info >> 9) & 1) ? 8 : 4;
                const bool used = k == 0 ? ((info >> 10) & 1) != 0 : k <= ndirs;
                out[(long)(base + r) * X265_HIP_RING_COSTS + k] = used ? L.cost[i] : 0x7fffffff;
            }
        }
        wave_sync(); // the next batch rewrites this wave's record
        b = nb;
    }
}

// batch bounds (items per batch by the list's mean), see the kernel; the environment overrides are for experiments
static int rf_max_items() { static const int v = getenv("X265_RF_ITEMS") ? atoi(getenv("X265_RF_ITEMS")) : 1024; return v; }
static int rf_max_items_one() { static const int v = getenv("X265_RF_ITEMS1") ? atoi(getenv("X265_RF_ITEMS1")) : 256; return v; }

void launch_refine(const DeviceTables *tabs, const x265_hip_refine_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st)
{
    if (n <= 0)
        return;
    int rb = RF_BATCH;
    while (rb > 1 && (n + rb - 1) / rb < 2048)
        rb >>= 1;
    int blocks = (n + 3) / 4;
    if (blocks > RF_MAX_BLOCKS)
        blocks = RF_MAX_BLOCKS;
    RingSections secs = {};
    hipLaunchKernelGGL(refine_kernel, dim3(blocks), dim3(256), 0, st, tabs, reqs, n, out, counters, parity, rb, rf_max_items(), rf_max_items_one(), secs);
}

// the list as runs of one PU size each (at most four, largest blocks first for the best packing; any order works)
void launch_refine_sections(const DeviceTables *tabs, const x265_hip_refine_req *reqs, const x265_hip_refine_section *sections, int nsec, int32_t *out,
                            uint32_t *counters, int parity, hipStream_t st)
{
    RingSections secs = {};
    int n = 0, nbatch = 0;
    for (int s = 0; s < nsec; s++)
    {
        if (sections[s].count <= 0)
            continue;
        const int q = secs.nsec++;
        const int items = 3 * (sections[s].pu_w >> 2) * ((sections[s].pu_h + 7) >> 3);
        int rb = RF_BATCH;
        while (rb > 1 && items * rb > rf_max_items_one())
            rb >>= 1;
        secs.req0[q] = n, secs.bat0[q] = nbatch, secs.rb[q] = rb;
        n += sections[s].count;
        nbatch += (sections[s].count + rb - 1) / rb;
    }
    if (!secs.nsec)
        return;
    for (int q = secs.nsec; q < 5; q++)
        secs.req0[q] = n, secs.bat0[q] = nbatch;
    int blocks = (nbatch + 3) / 4;
    if (blocks > RF_MAX_BLOCKS)
        blocks = RF_MAX_BLOCKS;
    hipLaunchKernelGGL(refine_kernel, dim3(blocks), dim3(256), 0, st, tabs, reqs, n, out, counters, parity, 1, rf_max_items(), rf_max_items_one(), secs);
}

} // namespace x265hip
