/* x265_oracle.c -- TEST INFRASTRUCTURE ONLY (see x265_oracle.h).
 *
 * CPU restatement of the x265 ME cost path.  Written from the algorithm, not
 * from the reference text: straightforward integer loops, no SWAR, no
 * templates.  "ref:" comments cite /root/reference/source/<file>:<lines>.
 */
#include "x265_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* partitions                                         ref: common/primitives.h:38-46,103-112;
 *                                                         common/primitives.cpp:32-51        */

static const uint8_t k_part_w[X265O_NUM_PARTS] = { 4, 8, 8, 4, 16, 16, 8, 16, 12, 16, 4, 32, 32, 16, 32,
                                                    24, 32, 8, 64, 64, 32, 64, 48, 64, 16 };
static const uint8_t k_part_h[X265O_NUM_PARTS] = { 4, 8, 4, 8, 16, 8, 16, 12, 16, 4, 16, 32, 16, 32, 24,
                                                    32, 8, 32, 64, 32, 64, 48, 64, 16, 64 };

int x265o_partition_from_sizes(int w, int h)
{
    for (int p = 0; p < X265O_NUM_PARTS; p++)
        if (k_part_w[p] == w && k_part_h[p] == h)
            return p;
    return -1;
}

void x265o_partition_size(int part, int *w, int *h)
{
    *w = k_part_w[part];
    *h = k_part_h[part];
}

/* ------------------------------------------------------------------------- */
/* SAD                                                 ref: common/pixel.cpp:92-156 */

static int sad_wh(int w, int h, const uint8_t *a, intptr_t as, const uint8_t *b, intptr_t bs)
{
    int acc = 0;
    for (int y = 0; y < h; y++, a += as, b += bs)
        for (int x = 0; x < w; x++)
        {
            int d = (int)a[x] - (int)b[x];
            acc += d < 0 ? -d : d;
        }
    return acc;
}

int x265o_sad(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    return sad_wh(k_part_w[part], k_part_h[part], fenc, fstride, fref, rstride);
}

/* fenc stride is the fixed FENC_STRIDE = 64 (ref: common/common.h:45, pixel.cpp:126,150) */
void x265o_sad_x3(int part, const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2,
                  intptr_t rstride, int32_t *res)
{
    int w = k_part_w[part], h = k_part_h[part];
    res[0] = sad_wh(w, h, fenc, 64, r0, rstride);
    res[1] = sad_wh(w, h, fenc, 64, r1, rstride);
    res[2] = sad_wh(w, h, fenc, 64, r2, rstride);
}

void x265o_sad_x4(int part, const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2,
                  const uint8_t *r3, intptr_t rstride, int32_t *res)
{
    int w = k_part_w[part], h = k_part_h[part];
    res[0] = sad_wh(w, h, fenc, 64, r0, rstride);
    res[1] = sad_wh(w, h, fenc, 64, r1, rstride);
    res[2] = sad_wh(w, h, fenc, 64, r2, rstride);
    res[3] = sad_wh(w, h, fenc, 64, r3, rstride);
}

/* ------------------------------------------------------------------------- */
/* SATD                                                ref: common/pixel.cpp:181-288, 922-946
 *
 * The reference computes 4x4 (satd_4x4) or paired 8x4 (satd_8x4) Hadamard sums with
 * two 16-bit lanes packed in a uint32 and halves the sum of absolute coefficients once per
 * 4x4 / 8x4 call.  All 16 coefficients of a 4x4 Hadamard share the parity of the block's
 * difference sum, so every per-4x4 sum is even and the halving distributes: the result is
 * sum over 4x4 tiles of (sum|coef| / 2), whatever the grouping.  That is what is coded here. */

static int satd_4x4_tile(const uint8_t *a, intptr_t as, const uint8_t *b, intptr_t bs)
{
    int m[4][4];
    for (int y = 0; y < 4; y++)
    {
        int d0 = (int)a[y * as + 0] - (int)b[y * bs + 0];
        int d1 = (int)a[y * as + 1] - (int)b[y * bs + 1];
        int d2 = (int)a[y * as + 2] - (int)b[y * bs + 2];
        int d3 = (int)a[y * as + 3] - (int)b[y * bs + 3];
        int s01 = d0 + d1, t01 = d0 - d1, s23 = d2 + d3, t23 = d2 - d3;
        m[y][0] = s01 + s23;
        m[y][1] = t01 + t23;
        m[y][2] = s01 - s23;
        m[y][3] = t01 - t23;
    }
    int acc = 0;
    for (int x = 0; x < 4; x++)
    {
        int s01 = m[0][x] + m[1][x], t01 = m[0][x] - m[1][x];
        int s23 = m[2][x] + m[3][x], t23 = m[2][x] - m[3][x];
        acc += abs(s01 + s23) + abs(s01 - s23) + abs(t01 + t23) + abs(t01 - t23);
    }
    return acc >> 1;
}

int x265o_satd(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    int w = k_part_w[part], h = k_part_h[part], acc = 0;
    for (int y = 0; y < h; y += 4)
        for (int x = 0; x < w; x += 4)
            acc += satd_4x4_tile(fenc + y * fstride + x, fstride, fref + y * rstride + x, rstride);
    return acc;
}

/* ------------------------------------------------------------------------- */
/* mode-decision distortions   ref: common/pixel.cpp:158-177 (sse), 290-376 (sa8d), 1095, 1117-1141 (tables)
 *
 * _sa8d_8x8 is the sum of the absolute coefficients of an 8x8 Hadamard transform of the differences; the
 * reference runs it on two 16-bit lanes per uint32, which is exact for 8-bit input (coefficients are at most
 * 64 * 255, a lane's eight absolute values sum to at most sqrt(8) * 64 * 255 < 65536).  Rounding: once per 8x8
 * block (sa8d_8x8, sa8d8<>) or once per 16x16 block over its four raw sums (sa8d_16x16, sa8d16<>). */

static int sa8d_8x8_raw(const uint8_t *a, intptr_t as, const uint8_t *b, intptr_t bs)
{
    int m[8][8];
    for (int y = 0; y < 8; y++)
    {
        for (int x = 0; x < 8; x++)
            m[y][x] = (int)a[y * as + x] - (int)b[y * bs + x];
        for (int s = 1; s < 8; s <<= 1)
            for (int x = 0; x < 8; x++)
                if (!(x & s))
                {
                    int u = m[y][x], v = m[y][x | s];
                    m[y][x] = u + v;
                    m[y][x | s] = u - v;
                }
    }
    int acc = 0;
    for (int x = 0; x < 8; x++)
    {
        for (int s = 1; s < 8; s <<= 1)
            for (int y = 0; y < 8; y++)
                if (!(y & s))
                {
                    int u = m[y][x], v = m[y | s][x];
                    m[y][x] = u + v;
                    m[y | s][x] = u - v;
                }
        for (int y = 0; y < 8; y++)
            acc += abs(m[y][x]);
    }
    return acc;
}

/* sa8d_inter[part]: the table of pixel.cpp:1117-1141 (sa8d16<> where both sides are multiples of 16, sa8d8<>
 * where both are multiples of 8, satd for the rest) */
int x265o_sa8d_inter(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    int w = k_part_w[part], h = k_part_h[part], cost = 0;
    if (!(w & 15) && !(h & 15))
    {
        for (int y = 0; y < h; y += 16)
            for (int x = 0; x < w; x += 16)
            {
                int sum = 0;
                for (int k = 0; k < 4; k++)
                {
                    int ox = x + 8 * (k & 1), oy = y + 8 * (k >> 1);
                    sum += sa8d_8x8_raw(fenc + oy * fstride + ox, fstride, fref + oy * rstride + ox, rstride);
                }
                cost += (sum + 2) >> 2;
            }
        return cost;
    }
    if (!(w & 7) && !(h & 7))
    {
        for (int y = 0; y < h; y += 8)
            for (int x = 0; x < w; x += 8)
                cost += (sa8d_8x8_raw(fenc + y * fstride + x, fstride, fref + y * rstride + x, rstride) + 2) >> 2;
        return cost;
    }
    return x265o_satd(part, fenc, fstride, fref, rstride);
}

/* sse_pp[part]: ref common/pixel.cpp:158-177 */
int x265o_sse_pp(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    int w = k_part_w[part], h = k_part_h[part], sum = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int d = (int)fenc[y * fstride + x] - (int)fref[y * rstride + x];
            sum += d * d;
        }
    return sum;
}

static int block_metric(int metric, int part, int w, int h, const uint8_t *f, intptr_t fs, const uint8_t *r, intptr_t rs);

/* ------------------------------------------------------------------------- */
/* 8-tap luma interpolation         ref: common/ipfilter.cpp:76-160,162-201,241-282,399-406;
 *                                       Lib/TLibCommon/TComRom.cpp:335-341; TComRom.h:117-121 */

static const int16_t k_luma_taps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};

static inline uint8_t clip_u8_from_i16(int v)
{
    /* the reference narrows to int16_t before clamping (ipfilter.cpp:105-109) */
    int16_t s = (int16_t)v;
    return (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
}

void x265o_luma_hpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx)
{
    const int16_t *c = k_luma_taps[coeffIdx];
    int w = k_part_w[part], h = k_part_h[part];
    for (int y = 0; y < h; y++, src += sstride, dst += dstride)
        for (int x = 0; x < w; x++)
        {
            int s = 0;
            for (int k = 0; k < 8; k++)
                s += src[x - 3 + k] * c[k];
            dst[x] = clip_u8_from_i16((s + 32) >> 6);
        }
}

void x265o_luma_vpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx)
{
    const int16_t *c = k_luma_taps[coeffIdx];
    int w = k_part_w[part], h = k_part_h[part];
    for (int y = 0; y < h; y++, src += sstride, dst += dstride)
        for (int x = 0; x < w; x++)
        {
            int s = 0;
            for (int k = 0; k < 8; k++)
                s += src[x + (k - 3) * sstride] * c[k];
            dst[x] = clip_u8_from_i16((s + 32) >> 6);
        }
}

/* 8-bit build: headRoom = 14-8 = 6, shift = 6-6 = 0, offset = -8192 (ipfilter.cpp:121-123) */
void x265o_luma_hps(int part, const uint8_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx,
                    int isRowExt)
{
    const int16_t *c = k_luma_taps[coeffIdx];
    int w = k_part_w[part], h = k_part_h[part];
    if (isRowExt)
    {
        src -= 3 * sstride;
        h += 7;
    }
    for (int y = 0; y < h; y++, src += sstride, dst += dstride)
        for (int x = 0; x < w; x++)
        {
            int s = 0;
            for (int k = 0; k < 8; k++)
                s += src[x - 3 + k] * c[k];
            dst[x] = (int16_t)(s - 8192);
        }
}

/* 8-bit build: shift = 6+6 = 12, offset = (1<<11) + (8192<<6) (ipfilter.cpp:244-247) */
void x265o_luma_vsp(int part, const int16_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx)
{
    const int16_t *c = k_luma_taps[coeffIdx];
    int w = k_part_w[part], h = k_part_h[part];
    for (int y = 0; y < h; y++, src += sstride, dst += dstride)
        for (int x = 0; x < w; x++)
        {
            int s = 0;
            for (int k = 0; k < 8; k++)
                s += src[x + (k - 3) * sstride] * c[k];
            dst[x] = clip_u8_from_i16((s + 2048 + (8192 << 6)) >> 12);
        }
}

/* ref: common/pixel.cpp:726-739 (blockcopy_pp_c<bx,by>) */
void x265o_luma_copy_pp(int part, uint8_t *dst, intptr_t dstride, const uint8_t *src, intptr_t sstride)
{
    int w = k_part_w[part], h = k_part_h[part];
    for (int y = 0; y < h; y++)
        memcpy(dst + y * dstride, src + y * sstride, (size_t)w);
}

/* ------------------------------------------------------------------------- */
/* MV bit-cost table           ref: encoder/bitcost.cpp:31-80; Lib/TLibCommon/TComRom.cpp:127-143 */

static const double k_lambda[70] = {
    0.2500, 0.2806, 0.3150, 0.3536, 0.3969, 0.4454, 0.5000, 0.5612, 0.6300, 0.7071,
    0.7937, 0.8909, 1.0000, 1.1225, 1.2599, 1.4142, 1.5874, 1.7818, 2.0000, 2.2449,
    2.5198, 2.8284, 3.1748, 3.5636, 4.0000, 4.4898, 5.0397, 5.6569, 6.3496, 7.1272,
    8.0000, 8.9797, 10.0794, 11.3137, 12.6992, 14.2544, 16.0000, 17.9594, 20.1587, 22.6274,
    25.3984, 28.5088, 32.0000, 35.9188, 40.3175, 45.2548, 50.7968, 57.0175, 64.0000, 71.8376,
    80.6349, 90.5097, 101.5937, 114.0350, 128.0000, 143.6751, 161.2699, 181.0193, 203.1873, 228.0701,
    256.0000, 287.3503, 322.5398, 362.0387, 406.3747, 456.1401, 512.0000, 574.7006, 645.0796, 724.0773
};

void x265o_bitcost_table(int qp, uint16_t *out)
{
    /* bits[0] = 0.718, bits[i] = log(i+1) * (2/ln 2) + 1.718, stored as float; the reference's
     * unqualified log() calls resolve to the double overload, so the logarithm and the
     * multiply-add are evaluated in double and only the stored results are single precision.
     * cost[+-i] = (uint16) min(bits[i] * lambda + 0.5, 65535) with lambda in double. */
    const float two_over_ln2 = (float)(2.0f / log((double)2.0f));
    const double lambda = k_lambda[qp];
    for (int i = 0; i < 32768; i++)
    {
        float bits = i ? (float)(log((double)(float)(i + 1)) * two_over_ln2 + 1.718f) : 0.718f;
        double v = bits * lambda + 0.5f;
        uint16_t c = (uint16_t)(v < 65535.0 ? v : 65535.0);
        out[32768 + i] = c;
        out[32768 - i] = c;
    }
    /* entry -32768 is never written by the reference loop (i < BC_MAX_MV); index 0 of our
     * array corresponds to it.  Give it the value of its neighbour so the array is defined. */
    out[0] = out[1];
}

/* ------------------------------------------------------------------------- */
/* search context                                                            */

typedef struct
{
    uint8_t fenc[64 * 64]; /* PU copied at stride 64   ref: encoder/motion.cpp:110-126 */
    int w, h, part;
    const uint8_t *fref;   /* ref plane at the PU origin   ref: motion.cpp:537 */
    intptr_t stride;
    const uint16_t *tab;   /* 65536 entries, centre at +32768 */
    int mvpx, mvpy;        /* setMVP bias                  ref: encoder/bitcost.h:42 */
    const uint8_t *lowres[4]; /* ReferencePlanes::lowresPlane[] at the PU origin when ref->isLowres, else NULL */
} me_ctx;

/* mvcost() sums two uint16 entries and truncates to uint16 (bitcost.h:45).  Indices are
 * wrapped to the table so that the reference's out-of-table reads (undefined there) are at
 * least defined here; in-range inputs never wrap. */
static inline int tab_at(const uint16_t *tab, int mvp, int v)
{
    return tab[(uint16_t)(v - mvp + 32768)];
}

static inline int mvcost_q(const me_ctx *c, int qx, int qy)
{
    return (uint16_t)(tab_at(c->tab, c->mvpx, qx) + tab_at(c->tab, c->mvpy, qy));
}

static inline int16_t s16(int v) { return (int16_t)v; }

/* Work model for the roofline of the search kernel (bench.py): the candidates a search actually evaluates, priced
 * at SURVEY.md 8(d)'s algorithmic lane-ops: SAD 3 per pixel, SATD 111 per 4x4 tile, hpp / vpp 8 MAC + 3 per pixel,
 * hps + vsp 8 (H + 7) / H + 8 MAC + 3 per pixel.  Per thread; summed by x265o_me_batch_ops. */
static __thread double t_ops_sad, t_ops_satd, t_ops_interp;
static __thread long t_n_sad, t_n_sub;

/* cost of a full-pel position given in full-pel units: SAD + mvcost(mv << 2)
 * (ref: COST_MV, motion.cpp:179-185).  MV << 2 is an int16 operation (mv.h:63). */
static inline int sad_at(const me_ctx *c, int mx, int my)
{
    t_ops_sad += 3.0 * c->w * c->h;
    t_n_sad++;
    return sad_wh(c->w, c->h, c->fenc, 64, c->fref + mx + my * c->stride, c->stride);
}

static inline int cost_fpel(const me_ctx *c, int mx, int my)
{
    return sad_at(c, mx, my) + mvcost_q(c, s16(mx << 2), s16(my << 2));
}

/* ref: ReferencePlanes::lowresQPelCost, common/lowres.h:73-93; pixelavg_pp, common/pixel.cpp:546-560.
 * The reference always averages an 8x8 block here (the lookahead's only block size). */
static int lowres_qpel_cost(const me_ctx *c, int qx, int qy, int use_satd)
{
    const uint8_t *cmp;
    intptr_t cmpstride;
    uint8_t buf[8 * 8];
    if ((qx | qy) & 1)
    {
        int hpelA = (qy & 2) | ((qx & 2) >> 1);
        const uint8_t *a = c->lowres[hpelA] + (qx >> 2) + (qy >> 2) * c->stride;
        int bx = s16(qx + (qx & 1) * 2), by = s16(qy + (qy & 1) * 2);
        int hpelB = (by & 2) | ((bx & 2) >> 1);
        const uint8_t *b = c->lowres[hpelB] + (bx >> 2) + (by >> 2) * c->stride;
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++)
                buf[y * 8 + x] = (uint8_t)((a[y * c->stride + x] + b[y * c->stride + x] + 1) >> 1);
        cmp = buf;
        cmpstride = 8;
    }
    else
    {
        int hpel = (qy & 2) | ((qx & 2) >> 1);
        cmp = c->lowres[hpel] + (qx >> 2) + (qy >> 2) * c->stride;
        cmpstride = c->stride;
    }
    return use_satd ? x265o_satd(c->part, c->fenc, 64, cmp, cmpstride) : sad_wh(c->w, c->h, c->fenc, 64, cmp, cmpstride);
}

/* ref: MotionEstimate::subpelCompare, motion.cpp:1126-1164 (lowres references: lowresQPelCost instead,
 * motion.cpp:556-557, 587-588, 1055-1080) */
static int subpel_cost(const me_ctx *c, int qx, int qy, int use_satd)
{
    if (c->lowres[0])
        return lowres_qpel_cost(c, qx, qy, use_satd);
    int xf = qx & 3, yf = qy & 3;
    {
        const double px = (double)c->w * c->h;
        if (use_satd)
            t_ops_satd += px * (111.0 / 16);
        else
            t_ops_sad += 3.0 * px;
        t_ops_interp += px * ((xf && yf) ? 8.0 * (c->h + 7) / c->h + 8 + 3 : (xf || yf) ? 11.0 : 0.0);
        t_n_sub++;
    }
    const uint8_t *p = c->fref + (qx >> 2) + (qy >> 2) * c->stride;
    uint8_t buf[64 * 64];
    const uint8_t *cmp = buf;
    intptr_t cmpstride = 64;
    if (!(xf | yf))
    {
        cmp = p;
        cmpstride = c->stride;
    }
    else if (!yf)
        x265o_luma_hpp(c->part, p, c->stride, buf, 64, xf);
    else if (!xf)
        x265o_luma_vpp(c->part, p, c->stride, buf, 64, yf);
    else
    {
        int16_t immed[64 * (64 + 8)];
        x265o_luma_hps(c->part, p, c->stride, immed, c->w, xf, 1);
        x265o_luma_vsp(c->part, immed + 3 * c->w, c->w, buf, 64, yf);
    }
    return block_metric(use_satd, c->part, c->w, c->h, c->fenc, 64, cmp, cmpstride);
}

/* the comparison subpelCompare is handed (motion.cpp:1126): 0 sad, 1 satd, 2 sa8d_inter, 3 sse_pp */
static int block_metric(int metric, int part, int w, int h, const uint8_t *f, intptr_t fs, const uint8_t *r, intptr_t rs)
{
    switch (metric)
    {
    case X265O_SATD: return x265o_satd(part, f, fs, r, rs);
    case X265O_SA8D: return x265o_sa8d_inter(part, f, fs, r, rs);
    case X265O_SSE: return x265o_sse_pp(part, f, fs, r, rs);
    default: return sad_wh(w, h, f, fs, r, rs);
    }
}

typedef struct
{
    int16_t x, y;
} mv_t;

static inline int in_range(int x, int y, mv_t lo, mv_t hi)
{
    return x >= lo.x && x <= hi.x && y >= lo.y && y <= hi.y;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* tables                                                ref: motion.cpp:129-149 */
static const int8_t k_hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
static const uint8_t k_mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
static const int8_t k_square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
static const int8_t k_hex4[16][2] = { { 0, -4 }, { 0, 4 }, { -2, -3 }, { 2, -3 }, { -4, -2 }, { 4, -2 }, { -4, -1 }, { 4, -1 },
                                      { -4, 0 }, { 4, 0 }, { -4, 1 }, { 4, 1 }, { -4, 2 }, { 4, 2 }, { -2, 3 }, { 2, 3 } };
static const int8_t k_twopoint[16][2] = { { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 },
                                          { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 } };
/* ref: motion.cpp:47-57 */
static const struct { int hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd; } k_workload[8] = {
    { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
    { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 },
};

/* search state shared by the pattern helpers */
typedef struct
{
    const me_ctx *c;
    mv_t lo, hi;   /* full-pel range */
    int bcost;
    mv_t bmv;      /* full-pel during the integer stage */
    int point, dist; /* STAR bookkeeping */
} srch;

/* one scalar candidate, strict less (COPY2_IF_LT, common.h:123-128) */
static inline void try_mv(srch *s, int mx, int my)
{
    int cost = cost_fpel(s->c, mx, my);
    if (cost < s->bcost)
    {
        s->bcost = cost;
        s->bmv.x = s16(mx);
        s->bmv.y = s16(my);
    }
}

static inline void try_mv_pd(srch *s, int mx, int my, int point, int dist)
{
    int cost = cost_fpel(s->c, mx, my);
    if (cost < s->bcost)
    {
        s->bcost = cost;
        s->bmv.x = s16(mx);
        s->bmv.y = s16(my);
        s->point = point;
        s->dist = dist;
    }
}

/* four candidates around origin o, in the listed order (COST_MV_X4, motion.cpp:218-235) */
static void try_x4(srch *s, mv_t o, const int d[4][2])
{
    for (int k = 0; k < 4; k++)
        try_mv(s, s16(o.x + d[k][0]), s16(o.y + d[k][1]));
}

/* ref: StarPatternSearch, motion.cpp:284-524.  Coordinates are int16 in the reference.
 * The reference has a sad_x4 fast path when the whole ring is inside [lo,hi] and a guarded scalar
 * path otherwise; both visit the same points in the same order, so only the guarded form is
 * written here (inside-ness of the outer points implies it for the inner ones). */
static void star_pattern(srch *s, int early_exit_iters, int merange)
{
    const mv_t o = s->bmv;
    const mv_t lo = s->lo, hi = s->hi;
    int saved = s->bcost;
    int rounds = 0;

    for (int dist = 1; dist <= 8; dist <<= 1)
    {
        const int half = dist >> 1;
        const int16_t top = s16(o.y - dist), bottom = s16(o.y + dist), left = s16(o.x - dist), right = s16(o.x + dist);
        const int16_t top2 = s16(o.y - half), bottom2 = s16(o.y + half), left2 = s16(o.x - half), right2 = s16(o.x + half);
        saved = s->bcost;
        /* visiting order 2,1,3,4,5,6,8,7; points 1,3,6,8 exist only for dist >= 2 (:317-419) */
        if (top >= lo.y)
            try_mv_pd(s, o.x, top, 2, dist);
        if (dist > 1 && top2 >= lo.y)
        {
            if (left2 >= lo.x)
                try_mv_pd(s, left2, top2, 1, half);
            if (right2 <= hi.x)
                try_mv_pd(s, right2, top2, 3, half);
        }
        if (left >= lo.x)
            try_mv_pd(s, left, o.y, 4, dist);
        if (right <= hi.x)
            try_mv_pd(s, right, o.y, 5, dist);
        if (dist > 1 && bottom2 <= hi.y)
        {
            if (left2 >= lo.x)
                try_mv_pd(s, left2, bottom2, 6, half);
            if (right2 <= hi.x)
                try_mv_pd(s, right2, bottom2, 8, half);
        }
        if (bottom <= hi.y)
            try_mv_pd(s, o.x, bottom, 7, dist);
        if (s->bcost < saved)
            rounds = 0;
        else if (++rounds >= early_exit_iters)
            return;
    }

    for (int dist = 16; dist <= (int16_t)merange; dist <<= 1)
    {
        const int q = dist >> 2;
        const int16_t top = s16(o.y - dist), bottom = s16(o.y + dist), left = s16(o.x - dist), right = s16(o.x + dist);
        saved = s->bcost;
        if (top >= lo.y)
            try_mv_pd(s, o.x, top, 0, dist);
        if (left >= lo.x)
            try_mv_pd(s, left, o.y, 0, dist);
        if (right <= hi.x)
            try_mv_pd(s, right, o.y, 0, dist);
        if (bottom <= hi.y)
            try_mv_pd(s, o.x, bottom, 0, dist);
        for (int idx = 1; idx < 4; idx++)
        {
            const int16_t yt = s16(top + q * idx), yb = s16(bottom - q * idx);
            const int16_t xl = s16(o.x - q * idx), xr = s16(o.x + q * idx);
            if (yt >= lo.y)
            {
                if (xl >= lo.x)
                    try_mv_pd(s, xl, yt, 0, dist);
                if (xr <= hi.x)
                    try_mv_pd(s, xr, yt, 0, dist);
            }
            if (yb <= hi.y)
            {
                if (xl >= lo.x)
                    try_mv_pd(s, xl, yb, 0, dist);
                if (xr <= hi.x)
                    try_mv_pd(s, xr, yb, 0, dist);
            }
        }
        if (s->bcost < saved)
            rounds = 0;
        else if (++rounds >= early_exit_iters)
            return;
    }
}

/* UMH cross (CROSS macro, motion.cpp:258-282) around origin o */
static void umh_cross(srch *s, mv_t o, int start, int x_max, int y_max)
{
    const mv_t lo = s->lo, hi = s->hi;
    int16_t i = s16(start);
    if (x_max <= imin(hi.x - o.x, o.x - lo.x))
        for (; i < x_max - 2; i = s16(i + 4))
        {
            const int d[4][2] = { { i, 0 }, { -i, 0 }, { i + 2, 0 }, { -i - 2, 0 } };
            try_x4(s, o, d);
        }
    for (; i < x_max; i = s16(i + 2))
    {
        if (o.x + i <= hi.x)
            try_mv(s, o.x + i, o.y);
        if (o.x - i >= lo.x)
            try_mv(s, o.x - i, o.y);
    }
    i = s16(start);
    if (y_max <= imin(hi.y - o.y, o.y - lo.y))
        for (; i < y_max - 2; i = s16(i + 4))
        {
            const int d[4][2] = { { 0, i }, { 0, -i }, { 0, i + 2 }, { 0, -i - 2 } };
            try_x4(s, o, d);
        }
    for (; i < y_max; i = s16(i + 2))
    {
        if (o.y + i <= hi.y)
            try_mv(s, o.x, o.y + i);
        if (o.y - i >= lo.y)
            try_mv(s, o.x, o.y - i);
    }
}

/* ref: MotionEstimate::motionEstimate, motion.cpp:526-1124.  With X265O_LOWRES_REF set in j->method the
 * reference is a Lowres (ref->isLowres): refpl[0..3] are lowresPlane[0..3] (same stride), the block is the
 * lookahead's 8x8 CU and sub-pel positions come from the four half-pel planes. */
static int motion_estimate(const x265o_plane *fencpl, const x265o_plane *refpl, const uint16_t *costtab,
                           const x265o_me_job *j, mv_t *out)
{
    me_ctx c;
    const int lowres = (j->method & X265O_LOWRES_REF) != 0;
    const int method = j->method & ~X265O_LOWRES_REF;
    for (int i = 0; i < 4; i++)
        c.lowres[i] = lowres ? refpl[i].origin + j->pu_x + (intptr_t)j->pu_y * refpl[0].stride : 0;
    c.w = j->pu_w;
    c.h = j->pu_h;
    c.part = x265o_partition_from_sizes(c.w, c.h);
    c.stride = refpl->stride;
    c.fref = refpl->origin + j->pu_x + (intptr_t)j->pu_y * refpl->stride;
    c.tab = costtab;
    c.mvpx = j->qmvp_x; /* setMVP, bitcost.h:42 */
    c.mvpy = j->qmvp_y;
    /* setSourcePU: copy the PU into the 64-stride cache (motion.cpp:125) */
    x265o_luma_copy_pp(c.part, c.fenc, 64, fencpl->origin + j->pu_x + (intptr_t)j->pu_y * fencpl->stride,
                       fencpl->stride);

    const mv_t lo = { j->mvmin_x, j->mvmin_y }, hi = { j->mvmax_x, j->mvmax_y };
    const int qlox = s16(lo.x << 2), qloy = s16(lo.y << 2), qhix = s16(hi.x << 2), qhiy = s16(hi.y << 2);
    int merange = j->merange;

    /* clipped quarter-pel predictor, measured with SAD and no mvcost (motion.cpp:552-559).
     * MV::clipped = min with max first, then max with min (mv.h:89-94). */
    mv_t pmv;
    pmv.x = s16(j->qmvp_x > qhix ? qhix : j->qmvp_x);
    pmv.x = s16(pmv.x < qlox ? qlox : pmv.x);
    pmv.y = s16(j->qmvp_y > qhiy ? qhiy : j->qmvp_y);
    pmv.y = s16(pmv.y < qloy ? qloy : pmv.y);
    mv_t bestpre = pmv;
    int bprecost = subpel_cost(&c, pmv.x, pmv.y, 0);

    srch s;
    s.c = &c;
    s.lo = lo;
    s.hi = hi;
    s.point = 0;
    s.dist = 0;
    /* full-pel start = (pmv + 2) >> 2, re-measured with mvcost if pmv was fractional (:562-567) */
    s.bmv.x = s16(s16(pmv.x + 2) >> 2);
    s.bmv.y = s16(s16(pmv.y + 2) >> 2);
    s.bcost = bprecost;
    if ((pmv.x | pmv.y) & 3)
        s.bcost = cost_fpel(&c, s.bmv.x, s.bmv.y);

    /* zero MV (:570-578) */
    if (pmv.x | pmv.y)
    {
        int cost = sad_at(&c, 0, 0) + mvcost_q(&c, 0, 0);
        if (cost < s.bcost)
        {
            s.bcost = cost;
            s.bmv.x = s.bmv.y = 0;
        }
    }

    /* candidate predictors at quarter-pel, SAD + mvcost, compete with bprecost only (:581-598) */
    for (int i = 0; i < j->num_mvc; i++)
    {
        int mx = j->mvc[2 * i], my = j->mvc[2 * i + 1];
        mx = mx > qhix ? qhix : mx;
        mx = mx < qlox ? qlox : mx;
        my = my > qhiy ? qhiy : my;
        my = my < qloy ? qloy : my;
        if ((mx | my) && !(mx == pmv.x && my == pmv.y) && !(mx == bestpre.x && my == bestpre.y))
        {
            int cost = subpel_cost(&c, mx, my, 0) + mvcost_q(&c, mx, my);
            if (cost < bprecost)
            {
                bprecost = cost;
                bestpre.x = s16(mx);
                bestpre.y = s16(my);
            }
        }
    }

    pmv.x = s16(s16(pmv.x + 2) >> 2);
    pmv.y = s16(s16(pmv.y + 2) >> 2);
    mv_t omv = s.bmv;

    switch (method)
    {
    case 0: /* DIA, motion.cpp:605-626: packed (cost<<4)+code, centre wins ties, lowest code next */
    {
        int i = merange;
        do
        {
            int best = s.bcost << 4;
            static const int dx[4] = { 0, 0, -1, 1 }, dy[4] = { -1, 1, 0, 0 }, code[4] = { 1, 3, 4, 12 };
            int k, sel = -1;
            for (k = 0; k < 4; k++)
            {
                int v = (cost_fpel(&c, s16(s.bmv.x + dx[k]), s16(s.bmv.y + dy[k])) << 4) + code[k];
                if (v < best)
                {
                    best = v;
                    sel = k;
                }
            }
            if (sel < 0)
                break;
            s.bmv.x = s16(s.bmv.x + dx[sel]);
            s.bmv.y = s16(s.bmv.y + dy[sel]);
            s.bcost = best >> 4;
        }
        while (--i && in_range(s.bmv.x, s.bmv.y, lo, hi));
        break;
    }

    case 2: /* UMH, motion.cpp:702-885; falls into the hexagon stage when bmv is in range */
    {
        int ucost1, ucost2;
        int cross_start = 1;
        int leave = 0;
        static const int dia[4][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 } };
        const int sscale = (c.h * c.h) >> 4; /* size_scale, motion.cpp:60-66 */
#define THRESH(v) (s.bcost < (((v) >> 4) * sscale))

        omv = s.bmv;
        ucost1 = s.bcost;
        try_x4(&s, pmv, dia);
        if (pmv.x | pmv.y)
        {
            mv_t z = { 0, 0 };
            try_x4(&s, z, dia);
        }
        ucost2 = s.bcost;
        if ((s.bmv.x | s.bmv.y) && !(s.bmv.x == pmv.x && s.bmv.y == pmv.y))
            try_x4(&s, s.bmv, dia);
        if (s.bcost == ucost2)
            cross_start = 3;

        omv = s.bmv;
        if (s.bcost == ucost2 && THRESH(2000))
        {
            static const int o1[4][2] = { { 0, -2 }, { -1, -1 }, { 1, -1 }, { -2, 0 } };
            static const int o2[4][2] = { { 2, 0 }, { -1, 1 }, { 1, 1 }, { 0, 2 } };
            try_x4(&s, omv, o1);
            try_x4(&s, omv, o2);
            if (s.bcost == ucost1 && THRESH(500))
                leave = 1;
            else if (s.bcost == ucost2)
            {
                int16_t range = s16((merange >> 1) | 1);
                static const int o3[4][2] = { { -1, -2 }, { 1, -2 }, { -2, -1 }, { 2, -1 } };
                static const int o4[4][2] = { { -2, 1 }, { 2, 1 }, { -1, 2 }, { 1, 2 } };
                umh_cross(&s, omv, 3, range, range);
                try_x4(&s, omv, o3);
                try_x4(&s, omv, o4);
                if (s.bcost == ucost2)
                    leave = 1;
                else
                    cross_start = range + 2;
            }
        }
        if (leave)
            break;

        if (j->num_mvc)
        {
            static const uint8_t range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };
            int mvd, denom = 1;
            const int is64 = (c.w == 64 && c.h == 64);
            if (j->num_mvc == 1)
                mvd = is64 ? 25 : abs(j->qmvp_x - j->mvc[0]) + abs(j->qmvp_y - j->mvc[1]);
            else
            {
                denom = j->num_mvc - 1;
                mvd = 0;
                if (!is64)
                {
                    mvd = abs(j->qmvp_x - j->mvc[0]) + abs(j->qmvp_y - j->mvc[1]);
                    denom++;
                }
                for (int i = 0; i < j->num_mvc - 1; i++)
                    mvd += abs(j->mvc[2 * i] - j->mvc[2 * i + 2]) + abs(j->mvc[2 * i + 1] - j->mvc[2 * i + 3]);
            }
            int sad_ctx = THRESH(1000) ? 0 : THRESH(2000) ? 1 : THRESH(4000) ? 2 : 3;
            int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            merange = (merange * range_mul[mvd_ctx][sad_ctx]) >> 2;
        }
#undef THRESH

        {
            static const int o5[4][2] = { { -2, -2 }, { -2, 2 }, { 2, -2 }, { 2, 2 } };
            umh_cross(&s, omv, cross_start, merange, merange >> 1);
            try_x4(&s, omv, o5);
        }

        /* hexagon grid, rings i = 1 .. merange/4 around the new centre (:802-881).  The in-range
         * branch adds the two table entries as ints (no uint16 truncation of their sum). */
        omv = s.bmv;
        {
            uint16_t i = 1;
            do
            {
                int room = imin(imin(hi.x - omv.x, omv.x - lo.x), imin(hi.y - omv.y, omv.y - lo.y));
                if (4 * i > room)
                {
                    for (int k = 0; k < 16; k++)
                    {
                        int16_t mx = s16(omv.x + s16(k_hex4[k][0] * (int16_t)i));
                        int16_t my = s16(omv.y + s16(k_hex4[k][1] * (int16_t)i));
                        if (in_range(mx, my, lo, hi))
                            try_mv(&s, mx, my);
                    }
                }
                else
                {
                    int dirx = 0, diry = 0, moved = 0;
                    for (int k = 0; k < 16; k++)
                    {
                        int ox = k_hex4[k][0] * i, oy = k_hex4[k][1] * i;
                        int cost = sad_at(&c, omv.x + ox, omv.y + oy)
                                   + tab_at(c.tab, c.mvpx, omv.x * 4 + ox * 4) + tab_at(c.tab, c.mvpy, omv.y * 4 + oy * 4);
                        if (cost < s.bcost)
                        {
                            s.bcost = cost;
                            dirx = k_hex4[k][0];
                            diry = k_hex4[k][1];
                            moved = 1;
                        }
                    }
                    if (moved)
                    {
                        s.bmv.x = s16(omv.x + i * dirx);
                        s.bmv.y = s16(omv.y + i * diry);
                    }
                }
            }
            while (++i <= merange >> 2);
        }
        if (!in_range(s.bmv.x, s.bmv.y, lo, hi))
            break;
    }
    /* fall through */
    case 1: /* HEX, motion.cpp:628-700 */
    {
        /* first full hexagon: codes 2..7 in the order (-2,0)(-1,2)(1,2)(2,0)(1,-2)(-1,-2) */
        static const int first[6][2] = { { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 } };
        int best = s.bcost << 3;
        for (int k = 0; k < 6; k++)
        {
            int v = (cost_fpel(&c, s16(s.bmv.x + first[k][0]), s16(s.bmv.y + first[k][1])) << 3) + (k + 2);
            if (v < best)
                best = v;
        }
        if (best & 7)
        {
            int dir = (best & 7) - 2;
            s.bmv.x = s16(s.bmv.x + k_hex2[dir + 1][0]);
            s.bmv.y = s16(s.bmv.y + k_hex2[dir + 1][1]);
            for (int i = (merange >> 1) - 1; i > 0 && in_range(s.bmv.x, s.bmv.y, lo, hi); i--)
            {
                best &= ~7;
                for (int k = 0; k < 3; k++)
                {
                    int v = (cost_fpel(&c, s16(s.bmv.x + k_hex2[dir + k][0]), s16(s.bmv.y + k_hex2[dir + k][1])) << 3) + (k + 1);
                    if (v < best)
                        best = v;
                }
                if (!(best & 7))
                    break;
                dir += (best & 7) - 2;
                dir = k_mod6m1[dir + 1];
                s.bmv.x = s16(s.bmv.x + k_hex2[dir + 1][0]);
                s.bmv.y = s16(s.bmv.y + k_hex2[dir + 1][1]);
            }
        }
        s.bcost = best >> 3;

        /* square refine, 8 neighbours, fixed centre, lowest index wins ties (:686-698) */
        int dir = 0;
        for (int k = 1; k <= 8; k++)
        {
            int cost = cost_fpel(&c, s16(s.bmv.x + k_square1[k][0]), s16(s.bmv.y + k_square1[k][1]));
            if (cost < s.bcost)
            {
                s.bcost = cost;
                dir = k;
            }
        }
        s.bmv.x = s16(s.bmv.x + k_square1[dir][0]);
        s.bmv.y = s16(s.bmv.y + k_square1[dir][1]);
        break;
    }

    case 3: /* STAR, motion.cpp:887-995 */
    {
        s.point = 0;
        s.dist = 0;
        star_pattern(&s, 3, merange);
        int done = 0;
        if (s.dist == 1)
        {
            if (s.point)
            {
                int saved = s.bcost;
                int x1 = s16(s.bmv.x + k_twopoint[(s.point - 1) * 2][0]), y1 = s16(s.bmv.y + k_twopoint[(s.point - 1) * 2][1]);
                int x2 = s16(s.bmv.x + k_twopoint[(s.point - 1) * 2 + 1][0]), y2 = s16(s.bmv.y + k_twopoint[(s.point - 1) * 2 + 1][1]);
                if (in_range(x1, y1, lo, hi))
                    try_mv(&s, x1, y1);
                if (in_range(x2, y2, lo, hi))
                    try_mv(&s, x2, y2);
                if (s.bcost == saved)
                    done = 1;
            }
            else
                done = 1;
        }
        if (done)
            break;

        if (s.dist > 5)
        {
            /* raster at step 5; groups of four when three more steps fit, and the 4th point of a
             * group is charged mvcost(tmv << 3) -- a reference quirk kept on purpose (:951) */
            for (int ty = lo.y; ty <= hi.y; ty += 5)
            {
                int16_t tx = lo.x;
                for (; tx <= hi.x; tx = s16(tx + 5))
                {
                    if (tx + 15 <= hi.x)
                    {
                        for (int k = 0; k < 4; k++)
                        {
                            int16_t px = s16(tx + 5 * k);
                            int sh = k == 3 ? 3 : 2;
                            int cost = sad_at(&c, px, ty) + mvcost_q(&c, s16(px << sh), s16(ty << sh));
                            if (cost < s.bcost)
                            {
                                s.bcost = cost;
                                s.bmv.x = px;
                                s.bmv.y = s16(ty);
                            }
                        }
                        tx = s16(tx + 15);
                    }
                    else
                        try_mv(&s, tx, ty);
                }
            }
        }

        while (s.dist > 0)
        {
            s.dist = 0;
            s.point = 0;
            star_pattern(&s, 32, merange);
            if (s.dist == 1)
            {
                if (!s.point)
                    break;
                int x1 = s16(s.bmv.x + k_twopoint[(s.point - 1) * 2][0]), y1 = s16(s.bmv.y + k_twopoint[(s.point - 1) * 2][1]);
                int x2 = s16(s.bmv.x + k_twopoint[(s.point - 1) * 2 + 1][0]), y2 = s16(s.bmv.y + k_twopoint[(s.point - 1) * 2 + 1][1]);
                if (in_range(x1, y1, lo, hi))
                    try_mv(&s, x1, y1);
                if (in_range(x2, y2, lo, hi))
                    try_mv(&s, x2, y2);
                break;
            }
        }
        break;
    }

    case 4: /* FULL, motion.cpp:997-1032: raster order, first minimum wins */
        for (int ty = lo.y; ty <= hi.y; ty++)
            for (int tx = lo.x; tx <= hi.x; tx++)
                try_mv(&s, tx, ty);
        break;

    default:
        break;
    }

    /* pre-test winner bypasses the integer result (:1039-1045) */
    int bcost = s.bcost;
    int bx, by;
    if (bprecost < bcost)
    {
        bx = bestpre.x;
        by = bestpre.y;
        bcost = bprecost;
    }
    else
    {
        bx = s16(s.bmv.x << 2);
        by = s16(s.bmv.y << 2);
    }

    if (!bcost)
    {
        bcost = mvcost_q(&c, bx, by); /* :1049-1054 */
    }
    else if (lowres)
    {
        /* motion.cpp:1055-1080: one half-pel ring with SAD, SATD re-measure, one quarter-pel ring with SATD */
        int bdir = 0;
        for (int i = 1; i <= k_workload[j->subme].hpel_dirs; i++)
        {
            int qx = s16(bx + k_square1[i][0] * 2), qy = s16(by + k_square1[i][1] * 2);
            int cost = subpel_cost(&c, qx, qy, 0) + mvcost_q(&c, qx, qy);
            if (cost < bcost)
            {
                bcost = cost;
                bdir = i;
            }
        }
        bx = s16(bx + k_square1[bdir][0] * 2);
        by = s16(by + k_square1[bdir][1] * 2);
        bcost = subpel_cost(&c, bx, by, 1) + mvcost_q(&c, bx, by);
        bdir = 0;
        for (int i = 1; i <= k_workload[j->subme].qpel_dirs; i++)
        {
            int qx = s16(bx + k_square1[i][0]), qy = s16(by + k_square1[i][1]);
            int cost = subpel_cost(&c, qx, qy, 1) + mvcost_q(&c, qx, qy);
            if (cost < bcost)
            {
                bcost = cost;
                bdir = i;
            }
        }
        bx = s16(bx + k_square1[bdir][0]);
        by = s16(by + k_square1[bdir][1]);
    }
    else
    {
        const int hsatd = k_workload[j->subme].hpel_satd;
        if (hsatd)
            bcost = subpel_cost(&c, bx, by, 1) + mvcost_q(&c, bx, by);
        for (int it = 0; it < k_workload[j->subme].hpel_iters; it++)
        {
            int bdir = 0;
            for (int i = 1; i <= k_workload[j->subme].hpel_dirs; i++)
            {
                int qx = s16(bx + k_square1[i][0] * 2), qy = s16(by + k_square1[i][1] * 2);
                int cost = subpel_cost(&c, qx, qy, hsatd) + mvcost_q(&c, qx, qy);
                if (cost < bcost)
                {
                    bcost = cost;
                    bdir = i;
                }
            }
            bx = s16(bx + k_square1[bdir][0] * 2);
            by = s16(by + k_square1[bdir][1] * 2);
        }
        if (!hsatd)
            bcost = subpel_cost(&c, bx, by, 1) + mvcost_q(&c, bx, by);
        for (int it = 0; it < k_workload[j->subme].qpel_iters; it++)
        {
            int bdir = 0;
            for (int i = 1; i <= k_workload[j->subme].qpel_dirs; i++)
            {
                int qx = s16(bx + k_square1[i][0]), qy = s16(by + k_square1[i][1]);
                int cost = subpel_cost(&c, qx, qy, 1) + mvcost_q(&c, qx, qy);
                if (cost < bcost)
                {
                    bcost = cost;
                    bdir = i;
                }
            }
            bx = s16(bx + k_square1[bdir][0]);
            by = s16(by + k_square1[bdir][1]);
        }
    }
    out->x = s16(bx);
    out->y = s16(by);
    return bcost;
}

/* ------------------------------------------------------------------------- */
/* public single-call helpers                                                */

int x265o_subpel_cost(const x265o_plane *fenc, const x265o_plane *ref, int x, int y, int w, int h, int qmvx,
                      int qmvy, int metric)
{
    me_ctx c;
    c.lowres[0] = c.lowres[1] = c.lowres[2] = c.lowres[3] = 0;
    c.w = w;
    c.h = h;
    c.part = x265o_partition_from_sizes(w, h);
    c.stride = ref->stride;
    c.fref = ref->origin + x + (intptr_t)y * ref->stride;
    c.tab = 0;
    c.mvpx = c.mvpy = 0;
    x265o_luma_copy_pp(c.part, c.fenc, 64, fenc->origin + x + (intptr_t)y * fenc->stride, fenc->stride);
    return subpel_cost(&c, qmvx, qmvy, metric);
}

/* Bidirectional candidate, ref: TEncSearch::predInterSearch, Lib/TLibEncoder/TEncSearch.cpp:2360-2412:
 * two uni-directional luma predictions (xPredInterLumaBlk, Lib/TLibCommon/TComPrediction.cpp:400-433: copy /
 * luma_hpp / luma_vpp / luma_hps + luma_vsp, the pixels subpelCompare compares), pixelavg_pp
 * (common/pixel.cpp:546-560), then sad / satd against the block. */
static void predict_luma(int part, int w, const uint8_t *p, intptr_t stride, int qx, int qy, uint8_t *dst)
{
    int xf = qx & 3, yf = qy & 3;
    p += (qx >> 2) + (qy >> 2) * stride;
    if (!(xf | yf))
        x265o_luma_copy_pp(part, dst, 64, p, stride);
    else if (!yf)
        x265o_luma_hpp(part, p, stride, dst, 64, xf);
    else if (!xf)
        x265o_luma_vpp(part, p, stride, dst, 64, yf);
    else
    {
        int16_t immed[64 * (64 + 8)];
        x265o_luma_hps(part, p, stride, immed, w, xf, 1);
        x265o_luma_vsp(part, immed + 3 * w, w, dst, 64, yf);
    }
}

static void predict_luma_short(int part, int w, int h, const uint8_t *p, intptr_t stride, int qx, int qy, int16_t *dst);

int x265o_bidir_cost(const x265o_plane *fenc, const x265o_plane *ref0, const x265o_plane *ref1, int x, int y, int w,
                     int h, int qmv0x, int qmv0y, int qmv1x, int qmv1y, int metric)
{
    uint8_t p0[64 * 64], p1[64 * 64], avg[64 * 64];
    int part = x265o_partition_from_sizes(w, h);
    const uint8_t *f = fenc->origin + x + (intptr_t)y * fenc->stride;
    if (metric & X265O_MC_AVG)
    {
        /* the prediction motion compensation makes of the pair (xPredInterBi, TComPrediction.cpp:330-355): what
         * xMergeEstimation compares (TEncSearch.cpp:2209-2210) */
        int16_t s0[64 * 64], s1[64 * 64];
        predict_luma_short(part, w, h, ref0->origin + x + (intptr_t)y * ref0->stride, ref0->stride, qmv0x, qmv0y, s0);
        predict_luma_short(part, w, h, ref1->origin + x + (intptr_t)y * ref1->stride, ref1->stride, qmv1x, qmv1y, s1);
        x265o_luma_addavg(part, s0, s1, avg, 64, 64, 64);
        return block_metric(metric & 3, part, w, h, f, fenc->stride, avg, 64);
    }
    predict_luma(part, w, ref0->origin + x + (intptr_t)y * ref0->stride, ref0->stride, qmv0x, qmv0y, p0);
    predict_luma(part, w, ref1->origin + x + (intptr_t)y * ref1->stride, ref1->stride, qmv1x, qmv1y, p1);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++)
            avg[r * 64 + c] = (uint8_t)((p0[r * 64 + c] + p1[r * 64 + c] + 1) >> 1);
    return block_metric(metric & 3, part, w, h, f, fenc->stride, avg, 64);
}

/* ------------------------------------------------------------------------- */
/* motion-compensated luma prediction: the 14-bit intermediates of bi-prediction and luma_addAvg       */

/* luma_p2s, ref: filterConvertPelToShort_c<MAX_CU_SIZE>, common/ipfilter.cpp:37-54, 510 (dst stride 64) */
void x265o_luma_p2s(const uint8_t *src, intptr_t sstride, int16_t *dst, int width, int height)
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            dst[y * 64 + x] = (int16_t)((src[y * sstride + x] << 6) - 8192);
}

/* luma_vps[part], ref: interp_vert_ps_c<8, W, H>, common/ipfilter.cpp:203-239: shift 0, offset -8192 */
void x265o_luma_vps(int part, const uint8_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx)
{
    int w = k_part_w[part], h = k_part_h[part];
    const int16_t *c = k_luma_taps[coeffIdx];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++)
                sum += src[(y + k - 3) * sstride + x] * c[k];
            dst[y * dstride + x] = (int16_t)(sum - 8192);
        }
}

/* luma_vss[part], ref: interp_vert_ss_c<8, W, H>, common/ipfilter.cpp:284-317: shift 6, no offset */
void x265o_luma_vss(int part, const int16_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx)
{
    int w = k_part_w[part], h = k_part_h[part];
    const int16_t *c = k_luma_taps[coeffIdx];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++)
                sum += src[(y + k - 3) * sstride + x] * c[k];
            dst[y * dstride + x] = (int16_t)(sum >> 6);
        }
}

/* luma_hvpp[part], ref: interp_hv_pp_c<8, W, H>, common/ipfilter.cpp:362-369 */
void x265o_luma_hvpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int idxX, int idxY)
{
    int16_t immed[(64 + 8) * (64 + 8)];
    int w = k_part_w[part];
    x265o_luma_hps(part, src, sstride, immed, w, idxX, 1);
    x265o_luma_vsp(part, immed + 3 * w, w, dst, dstride, idxY);
}

/* luma_addAvg[part], ref: addAvg<W, H>, common/pixel.cpp:819-839: shift 7, offset 64 + 2 * 8192 */
void x265o_luma_addavg(int part, const int16_t *src0, const int16_t *src1, uint8_t *dst, intptr_t sstride0, intptr_t sstride1,
                       intptr_t dstride)
{
    int w = k_part_w[part], h = k_part_h[part];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int v = (src0[y * sstride0 + x] + src1[y * sstride1 + x] + 64 + 2 * 8192) >> 7;
            dst[y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

/* ref: TComPrediction::xPredInterLumaBlk to a ShortYuv (stride 64), TComPrediction.cpp:436-473 */
static void predict_luma_short(int part, int w, int h, const uint8_t *p, intptr_t stride, int qx, int qy, int16_t *dst)
{
    int xf = qx & 3, yf = qy & 3;
    p += (qx >> 2) + (qy >> 2) * stride;
    if (!(xf | yf))
        x265o_luma_p2s(p, stride, dst, w, h);
    else if (!yf)
        x265o_luma_hps(part, p, stride, dst, 64, xf, 0);
    else if (!xf)
        x265o_luma_vps(part, p, stride, dst, 64, yf);
    else
    {
        int16_t immed[64 * (64 + 8)];
        x265o_luma_hps(part, p, stride, immed, w, xf, 1);
        x265o_luma_vss(part, immed + 3 * w, w, dst, 64, yf);
    }
}

/* ref: weight_sp_c, common/pixel.cpp:492-508 (IF_INTERNAL_OFFS = 8192) */
void x265o_weight_sp(const int16_t *src, uint8_t *dst, intptr_t sstride, intptr_t dstride, int width, int height, int w0,
                     int round, int shift, int offset)
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            int v = ((w0 * (src[y * sstride + x] + 8192) + round) >> shift) + offset;
            dst[y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

/* ref: TComWeightPrediction::addWeightBi on two ShortYuv, luma, TComWeightPrediction.cpp:203-255 (weightBidirY :46-49)
 * with the parameters getWpScaling derives for a block that uses both lists (:534-547): w = inputWeight,
 * o = inputOffset (8-bit samples: x 1), shift = log2WeightDenom of list 0's entry, + IF_INTERNAL_PREC - X265_DEPTH + 1,
 * round = 1 << (shift - 1) (bRound is true at both call sites, :581, :617) */
void x265o_add_weight_bi(const int16_t *src0, const int16_t *src1, uint8_t *dst, intptr_t sstride0, intptr_t sstride1,
                         intptr_t dstride, int width, int height, int w0, int o0, int w1, int o1, int log2_denom)
{
    int offset = o0 + o1, shift = log2_denom + 6 + 1, round = 1 << (shift - 1);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            int v = (w0 * (src0[y * sstride0 + x] + 8192) + w1 * (src1[y * sstride1 + x] + 8192) + round + offset * (1 << (shift - 1))) >> shift;
            dst[y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

/* ref: TComPrediction::xPredInterUni / xPredInterBi for luma, TComPrediction.cpp:330-387; weighted
 * uni-directional prediction: motionCompensation's getUseWP branch, TComPrediction.cpp:265-279, with
 * TComWeightPrediction::addWeightUni, TComWeightPrediction.cpp:433-458; weighted bi-prediction: xPredInterBi's
 * getWPBiPred branch, TComPrediction.cpp:347-350, with xWeightedPredictionBi, TComWeightPrediction.cpp:609-618 */
static void predict_one(const x265o_plane *planes, const x265o_pred_req *r, uint8_t *dst, intptr_t dstride)
{
    const x265o_plane *a = &planes[r->ref0_plane];
    int part = x265o_partition_from_sizes(r->pu_w, r->pu_h);
    uint8_t tmp[64 * 64];
    if (r->ref1_plane == X265O_PRED_WEIGHTED)
    {
        int16_t s0[64 * 64];
        int shift = r->log2_denom + 6; /* IF_INTERNAL_PREC - X265_DEPTH */
        predict_luma_short(part, r->pu_w, r->pu_h, a->origin + r->pu_x + (intptr_t)r->pu_y * a->stride, a->stride, r->qmv0_x,
                           r->qmv0_y, s0);
        x265o_weight_sp(s0, dst, 64, dstride, r->pu_w, r->pu_h, r->qmv1_x, 1 << (shift - 1), shift, r->qmv1_y);
    }
    else if (r->ref1_plane < 0)
    {
        predict_luma(part, r->pu_w, a->origin + r->pu_x + (intptr_t)r->pu_y * a->stride, a->stride, r->qmv0_x, r->qmv0_y, tmp);
        for (int y = 0; y < r->pu_h; y++)
            memcpy(dst + y * dstride, tmp + y * 64, r->pu_w);
    }
    else
    {
        const x265o_plane *b = &planes[r->ref1_plane];
        int16_t s0[64 * 64], s1[64 * 64];
        predict_luma_short(part, r->pu_w, r->pu_h, a->origin + r->pu_x + (intptr_t)r->pu_y * a->stride, a->stride, r->qmv0_x,
                           r->qmv0_y, s0);
        predict_luma_short(part, r->pu_w, r->pu_h, b->origin + r->pu_x + (intptr_t)r->pu_y * b->stride, b->stride, r->qmv1_x,
                           r->qmv1_y, s1);
        if (r->weighted_bi)
            x265o_add_weight_bi(s0, s1, dst, 64, 64, dstride, r->pu_w, r->pu_h, r->wp_w0, r->wp_o0, r->wp_w1, r->wp_o1, r->log2_denom);
        else
            x265o_luma_addavg(part, s0, s1, dst, 64, 64, dstride);
    }
}

void x265o_predict_batch(const x265o_plane *planes, const x265o_pred_req *reqs, int n)
{
    for (int i = 0; i < n; i++)
    {
        const x265o_plane *d = &planes[reqs[i].dst_plane];
        predict_one(planes, &reqs[i], (uint8_t *)d->origin + reqs[i].pu_x + (intptr_t)reqs[i].pu_y * d->stride, d->stride);
    }
}

/* ref: what TEncSearch::xMergeEstimation measures on a candidate, TEncSearch.cpp:2209-2210: motionCompensation into a
 * temporary block, then MotionEstimate::bufSATD (motion.h:79-82: satd[partEnum] against the source PU).  dst_plane of
 * the request names the source picture. */
void x265o_mc_satd_batch(const x265o_plane *planes, const x265o_pred_req *reqs, int n, int32_t *out)
{
    for (int i = 0; i < n; i++)
    {
        const x265o_pred_req *r = &reqs[i];
        const x265o_plane *f = &planes[r->dst_plane];
        uint8_t pred[64 * 64];
        predict_one(planes, r, pred, 64);
        out[i] = x265o_satd(x265o_partition_from_sizes(r->pu_w, r->pu_h), f->origin + r->pu_x + (intptr_t)r->pu_y * f->stride, f->stride,
                            pred, 64);
    }
}

static void clip_mv(int *qx, int *qy, int cu_x, int cu_y, int pic_w, int pic_h);

/* ref: TEncSearch::xMergeEstimation, Lib/TLibEncoder/TEncSearch.cpp:2160-2226, behind getInterMergeCandidates and the
 * conversion under isBipredRestriction (the caller's).  Per candidate (:2194-2219): the frame threads' guard on mv.y, the
 * candidate's motion written into the CU and predicted by motionCompensation(cu, .., REF_PIC_LIST_X, ..)
 * (TComPrediction.cpp:252-297), bufSATD, getTUBits (TEncSearch.h:94-97), RDCost::getCost (rdcost.h:112-115), first
 * minimum. */
void x265o_merge_estimate_batch(const x265o_plane *planes, const x265o_merge_pu *pus, int n, int pic_w, int pic_h,
                                x265o_merge_result *out)
{
    for (int i = 0; i < n; i++)
    {
        const x265o_merge_pu *p = &pus[i];
        x265o_merge_result *o = &out[i];
        memset(o, 0, sizeof(*o));
        o->cost = UINT32_MAX;
        for (int c = 0; c < X265O_MRG_MAX_NUM_CANDS; c++)
            o->satd[c] = -1;
        for (int c = 0; c < p->num_cand; c++)
        {
            const x265o_merge_cand *m = &p->cand[c];
            if (m->mv[1] >= p->mvy_limit || m->mv[3] >= p->mvy_limit) /* :2197-2202 */
                continue;
            int ref[2] = { m->ref_plane[0], m->ref_plane[1] }; /* refIdx >= 0 <=> a plane */
            int mv[2][2] = { { m->mv[0], m->mv[1] }, { m->mv[2], m->mv[3] } };
            x265o_pred_req r;
            memset(&r, 0, sizeof(r));
            r.dst_plane = p->fenc_plane;
            r.pu_x = p->pu_x, r.pu_y = p->pu_y, r.pu_w = p->pu_w, r.pu_h = p->pu_h;
            r.ref1_plane = -1;
            int lists[2], nl = 0; /* the lists motionCompensation predicts from */
            int weighted_uni = 0;
            if (p->weighted == 1)
            {
                /* a P slice: list = REF_PIC_LIST_0 (:261-262), through the intermediate and addWeightUni (:265-279) */
                if (ref[0] < 0)
                    continue;
                lists[nl++] = 0;
                weighted_uni = 1;
            }
            else if (ref[0] >= 0 && ref[1] >= 0)
            {
                /* xCheckIdenticalMotion (:232-250): the same picture and vector on both lists, never with weighted
                 * bi-prediction -> xPredInterUni of list 0 (:287-290) */
                int same = p->weighted != 2 && ref[0] == ref[1] && mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1];
                lists[nl++] = 0;
                if (!same)
                    lists[nl++] = 1;
            }
            else if (ref[0] >= 0 || ref[1] >= 0)
            {
                /* xPredInterBi with one list (:356-386): weighted where the PPS has weighted bi-prediction */
                lists[nl++] = ref[0] >= 0 ? 0 : 1;
                weighted_uni = p->weighted == 2;
            }
            else
                continue;
            for (int l = 0; l < 2; l++)
                clip_mv(&mv[l][0], &mv[l][1], p->cu_x, p->cu_y, pic_w, pic_h); /* xPredInterUni :305-306, :321-322 */
            r.ref0_plane = ref[lists[0]];
            r.qmv0_x = (int16_t)mv[lists[0]][0], r.qmv0_y = (int16_t)mv[lists[0]][1];
            if (nl == 2)
            {
                r.ref1_plane = ref[1];
                r.qmv1_x = (int16_t)mv[1][0], r.qmv1_y = (int16_t)mv[1][1];
                if (p->weighted == 2) /* xWeightedPredictionBi, :347-350 */
                {
                    r.weighted_bi = 1;
                    r.wp_w0 = m->wp_w[0], r.wp_o0 = m->wp_o[0], r.wp_w1 = m->wp_w[1], r.wp_o1 = m->wp_o[1];
                    r.log2_denom = m->log2_denom[0];
                }
            }
            else if (weighted_uni)
            {
                r.ref1_plane = X265O_PRED_WEIGHTED;
                r.qmv1_x = m->wp_w[lists[0]], r.qmv1_y = m->wp_o[lists[0]];
                r.log2_denom = m->log2_denom[lists[0]];
            }
            int32_t satd;
            x265o_mc_satd_batch(planes, &r, 1, &satd);
            uint32_t bits = (uint32_t)(c + (c < p->num_cand - 1));
            uint32_t cost = (uint32_t)satd + (uint32_t)((bits * p->lambda_sad + 128) >> 8);
            o->satd[c] = satd;
            if (cost < o->cost)
            {
                o->cost = cost;
                o->bits = bits;
                o->index = (uint8_t)c;
                o->inter_dir = (uint8_t)((ref[0] >= 0 ? 1 : 0) | (ref[1] >= 0 ? 2 : 0));
            }
        }
    }
}

/* ref: ReferencePlanes::lowresMC, common/lowres.h:53-73: the half-pel plane the vector selects, or the rounded
 * average of two of them for quarter-pel vectors (always an 8x8 block) */
static const uint8_t *lowres_mc(const x265o_plane *four, int x, int y, int qx, int qy, uint8_t *buf, intptr_t *stride)
{
    intptr_t ls = four[0].stride;
    intptr_t off = x + (intptr_t)y * ls;
    if ((qx | qy) & 1)
    {
        int hpelA = (qy & 2) | ((qx & 2) >> 1);
        const uint8_t *a = four[hpelA].origin + off + (qx >> 2) + (qy >> 2) * ls;
        int bx = s16(qx + (qx & 1) * 2), by = s16(qy + (qy & 1) * 2);
        int hpelB = (by & 2) | ((bx & 2) >> 1);
        const uint8_t *b = four[hpelB].origin + off + (bx >> 2) + (by >> 2) * ls;
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++)
                buf[r * 8 + c] = (uint8_t)((a[r * ls + c] + b[r * ls + c] + 1) >> 1);
        *stride = 8;
        return buf;
    }
    *stride = ls;
    return four[(qy & 2) | ((qx & 2) >> 1)].origin + off + (qx >> 2) + (qy >> 2) * ls;
}

/* The lookahead's bidirectional cost, ref: EstimateRow::estimateCUCost, encoder/slicetype.cpp:1585-1603:
 * lowresMC of both lists, pixelavg_pp[LUMA_8x8], satd[LUMA_8x8] against the lowres source block */
static int lowres_bidir_cost(const x265o_plane *planes, const x265o_bidir_req *r)
{
    uint8_t b0[64], b1[64], avg[64];
    intptr_t s0, s1;
    const uint8_t *p0 = lowres_mc(&planes[r->ref0_plane], r->pu_x, r->pu_y, r->qmv0_x, r->qmv0_y, b0, &s0);
    const uint8_t *p1 = lowres_mc(&planes[r->ref1_plane], r->pu_x, r->pu_y, r->qmv1_x, r->qmv1_y, b1, &s1);
    x265o_pixelavg_pp(1 /* LUMA_8x8 */, avg, 8, p0, s0, p1, s1);
    const x265o_plane *f = &planes[r->fenc_plane];
    const uint8_t *fe = f->origin + r->pu_x + (intptr_t)r->pu_y * f->stride;
    return block_metric(r->metric & 3, 1, 8, 8, fe, f->stride, avg, 8);
}

void x265o_bidir_cost_batch(const x265o_plane *planes, const x265o_bidir_req *reqs, int n, int32_t *out)
{
    for (int i = 0; i < n; i++)
    {
        const x265o_bidir_req *r = &reqs[i];
        if (r->metric & X265O_LOWRES_REF)
        {
            out[i] = lowres_bidir_cost(planes, r);
            continue;
        }
        out[i] = x265o_bidir_cost(&planes[r->fenc_plane], &planes[r->ref0_plane], &planes[r->ref1_plane], r->pu_x, r->pu_y,
                                  r->pu_w, r->pu_h, r->qmv0_x, r->qmv0_y, r->qmv1_x, r->qmv1_y, r->metric);
    }
}

/* ------------------------------------------------------------------------- */
/* lookahead planes                                                          */

/* ref: frame_init_lowres_core, common/pixel.cpp:595-618 */
void x265o_frame_init_lowres(const uint8_t *src0, uint8_t *dst0, uint8_t *dsth, uint8_t *dstv, uint8_t *dstc,
                             intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
    for (int y = 0; y < height; y++)
    {
        const uint8_t *src1 = src0 + src_stride, *src2 = src1 + src_stride;
        for (int x = 0; x < width; x++)
        {
#define X265O_F(a, b, c, d) ((((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1)
            dst0[x] = (uint8_t)X265O_F(src0[2 * x], src1[2 * x], src0[2 * x + 1], src1[2 * x + 1]);
            dsth[x] = (uint8_t)X265O_F(src0[2 * x + 1], src1[2 * x + 1], src0[2 * x + 2], src1[2 * x + 2]);
            dstv[x] = (uint8_t)X265O_F(src1[2 * x], src2[2 * x], src1[2 * x + 1], src2[2 * x + 1]);
            dstc[x] = (uint8_t)X265O_F(src1[2 * x + 1], src2[2 * x + 1], src1[2 * x + 2], src2[2 * x + 2]);
#undef X265O_F
        }
        src0 += src_stride * 2;
        dst0 += dst_stride;
        dsth += dst_stride;
        dstv += dst_stride;
        dstc += dst_stride;
    }
}

/* ref: extendPicBorder, common/pixel.cpp:893-911 (rows first: extendRowBorder replicates the first and last
 * sample of every row into the margins; then whole rows of `stride` bytes are copied up and down) */
void x265o_extend_pic_border(uint8_t *pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    for (int y = 0; y < height; y++)
    {
        uint8_t *row = pic + y * stride;
        memset(row - marginX, row[0], marginX);
        memset(row + width, row[width - 1], marginX);
    }
    uint8_t *top = pic - marginX;
    for (int y = 0; y < marginY; y++)
        memcpy(top - (y + 1) * stride, top, stride);
    uint8_t *bot = pic - marginX + (height - 1) * stride;
    for (int y = 0; y < marginY; y++)
        memcpy(bot + (y + 1) * stride, bot, stride);
}

/* ref: pixelavg_pp<lx, ly>, common/pixel.cpp:546-560 (the weight argument is unused there) */
void x265o_pixelavg_pp(int part, uint8_t *dst, intptr_t dstride, const uint8_t *src0, intptr_t sstride0,
                       const uint8_t *src1, intptr_t sstride1)
{
    int w = k_part_w[part], h = k_part_h[part];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[y * dstride + x] = (uint8_t)((src0[y * sstride0 + x] + src1[y * sstride1 + x] + 1) >> 1);
}

/* ref: weight_pp_c, common/pixel.cpp:510-529 (8-bit: IF_INTERNAL_PREC - X265_DEPTH = 6) */
void x265o_weight_pp(const uint8_t *src, uint8_t *dst, intptr_t sstride, intptr_t dstride, int width, int height,
                     int w0, int round, int shift, int offset)
{
    for (int y = 0; y < height; y++)
    {
        for (int x = 0; x < width; x++)
        {
            int16_t val = (int16_t)(src[x] << 6);
            int v = ((w0 * val + round) >> shift) + offset;
            dst[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        src += sstride;
        dst += dstride;
    }
}

/* ------------------------------------------------------------------------- */
/* batched drivers (pthread fan-out over contiguous slices)                  */

typedef struct
{
    const x265o_plane *planes;
    const uint16_t *const *tabs;
    const x265o_me_job *jobs;
    x265o_me_result *mres;
    const x265o_cost_req *reqs;
    int32_t *cres;
    int lo, hi;
    double ops[5]; /* me_slice: this slice's work model (sad, satd, interpolation lane-ops; sad and sub-pel candidates) */
} slice_arg;

static void *me_slice(void *p)
{
    slice_arg *a = (slice_arg *)p;
    t_ops_sad = t_ops_satd = t_ops_interp = 0;
    t_n_sad = t_n_sub = 0;
    for (int i = a->lo; i < a->hi; i++)
    {
        const x265o_me_job *j = &a->jobs[i];
        mv_t mv;
        a->mres[i].cost = motion_estimate(&a->planes[j->fenc_plane], &a->planes[j->ref_plane], a->tabs[j->qp], j, &mv);
        a->mres[i].mvx = mv.x;
        a->mres[i].mvy = mv.y;
    }
    a->ops[0] = t_ops_sad;
    a->ops[1] = t_ops_satd;
    a->ops[2] = t_ops_interp;
    a->ops[3] = (double)t_n_sad;
    a->ops[4] = (double)t_n_sub;
    return 0;
}

static void *cost_slice(void *p)
{
    slice_arg *a = (slice_arg *)p;
    for (int i = a->lo; i < a->hi; i++)
    {
        const x265o_cost_req *r = &a->reqs[i];
        a->cres[i] = x265o_subpel_cost(&a->planes[r->fenc_plane], &a->planes[r->ref_plane], r->pu_x, r->pu_y, r->pu_w,
                                       r->pu_h, r->qmv_x, r->qmv_y, r->metric);
    }
    return 0;
}

static void fan_out(void *(*fn)(void *), slice_arg *proto, int n, int nthreads)
{
    memset(proto->ops, 0, sizeof(proto->ops));
    if (nthreads <= 1 || n < 2 * nthreads)
    {
        proto->lo = 0;
        proto->hi = n;
        fn(proto);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    slice_arg *args = (slice_arg *)malloc(sizeof(slice_arg) * nthreads);
    for (int t = 0; t < nthreads; t++)
    {
        args[t] = *proto;
        args[t].lo = (int)((int64_t)n * t / nthreads);
        args[t].hi = (int)((int64_t)n * (t + 1) / nthreads);
        pthread_create(&th[t], 0, fn, &args[t]);
    }
    for (int t = 0; t < nthreads; t++)
    {
        pthread_join(th[t], 0);
        for (int k = 0; k < 5; k++)
            proto->ops[k] += args[t].ops[k];
    }
    free(args);
    free(th);
}

void x265o_me_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_me_job *jobs, int n,
                    x265o_me_result *out, int nthreads)
{
    slice_arg a;
    memset(&a, 0, sizeof(a));
    a.planes = planes;
    a.tabs = costtabs;
    a.jobs = jobs;
    a.mres = out;
    fan_out(me_slice, &a, n, nthreads);
}

/* ------------------------------------------------------------------------- */
/* predInterSearch's (list, reference) iteration                              */

/* ref: TComDataCU::clipMv, Lib/TLibCommon/TComDataCU.cpp:2502-2514 (g_maxCUSize 64, offset 8) */
static void clip_mv(int *qx, int *qy, int cu_x, int cu_y, int pic_w, int pic_h)
{
    int xmax = (pic_w + 8 - cu_x - 1) << 2, xmin = (-64 - 8 - cu_x + 1) << 2;
    int ymax = (pic_h + 8 - cu_y - 1) << 2, ymin = (-64 - 8 - cu_y + 1) << 2;
    *qx = imin(xmax, *qx > xmin ? *qx : xmin);
    *qy = imin(ymax, *qy > ymin ? *qy : ymin);
}

/* ref: TEncSearch::predInterSearch, Lib/TLibEncoder/TEncSearch.cpp:2315-2340 (AMVP pick by least SAD of the clipped
 * candidates' predictions, first minimum), xSetSearchRange :2600-2617, then motionEstimate */
void x265o_inter_search_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_inter_job *jobs, int n,
                              int pic_w, int pic_h, x265o_inter_result *out)
{
    for (int i = 0; i < n; i++)
    {
        const x265o_inter_job *q = &jobs[i];
        x265o_inter_result *r = &out[i];
        memset(r, 0, sizeof(*r));
        int best = INT_MAX, pick = 0;
        r->amvp_sad[0] = r->amvp_sad[1] = INT_MAX;
        for (int k = 0; k < q->num_amvp && k < 2; k++)
        {
            int qx = q->amvp[2 * k], qy = q->amvp[2 * k + 1];
            clip_mv(&qx, &qy, q->cu_x, q->cu_y, pic_w, pic_h);
            int cost = x265o_subpel_cost(&planes[q->fenc_plane], &planes[q->pred_plane], q->pu_x, q->pu_y, q->pu_w, q->pu_h, qx, qy, X265O_SAD);
            r->amvp_sad[k] = cost;
            if (best > cost)
            {
                best = cost;
                pick = k;
            }
        }
        x265o_me_job m;
        memset(&m, 0, sizeof(m));
        m.fenc_plane = q->fenc_plane;
        m.ref_plane = q->ref_plane;
        m.pu_x = q->pu_x, m.pu_y = q->pu_y, m.pu_w = q->pu_w, m.pu_h = q->pu_h;
        m.method = q->method, m.subme = q->subme, m.merange = q->merange, m.qp = q->qp, m.num_mvc = q->num_mvc;
        memcpy(m.mvc, q->mvc, sizeof(m.mvc));
        m.qmvp_x = q->amvp[2 * pick], m.qmvp_y = q->amvp[2 * pick + 1];
        int px = m.qmvp_x, py = m.qmvp_y;
        clip_mv(&px, &py, q->cu_x, q->cu_y, pic_w, pic_h);
        int d = s16(q->merange << 2);
        int lox = s16(px - d), loy = s16(py - d), hix = s16(px + d), hiy = s16(py + d);
        clip_mv(&lox, &loy, q->cu_x, q->cu_y, pic_w, pic_h);
        clip_mv(&hix, &hiy, q->cu_x, q->cu_y, pic_w, pic_h);
        m.mvmin_x = (int16_t)(lox >> 2), m.mvmax_x = (int16_t)(hix >> 2);
        m.mvmin_y = (int16_t)imin(loy >> 2, q->ref_lag), m.mvmax_y = (int16_t)imin(hiy >> 2, q->ref_lag);
        mv_t mv;
        r->cost = motion_estimate(&planes[m.fenc_plane], &planes[m.ref_plane], costtabs[m.qp], &m, &mv);
        r->mvx = mv.x, r->mvy = mv.y;
        r->mvmin_x = m.mvmin_x, r->mvmin_y = m.mvmin_y, r->mvmax_x = m.mvmax_x, r->mvmax_y = m.mvmax_y;
        r->mvp_x = m.qmvp_x, r->mvp_y = m.qmvp_y;
        r->mvp_idx = (uint8_t)pick;
    }
}

/* ------------------------------------------------------------------------- */
/* predInterSearch behind its iterations: bits, xCheckBestMVP, the bidirectional candidate, the choice */

/* ref: BitCost::CalculateLogs, encoder/bitcost.cpp:68-80: s_bitsizes[0] = 0.718f, [i] = log(i + 1) * (2 / ln 2) + 1.718f
 * (the logarithm in double, as in x265o_bitcost_table) */
static float g_bitsizes[32769];
static void bitsizes_init(void)
{
    static int done = 0;
    if (done)
        return;
    const float two_over_ln2 = (float)(2.0f / log((double)2.0f));
    g_bitsizes[0] = 0.718f;
    for (int i = 1; i <= 32768; i++)
        g_bitsizes[i] = (float)(log((double)(float)(i + 1)) * two_over_ln2 + 1.718f);
    done = 1;
}

/* ref: BitCost::bitcost, encoder/bitcost.h:48-52 (differences beyond the table's 16383 quarter samples, which the
 * reference reads out of bounds, take its last entry here) */
int x265o_mv_bits(int x, int y, int px, int py)
{
    bitsizes_init();
    int ix = (abs(x - px) << 1) + (x < px), iy = (abs(y - py) << 1) + (y < py);
    if (ix > 32768)
        ix = 32768;
    if (iy > 32768)
        iy = 32768;
    volatile float sum = g_bitsizes[ix] + g_bitsizes[iy];
    volatile float r = sum + 0.5f;
    return (uint16_t)r;
}

/* ref: RDCost::getCost, encoder/rdcost.h:112-115 */
static uint32_t rd_cost_of_bits(uint32_t bits, uint32_t lambda_sad) { return (uint32_t)(((uint64_t)bits * lambda_sad + 128) >> 8); }

/* ref: TEncSearch::xCheckBestMVP, TEncSearch.cpp:2565-2598 (AMVP_MAX_NUM_CANDS = 2, MVP_IDX_BITS = 1) */
void x265o_check_best_mvp(const int16_t cand[4], int mvx, int mvy, int *mvp_idx, uint32_t *bits, uint32_t *cost, uint32_t lambda_sad)
{
    const int cur = *mvp_idx;
    const int orig = x265o_mv_bits(mvx, mvy, cand[2 * cur], cand[2 * cur + 1]) + 1;
    int best_bits = orig, best = cur;
    for (int i = 0; i < 2; i++)
    {
        if (i == cur)
            continue;
        int b = x265o_mv_bits(mvx, mvy, cand[2 * i], cand[2 * i + 1]) + 1;
        if (b < best_bits)
        {
            best_bits = b;
            best = i;
        }
    }
    if (best != cur)
    {
        const uint32_t before = *bits;
        *mvp_idx = best;
        *bits = before - orig + best_bits;
        *cost = (*cost - rd_cost_of_bits(before, lambda_sad)) + rd_cost_of_bits(*bits, lambda_sad);
    }
}

void x265o_inter_decide_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_inter_job *jobs,
                              const x265o_inter_result *res, const x265o_inter_pu *pus, int npu, int pic_w, int pic_h,
                              x265o_inter_pu_result *out)
{
    for (int u = 0; u < npu; u++)
    {
        const x265o_inter_pu *p = &pus[u];
        x265o_inter_pu_result *o = &out[u];
        memset(o, 0, sizeof(*o));
        o->list[0].cost = o->list[1].cost = o->bidir_cost = UINT32_MAX;
        const x265o_inter_job *jl[2] = { 0, 0 }; /* the job of each list's best reference */
        int k = p->first_job;
        /* ref: TEncSearch.cpp:2303-2358 */
        for (int l = 0; l < 2; l++)
            for (int ref = 0; ref < p->num_ref[l]; ref++, k++)
            {
                const x265o_inter_job *q = &jobs[k];
                const x265o_inter_result *r = &res[k];
                uint32_t bits = p->list_sel_bits[l] + 1 + (uint32_t)(ref + (ref < p->num_ref[l] - 1)); /* MVP_IDX_BITS, getTUBits */
                bits += (uint32_t)x265o_mv_bits(r->mvx, r->mvy, r->mvp_x, r->mvp_y);
                const int mvc = (uint16_t)(tab_at(costtabs[q->qp], r->mvp_x, r->mvx) + tab_at(costtabs[q->qp], r->mvp_y, r->mvy));
                uint32_t cost = (uint32_t)(r->cost - mvc) + rd_cost_of_bits(bits, p->lambda_sad);
                int idx = r->mvp_idx;
                x265o_check_best_mvp(q->amvp, r->mvx, r->mvy, &idx, &bits, &cost, p->lambda_sad);
                if (cost < o->list[l].cost)
                {
                    x265o_inter_side *s = &o->list[l];
                    s->cost = cost, s->bits = (int32_t)bits;
                    s->mvx = r->mvx, s->mvy = r->mvy;
                    s->mvp_x = q->amvp[2 * idx], s->mvp_y = q->amvp[2 * idx + 1];
                    s->ref = (int8_t)ref, s->mvp_idx = (int8_t)idx;
                    jl[l] = q;
                }
            }
        /* ref: TEncSearch.cpp:2360-2439 */
        if (!p->no_bidir && o->list[0].cost != UINT32_MAX && o->list[1].cost != UINT32_MAX)
        {
            const x265o_inter_job *q0 = jl[0], *q1 = jl[1];
            const int sel = p->list_sel_bits[2] - (p->list_sel_bits[0] + p->list_sel_bits[1]);
            o->bidir[0] = o->list[0];
            o->bidir[1] = o->list[1];
            int satd = x265o_bidir_cost(&planes[q0->fenc_plane], &planes[q0->pred_plane], &planes[q1->pred_plane], q0->pu_x, q0->pu_y, q0->pu_w,
                                        q0->pu_h, o->list[0].mvx, o->list[0].mvy, o->list[1].mvx, o->list[1].mvy, X265O_SATD);
            o->bidir_bits = o->list[0].bits + o->list[1].bits + sel;
            o->bidir_cost = (uint32_t)satd + rd_cost_of_bits((uint32_t)o->bidir_bits, p->lambda_sad);
            int try_zero = (o->list[0].mvx | o->list[0].mvy | o->list[1].mvx | o->list[1].mvy) != 0;
            if (try_zero)
            {
                /* xSetSearchRange around the zero vector with merange = the larger picture side (:2600-2617) */
                const int d = s16(imax(pic_w, pic_h) << 2);
                int px = 0, py = 0;
                clip_mv(&px, &py, q0->cu_x, q0->cu_y, pic_w, pic_h);
                int lox = s16(px - d), loy = s16(py - d), hix = s16(px + d), hiy = s16(py + d);
                clip_mv(&lox, &loy, q0->cu_x, q0->cu_y, pic_w, pic_h);
                clip_mv(&hix, &hiy, q0->cu_x, q0->cu_y, pic_w, pic_h);
                int minx = lox >> 2, maxx = hix >> 2, miny = imin(loy >> 2, q0->ref_lag), maxy = imin(hiy >> 2, q0->ref_lag);
                maxy = s16(maxy + 2);
                minx = s16(minx << 2), miny = s16(miny << 2), maxx = s16(maxx << 2), maxy = s16(maxy << 2);
                for (int l = 0; l < 2; l++)
                    try_zero &= o->list[l].mvp_x >= minx && o->list[l].mvp_x <= maxx && o->list[l].mvp_y >= miny && o->list[l].mvp_y <= maxy;
            }
            o->zero_tried = (uint8_t)try_zero;
            if (try_zero)
            {
                /* the coincident blocks of the two searched planes (m_mref[l][ref]->fpelPlane) */
                satd = x265o_bidir_cost(&planes[q0->fenc_plane], &planes[q0->ref_plane], &planes[q1->ref_plane], q0->pu_x, q0->pu_y, q0->pu_w, q0->pu_h,
                                        0, 0, 0, 0, X265O_SATD);
                uint32_t bits[2], cost;
                int idx[2];
                for (int l = 0; l < 2; l++)
                {
                    const x265o_inter_side *s = &o->list[l];
                    idx[l] = s->mvp_idx;
                    bits[l] = (uint32_t)(s->bits - x265o_mv_bits(s->mvx, s->mvy, s->mvp_x, s->mvp_y) + x265o_mv_bits(0, 0, s->mvp_x, s->mvp_y));
                }
                cost = (uint32_t)satd + rd_cost_of_bits(bits[0], p->lambda_sad) + rd_cost_of_bits(bits[1], p->lambda_sad);
                x265o_check_best_mvp(q0->amvp, 0, 0, &idx[0], &bits[0], &cost, p->lambda_sad);
                x265o_check_best_mvp(q1->amvp, 0, 0, &idx[1], &bits[1], &cost, p->lambda_sad);
                if (cost < o->bidir_cost)
                {
                    const x265o_inter_job *qq[2] = { q0, q1 };
                    for (int l = 0; l < 2; l++)
                    {
                        o->bidir[l].mvx = o->bidir[l].mvy = 0;
                        o->bidir[l].mvp_x = qq[l]->amvp[2 * idx[l]], o->bidir[l].mvp_y = qq[l]->amvp[2 * idx[l] + 1];
                        o->bidir[l].mvp_idx = (int8_t)idx[l];
                        o->bidir[l].bits = (int32_t)bits[l];
                    }
                    o->bidir_cost = cost;
                    o->bidir_bits = (int32_t)(bits[0] + bits[1]) + sel;
                }
            }
        }
        /* ref: TEncSearch.cpp:2449-2499 without the merge candidate */
        o->inter_dir = (o->bidir_cost < o->list[0].cost && o->bidir_cost < o->list[1].cost) ? 3 : (o->list[0].cost <= o->list[1].cost ? 1 : 2);
    }
}

/* ------------------------------------------------------------------------- */
/* the lookahead's whole-frame motion search                                   */

/* ref: median / median_mv, encoder/slicetype.cpp:40-55 (int16 arithmetic, branch-free) */
static int16_t look_median(int16_t a, int16_t b, int16_t c)
{
    int16_t t = (int16_t)((a - b) & ((a - b) >> 31));
    a = (int16_t)(a - t);
    b = (int16_t)(b + t);
    b = (int16_t)(b - ((b - c) & ((b - c) >> 31)));
    b = (int16_t)(b + ((a - b) & ((a - b) >> 31)));
    return b;
}

typedef struct
{
    const x265o_plane *planes;
    const uint16_t *const *tabs;
    const x265o_lookahead_job *jobs;
    x265o_me_result *out;
    int w, h, lo, hi;
} look_arg;

/* ref: EstimateRow::estimateCUCost, encoder/slicetype.cpp:1512-1584, the search of one list: bounds :1540-1544,
 * candidates from the CUs already estimated :1556-1573, predictor :1575-1580, motionEstimate :1582; the CUs in the order
 * CostEstimate::processRow's dependencies allow (:1458-1499): reverse raster is one such order */
static void *look_slice(void *p)
{
    look_arg *a = (look_arg *)p;
    const int W = a->w, H = a->h;
    for (int i = a->lo; i < a->hi; i++)
    {
        const x265o_lookahead_job *lj = &a->jobs[i];
        x265o_me_result *res = a->out + (size_t)i * W * H;
        for (int cuy = H - 1; cuy >= 0; cuy--)
            for (int cux = W - 1; cux >= 0; cux--)
            {
                x265o_me_result *mv = res + cuy * W + cux; /* fenc_mv */
                x265o_me_job j;
                memset(&j, 0, sizeof(j));
                j.fenc_plane = lj->fenc_plane;
                j.ref_plane = lj->ref_plane;
                j.pu_x = (int16_t)(8 * cux), j.pu_y = (int16_t)(8 * cuy);
                j.pu_w = j.pu_h = 8;
                j.method = (uint8_t)(lj->method | X265O_LOWRES_REF);
                j.subme = lj->subme, j.merange = lj->merange, j.qp = lj->qp;
                j.mvmin_x = (int16_t)(-cux * 8 - 8), j.mvmin_y = (int16_t)(-cuy * 8 - 8);
                j.mvmax_x = (int16_t)((W - cux - 1) * 8 + 8), j.mvmax_y = (int16_t)((H - cuy - 1) * 8 + 8);
                int numc = 0; /* mvc[0] = mvc[2] = 0 by the memset */
#define MVC(m) (j.mvc[2 * numc] = (m).mvx, j.mvc[2 * numc + 1] = (m).mvy, numc++)
                if (cux < W - 1)
                    MVC(mv[1]);
                if (cuy < H - 1)
                {
                    MVC(mv[W]);
                    if (cux > 0)
                        MVC(mv[W - 1]);
                    if (cux < W - 1)
                        MVC(mv[W + 1]);
                }
#undef MVC
                if (numc <= 1)
                    j.qmvp_x = j.mvc[0], j.qmvp_y = j.mvc[1];
                else
                {
                    j.qmvp_x = look_median(j.mvc[0], j.mvc[2], j.mvc[4]);
                    j.qmvp_y = look_median(j.mvc[1], j.mvc[3], j.mvc[5]);
                }
                j.num_mvc = (uint8_t)numc;
                mv_t best;
                mv->cost = motion_estimate(&a->planes[j.fenc_plane], &a->planes[j.ref_plane], a->tabs[j.qp], &j, &best);
                mv->mvx = best.x, mv->mvy = best.y;
            }
    }
    return 0;
}

/* out[(i * height_in_cu + cuy) * width_in_cu + cux]; searches are independent of each other: threads take whole ones */
void x265o_lookahead_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_lookahead_job *jobs, int n,
                           int width_in_cu, int height_in_cu, x265o_me_result *out, int nthreads)
{
    if (nthreads > n)
        nthreads = n;
    if (nthreads < 1)
        nthreads = 1;
    pthread_t th[64];
    look_arg args[64];
    if (nthreads > 64)
        nthreads = 64;
    for (int t = 0; t < nthreads; t++)
    {
        look_arg a = { planes, costtabs, jobs, out, width_in_cu, height_in_cu, (int)((int64_t)n * t / nthreads),
                       (int)((int64_t)n * (t + 1) / nthreads) };
        args[t] = a;
        if (nthreads == 1)
            look_slice(&args[t]);
        else
            pthread_create(&th[t], 0, look_slice, &args[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++)
            pthread_join(th[t], 0);
}

/* x265o_me_batch plus the work model of the searches: ops5 = { sad lane-ops, satd lane-ops, interpolation lane-ops,
 * full-pel sad candidates, sub-pel (subpelCompare) candidates } summed over the list */
void x265o_me_batch_ops(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_me_job *jobs, int n,
                        x265o_me_result *out, int nthreads, double *ops5)
{
    slice_arg a;
    memset(&a, 0, sizeof(a));
    a.planes = planes;
    a.tabs = costtabs;
    a.jobs = jobs;
    a.mres = out;
    fan_out(me_slice, &a, n, nthreads);
    memcpy(ops5, a.ops, sizeof(a.ops));
}

void x265o_cost_batch(const x265o_plane *planes, const x265o_cost_req *reqs, int n, int32_t *out, int nthreads)
{
    slice_arg a;
    memset(&a, 0, sizeof(a));
    a.planes = planes;
    a.reqs = reqs;
    a.cres = out;
    fan_out(cost_slice, &a, n, nthreads);
}

/* ------------------------------------------------------------------------- */
/* config-2 hierarchical full search, one CTU                                 */

void x265o_ctu_fullsearch(const x265o_plane *fenc, const x265o_plane *ref, const uint16_t *costtab, int ctu_x,
                          int ctu_y, int mvmin_x, int mvmin_y, int mvmax_x, int mvmax_y, int qmvp_x, int qmvp_y,
                          x265o_me_result *out)
{
    /* PU order: level 0 = 64x64 (1), level 1 = 32x32 (4), level 2 = 16x16 (16), level 3 = 8x8 (64);
     * inside a level PUs are in raster order of their position in the CTU. */
    int idx = 0;
    for (int lvl = 0; lvl < 4; lvl++)
    {
        int size = 64 >> lvl, n = 1 << lvl;
        for (int by = 0; by < n; by++)
            for (int bx = 0; bx < n; bx++, idx++)
            {
                const uint8_t *f = fenc->origin + (ctu_x + bx * size) + (intptr_t)(ctu_y + by * size) * fenc->stride;
                const uint8_t *r = ref->origin + (ctu_x + bx * size) + (intptr_t)(ctu_y + by * size) * ref->stride;
                int best = INT_MAX, bmx = 0, bmy = 0;
                for (int my = mvmin_y; my <= mvmax_y; my++)
                    for (int mx = mvmin_x; mx <= mvmax_x; mx++)
                    {
                        int cost = sad_wh(size, size, f, fenc->stride, r + mx + my * ref->stride, ref->stride)
                                   + (uint16_t)(tab_at(costtab, qmvp_x, s16(mx << 2)) + tab_at(costtab, qmvp_y, s16(my << 2)));
                        if (cost < best)
                        {
                            best = cost;
                            bmx = mx;
                            bmy = my;
                        }
                    }
                out[idx].cost = best;
                out[idx].mvx = s16(bmx << 2);
                out[idx].mvy = s16(bmy << 2);
            }
    }
}
