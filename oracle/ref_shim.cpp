/* ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" wrapper around the REAL reference implementation (the C primitives filled by
 * Setup_C_Primitives and the MotionEstimate class), compiled by oracle/Makefile together with
 * the reference's own sources where they lie under /root/reference.  It contains no arithmetic
 * of its own: every function forwards to the reference.  Same signatures as x265_oracle.h with
 * the x265r_ prefix so tests can swap the two libraries.
 *
 * Built only when /root/reference exists (this container); output goes to oracle/_ref/.
 */
#include "TLibCommon/TComRom.h"
#include "primitives.h"
#include "common.h"
#include "lowres.h"
#include "motion.h"
#include "bitcost.h"
#include "x265.h"

#include <pthread.h>
#include <string.h>

#include "x265_oracle.h" /* job / result / plane structs only */

using namespace x265;

namespace {
EncoderPrimitives g_c;
pthread_once_t g_once = PTHREAD_ONCE_INIT;

void init_once()
{
    memset(&g_c, 0, sizeof(g_c));
    Setup_C_Primitives(g_c);
    /* MotionEstimate reads the global table; give it the C primitives as well */
    Setup_C_Primitives(primitives);
    initROM();
}

inline void init() { pthread_once(&g_once, init_once); }

/* exposes the protected cost-table pointer of BitCost */
struct CostPeek : public BitCost
{
    const uint16_t *centre(int qp)
    {
        setQP(qp);
        return m_cost;
    }
};
}

extern "C" {
int x265r_partition_from_sizes(int w, int h)
{
    return partitionFromSizes(w, h);
}

int x265r_sad(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    init();
    return g_c.sad[part]((pixel *)fenc, fstride, (pixel *)fref, rstride);
}

void x265r_sad_x3(int part, const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2,
                  intptr_t rstride, int32_t *res)
{
    init();
    g_c.sad_x3[part]((pixel *)fenc, (pixel *)r0, (pixel *)r1, (pixel *)r2, rstride, res);
}

void x265r_sad_x4(int part, const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2,
                  const uint8_t *r3, intptr_t rstride, int32_t *res)
{
    init();
    g_c.sad_x4[part]((pixel *)fenc, (pixel *)r0, (pixel *)r1, (pixel *)r2, (pixel *)r3, rstride, res);
}

int x265r_satd(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    init();
    return g_c.satd[part]((pixel *)fenc, fstride, (pixel *)fref, rstride);
}

int x265r_sa8d_inter(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    init();
    return g_c.sa8d_inter[part]((pixel *)fenc, fstride, (pixel *)fref, rstride);
}

int x265r_sse_pp(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride)
{
    init();
    return g_c.sse_pp[part]((pixel *)fenc, fstride, (pixel *)fref, rstride);
}

void x265r_luma_hpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx)
{
    init();
    g_c.luma_hpp[part]((pixel *)src, sstride, dst, dstride, coeffIdx);
}

void x265r_luma_vpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx)
{
    init();
    g_c.luma_vpp[part]((pixel *)src, sstride, dst, dstride, coeffIdx);
}

void x265r_luma_hps(int part, const uint8_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx,
                    int isRowExt)
{
    init();
    g_c.luma_hps[part]((pixel *)src, sstride, dst, dstride, coeffIdx, isRowExt);
}

void x265r_luma_vsp(int part, const int16_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx)
{
    init();
    g_c.luma_vsp[part]((int16_t *)src, sstride, dst, dstride, coeffIdx);
}

void x265r_luma_copy_pp(int part, uint8_t *dst, intptr_t dstride, const uint8_t *src, intptr_t sstride)
{
    init();
    g_c.luma_copy_pp[part](dst, dstride, (pixel *)src, sstride);
}

void x265r_bitcost_table(int qp, uint16_t *out)
{
    init();
    CostPeek p;
    const uint16_t *c = p.centre(qp);
    /* the reference never writes entry -32768 (bitcost.cpp:48); mirror x265o's convention */
    memcpy(out + 1, c - 32767, 65535 * sizeof(uint16_t));
    out[0] = out[1];
}

/* BitCost::bitcost (bitcost.h:48-52) with the predictor set by setMVP, as predInterSearch and xCheckBestMVP use it */
int x265r_mv_bits(int x, int y, int px, int py)
{
    init();
    CostPeek p;
    p.centre(30); /* (CalculateLogs: the bit sizes do not depend on the QP) */
    p.setMVP(MV(px, py));
    return p.bitcost(MV(x, y));
}

static void run_me(MotionEstimate &me, const x265o_plane *planes, const x265o_me_job *j, x265o_me_result *r)
{
    const x265o_plane &f = planes[j->fenc_plane];
    const x265o_plane &p = planes[j->ref_plane];
    ReferencePlanes ref;
    ref.fpelPlane = (pixel *)p.origin;
    ref.lumaStride = (int)p.stride;
    ref.isLowres = (j->method & X265O_LOWRES_REF) != 0;
    ref.isWeighted = false;
    if (ref.isLowres) /* a Lowres: planes[ref_plane .. ref_plane + 3] are lowresPlane[0..3] (lowres.cpp:66-69, 174) */
        for (int i = 0; i < 4; i++)
            ref.lowresPlane[i] = (pixel *)planes[j->ref_plane + i].origin;
    me.setSourcePlane((pixel *)f.origin, f.stride);
    me.setSearchMethod(j->method & ~X265O_LOWRES_REF);
    me.setSubpelRefine(j->subme);
    me.setQP(j->qp);
    me.setSourcePU(j->pu_x + j->pu_y * (int)f.stride, j->pu_w, j->pu_h);
    /* blockOffset is shared between fenc and ref planes in the reference (motion.cpp:537);
     * both planes therefore must have the same stride */
    MV mvmin(j->mvmin_x, j->mvmin_y), mvmax(j->mvmax_x, j->mvmax_y), mvp(j->qmvp_x, j->qmvp_y), outmv;
    MV mvc[X265O_MAX_MVC];
    for (int i = 0; i < j->num_mvc; i++)
        mvc[i] = MV(j->mvc[2 * i], j->mvc[2 * i + 1]);
    r->cost = me.motionEstimate(&ref, mvmin, mvmax, mvp, j->num_mvc, mvc, j->merange, outmv);
    r->mvx = outmv.x;
    r->mvy = outmv.y;
}

struct MeSlice
{
    const x265o_plane *planes;
    const x265o_me_job *jobs;
    x265o_me_result *out;
    int lo, hi;
};

static void *me_slice(void *p)
{
    MeSlice *s = (MeSlice *)p;
    MotionEstimate me; /* one per worker, as cturow.h:59 does */
    for (int i = s->lo; i < s->hi; i++)
        run_me(me, s->planes, &s->jobs[i], &s->out[i]);
    return 0;
}

/* costtabs is ignored: the reference builds its own tables from job->qp */
void x265r_me_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_me_job *jobs, int n,
                    x265o_me_result *out, int nthreads)
{
    (void)costtabs;
    init();
    {
        /* build every needed cost table up front, single-threaded */
        CostPeek p;
        for (int i = 0; i < n; i++)
            p.centre(jobs[i].qp);
    }
    if (nthreads <= 1 || n < 2 * nthreads)
    {
        MeSlice s = { planes, jobs, out, 0, n };
        me_slice(&s);
        return;
    }
    pthread_t *th = new pthread_t[nthreads];
    MeSlice *sl = new MeSlice[nthreads];
    for (int t = 0; t < nthreads; t++)
    {
        MeSlice s = { planes, jobs, out, (int)((int64_t)n * t / nthreads), (int)((int64_t)n * (t + 1) / nthreads) };
        sl[t] = s;
        pthread_create(&th[t], 0, me_slice, &sl[t]);
    }
    for (int t = 0; t < nthreads; t++)
        pthread_join(th[t], 0);
    delete[] sl;
    delete[] th;
}

void x265r_frame_init_lowres(const uint8_t *src0, uint8_t *dst0, uint8_t *dsth, uint8_t *dstv, uint8_t *dstc,
                             intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
    init();
    g_c.frame_init_lowres_core((pixel *)src0, dst0, dsth, dstv, dstc, src_stride, dst_stride, width, height);
}

void x265r_extend_pic_border(uint8_t *pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    init();
    extendPicBorder(pic, (int)stride, width, height, marginX, marginY); /* uses primitives.extendRowBorder */
}

/* the call sequence of TComPrediction::xPredInterLumaBlk (TComPrediction.cpp:400-433) on the reference's primitives */
static void ref_predict_luma(int part, int w, pixel *p, intptr_t stride, int qx, int qy, pixel *dst)
{
    int xf = qx & 3, yf = qy & 3;
    p += (qx >> 2) + (qy >> 2) * stride;
    if (!(xf | yf))
        g_c.luma_copy_pp[part](dst, 64, p, stride);
    else if (!yf)
        g_c.luma_hpp[part](p, stride, dst, 64, xf);
    else if (!xf)
        g_c.luma_vpp[part](p, stride, dst, 64, yf);
    else
    {
        static thread_local int16_t immed[64 * (64 + 8)];
        g_c.luma_hps[part](p, stride, immed, w, xf, 1);
        g_c.luma_vsp[part](immed + 3 * w, w, dst, 64, yf);
    }
}

static pixelcmp_t ref_cmp(int metric, int part)
{
    return metric == X265O_SSE ? g_c.sse_pp[part] : metric == X265O_SA8D ? g_c.sa8d_inter[part] : metric ? g_c.satd[part] : g_c.sad[part];
}

static void ref_predict_luma_short(int part, int w, int h, pixel *p, intptr_t stride, int qx, int qy, int16_t *dst);

/* TEncSearch.cpp:2360-2412: pixelavg_pp of the two predictions, then satd / sad of the block */
void x265r_bidir_cost_batch(const x265o_plane *planes, const x265o_bidir_req *reqs, int n, int32_t *out)
{
    init();
    static thread_local pixel p0[64 * 64], p1[64 * 64], avg[64 * 64];
    for (int i = 0; i < n; i++)
    {
        const x265o_bidir_req *r = &reqs[i];
        const x265o_plane &f = planes[r->fenc_plane], &a = planes[r->ref0_plane], &b = planes[r->ref1_plane];
        if (r->metric & X265O_LOWRES_REF)
        {
            /* EstimateRow::estimateCUCost, slicetype.cpp:1585-1595, on the reference's ReferencePlanes::lowresMC */
            ReferencePlanes r0, r1;
            for (int k = 0; k < 4; k++)
            {
                r0.lowresPlane[k] = (pixel *)planes[r->ref0_plane + k].origin;
                r1.lowresPlane[k] = (pixel *)planes[r->ref1_plane + k].origin;
            }
            r0.lumaStride = (int)a.stride;
            r1.lumaStride = (int)b.stride;
            pixel buf0[64], buf1[64], ref[64];
            intptr_t s0 = 8, s1 = 8;
            pixel *src0 = r0.lowresMC(r->pu_x + (intptr_t)r->pu_y * a.stride, MV(r->qmv0_x, r->qmv0_y), buf0, s0);
            pixel *src1 = r1.lowresMC(r->pu_x + (intptr_t)r->pu_y * b.stride, MV(r->qmv1_x, r->qmv1_y), buf1, s1);
            g_c.pixelavg_pp[LUMA_8x8](ref, 8, src0, s0, src1, s1, 32);
            pixel *fe8 = (pixel *)f.origin + r->pu_x + (intptr_t)r->pu_y * f.stride;
            out[i] = ref_cmp(r->metric & 3, LUMA_8x8)(fe8, f.stride, ref, 8);
            continue;
        }
        int part = partitionFromSizes(r->pu_w, r->pu_h);
        if (r->metric & X265O_MC_AVG)
        {
            /* motionCompensation of a bi-directional block + bufSATD (TEncSearch.cpp:2209-2210) */
            static thread_local int16_t s0[64 * 64], s1[64 * 64];
            ref_predict_luma_short(part, r->pu_w, r->pu_h, (pixel *)a.origin + r->pu_x + (intptr_t)r->pu_y * a.stride, a.stride,
                                   r->qmv0_x, r->qmv0_y, s0);
            ref_predict_luma_short(part, r->pu_w, r->pu_h, (pixel *)b.origin + r->pu_x + (intptr_t)r->pu_y * b.stride, b.stride,
                                   r->qmv1_x, r->qmv1_y, s1);
            g_c.luma_addAvg[part](s0, s1, avg, 64, 64, 64);
            out[i] = ref_cmp(r->metric & 3, part)((pixel *)f.origin + r->pu_x + (intptr_t)r->pu_y * f.stride, f.stride, avg, 64);
            continue;
        }
        ref_predict_luma(part, r->pu_w, (pixel *)a.origin + r->pu_x + (intptr_t)r->pu_y * a.stride, a.stride, r->qmv0_x, r->qmv0_y, p0);
        ref_predict_luma(part, r->pu_w, (pixel *)b.origin + r->pu_x + (intptr_t)r->pu_y * b.stride, b.stride, r->qmv1_x, r->qmv1_y, p1);
        g_c.pixelavg_pp[part](avg, 64, p0, 64, p1, 64, 32);
        pixel *fe = (pixel *)f.origin + r->pu_x + (intptr_t)r->pu_y * f.stride;
        out[i] = ref_cmp(r->metric & 3, part)(fe, f.stride, avg, 64);
    }
}

void x265r_luma_p2s(const uint8_t *src, intptr_t sstride, int16_t *dst, int width, int height)
{
    init();
    g_c.luma_p2s((pixel *)src, sstride, dst, width, height);
}

void x265r_luma_vps(int part, const uint8_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx)
{
    init();
    g_c.luma_vps[part]((pixel *)src, sstride, dst, dstride, coeffIdx);
}

void x265r_luma_vss(int part, const int16_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx)
{
    init();
    g_c.luma_vss[part]((int16_t *)src, sstride, dst, dstride, coeffIdx);
}

void x265r_luma_hvpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int idxX, int idxY)
{
    init();
    g_c.luma_hvpp[part]((pixel *)src, sstride, dst, dstride, idxX, idxY);
}

void x265r_luma_addavg(int part, const int16_t *src0, const int16_t *src1, uint8_t *dst, intptr_t sstride0, intptr_t sstride1,
                       intptr_t dstride)
{
    init();
    g_c.luma_addAvg[part]((int16_t *)src0, (int16_t *)src1, dst, sstride0, sstride1, dstride);
}

/* the call sequence of TComPrediction::xPredInterLumaBlk to a ShortYuv (TComPrediction.cpp:436-473) */
static void ref_predict_luma_short(int part, int w, int h, pixel *p, intptr_t stride, int qx, int qy, int16_t *dst)
{
    int xf = qx & 3, yf = qy & 3;
    p += (qx >> 2) + (qy >> 2) * stride;
    if (!(xf | yf))
        g_c.luma_p2s(p, stride, dst, w, h);
    else if (!yf)
        g_c.luma_hps[part](p, stride, dst, 64, xf, 0);
    else if (!xf)
        g_c.luma_vps[part](p, stride, dst, 64, yf);
    else
    {
        static thread_local int16_t immed[64 * (64 + 8)];
        g_c.luma_hps[part](p, stride, immed, w, xf, 1);
        g_c.luma_vss[part](immed + 3 * w, w, dst, 64, yf);
    }
}

/* TComPrediction::xPredInterUni / xPredInterBi for luma (TComPrediction.cpp:330-387), on the reference's primitives */
void x265r_predict_batch(const x265o_plane *planes, const x265o_pred_req *reqs, int n)
{
    init();
    static thread_local int16_t s0[64 * 64], s1[64 * 64];
    for (int i = 0; i < n; i++)
    {
        const x265o_pred_req *r = &reqs[i];
        const x265o_plane &d = planes[r->dst_plane], &a = planes[r->ref0_plane];
        int part = partitionFromSizes(r->pu_w, r->pu_h);
        pixel *dst = (pixel *)d.origin + r->pu_x + (intptr_t)r->pu_y * d.stride;
        pixel *pa = (pixel *)a.origin + r->pu_x + (intptr_t)r->pu_y * a.stride;
        if (r->ref1_plane == X265O_PRED_WEIGHTED)
        {
            /* motionCompensation with getUseWP (TComPrediction.cpp:265-279) + addWeightUni (TComWeightPrediction.cpp:433-458) */
            int shift = r->log2_denom + (IF_INTERNAL_PREC - X265_DEPTH);
            ref_predict_luma_short(part, r->pu_w, r->pu_h, pa, a.stride, r->qmv0_x, r->qmv0_y, s0);
            g_c.weight_sp(s0, dst, 64, d.stride, r->pu_w, r->pu_h, r->qmv1_x, shift ? 1 << (shift - 1) : 0, shift, r->qmv1_y);
        }
        else if (r->ref1_plane < 0)
        {
            /* xPredInterLumaBlk with the destination's own stride */
            int xf = r->qmv0_x & 3, yf = r->qmv0_y & 3;
            pixel *p = pa + (r->qmv0_x >> 2) + (r->qmv0_y >> 2) * a.stride;
            if (!(xf | yf))
                g_c.luma_copy_pp[part](dst, d.stride, p, a.stride);
            else if (!yf)
                g_c.luma_hpp[part](p, a.stride, dst, d.stride, xf);
            else if (!xf)
                g_c.luma_vpp[part](p, a.stride, dst, d.stride, yf);
            else
            {
                static thread_local int16_t immed[64 * (64 + 8)];
                g_c.luma_hps[part](p, a.stride, immed, r->pu_w, xf, 1);
                g_c.luma_vsp[part](immed + 3 * r->pu_w, r->pu_w, dst, d.stride, yf);
            }
        }
        else
        {
            if (r->weighted_bi)
            {
                /* xWeightedPredictionBi is a member of the encoder's prediction class, not a primitive: it is run by
                 * oracle/_ref/enc_capture --wpbi, which links the whole encoder (tests/golden/wpbi.npz) */
                fprintf(stderr, "x265r_predict_batch: weighted bi-prediction is not reachable through the primitive table\n");
                abort();
            }
            const x265o_plane &b = planes[r->ref1_plane];
            pixel *pb = (pixel *)b.origin + r->pu_x + (intptr_t)r->pu_y * b.stride;
            ref_predict_luma_short(part, r->pu_w, r->pu_h, pa, a.stride, r->qmv0_x, r->qmv0_y, s0);
            ref_predict_luma_short(part, r->pu_w, r->pu_h, pb, b.stride, r->qmv1_x, r->qmv1_y, s1);
            g_c.luma_addAvg[part](s0, s1, dst, 64, 64, d.stride);
        }
    }
}

void x265r_pixelavg_pp(int part, uint8_t *dst, intptr_t dstride, const uint8_t *src0, intptr_t sstride0,
                       const uint8_t *src1, intptr_t sstride1)
{
    init();
    g_c.pixelavg_pp[part](dst, dstride, (pixel *)src0, sstride0, (pixel *)src1, sstride1, 32);
}

void x265r_weight_pp(const uint8_t *src, uint8_t *dst, intptr_t sstride, intptr_t dstride, int width, int height,
                     int w0, int round, int shift, int offset)
{
    init();
    g_c.weight_pp((pixel *)src, dst, sstride, dstride, width, height, w0, round, shift, offset);
}

int x265r_subpel_cost(const x265o_plane *fenc, const x265o_plane *ref, int x, int y, int w, int h, int qmvx,
                      int qmvy, int metric)
{
    init();
    MotionEstimate me;
    ReferencePlanes rp;
    rp.fpelPlane = (pixel *)ref->origin;
    rp.lumaStride = (int)ref->stride;
    me.setSourcePlane((pixel *)fenc->origin, fenc->stride);
    me.setSourcePU(x + y * (int)fenc->stride, w, h);
    int part = partitionFromSizes(w, h);
    pixelcmp_t cmp = metric == X265O_SSE ? primitives.sse_pp[part] : metric == X265O_SA8D ? primitives.sa8d_inter[part]
                   : metric ? primitives.satd[part] : primitives.sad[part];
    return me.subpelCompare(&rp, MV(qmvx, qmvy), cmp);
}

struct CostSlice
{
    const x265o_plane *planes;
    const x265o_cost_req *reqs;
    int32_t *out;
    int first, last;
};

/* one MotionEstimate per thread, as the encoder keeps one per CTU row (cturow.h:59) */
static void *cost_slice(void *arg)
{
    CostSlice *s = (CostSlice *)arg;
    MotionEstimate me;
    ReferencePlanes rp;
    for (int i = s->first; i < s->last; i++)
    {
        const x265o_cost_req *r = &s->reqs[i];
        const x265o_plane *fenc = &s->planes[r->fenc_plane], *ref = &s->planes[r->ref_plane];
        rp.fpelPlane = (pixel *)ref->origin;
        rp.lumaStride = (int)ref->stride;
        me.setSourcePlane((pixel *)fenc->origin, fenc->stride);
        me.setSourcePU(r->pu_x + r->pu_y * (int)fenc->stride, r->pu_w, r->pu_h);
        const int part = partitionFromSizes(r->pu_w, r->pu_h);
        pixelcmp_t cmp = r->metric == X265O_SSE ? primitives.sse_pp[part] : r->metric == X265O_SA8D ? primitives.sa8d_inter[part]
                       : r->metric ? primitives.satd[part] : primitives.sad[part];
        s->out[i] = me.subpelCompare(&rp, MV(r->qmv_x, r->qmv_y), cmp);
    }
    return 0;
}

void x265r_cost_batch(const x265o_plane *planes, const x265o_cost_req *reqs, int n, int32_t *out, int nthreads)
{
    init();
    if (nthreads <= 1 || n < 2 * nthreads)
    {
        CostSlice s = { planes, reqs, out, 0, n };
        cost_slice(&s);
        return;
    }
    pthread_t *th = new pthread_t[nthreads];
    CostSlice *sl = new CostSlice[nthreads];
    for (int t = 0; t < nthreads; t++)
    {
        CostSlice s = { planes, reqs, out, (int)((int64_t)n * t / nthreads), (int)((int64_t)n * (t + 1) / nthreads) };
        sl[t] = s;
        pthread_create(&th[t], 0, cost_slice, &sl[t]);
    }
    for (int t = 0; t < nthreads; t++)
        pthread_join(th[t], 0);
    delete[] th;
    delete[] sl;
}

/* config-2 building block via the reference: motionEstimate's FULL raster is not separable from
 * its pre-tests, so this drives the reference sad primitive in the same raster order instead. */
void x265r_ctu_fullsearch(const x265o_plane *fenc, const x265o_plane *ref, const uint16_t *costtab, int ctu_x,
                          int ctu_y, int mvmin_x, int mvmin_y, int mvmax_x, int mvmax_y, int qmvp_x, int qmvp_y,
                          x265o_me_result *out)
{
    init();
    int idx = 0;
    for (int lvl = 0; lvl < 4; lvl++)
    {
        int size = 64 >> lvl, n = 1 << lvl;
        int part = partitionFromSizes(size, size);
        for (int by = 0; by < n; by++)
            for (int bx = 0; bx < n; bx++, idx++)
            {
                pixel *f = (pixel *)fenc->origin + (ctu_x + bx * size) + (intptr_t)(ctu_y + by * size) * fenc->stride;
                pixel *r = (pixel *)ref->origin + (ctu_x + bx * size) + (intptr_t)(ctu_y + by * size) * ref->stride;
                int best = 0x7fffffff, bmx = 0, bmy = 0;
                for (int my = mvmin_y; my <= mvmax_y; my++)
                    for (int mx = mvmin_x; mx <= mvmax_x; mx++)
                    {
                        MV q = MV(mx, my) << 2;
                        uint16_t mvc = (uint16_t)(costtab[(uint16_t)(q.x - qmvp_x + 32768)] +
                                                  costtab[(uint16_t)(q.y - qmvp_y + 32768)]);
                        int cost = g_c.sad[part](f, fenc->stride, r + mx + my * ref->stride, ref->stride) + mvc;
                        if (cost < best)
                        {
                            best = cost;
                            bmx = mx;
                            bmy = my;
                        }
                    }
                out[idx].cost = best;
                out[idx].mvx = (int16_t)(bmx << 2);
                out[idx].mvy = (int16_t)(bmy << 2);
            }
    }
}
}

/* ns per call of one reference C primitive, TestBench-style loop with the argument shapes of
 * test/pixelharness.cpp:1480-1521 and test/ipfilterharness.cpp:703-757 (fenc stride 64, reference stride 64 + a few).
 * kind: 0 sad, 1 sad_x3, 2 sad_x4, 3 satd, 4 luma_hpp, 5 luma_vpp, 6 luma_hps (row extension) + luma_vsp */
#include <time.h>
extern "C" double x265r_time_primitive(int kind, int part, int iters)
{
    init();
    static pixel fenc[64 * 64 + 64], frefbuf[80 * 200], dst[64 * 64];
    static int16_t immed[64 * (64 + 8)];
    static int seeded = 0;
    if (!seeded)
    {
        unsigned s = 12345;
        for (size_t i = 0; i < sizeof(fenc); i++)
            fenc[i] = (pixel)((s = s * 1664525u + 1013904223u) >> 24);
        for (size_t i = 0; i < sizeof(frefbuf); i++)
            frefbuf[i] = (pixel)((s = s * 1664525u + 1013904223u) >> 24);
        seeded = 1;
    }
    const intptr_t rs = 80;
    pixel *fref = frefbuf + 8 * rs + 8;
    int w = 0, h = 0;
    for (int ww = 4; ww <= 64; ww += 4)
        for (int hh = 4; hh <= 64; hh += 4)
            if (partitionFromSizes(ww, hh) == part)
                w = ww, h = hh;
    (void)h;
    ALIGN_VAR_16(int32_t, res[4]);
    volatile int sink = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < iters; i++)
    {
        pixel *r = fref + (i & 7);
        switch (kind)
        {
        case 0: sink += g_c.sad[part](fenc, 64, r, rs); break;
        case 1: g_c.sad_x3[part](fenc, r, r + 1, r - 1, rs, res); sink += res[0]; break;
        case 2: g_c.sad_x4[part](fenc, r, r + 1, r - 1, r - rs, rs, res); sink += res[0]; break;
        case 3: sink += g_c.satd[part](fenc, 64, r, rs); break;
        case 4: g_c.luma_hpp[part](r, rs, dst, 64, 1 + (i % 3)); sink += dst[0]; break;
        case 5: g_c.luma_vpp[part](r, rs, dst, 64, 1 + (i % 3)); sink += dst[0]; break;
        default:
            g_c.luma_hps[part](r, rs, immed, w, 1 + (i % 3), 1);
            g_c.luma_vsp[part](immed + 3 * w, w, dst, 64, 3 - (i % 3));
            sink += dst[0];
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / iters;
}
