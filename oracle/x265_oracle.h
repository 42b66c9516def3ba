/* x265_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-thread-per-job CPU restatement of the x265 motion-estimation
 * cost path (SAD / SATD / 8-tap luma interpolation / MV bit cost / the
 * MotionEstimate search state machine).  It exists so that the HIP path can be
 * checked bit-for-bit on a machine that has no copy of the reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under x265_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  Every function here is compared against the real
 * reference C primitives / MotionEstimate (built from /root/reference by
 * oracle/Makefile into oracle/_ref/libx265ref.so) by tests/test_oracle_vs_ref.py
 * when that library is present, and against tests/golden/ fixtures (generated
 * from that same library by oracle/gen_golden.py) everywhere else.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/source).
 */
#ifndef X265_ORACLE_H
#define X265_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265O_NUM_PARTS 25
#define X265O_MAX_MVC   12
#define X265O_MC_AVG 0x40 /* OR-ed into x265o_bidir_req::metric: luma_addAvg of 14-bit intermediates instead of pixelavg_pp */
#define X265O_LOWRES_REF 0x80 /* OR-ed into x265o_me_job::method: the reference is a Lowres (ref->isLowres) */

/* Same byte layout as x265_hip_me_job in include/x265_hip.h (declared again
 * here on purpose: the oracle shares no header with the product). */
typedef struct x265o_me_job
{
    int32_t fenc_plane;
    int32_t ref_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h;
    uint8_t method;     /* 0 DIA 1 HEX 2 UMH 3 STAR 4 FULL  (x265.h:143-147) */
    uint8_t subme;      /* 0..7                              (x265.h:192)     */
    int16_t mvmin_x, mvmin_y, mvmax_x, mvmax_y; /* full-pel, inclusive */
    int16_t qmvp_x, qmvp_y;                     /* quarter-pel         */
    int16_t merange;
    uint8_t qp;
    uint8_t num_mvc;
    int16_t mvc[2 * X265O_MAX_MVC];             /* quarter-pel x,y pairs */
} x265o_me_job;

typedef struct x265o_me_result
{
    int32_t cost;
    int16_t mvx, mvy; /* quarter-pel */
} x265o_me_result;

typedef struct x265o_plane
{
    const uint8_t *origin; /* pointer to picture sample (0,0) inside a padded buffer */
    intptr_t stride;
} x265o_plane;

/* one batched cost request: block (x,y,w,h) of fenc vs ref displaced by qmv */
typedef struct x265o_cost_req
{
    int32_t fenc_plane;
    int32_t ref_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h;
    uint8_t metric; /* 0 SAD 1 SATD */
    uint8_t pad;
    int16_t qmv_x, qmv_y;
} x265o_cost_req;

/* block vs the average of two predictions (same layout as x265_hip_bidir_req) */
typedef struct x265o_bidir_req
{
    int32_t fenc_plane;
    int32_t ref0_plane, ref1_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h;
    uint8_t metric;
    uint8_t pad;
    int16_t qmv0_x, qmv0_y, qmv1_x, qmv1_y;
} x265o_bidir_req;

int  x265o_partition_from_sizes(int w, int h);
void x265o_partition_size(int part, int *w, int *h);

int  x265o_sad(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride);
void x265o_sad_x3(int part, const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2,
                  intptr_t rstride, int32_t *res);
void x265o_sad_x4(int part, const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2,
                  const uint8_t *r3, intptr_t rstride, int32_t *res);
int  x265o_satd(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride);
/* mode-decision distortions: sa8d_inter[part] (pixel.cpp:290-376, 1117-1141), sse_pp[part] (pixel.cpp:158-177, 1095) */
int  x265o_sa8d_inter(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride);
int  x265o_sse_pp(int part, const uint8_t *fenc, intptr_t fstride, const uint8_t *fref, intptr_t rstride);
/* metric codes of the cost requests */
enum { X265O_SAD = 0, X265O_SATD = 1, X265O_SA8D = 2, X265O_SSE = 3 };

void x265o_luma_hpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx);
void x265o_luma_vpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx);
void x265o_luma_hps(int part, const uint8_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx,
                    int isRowExt);
void x265o_luma_vsp(int part, const int16_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int coeffIdx);
void x265o_luma_copy_pp(int part, uint8_t *dst, intptr_t dstride, const uint8_t *src, intptr_t sstride);

/* fills out[65536]; out[32768 + d] is the lambda-scaled cost of MVD component d */
void x265o_bitcost_table(int qp, uint16_t *out);

/* MotionEstimate::subpelCompare restated: cost of the block at qmv, metric 0 SAD / 1 SATD */
int  x265o_subpel_cost(const x265o_plane *fenc, const x265o_plane *ref, int x, int y, int w, int h,
                       int qmvx, int qmvy, int metric);

/* Batched front-ends.  costtabs[qp] points at 65536 entries (centre at +32768).
 * nthreads <= 1 runs inline. */
/* one (list, reference) iteration of TEncSearch::predInterSearch (TEncSearch.cpp:2306-2358), same layout as
 * x265_hip_inter_job / x265_hip_inter_result */
typedef struct x265o_inter_job
{
    int32_t fenc_plane, ref_plane, pred_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h, method, subme;
    int16_t cu_x, cu_y;
    int16_t amvp[4];
    int16_t merange, ref_lag;
    uint8_t num_amvp, qp, num_mvc, pad;
    int16_t mvc[2 * X265O_MAX_MVC];
} x265o_inter_job;
typedef struct x265o_inter_result
{
    int32_t cost;
    int16_t mvx, mvy;
    int32_t amvp_sad[2];
    int16_t mvmin_x, mvmin_y, mvmax_x, mvmax_y;
    int16_t mvp_x, mvp_y;
    uint8_t mvp_idx, pad[3];
} x265o_inter_result;
void x265o_inter_search_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_inter_job *jobs, int n,
                              int pic_w, int pic_h, x265o_inter_result *out);

/* What predInterSearch does with a prediction unit's iterations (TEncSearch.cpp:2342-2358, 2360-2439, 2449-2499):
 * bits and cost of every iteration with xCheckBestMVP's second look at the AMVP index (:2565-2598), the best reference per
 * list, the bidirectional candidate with its zero-vector variant, and the choice between the three.  Same layout as
 * x265_hip_inter_pu / x265_hip_inter_side / x265_hip_inter_pu_result. */
typedef struct x265o_inter_pu
{
    int32_t first_job;        /* its iterations: jobs first_job .., list 0's references in order, then list 1's */
    uint8_t num_ref[2];       /* numRefIdx[l]; num_ref[1] = 0 in a P slice */
    uint8_t list_sel_bits[3]; /* xGetBlkBits */
    uint8_t no_bidir;         /* a P slice, or TComDataCU::isBipredRestriction */
    uint8_t pad[2];
    uint32_t lambda_sad;      /* RDCost::m_lambdaSAD */
} x265o_inter_pu;
typedef struct x265o_inter_side
{
    uint32_t cost; /* UINT32_MAX: none */
    int32_t bits;
    int16_t mvx, mvy, mvp_x, mvp_y;
    int8_t ref, mvp_idx;
    uint8_t pad[2];
} x265o_inter_side;
typedef struct x265o_inter_pu_result
{
    x265o_inter_side list[2];  /* the best iteration of each list */
    x265o_inter_side bidir[2]; /* the two halves of the bidirectional candidate (cost, bits: as list[] or the zero-vector form's) */
    uint32_t bidir_cost;       /* UINT32_MAX: not tried */
    int32_t bidir_bits;
    uint8_t inter_dir;         /* the cheapest of the three: 1 list 0, 2 list 1, 3 both (merge is the caller's to compare) */
    uint8_t zero_tried, pad[2];
} x265o_inter_pu_result;
/* BitCost::bitcost (bitcost.h:48-52) of vector (x, y) against predictor (px, py), quarter-pel */
int x265o_mv_bits(int x, int y, int px, int py);
/* xCheckBestMVP (TEncSearch.cpp:2565-2598): cand = m_mvCand[0..1] as x y pairs */
void x265o_check_best_mvp(const int16_t cand[4], int mvx, int mvy, int *mvp_idx, uint32_t *bits, uint32_t *cost, uint32_t lambda_sad);
void x265o_inter_decide_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_inter_job *jobs,
                              const x265o_inter_result *res, const x265o_inter_pu *pus, int npu, int pic_w, int pic_h,
                              x265o_inter_pu_result *out);

/* the motion search of a whole lookahead frame cost estimate, one (picture, reference) pair per job: EstimateRow::
 * estimateCUCost's list loop (encoder/slicetype.cpp:1540-1584) over every 8x8 CU; same layout as x265_hip_lookahead_job */
typedef struct x265o_lookahead_job
{
    int32_t fenc_plane, ref_plane;
    int16_t merange;
    uint8_t method, subme, qp, pad[3];
} x265o_lookahead_job;
void x265o_lookahead_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_lookahead_job *jobs, int n,
                           int width_in_cu, int height_in_cu, x265o_me_result *out, int nthreads);

void x265o_me_batch_ops(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_me_job *jobs, int n,
                        x265o_me_result *out, int nthreads, double *ops5);
void x265o_cost_batch(const x265o_plane *planes, const x265o_cost_req *reqs, int n, int32_t *out, int nthreads);
void x265o_me_batch(const x265o_plane *planes, const uint16_t *const *costtabs, const x265o_me_job *jobs, int n,
                    x265o_me_result *out, int nthreads);

/* bidirectional candidate: TEncSearch.cpp:2360-2412 (two xPredInterLumaBlk predictions, pixelavg_pp, sad / satd) */
int  x265o_bidir_cost(const x265o_plane *fenc, const x265o_plane *ref0, const x265o_plane *ref1, int x, int y, int w,
                      int h, int qmv0x, int qmv0y, int qmv1x, int qmv1y, int metric);
void x265o_bidir_cost_batch(const x265o_plane *planes, const x265o_bidir_req *reqs, int n, int32_t *out);

/* motion-compensated luma prediction (TComPrediction.cpp:330-473): the remaining interpolation primitives
 * (ipfilter.cpp:37-54, 203-239, 284-317, 362-369), luma_addAvg (pixel.cpp:819-839) and the batch that writes
 * predictions into planes[dst_plane] (uni-directional when ref1_plane < 0; ref1_plane == X265O_PRED_WEIGHTED:
 * weighted uni-directional prediction, w0 = qmv1_x, offset = qmv1_y, see x265o_weight_sp; weighted_bi != 0 with
 * ref1_plane >= 0: weighted bi-prediction, see x265o_add_weight_bi) */
#define X265O_PRED_WEIGHTED (-2)
typedef struct x265o_pred_req
{
    int32_t dst_plane;
    int32_t ref0_plane, ref1_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h;
    uint8_t log2_denom;  /* the weighted forms: log2WeightDenom */
    uint8_t weighted_bi; /* != 0 with ref1_plane >= 0: xWeightedPredictionBi with the four values below */
    int16_t qmv0_x, qmv0_y, qmv1_x, qmv1_y;
    int16_t wp_w0, wp_o0, wp_w1, wp_o1; /* inputWeight / inputOffset of list 0's and list 1's reference */
} x265o_pred_req;
void x265o_luma_p2s(const uint8_t *src, intptr_t sstride, int16_t *dst, int width, int height);
void x265o_luma_vps(int part, const uint8_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx);
void x265o_luma_vss(int part, const int16_t *src, intptr_t sstride, int16_t *dst, intptr_t dstride, int coeffIdx);
void x265o_luma_hvpp(int part, const uint8_t *src, intptr_t sstride, uint8_t *dst, intptr_t dstride, int idxX, int idxY);
void x265o_luma_addavg(int part, const int16_t *src0, const int16_t *src1, uint8_t *dst, intptr_t sstride0, intptr_t sstride1,
                       intptr_t dstride);
void x265o_weight_sp(const int16_t *src, uint8_t *dst, intptr_t sstride, intptr_t dstride, int width, int height, int w0,
                     int round, int shift, int offset);
void x265o_add_weight_bi(const int16_t *src0, const int16_t *src1, uint8_t *dst, intptr_t sstride0, intptr_t sstride1,
                         intptr_t dstride, int width, int height, int w0, int o0, int w1, int o1, int log2_denom);
void x265o_predict_batch(const x265o_plane *planes, const x265o_pred_req *reqs, int n);
void x265o_mc_satd_batch(const x265o_plane *planes, const x265o_pred_req *reqs, int n, int32_t *out);

/* TEncSearch::xMergeEstimation (TEncSearch.cpp:2160-2226): same layout as x265_hip_merge_pu / _result */
#define X265O_MRG_MAX_NUM_CANDS 5
typedef struct x265o_merge_cand
{
    int32_t ref_plane[2];
    int16_t mv[4];
    int16_t wp_w[2], wp_o[2];
    uint8_t log2_denom[2];
    uint8_t pad[2];
} x265o_merge_cand;
typedef struct x265o_merge_pu
{
    int32_t fenc_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h, num_cand, weighted;
    int16_t cu_x, cu_y, mvy_limit;
    uint16_t pad;
    uint32_t lambda_sad;
    x265o_merge_cand cand[X265O_MRG_MAX_NUM_CANDS];
} x265o_merge_pu;
typedef struct x265o_merge_result
{
    uint32_t cost, bits;
    uint8_t index, inter_dir, pad[2];
    int32_t satd[X265O_MRG_MAX_NUM_CANDS];
} x265o_merge_result;
void x265o_merge_estimate_batch(const x265o_plane *planes, const x265o_merge_pu *pus, int n, int pic_w, int pic_h,
                                x265o_merge_result *out);

/* lookahead planes: frame_init_lowres_core (pixel.cpp:595-618) and extendPicBorder (pixel.cpp:893-911) */
void x265o_frame_init_lowres(const uint8_t *src0, uint8_t *dst0, uint8_t *dsth, uint8_t *dstv, uint8_t *dstc,
                             intptr_t src_stride, intptr_t dst_stride, int width, int height);
void x265o_extend_pic_border(uint8_t *pic, intptr_t stride, int width, int height, int marginX, int marginY);
void x265o_pixelavg_pp(int part, uint8_t *dst, intptr_t dstride, const uint8_t *src0, intptr_t sstride0,
                       const uint8_t *src1, intptr_t sstride1);
/* weight_pp (pixel.cpp:510-529): the weighted copy of a reference plane, MotionReference::applyWeight's kernel */
void x265o_weight_pp(const uint8_t *src, uint8_t *dst, intptr_t sstride, intptr_t dstride, int width, int height,
                     int w0, int round, int shift, int offset);

/* Hierarchical full search used by config 2: for one 64x64 CTU at (ctu_x,ctu_y), every square PU of
 * sizes 64/32/16/8 (85 PUs, z-order within each level, levels 64,32,16,8) is searched over
 * [mvmin,mvmax] in raster order with SAD + mvcost; equivalent to motionEstimate(FULL) restricted to
 * its integer stage with bcost initialised to INT_MAX.  out: 85 results (cost, full-pel mv<<2). */
void x265o_ctu_fullsearch(const x265o_plane *fenc, const x265o_plane *ref, const uint16_t *costtab,
                          int ctu_x, int ctu_y, int mvmin_x, int mvmin_y, int mvmax_x, int mvmax_y,
                          int qmvp_x, int qmvp_y, x265o_me_result *out);

#ifdef __cplusplus
}
#endif
#endif
