/* x265_hip.h -- C ABI of the MI355X (gfx950) implementation of x265's motion-estimation cost path.
 *
 * This is the drop-in boundary: plain C, pointers and sizes only.  Two groups of entry points:
 *
 *  (1) Primitive level -- the exact signatures of the reference's function-pointer table
 *      (struct EncoderPrimitives, source/common/primitives.h:191-283 of the reference) for the
 *      entries on the ME cost path.  x265_hip_setup_primitives() fills a table of such pointers;
 *      a maintainer's Setup_HIP_Primitives() copies them over the C/asm entries.
 *      They take HOST pointers, stage the operands through the GPU and return synchronously, so
 *      they are bit-exact drop-ins but not fast; throughput comes from group (2).
 *
 *  (2) Batched level -- whole-CTU-row / whole-frame candidate sets per launch with planes resident
 *      in HBM.  These have no counterpart in the reference (it calls one primitive per candidate
 *      from MotionEstimate::motionEstimate, source/encoder/motion.cpp:526-1124); each one cites the
 *      reference code whose results it reproduces.
 *
 * Every function returns 0 on success or a negative X265_HIP_E* code; none throws.  All functions
 * are thread-safe; a context serialises its own launches on one HIP stream.
 *
 * There is no CPU fallback: if no gfx950 device is usable the calls fail with X265_HIP_ENODEV.
 */
#ifndef X265_HIP_H
#define X265_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265_HIP_OK        0
#define X265_HIP_ENODEV   -1 /* no usable gfx950 device / HIP runtime error at init */
#define X265_HIP_EINVAL   -2 /* bad argument (null pointer, unknown plane id, bad partition ...) */
#define X265_HIP_ENOMEM   -3 /* device or pinned-host allocation failed */
#define X265_HIP_ERUNTIME -4 /* a HIP call failed; x265_hip_last_error() has the text */
#define X265_HIP_ERANGE   -5 /* request exceeds a documented limit */

/* LumaPartitions, in the reference's index order (source/common/primitives.h:38-46): the index is ABI. */
enum
{
    X265_HIP_LUMA_4x4, X265_HIP_LUMA_8x8, X265_HIP_LUMA_8x4, X265_HIP_LUMA_4x8,
    X265_HIP_LUMA_16x16, X265_HIP_LUMA_16x8, X265_HIP_LUMA_8x16, X265_HIP_LUMA_16x12, X265_HIP_LUMA_12x16,
    X265_HIP_LUMA_16x4, X265_HIP_LUMA_4x16,
    X265_HIP_LUMA_32x32, X265_HIP_LUMA_32x16, X265_HIP_LUMA_16x32, X265_HIP_LUMA_32x24, X265_HIP_LUMA_24x32,
    X265_HIP_LUMA_32x8, X265_HIP_LUMA_8x32,
    X265_HIP_LUMA_64x64, X265_HIP_LUMA_64x32, X265_HIP_LUMA_32x64, X265_HIP_LUMA_64x48, X265_HIP_LUMA_48x64,
    X265_HIP_LUMA_64x16, X265_HIP_LUMA_16x64,
    X265_HIP_NUM_LUMA_PARTITIONS
};

/* search methods, values of the reference's X265_*_SEARCH (source/x265.h:143-147) */
enum { X265_HIP_DIA = 0, X265_HIP_HEX = 1, X265_HIP_UMH = 2, X265_HIP_STAR = 3, X265_HIP_FULL = 4 };
/* block comparisons; SA8D and SSE (the mode-decision distortions sa8d_inter / sse_pp) only in x265_hip_dist_batch */
enum { X265_HIP_SAD = 0, X265_HIP_SATD = 1, X265_HIP_SA8D = 2, X265_HIP_SSE = 3 };
/* OR-ed into x265_hip_me_job::method: the reference is a Lowres (ReferencePlanes::isLowres, lowres.h:42):
 * ref_plane .. ref_plane + 3 are lowresPlane[0..3] with one stride, the block must be the lookahead's 8x8 CU,
 * and sub-pel costs come from lowresQPelCost (lowres.h:73-93) along motion.cpp:556-557, 587-588, 1055-1080 */
#define X265_HIP_LOWRES_REF 0x80
/* OR-ed into x265_hip_bidir_req::metric: combine the two predictions as motion compensation does (see there) */
#define X265_HIP_MC_AVG 0x40

#define X265_HIP_MAX_MVC    12  /* AMVP + spatial candidates, <= 11 in the reference (TEncSearch.cpp:2311) */
#define X265_HIP_MAX_PLANES 512 /* a 64-picture GOP segment with its four lookahead planes per picture is 320 */
#define X265_HIP_MAX_QP     82  /* BitCost::BC_MAX_QP, source/encoder/bitcost.h:72 */
#define X265_HIP_FENC_STRIDE 64 /* FENC_STRIDE, source/common/common.h:45 */

typedef struct x265_hip_ctx x265_hip_ctx;

/* ---- (0) context ------------------------------------------------------------------------------ */

int x265_hip_device_count(void);
int x265_hip_ctx_create(int device, x265_hip_ctx **out);
void x265_hip_ctx_destroy(x265_hip_ctx *ctx);
const char *x265_hip_last_error(void); /* thread-local text of the last failure */
/* the context's hipStream_t, for callers that time or order work with HIP events */
void *x265_hip_ctx_stream(x265_hip_ctx *ctx);
int x265_hip_ctx_sync(x265_hip_ctx *ctx);
/* Record-check mode, a debugging aid for callers of the *_device entry points.  Those take their records on trust
 * (a record that names a missing plane or reaches outside one makes the kernel fault the GPU).  With the mode on --
 * by this call, or with X265_HIP_CHECK_RECORDS=1 in the environment when the context is created -- every *_device
 * call first waits for the context's stream, copies the records back to the host and puts them through the checks of
 * the host-array form; a bad record then returns X265_HIP_EINVAL / X265_HIP_ERANGE with a message and nothing is
 * launched.  It costs a synchronisation and a copy per call: not for production runs.  No counterpart in the reference.
 *
 * Mode 2 (X265_HIP_CHECK_RECORDS=2) is the guard on the device, one thread per record, the same checks against the
 * plane directory in device memory; nothing waits and nothing is copied back, and the call itself returns X265_HIP_OK.
 * x265_hip_me_batch_device checks every job in the kernel that sorts its list: a job that fails is searched by no
 * kernel and gets the result (cost -1, mv 0 0).  x265_hip_cost_batch_device, x265_hip_dist_batch_device,
 * x265_hip_refine_batch_device, x265_hip_refine_sections_device, x265_hip_bidir_cost_batch_device,
 * x265_hip_mc_satd_batch_device and x265_hip_fullsearch_device run a kernel that
 * writes a checked copy of the list into a buffer of the context (a record that fails is replaced by a harmless one of
 * its size) in front of the cost kernel, which reads the copy, and one behind it that sets the results of the
 * replaced records to -1 (all nine of a ring; a CTU's 85 results to cost -1, mv 0 0); the caller's list is not written.  The refusals are counted on the
 * device: x265_hip_ctx_guard_status waits for the stream and returns (and clears) their number and the lowest index of
 * one in its list; x265_hip_ctx_sync in this mode returns X265_HIP_ERANGE with a message if there were any since the
 * last look.  The remaining *_device entry points (predict, inter search / decide, merge estimation, lookahead, and
 * the one that allocates nothing) have no guard and keep mode 1's copy-back in mode 2; so do the guarded request lists while
 * the context holds no plane that a replaced record could point at (64x64 samples at the origin with the filters'
 * margins). */
int x265_hip_ctx_check_records(x265_hip_ctx *ctx, int mode);
int x265_hip_ctx_guard_status(x265_hip_ctx *ctx, uint32_t *refused, uint32_t *first_index);

/* ---- (1) primitive level ---------------------------------------------------------------------- */

/* signatures of source/common/primitives.h:123-127,165-168 with pixel = uint8_t */
typedef int (*x265_hip_pixelcmp_t)(uint8_t *fenc, intptr_t fencstride, uint8_t *fref, intptr_t frefstride);
typedef void (*x265_hip_pixelcmp_x3_t)(uint8_t *fenc, uint8_t *fref0, uint8_t *fref1, uint8_t *fref2,
                                       intptr_t frefstride, int32_t *res);
typedef void (*x265_hip_pixelcmp_x4_t)(uint8_t *fenc, uint8_t *fref0, uint8_t *fref1, uint8_t *fref2, uint8_t *fref3,
                                       intptr_t frefstride, int32_t *res);
typedef void (*x265_hip_filter_pp_t)(uint8_t *src, intptr_t srcStride, uint8_t *dst, intptr_t dstStride, int coeffIdx);
typedef void (*x265_hip_filter_hps_t)(uint8_t *src, intptr_t srcStride, int16_t *dst, intptr_t dstStride, int coeffIdx,
                                      int isRowExt);
typedef void (*x265_hip_filter_sp_t)(int16_t *src, intptr_t srcStride, uint8_t *dst, intptr_t dstStride, int coeffIdx);

/* primitives.h:134, 154, 157-158 */
typedef void (*x265_hip_pixelavg_pp_t)(uint8_t *dst, intptr_t dstride, uint8_t *src0, intptr_t sstride0, uint8_t *src1,
                                       intptr_t sstride1, int weight);
typedef void (*x265_hip_weightp_pp_t)(uint8_t *src, uint8_t *dst, intptr_t srcStride, intptr_t dstStride, int width,
                                      int height, int w0, int round, int shift, int offset);
typedef void (*x265_hip_downscale_t)(uint8_t *src0, uint8_t *dstf, uint8_t *dsth, uint8_t *dstv, uint8_t *dstc,
                                     intptr_t src_stride, intptr_t dst_stride, int width, int height);
/* primitives.h:159 */
typedef void (*x265_hip_extendCURowBorder_t)(uint8_t *txt, intptr_t stride, int width, int height, int marginX);

/* primitives.h:167, 169-171, 180 */
typedef void (*x265_hip_filter_ps_t)(uint8_t *src, intptr_t srcStride, int16_t *dst, intptr_t dstStride, int coeffIdx);
typedef void (*x265_hip_filter_ss_t)(int16_t *src, intptr_t srcStride, int16_t *dst, intptr_t dstStride, int coeffIdx);
typedef void (*x265_hip_filter_hv_pp_t)(uint8_t *src, intptr_t srcStride, uint8_t *dst, intptr_t dstStride, int idxX, int idxY);
typedef void (*x265_hip_filter_p2s_t)(uint8_t *src, intptr_t srcStride, int16_t *dst, int width, int height);
typedef void (*x265_hip_addAvg_t)(int16_t *src0, int16_t *src1, uint8_t *dst, intptr_t src0Stride, intptr_t src1Stride,
                                  intptr_t dstStride);

/* the slice of EncoderPrimitives this library replaces; member names are the reference's */
typedef struct x265_hip_primitives
{
    x265_hip_pixelcmp_t    sad[X265_HIP_NUM_LUMA_PARTITIONS];      /* primitives.h:193; C: pixel.cpp:92-109  */
    x265_hip_pixelcmp_x3_t sad_x3[X265_HIP_NUM_LUMA_PARTITIONS];   /* primitives.h:194; C: pixel.cpp:111-131 */
    x265_hip_pixelcmp_x4_t sad_x4[X265_HIP_NUM_LUMA_PARTITIONS];   /* primitives.h:195; C: pixel.cpp:133-156 */
    x265_hip_pixelcmp_t    satd[X265_HIP_NUM_LUMA_PARTITIONS];     /* primitives.h:199; C: pixel.cpp:201-288 */
    x265_hip_filter_pp_t   luma_hpp[X265_HIP_NUM_LUMA_PARTITIONS]; /* primitives.h:221; C: ipfilter.cpp:76-115  */
    x265_hip_filter_hps_t  luma_hps[X265_HIP_NUM_LUMA_PARTITIONS]; /* primitives.h:222; C: ipfilter.cpp:117-160 */
    x265_hip_filter_pp_t   luma_vpp[X265_HIP_NUM_LUMA_PARTITIONS]; /* primitives.h:223; C: ipfilter.cpp:162-201 */
    x265_hip_filter_sp_t   luma_vsp[X265_HIP_NUM_LUMA_PARTITIONS]; /* primitives.h:225; C: ipfilter.cpp:241-282 */
    /* the steps either side of the path */
    x265_hip_pixelavg_pp_t pixelavg_pp[X265_HIP_NUM_LUMA_PARTITIONS]; /* primitives.h:234; C: pixel.cpp:546-560 */
    x265_hip_weightp_pp_t  weight_pp;                                 /* primitives.h:232; C: pixel.cpp:510-529 */
    x265_hip_downscale_t   frame_init_lowres_core;                    /* primitives.h:257; C: pixel.cpp:595-618 */
    /* the distortions mode decision takes of a prediction */
    x265_hip_pixelcmp_t    sa8d_inter[X265_HIP_NUM_LUMA_PARTITIONS];  /* primitives.h:200; C: pixel.cpp:290-376, 1117-1141 */
    x265_hip_pixelcmp_t    sse_pp[X265_HIP_NUM_LUMA_PARTITIONS];      /* primitives.h:196; C: pixel.cpp:158-177, 1095 */
    /* the rest of luma motion compensation (TComPrediction.cpp:400-473) */
    x265_hip_filter_ps_t    luma_vps[X265_HIP_NUM_LUMA_PARTITIONS];    /* primitives.h:224; C: ipfilter.cpp:203-239 */
    x265_hip_filter_ss_t    luma_vss[X265_HIP_NUM_LUMA_PARTITIONS];    /* primitives.h:226; C: ipfilter.cpp:284-317 */
    x265_hip_filter_hv_pp_t luma_hvpp[X265_HIP_NUM_LUMA_PARTITIONS];   /* primitives.h:227; C: ipfilter.cpp:362-369 */
    x265_hip_filter_p2s_t   luma_p2s;                                  /* primitives.h:228; C: ipfilter.cpp:37-54, 510 */
    x265_hip_addAvg_t       luma_addAvg[X265_HIP_NUM_LUMA_PARTITIONS]; /* primitives.h:235; C: pixel.cpp:819-839 */
    /* the margins of the rows a frame thread has finished (MotionReference::applyWeight, reference.cpp:99; the frame
     * filter, framefilter.cpp:221-223) */
    x265_hip_extendCURowBorder_t extendRowBorder;                      /* primitives.h:259; C: ipfilter.cpp:56-74, 516 */
} x265_hip_primitives;

/* Replaces the assignments Setup_C_PixelPrimitives / Setup_C_IPFilterPrimitives make for these
 * entries (pixel.cpp:914-946, 986; ipfilter.cpp:399-408).  Creates the process-wide default context
 * on `device` on first use.  On failure the table is left untouched and a negative code returned
 * (the reference keeps its C path in that case, primitives.cpp:137-141). */
int x265_hip_setup_primitives(x265_hip_primitives *table, int device);

/* partitionFromSizes, source/common/primitives.h:103-112: -1 for an invalid size */
int x265_hip_partition_from_sizes(int width, int height);

/* Optional: tell the primitive thunks that [host_base, host_base+bytes) is a plane that stays
 * unchanged; fref/src pointers inside it are then served from a device copy made here instead of
 * being staged on every call.  Unregister before the host memory changes or is freed. */
int x265_hip_register_host_plane(const uint8_t *host_base, size_t bytes);
int x265_hip_unregister_host_plane(const uint8_t *host_base);

/* ---- (2) batched level ------------------------------------------------------------------------
 * The entry points that take host arrays validate every record before anything is launched: plane ids,
 * partition sizes, metric / method / QP, and that every sample a request can reach (block, displacement or
 * search window, the 8-tap filters' margins) lies inside the plane's buffer; X265_HIP_EINVAL / X265_HIP_ERANGE
 * name the first offending record.  The `_device` forms cannot look at the records: for them the same
 * conditions are the caller's contract (a record that reaches outside a buffer faults the GPU). */

/* A plane is a padded luma picture in the layout of the reference's TComPicYuv
 * (source/Lib/TLibCommon/TComPicYuv.cpp:78-97): `origin_offset` is the byte offset of picture
 * sample (0,0) inside the buffer (= marginY*stride + marginX), i.e. what ReferencePlanes::fpelPlane
 * points at (source/encoder/reference.cpp:39-47).  Margins must cover every sample a search can
 * touch (the reference guarantees 96 / 80). */
int x265_hip_plane_upload(x265_hip_ctx *ctx, int plane_id, const uint8_t *host_buf, size_t bytes, int stride,
                          int origin_offset);
/* adopt memory already in HBM (not freed by the library); must be readable 16 bytes past `bytes` */
int x265_hip_plane_bind(x265_hip_ctx *ctx, int plane_id, const void *device_buf, size_t bytes, int stride,
                        int origin_offset);
int x265_hip_plane_release(x265_hip_ctx *ctx, int plane_id);
/* copy a plane's buffer back to the host (synchronous; `bytes` at most the size it was created with) */
int x265_hip_plane_download(x265_hip_ctx *ctx, int plane_id, uint8_t *host_buf, size_t bytes);

/* ---- asynchronous transfers (BASELINE configs[4]: a long sequence streamed through a GPU) -------
 * The reference overlaps pictures on the host: a frame encoder starts on a CTU row once the rows it references have
 * been reconstructed (source/encoder/frameencoder.cpp:1039-1099).  The counterpart here is overlapping the arrival
 * of the next closed GOP segment's pictures (and the departure of results) with the searches of the current one:
 * a second set of planes, a copy stream inside the context, two fences.  No reference interface is replaced.
 *   x265_hip_plane_alloc         an owned plane of `bytes` with the given layout, contents undefined (synchronises the
 *                                context once; the directory entry is set, so later uploads change no metadata)
 *   x265_hip_plane_upload_async  host_buf (pinned memory: x265_hip_host_alloc) -> that plane, on the copy stream;
 *                                `bytes` must be the plane's size.  Returns at once.
 *   x265_hip_download_async      device memory (e.g. a result array) -> pinned host memory, on the copy stream
 *   x265_hip_copy_wait_compute   transfers queued from now on start after every kernel queued so far on the context
 *   x265_hip_compute_wait_copy   kernels queued from now on start after every transfer queued so far
 *   x265_hip_copy_sync           block until the copy stream is idle
 * Ordering is the caller's: a plane must not be overwritten while queued searches read it, nor searched before its
 * upload is fenced.  x265_amd/runner.py (SegmentStreamer) shows the double-buffered loop. */
int x265_hip_host_alloc(size_t bytes, void **out);
void x265_hip_host_free(void *p);
int x265_hip_plane_alloc(x265_hip_ctx *ctx, int plane_id, size_t bytes, int stride, int origin_offset);
int x265_hip_plane_upload_async(x265_hip_ctx *ctx, int plane_id, const uint8_t *host_buf, size_t bytes);
int x265_hip_download_async(x265_hip_ctx *ctx, const void *d_src, void *host_dst, size_t bytes);
int x265_hip_copy_wait_compute(x265_hip_ctx *ctx);
int x265_hip_compute_wait_copy(x265_hip_ctx *ctx);
int x265_hip_copy_sync(x265_hip_ctx *ctx);

/* ---- lookahead planes ---------------------------------------------------
 * Lowres::create + Lowres::init (source/common/lowres.cpp:30-69, 164-174): the half-resolution plane and
 * its three half-pel companions (frame_init_lowres_core, source/common/pixel.cpp:595-618), edge-extended
 * by marginX / marginY (extendPicBorder, pixel.cpp:893-911), made on the device from plane `src_plane`
 * (a full-resolution picture of `width` x `height`; with W = roundup8(width / 2), L = roundup8(height / 2) the
 * kernel reads source rows 0 .. 2 L and columns 0 .. 2 W + 3, i.e. 2 L - height + 1 rows below and 2 W - width + 4
 * columns right of the picture must lie inside the plane's buffer: X265_HIP_ERANGE otherwise).
 * Fills plane ids dst_plane0 .. dst_plane0 + 3 = lowresPlane[0..3] (owned by the context, released with
 * x265_hip_plane_release).  Asynchronous on the context's stream. */
typedef struct x265_hip_lowres_geometry
{
    int width, lines;  /* half resolution, rounded up to multiples of the 8x8 lookahead CU */
    int stride;        /* width / 2 + 2 marginX, rounded up to a multiple of 32 (lowres.cpp:36-38) */
    int origin_offset; /* stride * marginY + marginX */
    size_t bytes;      /* stride * (lines + 2 marginY) */
} x265_hip_lowres_geometry;
int x265_hip_lowres_geometry_of(int width, int height, int margin_x, int margin_y, x265_hip_lowres_geometry *out);
int x265_hip_lowres_init(x265_hip_ctx *ctx, int src_plane, int width, int height, int margin_x, int margin_y,
                         int dst_plane0);

/* ---- reference-plane production ---------------------------------------------
 * extendPicBorder (source/common/pixel.cpp:893-911) in place on a plane whose picture area (`width` x
 * `height` at the plane's origin) has been written, e.g. a reconstructed picture bound with
 * x265_hip_plane_bind: columns -marginX .. width + marginX - 1 of rows -marginY .. height + marginY - 1. */
int x265_hip_plane_extend_border(x265_hip_ctx *ctx, int plane_id, int width, int height, int margin_x, int margin_y);
/* MotionReference::init + applyWeight over a whole picture (source/encoder/reference.cpp:57-121): plane
 * `dst_plane` (same geometry as `src_plane`, owned by the context) receives weight_pp (pixel.cpp:510-529) of
 * the picture area with weight `w0`, `offset` and `log2_denom` as in wpScalingParam (the interpolation-depth
 * correction of reference.cpp:95-98 is applied inside), then the border extension. */
int x265_hip_plane_weight(x265_hip_ctx *ctx, int src_plane, int dst_plane, int width, int height, int margin_x,
                          int margin_y, int w0, int offset, int log2_denom);
/* The same two for a range of picture rows, as the reference produces reference planes while a picture is being
 * reconstructed: MotionReference::applyWeight(rows, numRows) weights and extends the CTU rows that have become
 * available (source/encoder/reference.cpp:78-121), the frame filter extends each finished row's left and right margins
 * (primitives.extendRowBorder, source/encoder/framefilter.cpp:221; C: extendCURowColBorder, source/common/
 * ipfilter.cpp:56-74).  Picture rows row0 .. row0 + rows - 1 are produced with their side margins; a range that
 * starts at row 0 also fills the rows above the picture, one that ends at the last row the rows below it.  Calling
 * them for consecutive ranges that cover the picture gives byte for byte what the whole-picture forms give. */
int x265_hip_plane_extend_border_rows(x265_hip_ctx *ctx, int plane_id, int width, int height, int margin_x, int margin_y, int row0,
                                      int rows);
int x265_hip_plane_weight_rows(x265_hip_ctx *ctx, int src_plane, int dst_plane, int width, int height, int margin_x, int margin_y,
                               int w0, int offset, int log2_denom, int row0, int rows);

/* One lambda-scaled MVD cost row as BitCost::setQP builds it (source/encoder/bitcost.cpp:31-56):
 * tab[32768 + d] = cost of difference d, d in [-32768, 32767].  The table is an input: it is made
 * with host floating point in the reference and is never recomputed on the GPU. */
int x265_hip_costtab_upload(x265_hip_ctx *ctx, int qp, const uint16_t *tab65536);
/* host-side restatement of BitCost::setQP for qp 0..69 (bitcost.cpp:31-80, x265_lambda_tab of
 * TComRom.cpp:127-143): build fills tab65536, set_qp builds and uploads */
int x265_hip_costtab_build(int qp, uint16_t *tab65536);
int x265_hip_costtab_set_qp(x265_hip_ctx *ctx, int qp);

/* one (block, reference, quarter-pel MV, metric) cost = MotionEstimate::subpelCompare
 * (motion.cpp:1126-1164) followed by sad/satd; no mvcost added */
typedef struct x265_hip_cost_req
{
    int32_t fenc_plane;
    int32_t ref_plane;
    int16_t pu_x, pu_y; /* block origin in picture samples */
    uint8_t pu_w, pu_h; /* one of the 25 partitions */
    uint8_t metric;     /* X265_HIP_SAD / X265_HIP_SATD */
    uint8_t pad;
    int16_t qmv_x, qmv_y;
} x265_hip_cost_req;

int x265_hip_cost_batch(x265_hip_ctx *ctx, const x265_hip_cost_req *reqs, int n, int32_t *costs);

/* Distortion of a uni-directional luma prediction: the same request, the same predicted
 * pixels (xPredInterLumaBlk, TComPrediction.cpp:400-433), compared with any of the four block comparisons;
 * X265_HIP_SA8D is primitives.sa8d_inter[part] (MotionEstimate::bufSA8D, encoder/motion.h:89, motion.cpp:116;
 * for square blocks the same functions as primitives.sa8d[sizeIdx], which inter mode decision applies to the
 * prediction, encoder/compress.cpp:215, 265; C: common/pixel.cpp:290-376, tables at 1111-1141); X265_HIP_SSE is
 * primitives.sse_pp[part] (pixel.cpp:158-177, 1095; TEncSearch.cpp:2651, 2780 on luma). */
int x265_hip_dist_batch(x265_hip_ctx *ctx, const x265_hip_cost_req *reqs, int n, int32_t *costs);
int x265_hip_dist_batch_device(x265_hip_ctx *ctx, const void *d_reqs, int n, void *d_costs);

/* Bidirectional candidate: the block against the rounded average of two uni-directional
 * luma predictions, as TEncSearch::predInterSearch measures it (source/Lib/TLibEncoder/TEncSearch.cpp:2360-2412:
 * xPredInterLumaBlk per list, TComPrediction.cpp:400-433 = the pixels of subpelCompare; pixelavg_pp,
 * common/pixel.cpp:546-560; then satd / sad of the block).  No rate term is added. */
typedef struct x265_hip_bidir_req
{
    int32_t fenc_plane;
    int32_t ref0_plane, ref1_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h; /* one of the 25 partitions */
    uint8_t metric;     /* X265_HIP_SAD / SATD / SA8D / SSE, optionally | X265_HIP_MC_AVG or | X265_HIP_LOWRES_REF (below) */
    uint8_t pad;
    int16_t qmv0_x, qmv0_y, qmv1_x, qmv1_y;
} x265_hip_bidir_req;
/* With X265_HIP_MC_AVG in `metric` the two predictions are combined the way TComPrediction::xPredInterBi does
 * (14-bit intermediates + luma_addAvg, TComPrediction.cpp:330-355, 436-473; pixel.cpp:819-839) instead of
 * pixelavg_pp: with X265_HIP_SATD that is what TEncSearch::xMergeEstimation measures on a bi-directional merge
 * candidate of a slice without weighted bi-prediction (motionCompensation + bufSATD, TEncSearch.cpp:2209-2210; every
 * form of a candidate: x265_hip_merge_estimate_batch), with X265_HIP_SA8D / X265_HIP_SSE what mode decision measures
 * on a bi-predicted block (compress.cpp:215, 265).
 * With X265_HIP_LOWRES_REF in `metric` the request is the lookahead's bidirectional cost (EstimateRow::
 * estimateCUCost, source/encoder/slicetype.cpp:1585-1603): ref0_plane .. +3 and ref1_plane .. +3 are the
 * lowresPlane[0..3] of two Lowres references, the block is an 8x8 CU of the lowres source picture, the two
 * predictions are ReferencePlanes::lowresMC (common/lowres.h:53-73) and the average is pixelavg_pp[LUMA_8x8]. */

int x265_hip_bidir_cost_batch(x265_hip_ctx *ctx, const x265_hip_bidir_req *reqs, int n, int32_t *costs);
int x265_hip_bidir_cost_batch_device(x265_hip_ctx *ctx, const void *d_reqs, int n, void *d_costs);

/* Motion-compensated luma prediction, TComPrediction::xPredInterUni / xPredInterBi for
 * luma (source/Lib/TLibCommon/TComPrediction.cpp:330-387).  Uni-directional: xPredInterLumaBlk to pixels
 * (:400-433: luma_copy_pp / luma_hpp / luma_vpp / luma_hps + luma_vsp).  Bi-directional: both lists to 14-bit
 * intermediates (:436-473: luma_p2s / luma_hps / luma_vps / luma_hps + luma_vss; ipfilter.cpp:38-54, 117-160,
 * 204-239, 285-317) and luma_addAvg (common/pixel.cpp:819-839).  Weighted uni-directional prediction: with
 * ref1_plane == X265_HIP_PRED_WEIGHTED the block
 * goes the way motionCompensation takes whenever the PPS enables weighted prediction (TComPrediction.cpp:265-279):
 * the 14-bit intermediate of ref0, then TComWeightPrediction::addWeightUni (TComWeightPrediction.cpp:433-458,
 * 678-697) = weight_sp (common/pixel.cpp:492-508) with w0 = qmv1_x (wpScalingParam::inputWeight), offset = qmv1_y
 * (inputOffset), shift = log2_denom + 6 and round = 1 << (shift - 1).
 * Weighted bi-prediction (the PPS's weighted_bipred_flag, --weightb; xPredInterBi's getWPBiPred branch,
 * TComPrediction.cpp:347-350): with weighted_bi != 0 and ref1_plane >= 0 the two 14-bit intermediates are combined by
 * TComWeightPrediction::xWeightedPredictionBi / addWeightBi (TComWeightPrediction.cpp:203-255, 609-618) with the
 * parameters getWpScaling derives for two lists (:534-547): weights wp_w0, wp_w1 (inputWeight of each list's
 * reference), offset = wp_o0 + wp_o1 (inputOffset), shift = log2_denom + 7 (the denominator of list 0's entry) and
 * round = 1 << (shift - 1): clip((w0 (P0 + 8192) + w1 (P1 + 8192) + round + (offset << (shift - 1))) >> shift).
 * (A block of such a slice that uses one list goes through addWeightUni, the X265_HIP_PRED_WEIGHTED form.)
 * pu_w x pu_h samples are written at (pu_x, pu_y) of dst_plane, which must not be one of the
 * request's references; requests of one batch whose destination blocks overlap race. */
#define X265_HIP_PRED_WEIGHTED (-2)
typedef struct x265_hip_pred_req
{
    int32_t dst_plane;
    int32_t ref0_plane, ref1_plane; /* ref1_plane < 0: uni-directional prediction from ref0 */
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h; /* one of the 25 partitions */
    uint8_t log2_denom;  /* the weighted forms: wpScalingParam::log2WeightDenom, 0..7 */
    uint8_t weighted_bi; /* != 0 with ref1_plane >= 0: weighted bi-prediction with wp_w0 .. wp_o1 */
    int16_t qmv0_x, qmv0_y, qmv1_x, qmv1_y;
    int16_t wp_w0, wp_o0; /* weighted_bi: inputWeight (-128..255) and inputOffset (-128..127) of list 0's reference */
    int16_t wp_w1, wp_o1; /* ... of list 1's */
} x265_hip_pred_req;

int x265_hip_predict_batch(x265_hip_ctx *ctx, const x265_hip_pred_req *reqs, int n);
int x265_hip_predict_batch_device(x265_hip_ctx *ctx, const void *d_reqs, int n);

/* one call of MotionEstimate::motionEstimate (motion.h:93; motion.cpp:526-1124), non-lowres branch */
typedef struct x265_hip_me_job
{
    int32_t fenc_plane; /* setSourcePlane (motion.h:76-80) */
    int32_t ref_plane;  /* ReferencePlanes::fpelPlane / lumaStride (lowres.h:36-50) */
    int16_t pu_x, pu_y; /* setSourcePU offset, as picture coordinates (motion.cpp:110-126) */
    uint8_t pu_w, pu_h;
    uint8_t method;     /* setSearchMethod: X265_HIP_DIA .. X265_HIP_FULL, optionally | X265_HIP_LOWRES_REF */
    uint8_t subme;      /* setSubpelRefine: 0..7, workload[] of motion.cpp:47-57 */
    int16_t mvmin_x, mvmin_y, mvmax_x, mvmax_y; /* full-pel, inclusive */
    int16_t qmvp_x, qmvp_y;                     /* quarter-pel predictor */
    int16_t merange;
    uint8_t qp;         /* which uploaded cost table (BitCost::setQP) */
    uint8_t num_mvc;
    int16_t mvc[2 * X265_HIP_MAX_MVC]; /* quarter-pel candidates, x,y pairs */
} x265_hip_me_job;

typedef struct x265_hip_me_result
{
    int32_t cost;     /* the int motionEstimate returns */
    int16_t mvx, mvy; /* outQMv, quarter-pel */
} x265_hip_me_result;

/* host arrays in, host arrays out (H2D + kernel + D2H, synchronous) */
int x265_hip_me_batch(x265_hip_ctx *ctx, const x265_hip_me_job *jobs, int n, x265_hip_me_result *results);
/* the same on device-resident job / result arrays, asynchronous on the context stream.  Jobs on blocks of 128 4x4
 * tiles and more (64x64, 64x32, 32x64, 64x48, 48x64) against a full-resolution reference are listed on the device and
 * searched by a kernel of their own beside the other jobs' kernel, on a second stream of the context that is forked
 * from and joined to the context stream inside the call (X265_HIP_ME_ONE_STREAM=1 in the environment at context
 * creation keeps both on the one stream); results are the same either way.  The list lives in a buffer of the
 * context that grows (one synchronisation + allocation) when a call has more jobs than any before it. */
int x265_hip_me_batch_device(x265_hip_ctx *ctx, const void *d_jobs, int n, void *d_results);
int x265_hip_cost_batch_device(x265_hip_ctx *ctx, const void *d_reqs, int n, void *d_costs);

/* The motion search of a whole lookahead frame cost estimate: what CostEstimate::estimateFrameCost
 * (source/encoder/slicetype.cpp:1242-1340) has its rows do for one (picture, reference) pair, i.e. for one list of
 * EstimateRow::estimateCUCost (:1512-1584).  Every 8x8 CU of the half-resolution picture is searched in reverse raster
 * order; its search bounds are those of :1540-1544, its candidates the vectors just found for the CUs right of it,
 * below it, below left and below right (:1556-1573), its predictor their median or the single candidate (:1575-1580),
 * and the search is motionEstimate(ref, mvmin, mvmax, mvp, numc, mvc, merange, outmv) (:1582) on the Lowres reference.
 * results[(j * height_in_cu + cuy) * width_in_cu + cux] is what the reference leaves in fenc->lowresMvCosts[list][dist]
 * [cuXY] and fenc->lowresMvs[list][dist][cuXY].  The bidirectional and intra costs of estimateCUCost are separate
 * requests (x265_hip_bidir_cost_batch with X265_HIP_LOWRES_REF; intra is out of scope).
 * One launch runs the searches of a batch side by side, one wave per CU row (the reference's processRow, :1458-1499):
 * a search is a chain of width_in_cu + 2 (height_in_cu - 1) dependent steps whatever the hardware, so throughput
 * comes from the batch (all the (picture, reference) pairs a slice-type decision needs), not from one pair. */
typedef struct x265_hip_lookahead_job
{
    int32_t fenc_plane; /* lowresPlane[0] of the picture being estimated (frames[b]) */
    int32_t ref_plane;  /* lowresPlane[0..3] of the reference, or of its weighted copy (wfref0): four consecutive ids */
    int16_t merange;    /* EstimateRow::m_merange: 16 (slicetype.h:76) */
    uint8_t method;     /* X265_HIP_HEX in the reference (slicetype.h:73) */
    uint8_t subme;      /* 1 in the reference (slicetype.h:74) */
    uint8_t qp;         /* X265_LOOKAHEAD_QP = 12: which uploaded cost table */
    uint8_t pad[3];
} x265_hip_lookahead_job;

int x265_hip_lookahead_batch(x265_hip_ctx *ctx, const x265_hip_lookahead_job *jobs, int njobs, int width_in_cu, int height_in_cu,
                             x265_hip_me_result *results);
/* device-resident jobs and results, asynchronous on the context stream; the records are the caller's contract.  A
 * result whose cost is INT32_MAX means the launch gave up waiting (never observed; the host-array form reports it). */
int x265_hip_lookahead_batch_device(x265_hip_ctx *ctx, const void *d_jobs, int njobs, int width_in_cu, int height_in_cu,
                                    void *d_results);
/* The status word of the context's last lookahead launch: 0, or nonzero if a row of it gave up waiting for the row
 * below it (its CUs then hold cost INT32_MAX).  Waits for the context's stream.  The host-array form checks it itself. */
int x265_hip_lookahead_status(x265_hip_ctx *ctx, int *status);

/* One ring of the sub-pel refinement of MotionEstimate::motionEstimate (motion.cpp:1047-1119) per request: the costs
 * subpelCompare(ref, centre + square1[i] * step, cmp) (motion.cpp:1126-1164; square1 at :71-73) for i = 1 .. 4 or 1 .. 8,
 * step 2 (half-pel ring) or 1 (quarter-pel ring), and optionally the centre itself (the SATD re-measure of
 * motion.cpp:1082-1086).  Same pixels and same sad / satd as nine x265_hip_cost_req, but the interpolation is shared:
 * the candidates are a 3 x 3 grid, so three horizontal passes serve all nine (x265_amd/csrc/refine.hip).  No rate
 * term is added.  costs[9 i + k] is candidate k of request i in square1's order (k = 0 the centre, 1 (0,-1), 2 (0,1),
 * 3 (-1,0), 4 (1,0), 5 (-1,-1), 6 (-1,1), 7 (1,-1), 8 (1,1)); entries the pattern does not ask for are INT32_MAX. */
#define X265_HIP_RING_QPEL   0x01 /* candidates at centre + square1[i]     */
#define X265_HIP_RING_HPEL   0x02 /* candidates at centre + square1[i] * 2 */
#define X265_HIP_RING_8      0x10 /* i = 1 .. 8 (otherwise 1 .. 4) */
#define X265_HIP_RING_CENTRE 0x20 /* also the centre */
#define X265_HIP_RING_COSTS  9
typedef struct x265_hip_refine_req
{
    int32_t fenc_plane;
    int32_t ref_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h; /* one of the 25 partitions */
    uint8_t metric;     /* X265_HIP_SAD / X265_HIP_SATD */
    uint8_t pattern;    /* X265_HIP_RING_QPEL or X265_HIP_RING_HPEL, | X265_HIP_RING_8, | X265_HIP_RING_CENTRE */
    int16_t qmv_x, qmv_y; /* the centre, quarter-pel */
} x265_hip_refine_req;

int x265_hip_refine_batch(x265_hip_ctx *ctx, const x265_hip_refine_req *reqs, int n, int32_t *costs9);
int x265_hip_refine_batch_device(x265_hip_ctx *ctx, const void *d_reqs, int n, void *d_costs9);
/* The same for a list that is made of up to X265_HIP_MAX_SECTIONS runs of requests of one PU size each (an encoder that
 * works depth by depth has exactly such lists: its 64x64 PUs, then the 32x32 ones ...): section s promises that its
 * `count` consecutive requests all have the size pu_w x pu_h.  One launch then serves all runs, each with batches cut
 * for its size, and the short batches of the small sizes fill the end of the launch (at 4K 7 % faster than one launch
 * per size, at 1080p 25 %).  The host-array form above finds such runs itself.  No counterpart in the reference. */
#define X265_HIP_MAX_SECTIONS 4
typedef struct x265_hip_refine_section
{
    int32_t count;
    uint8_t pu_w, pu_h;
    uint8_t pad[2];
} x265_hip_refine_section;
int x265_hip_refine_sections_device(x265_hip_ctx *ctx, const void *d_reqs, const x265_hip_refine_section *sections, int nsections,
                                    void *d_costs9);

/* x265_hip_cost_batch_device for callers that cannot allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc, code that must not grow buffers).  No counterpart in the reference.  Same requests, same costs;
 * the kernel's batch counters live in a static device array of the library (512 KiB: 64 slots of 8 KiB; one copy per
 * GPU, held until the process ends) instead of the context's buffer, so the call allocates and frees nothing.  At
 * most X265_HIP_NOALLOC_MAX_REQUESTS requests per call (X265_HIP_ERANGE beyond).  A context takes a slot of its GPU's
 * array at its first call here and keeps it until x265_hip_ctx_destroy: 64 contexts per GPU at most can use this
 * entry point at a time (X265_HIP_ERANGE for the 65th), and calls of different contexts may overlap on one GPU; calls
 * on one context are ordered on its stream like every other call.  The call is not meant for HIP graph capture as it
 * stands: when the context's plane / cost-table directory has changed since its last call it copies the directory
 * and waits for the stream first, and record-check mode synchronises too.
 * Asynchronous on the context's stream like the other `_device` forms. */
#define X265_HIP_NOALLOC_MAX_REQUESTS (1 << 24)
int x265_hip_cost_batch_device_noalloc(x265_hip_ctx *ctx, const void *d_reqs, int n, void *d_costs);

/* One iteration of the (list, reference) loop of TEncSearch::predInterSearch (source/Lib/TLibEncoder/TEncSearch.cpp:
 * 2306-2358) per job, on the device from the AMVP candidates to the motion vector:
 *   1. every AMVP candidate is clipped (TComDataCU::clipMv, TLibCommon/TComDataCU.cpp:2502-2514) and the block
 *      predicted at it from the reconstructed reference (xPredInterLumaBlk, TComPrediction.cpp:400-433) is compared
 *      with the source block by SAD (bufSAD); the first minimum is the predictor (TEncSearch.cpp:2315-2336: the rate
 *      term calcRdSADCost adds is the same for every candidate);
 *   2. the search window around it: TEncSearch::xSetSearchRange (:2600-2617) with the frame-parallel clamp ref_lag
 *      (m_refLagPixels, :110: the search range with frame threads, the picture height without);
 *   3. MotionEstimate::motionEstimate with that predictor, window and the job's candidate vectors (:2340).
 * What stays with the caller of this entry point is the control plane around it: TComDataCU::fillMvpCand (the
 * candidates come from neighbouring CUs' final vectors); the bit counting of xGetBlkBits / BitCost::bitcost /
 * xCheckBestMVP (:2343-2347, 2565-2598) and the choice between the lists are x265_hip_inter_decide_batch below. */
typedef struct x265_hip_inter_job
{
    int32_t fenc_plane;
    int32_t ref_plane;  /* m_mref[list][ref]: what motionEstimate searches (the weighted copy where there is one) */
    int32_t pred_plane; /* getRefPic(list, ref)->getPicYuvRec(): what the AMVP candidates are predicted from */
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h, method, subme;
    int16_t cu_x, cu_y; /* TComDataCU::getCUPelX / getCUPelY: clipMv is relative to the CU */
    int16_t amvp[4];    /* AMVPInfo::m_mvCand[0..1], quarter-pel, x,y pairs */
    int16_t merange;
    int16_t ref_lag;    /* m_refLagPixels */
    uint8_t num_amvp;   /* AMVPInfo::m_num: 1 or 2 */
    uint8_t qp;
    uint8_t num_mvc;
    uint8_t pad;
    int16_t mvc[2 * X265_HIP_MAX_MVC];
} x265_hip_inter_job;

typedef struct x265_hip_inter_result
{
    int32_t cost;       /* what motionEstimate returns */
    int16_t mvx, mvy;   /* outmv */
    int32_t amvp_sad[2]; /* bufSAD of the prediction at each clipped candidate (INT32_MAX for an absent one) */
    int16_t mvmin_x, mvmin_y, mvmax_x, mvmax_y; /* the window of step 2, full-pel */
    int16_t mvp_x, mvp_y; /* the predictor chosen in step 1 (as given, not clipped) */
    uint8_t mvp_idx;
    uint8_t pad[3];
} x265_hip_inter_result;

/* pic_width / pic_height: SPS picture size in luma samples (clipMv).  The `_device` form keeps an internal array of
 * n motionEstimate jobs in device memory, grown on demand (so a call with more jobs than any before allocates). */
int x265_hip_inter_search_batch(x265_hip_ctx *ctx, const x265_hip_inter_job *jobs, int n, int pic_width, int pic_height,
                                x265_hip_inter_result *results);
int x265_hip_inter_search_batch_device(x265_hip_ctx *ctx, const void *d_jobs, int n, int pic_width, int pic_height,
                                       void *d_results);

/* The rest of predInterSearch for a prediction unit, behind its iterations (reference
 * source/Lib/TLibEncoder/TEncSearch.cpp:2342-2358, 2360-2439, 2449-2499), on the device from the iterations' results:
 *   4. bits and cost of every iteration -- list selection (xGetBlkBits) + MVP_IDX_BITS + getTUBits(ref) +
 *      BitCost::bitcost(outmv), cost = (motionEstimate's cost - mvcost(outmv)) + RDCost::getCost(bits) -- with
 *      xCheckBestMVP's second look at the AMVP index (:2565-2598), and the first cheapest reference of each list;
 *   5. the bidirectional candidate (B slices without TComDataCU::isBipredRestriction, both lists searched): SATD of the
 *      block against pixelavg_pp of the two predictions, and its zero-vector form where both predictors pass the range
 *      check of :2400-2412, each half through xCheckBestMVP;
 *   6. the cheapest of list 0, list 1 and both (:2449-2499; the merge candidate's cost is the caller's to compare).
 * BitCost's float table of bit sizes (bitcost.cpp:68-80) is made inside the library the way the reference makes it
 * and lives in device memory beside the cost tables.  What stays with the caller: fillMvpCand, xGetBlkBits' three
 * numbers (they depend on the partition and on the previous unit's mode), lambda, and writing the outcome into the CU;
 * merge estimation is x265_hip_merge_estimate_batch below. */
typedef struct x265_hip_inter_pu
{
    int32_t first_job;        /* its iterations are jobs first_job .. of the job list: list 0's references 0 .., then list 1's */
    uint8_t num_ref[2];       /* numRefIdx[l] (at most 16 each, one at least in all); num_ref[1] = 0 in a P slice */
    uint8_t list_sel_bits[3]; /* xGetBlkBits (TEncSearch.cpp:2301, 2507-2563) */
    uint8_t no_bidir;         /* a P slice, or TComDataCU::isBipredRestriction */
    uint8_t pad[2];
    uint32_t lambda_sad;      /* RDCost::m_lambdaSAD (rdcost.h:50-54) */
} x265_hip_inter_pu;

typedef struct x265_hip_inter_side /* MotionData of predInterSearch */
{
    uint32_t cost; /* UINT32_MAX: none */
    int32_t bits;
    int16_t mvx, mvy, mvp_x, mvp_y;
    int8_t ref, mvp_idx;
    uint8_t pad[2];
} x265_hip_inter_side;

typedef struct x265_hip_inter_pu_result
{
    x265_hip_inter_side list[2];  /* list[l] of the reference: the best iteration of each list */
    x265_hip_inter_side bidir[2]; /* bidir[l]: vectors, predictors and AMVP indices of the bidirectional candidate */
    uint32_t bidir_cost;          /* UINT32_MAX: no such candidate */
    int32_t bidir_bits;
    uint8_t inter_dir;            /* 1 list 0, 2 list 1, 3 both */
    uint8_t zero_tried;           /* the zero-vector form was measured */
    uint8_t pending, pad;         /* (internal; 0 in results) */
} x265_hip_inter_pu_result;

/* Steps 1-3 for every job (x265_hip_inter_search_batch), then steps 4-6 for every unit, chained on the context's stream
 * without a host round trip.  job_results (njobs records) may be NULL in the host-array form.  The `_device` form keeps
 * internal arrays in device memory, grown on demand; any_bidir = 0 treats every unit as if no_bidir were set (a P picture)
 * and skips the bidirectional kernels. */
int x265_hip_inter_decide_batch(x265_hip_ctx *ctx, const x265_hip_inter_job *jobs, int njobs, const x265_hip_inter_pu *pus, int npu,
                                int pic_width, int pic_height, x265_hip_inter_result *job_results, x265_hip_inter_pu_result *results);
int x265_hip_inter_decide_batch_device(x265_hip_ctx *ctx, const void *d_jobs, int njobs, const void *d_pus, int npu, int pic_width,
                                       int pic_height, void *d_job_results, void *d_results, int any_bidir);

/* SATD of a source block against the prediction TComPrediction::motionCompensation makes for it, in every form of a
 * prediction request (uni- and bi-directional, weighted or not): the request's dst_plane names the SOURCE picture, nothing
 * is written to it, and costs[i] = satd[partition](source block, prediction) -- what TEncSearch::xMergeEstimation measures
 * on a merge candidate (source/Lib/TLibEncoder/TEncSearch.cpp:2209-2210: motionCompensation into m_predTempYuv, then
 * MotionEstimate::bufSATD, encoder/motion.h:79-82). */
int x265_hip_mc_satd_batch(x265_hip_ctx *ctx, const x265_hip_pred_req *reqs, int n, int32_t *costs);
int x265_hip_mc_satd_batch_device(x265_hip_ctx *ctx, const void *d_reqs, int n, void *d_costs);

/* TEncSearch::xMergeEstimation for a prediction unit (TEncSearch.cpp:2160-2226), behind
 * TComDataCU::getInterMergeCandidates (neighbour CU state: the caller's, like fillMvpCand) and the conversion of
 * bidirectional candidates under isBipredRestriction (:2181-2191, also the caller's: such a candidate is handed over
 * with ref_plane[1] = -1).  Per candidate in order: passed over if a vector of either list reaches mvy_limit (the frame
 * threads' guard, :2197-2202); otherwise both vectors through TComDataCU::clipMv, the luma prediction as
 * motionCompensation(cu, .., REF_PIC_LIST_X, ..) makes it for the slice (TComPrediction.cpp:252-297: a P slice predicts
 * from list 0, through addWeightUni where the PPS has weighted prediction; a B slice from both lists by luma_addAvg, from
 * list 0 alone where xCheckIdenticalMotion holds -- the same plane and vector on both lists --, through
 * xWeightedPredictionBi where the PPS has weighted bi-prediction, or from its one list), SATD against the source block,
 * + RDCost::getCost(getTUBits(candidate, num_cand)); the first cheapest wins (:2213-2218). */
#define X265_HIP_MRG_MAX_NUM_CANDS 5
typedef struct x265_hip_merge_cand /* mvFieldNeighbours[c][0..1] */
{
    int32_t ref_plane[2];     /* getRefPic(l, refIdx)->getPicYuvRec() of each list as a plane; < 0: refIdx NOT_VALID */
    int16_t mv[4];            /* list 0's vector (x, y), list 1's: quarter-pel, as getInterMergeCandidates gave them */
    int16_t wp_w[2], wp_o[2]; /* weighted slices: inputWeight / inputOffset of each list's reference */
    uint8_t log2_denom[2];    /* ... and its log2WeightDenom */
    uint8_t pad[2];
} x265_hip_merge_cand;

typedef struct x265_hip_merge_pu
{
    int32_t fenc_plane; /* the source picture */
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h;
    uint8_t num_cand;   /* MergeData::maxNumMergeCand, 1..5 */
    uint8_t weighted;   /* 0: no weighted prediction in this slice; 1: a P slice of a PPS with getUseWP; 2: a B slice of
                         * a PPS with getWPBiPred */
    int16_t cu_x, cu_y; /* getCUPelX / getCUPelY (clipMv) */
    int16_t mvy_limit;  /* (searchRange + 1) * 4 with frame threads > 1, else 32767 */
    uint16_t pad;
    uint32_t lambda_sad; /* RDCost::m_lambdaSAD */
    x265_hip_merge_cand cand[X265_HIP_MRG_MAX_NUM_CANDS];
} x265_hip_merge_pu;

typedef struct x265_hip_merge_result
{
    uint32_t cost;  /* what xMergeEstimation returns; UINT32_MAX: every candidate was passed over */
    uint32_t bits;  /* MergeData::bits */
    uint8_t index;  /* MergeData::index */
    uint8_t inter_dir; /* MergeData::interDir: 1 list 0, 2 list 1, 3 both (from the candidate's ref_plane[]) */
    uint8_t pad[2];
    int32_t satd[X265_HIP_MRG_MAX_NUM_CANDS]; /* the candidates' SATDs (-1: passed over or absent) */
} x265_hip_merge_result;

/* The `_device` form keeps an internal array of 5 n prediction requests and their costs in device memory, grown on demand. */
int x265_hip_merge_estimate_batch(x265_hip_ctx *ctx, const x265_hip_merge_pu *pus, int n, int pic_width, int pic_height,
                                  x265_hip_merge_result *results);
int x265_hip_merge_estimate_batch_device(x265_hip_ctx *ctx, const void *d_pus, int n, int pic_width, int pic_height, void *d_results);

/* Exhaustive integer-pel search of every square PU (64,32,16,8: 85 per CTU) of the listed 64x64
 * CTUs over [mvmin, mvmax], SAD + mvcost, first minimum in raster order -- the integer stage of
 * motionEstimate with X265_FULL_SEARCH (motion.cpp:997-1032) started from bcost = INT_MAX, with
 * 8x8 SADs shared by all PU sizes.  One predictor and window per CTU.  Results: 85 per CTU in the
 * order 64x64, 32x32 (raster), 16x16 (raster), 8x8 (raster); mv in quarter-pel units. */
typedef struct x265_hip_ctu_job
{
    int32_t fenc_plane;
    int32_t ref_plane;
    int16_t ctu_x, ctu_y;
    int16_t mvmin_x, mvmin_y, mvmax_x, mvmax_y; /* (mvmax-mvmin) <= 131 in x and y */
    int16_t qmvp_x, qmvp_y;
    uint8_t qp;
    uint8_t pad[3];
} x265_hip_ctu_job;

#define X265_HIP_PUS_PER_CTU 85

int x265_hip_fullsearch(x265_hip_ctx *ctx, const x265_hip_ctu_job *jobs, int n, x265_hip_me_result *results);
int x265_hip_fullsearch_device(x265_hip_ctx *ctx, const void *d_jobs, int n, void *d_results);

#ifdef __cplusplus
}
#endif
#endif /* X265_HIP_H */
