/* enc_capture.cpp -- TEST INFRASTRUCTURE ONLY (golden capture, route 3 of SURVEY.md 8(c)).
 *
 * Runs the REAL reference encoder (its own sources compiled where they lie, see oracle/Makefile target
 * `capture`) over a raw 8-bit 4:2:0 clip with the medium preset and records what the encoder itself passes to and
 * receives from MotionEstimate::motionEstimate: every call is intercepted at link time
 * (ld --wrap on the member function's symbol; no reference source is changed or copied), its arguments, the
 * search state of the MotionEstimate object and the result are logged, and the planes the call reads (source
 * picture, reconstructed / weighted reference, or the four Lowres planes of the lookahead) are snapshotted the
 * first time each buffer content is seen.  oracle/gen_golden.py turns the two output files into
 * tests/golden/encoder_me.npz, which the oracle and the HIP path must reproduce call for call.
 *
 *   enc_capture in.yuv width height frames out_dir [max_full_calls [max_lowres_calls [first_full_call
 *               [pool_threads [frame_threads]]]]]
 * With both call limits 0 nothing is recorded and the program is the reference encoder timed end to end.
 *
 * Built with -DWITH_HIP and linked to libx265_hip.so (target `dropin`, binary enc_hip) the same program is the
 * drop-in demonstration (tests/tools/dropin_demo.sh); the environment selects what the GPU answers, and in each
 * mode the answers are compared with the reference's own before the encoder goes on with them:
 *   X265_USE_HIP=1         the primitive table is installed through source/common/hip/hip-primitives.cpp
 *   X265_USE_HIP_ME=1      every MotionEstimate::motionEstimate call, the lookahead's included -> x265_hip_me_batch
 *   X265_USE_HIP_MC=1      the luma block of every TComPrediction::motionCompensation call -> x265_hip_predict_batch
 *   X265_USE_HIP_MERGE=1   every TEncSearch::xMergeEstimation call is made again by x265_hip_merge_estimate_batch and compared
 *   X265_USE_HIP_PLANES=1  Lowres::init -> x265_hip_lowres_init, MotionReference::applyWeight -> x265_hip_plane_weight
 *   X265_USE_HIP_PLANES=2  ... and every applyWeight call that adds CU rows -> x265_hip_plane_weight_rows for those rows
 *                          (what frame threads > 1 do while the reference picture is still being reconstructed)
 *   X265_USE_HIP_LOOKAHEAD=1|2  the lookahead's motion search a whole frame at a time: the first lowres motionEstimate call
 *                          of a frame cost estimate's list (CU (W-1, H-1), slicetype.cpp:1458-1499) runs
 *                          x265_hip_lookahead_batch for the (picture, reference) pair, and this and every later call of
 *                          the pair is answered from its results.  1: the reference's own search also runs and is
 *                          compared; 2: it does not run at all (timing).
 * All four member functions are intercepted with ld --wrap; the bitstream hash printed at the end must not change.
 */
#include <atomic>
#include <map>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

/* the interception needs the search state the object keeps (fencplane, blockOffset, searchMethod, ...) */
#define protected public
#define private public
#include "common.h"
#include "primitives.h"
#include "lowres.h"
#include "motion.h"
#include "bitcost.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComDataCU.h"
#include "TLibEncoder/TEncSearch.h"
#include "encoder.h"
#include "rdcost.h"
#ifdef WITH_HIP
#include "TLibCommon/TComPrediction.h"
#endif
#undef protected
#undef private
#include "x265.h"
#include <time.h>

namespace x265 {
void Setup_HIP_Primitives(EncoderPrimitives &p, x265_param *param, int device); /* source/common/hip/hip-primitives.cpp */
}
using namespace x265;

#define ME_SYM "_ZN4x26514MotionEstimate14motionEstimateEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_"
int real_me(MotionEstimate *self, ReferencePlanes *ref, const MV &mvmin, const MV &mvmax, const MV &qmvp, int numc,
            const MV *mvc, int merange, MV &out) asm("__real_" ME_SYM);
int wrap_me(MotionEstimate *self, ReferencePlanes *ref, const MV &mvmin, const MV &mvmax, const MV &qmvp, int numc,
            const MV *mvc, int merange, MV &out) asm("__wrap_" ME_SYM);

/* TComDataCU::fillMvpCand is what predInterSearch calls right in front of the AMVP pick and motionEstimate
 * (TEncSearch.cpp:2312): its candidates, the CU's position (what clipMv is relative to) and the reconstructed picture
 * the candidates are predicted from are noted per thread and written out beside the motionEstimate call that follows. */
#define FM_SYM "_ZN4x26510TComDataCU11fillMvpCandEjjiiPNS_8AMVPInfoEPNS_2MVE"
int real_fm(TComDataCU *self, uint32_t partIdx, uint32_t partAddr, int picList, int refIdx, AMVPInfo *info, MV *mvc) asm("__real_" FM_SYM);
int wrap_fm(TComDataCU *self, uint32_t partIdx, uint32_t partAddr, int picList, int refIdx, AMVPInfo *info, MV *mvc) asm("__wrap_" FM_SYM);

/* TEncSearch::predInterSearch itself (called from TEncCu.cpp / compress.cpp): what it left in the CU for every prediction
 * unit -- merge or not, direction, vectors, references, AMVP indices after xCheckBestMVP and the bidirectional stage
 * (TEncSearch.cpp:2343-2347, 2360-2499) -- is written to pu.bin beside the motionEstimate calls the unit made. */
#define PI_SYM "_ZN4x26510TEncSearch15predInterSearchEPNS_10TComDataCUEPNS_7TComYuvEbb"
bool real_pi(TEncSearch *self, TComDataCU *cu, TComYuv *predYuv, bool bMergeOnly, bool bChroma) asm("__real_" PI_SYM);
bool wrap_pi(TEncSearch *self, TComDataCU *cu, TComYuv *predYuv, bool bMergeOnly, bool bChroma) asm("__wrap_" PI_SYM);

/* TEncSearch::xMergeEstimation (TEncSearch.cpp:2160-2226) is defined and called in one translation unit, so the linker
 * cannot intercept it: wrap_pi runs it once more, on the CU as predInterSearch left it, for every prediction unit of a
 * partition other than 2Nx2N (merge_probe), and writes the candidates getInterMergeCandidates gave it, what it measured
 * them against and what it chose to merge.bin (with X265_USE_HIP_MERGE=1 in the drop-in build the same record is answered
 * by x265_hip_merge_estimate_batch and compared). */

#ifdef WITH_HIP
#define MC_SYM "_ZN4x26514TComPrediction18motionCompensationEPNS_10TComDataCUEPNS_7TComYuvEiibb"
void real_mc(TComPrediction *self, TComDataCU *cu, TComYuv *predYuv, int list, int partIdx, bool bLuma, bool bChroma) asm("__real_" MC_SYM);
void wrap_mc(TComPrediction *self, TComDataCU *cu, TComYuv *predYuv, int list, int partIdx, bool bLuma, bool bChroma) asm("__wrap_" MC_SYM);
#define LI_SYM "_ZN4x2656Lowres4initEPNS_10TComPicYuvEii"
void real_li(Lowres *self, TComPicYuv *orig, int poc, int type) asm("__real_" LI_SYM);
void wrap_li(Lowres *self, TComPicYuv *orig, int poc, int type) asm("__wrap_" LI_SYM);
#define AW_SYM "_ZN4x26515MotionReference11applyWeightEii"
void real_aw(MotionReference *self, int rows, int numRows) asm("__real_" AW_SYM);
void wrap_aw(MotionReference *self, int rows, int numRows) asm("__wrap_" AW_SYM);
#endif

namespace {

#pragma pack(push, 1)
struct CallRec /* 92 bytes, read by oracle/gen_golden.py */
{
    int32_t fenc_plane, ref_plane; /* snapshot ids; a Lowres reference is ref_plane .. ref_plane + 3 */
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h, method, subme;
    int16_t mvmin_x, mvmin_y, mvmax_x, mvmax_y, qmvp_x, qmvp_y;
    int16_t merange;
    uint8_t num_mvc, qp;
    int16_t mvc[24];
    uint8_t lowres, pad[3];
    int32_t cost;
    int16_t mvx, mvy;
};
struct PlaneRec /* header of one snapshot in planes.bin, followed by `bytes` bytes */
{
    int32_t id, stride, origin, bytes;
};
#pragma pack(pop)
static_assert(sizeof(CallRec) == 92, "layout read by gen_golden.py");
#pragma pack(push, 1)
struct AmvpRec /* amvp.bin: one per CallRec, in the same order (num_amvp 0 for the lookahead's calls) */
{
    int32_t pred_plane; /* snapshot id of getRefPic(list, ref)->getPicYuvRec() */
    int16_t cu_x, cu_y;
    int16_t amvp[4];
    uint8_t num_amvp, pad[3];
};
#pragma pack(pop)
static_assert(sizeof(AmvpRec) == 20, "layout read by gen_golden.py");
#pragma pack(push, 1)
struct PuRec /* pu.bin: one per prediction unit that searched (read by oracle/gen_golden.py) */
{
    int32_t call[8];     /* indices into calls.bin of its motionEstimate calls, in order: list 0's references, then list 1's */
    uint8_t call_list[8], call_ref[8];
    uint8_t n_calls, is_p, part_size, part_idx, num_ref[2], bipred_restriction, complete; /* complete: every call is in calls.bin */
    uint8_t list_sel_bits[3]; /* xGetBlkBits for this unit (TEncSearch.cpp:2301) */
    uint8_t merge_flag, inter_dir, pad[3]; /* as predInterSearch left them */
    uint32_t lambda_sad;      /* RDCost::m_lambdaSAD */
    int16_t mv[4];            /* list 0 x y, list 1 x y */
    int8_t ref[2], mvp_idx[2];
};
#pragma pack(pop)
static_assert(sizeof(PuRec) == 80, "layout read by gen_golden.py");
#pragma pack(push, 1)
struct MergeCandRec
{
    int32_t ref_plane[2]; /* snapshot ids of getRefPic(l, refIdx)->getPicYuvRec(), -1: refIdx NOT_VALID */
    int16_t mv[4];
    int16_t wp_w[2], wp_o[2];
    uint8_t log2_denom[2], pad[2];
};
struct MergeRec /* merge.bin: the layout of x265_hip_merge_pu, then what xMergeEstimation returned and left in MergeData */
{
    int32_t fenc_plane;
    int16_t pu_x, pu_y;
    uint8_t pu_w, pu_h, num_cand, weighted;
    int16_t cu_x, cu_y, mvy_limit;
    uint16_t pad;
    uint32_t lambda_sad;
    MergeCandRec cand[5];
    uint32_t out_cost, out_bits;
    uint8_t out_index, out_dir, cand_dir[5], pad2;
};
#pragma pack(pop)
static_assert(sizeof(MergeRec) == 164 + 16, "layout read by gen_golden.py");
std::vector<MergeRec> g_merges;
std::vector<PuRec> g_pus;
struct PuCall
{
    int index, list, ref, part;
};
thread_local bool t_pu_active = false;
thread_local std::vector<PuCall> t_pu_calls;

struct AmvpNote
{
    bool valid;
    int list, ref, part;
    int cu_x, cu_y, num;
    MV cand[2];
    const pixel *recon;
    int recon_stride;
};
thread_local AmvpNote t_amvp;
std::vector<AmvpRec> g_amvp;

struct Snap
{
    int id;
    uint32_t sum;
};
std::mutex g_mu;
std::map<const pixel *, std::vector<Snap> > g_snaps; /* buffer base -> versions seen */
std::vector<CallRec> g_calls;
FILE *g_planes = nullptr;
int g_next_plane = 0, g_width = 0, g_height = 0;
long g_max_full = 4000, g_max_low = 2000, g_first_full = 0, g_seen_full = 0, g_n_full = 0, g_n_low = 0;

/* TComPicYuv / Lowres layout: margins 80 rows, 96 columns (TComPicYuv.cpp:78-97, lowres.cpp:33-48) */
const int MX = 96, MY = 80;

bool g_hip_sparse = false;

uint32_t sparse_sum(const pixel *p, size_t n)
{
    uint32_t s = 2166136261u;
    const size_t step = g_hip_sparse ? (n / 2048) | 1 : 29; /* in-situ runs take a sum per call: about 2048 samples */
    for (size_t i = 0; i < n; i += step)
        s = (s ^ p[i]) * 16777619u;
    return s;
}

/* id of the snapshot of the padded buffer whose sample (0,0) is `origin`; rows = picture rows incl. padding */
int snapshot(const pixel *origin, int stride, int rows)
{
    const pixel *base = origin - (intptr_t)MY * stride - MX;
    const size_t bytes = (size_t)stride * (rows + 2 * MY);
    const uint32_t sum = sparse_sum(base, bytes);
    std::vector<Snap> &v = g_snaps[base];
    for (const Snap &s : v)
        if (s.sum == sum)
            return s.id;
    PlaneRec h = { g_next_plane, stride, MY * stride + MX, (int32_t)bytes };
    fwrite(&h, sizeof(h), 1, g_planes);
    fwrite(base, 1, bytes, g_planes);
    v.push_back(Snap{ g_next_plane, sum });
    return g_next_plane++;
}

/* four consecutive ids for lowresPlane[0..3] (they are snapshotted together the first time plane 0 is new) */
int snapshot_lowres(pixel *const planes[4], int stride, int rows)
{
    const pixel *base0 = planes[0] - (intptr_t)MY * stride - MX;
    const size_t bytes = (size_t)stride * (rows + 2 * MY);
    uint32_t sum = 0;
    for (int k = 0; k < 4; k++)
        sum = sum * 31 + sparse_sum(planes[k] - (intptr_t)MY * stride - MX, bytes);
    std::vector<Snap> &v = g_snaps[base0];
    for (const Snap &s : v)
        if (s.sum == sum)
            return s.id;
    const int id = g_next_plane;
    for (int k = 0; k < 4; k++)
    {
        PlaneRec h = { g_next_plane++, stride, MY * stride + MX, (int32_t)bytes };
        fwrite(&h, sizeof(h), 1, g_planes);
        fwrite(planes[k] - (intptr_t)MY * stride - MX, 1, bytes, g_planes);
    }
    v.push_back(Snap{ id, sum });
    return id;
}

int qp_of(const uint16_t *m_cost)
{
    for (int q = 0; q < BitCost::BC_MAX_QP; q++)
        if (BitCost::s_costs[q] && (m_cost == BitCost::s_costs[q] || m_cost == BitCost::s_costs[q] + BitCost::BC_MAX_MV))
            return q;
    return 255;
}

} // namespace

#ifdef WITH_HIP
/* X265_USE_HIP_ME=1: every motionEstimate call of the encoder, the lookahead's included, is answered by x265_hip_me_batch (one
 * job per call: slow, but it is the encoder's own control flow around the batched entry point).  Planes are kept
 * on the device in an LRU of 60 slots keyed by buffer and content. */
#include "x265_hip.h"
namespace {
bool g_hip_me = false;
x265_hip_ctx *g_ctx = nullptr;
long g_hip_calls = 0, g_hip_mismatch = 0, g_tick = 0;
struct Slot
{
    const pixel *base;
    uint32_t sum;
    long used;
};
Slot g_slots[60];
bool g_qp_set[2][82];

int device_plane(const pixel *origin, int stride, int rows)
{
    const pixel *base = origin - (intptr_t)MY * stride - MX;
    const size_t bytes = (size_t)stride * (rows + 2 * MY);
    const uint32_t sum = sparse_sum(base, bytes);
    int lru = 0;
    for (int i = 0; i < 60; i++)
    {
        if (g_slots[i].base == base && g_slots[i].sum == sum)
        {
            g_slots[i].used = ++g_tick;
            return i;
        }
        if (g_slots[i].used < g_slots[lru].used)
            lru = i;
    }
    if (x265_hip_plane_upload(g_ctx, lru, base, bytes, stride, MY * stride + MX) != X265_HIP_OK)
    {
        fprintf(stderr, "plane_upload: %s\n", x265_hip_last_error());
        abort();
    }
    g_slots[lru] = Slot{ base, sum, ++g_tick };
    return lru;
}

/* The lookahead's calls (ReferencePlanes::isLowres) go to a second context: slots 0..15 hold source planes
 * (lowresPlane[0] of the picture being estimated), slots 16..63 twelve groups of lowresPlane[0..3]. */
x265_hip_ctx *g_ctx_low = nullptr;
long g_hip_low_calls = 0, g_hip_low_mismatch = 0;
Slot g_low_src[16], g_low_ref[12];

void upload_low(int id, const pixel *origin, int stride, int lines)
{
    const pixel *base = origin - (intptr_t)MY * stride - MX;
    if (x265_hip_plane_upload(g_ctx_low, id, base, (size_t)stride * (lines + 2 * MY), stride, MY * stride + MX) != X265_HIP_OK)
    {
        fprintf(stderr, "plane_upload (lowres): %s\n", x265_hip_last_error());
        abort();
    }
}

int find_slot(Slot *slots, int n, const pixel *base, uint32_t sum, bool &hit)
{
    int lru = 0;
    for (int i = 0; i < n; i++)
    {
        if (slots[i].base == base && slots[i].sum == sum)
        {
            slots[i].used = ++g_tick;
            hit = true;
            return i;
        }
        if (slots[i].used < slots[lru].used)
            lru = i;
    }
    slots[lru] = Slot{ base, sum, ++g_tick };
    hit = false;
    return lru;
}

int device_low_src(const pixel *origin, int stride, int lines)
{
    const pixel *base = origin - (intptr_t)MY * stride - MX;
    bool hit;
    const int i = find_slot(g_low_src, 16, base, sparse_sum(base, (size_t)stride * (lines + 2 * MY)), hit);
    if (!hit)
        upload_low(i, origin, stride, lines);
    return i;
}

int device_low_ref(pixel *const planes[4], int stride, int lines)
{
    const size_t bytes = (size_t)stride * (lines + 2 * MY);
    uint32_t sum = 0;
    for (int k = 0; k < 4; k++)
        sum = sum * 31 + sparse_sum(planes[k] - (intptr_t)MY * stride - MX, bytes);
    bool hit;
    const int g = find_slot(g_low_ref, 12, planes[0] - (intptr_t)MY * stride - MX, sum, hit);
    if (!hit)
        for (int k = 0; k < 4; k++)
            upload_low(16 + 4 * g + k, planes[k], stride, lines);
    return 16 + 4 * g;
}
} // namespace
#endif

#ifdef WITH_HIP
/* X265_USE_HIP_LOOKAHEAD: whole-frame lookahead searches (see the head of this file) */
namespace {
int g_hip_look = 0; /* 1 / 2: one launch per (picture, reference) pair, with / without the reference's own search beside it;
                     * 3 / 4: the same with the pairs of a picture batched ahead of their use (see look_trigger) */
long g_look_frames = 0, g_look_launches = 0, g_look_mismatch = 0, g_look_hits = 0, g_look_unused = 0;
std::atomic<long> g_look_calls(0);
double g_look_seconds = 0;
struct LookEntry
{
    uint32_t fsum, rsum; /* content of the source plane and of the four reference planes the results were made from */
    bool valid, used;
    std::vector<x265_hip_me_result> res;
};
typedef std::pair<const void *, const void *> LookKey; /* (source lowresPlane[0], reference lowresPlane[0]) */
std::map<LookKey, LookEntry> g_look;
std::map<const pixel *, Lowres *> g_live; /* every Lowres that went through Lowres::init, by its lowresPlane[0] */

uint32_t low_sum(const pixel *origin, int stride, int lines)
{
    return sparse_sum(origin - (intptr_t)MY * stride - MX, (size_t)stride * (lines + 2 * MY));
}

uint32_t low_sum4(pixel *const planes[4], int stride, int lines)
{
    uint32_t sum = 0;
    for (int k = 0; k < 4; k++)
        sum = sum * 31 + low_sum(planes[k], stride, lines);
    return sum;
}

/* The first lowres motionEstimate call of a frame cost estimate's list (CU (W - 1, H - 1)).  Modes 1 / 2 search the
 * pair.  Modes 3 / 4 first look for results made ahead of time from the same plane contents; failing that they
 * search the pair together with every other pair (this picture, picture within bframes + 1 of it) the lookahead may
 * ask for next -- the search of a pair depends on nothing but the two pictures (lowresMvs[list][dist] is keyed by
 * the pair) -- in one launch, which takes about as long as the launch for one pair. */
LookEntry *look_trigger(MotionEstimate *self, ReferencePlanes *ref, int merange, int W, int H, int lines)
{
    const int stride = (int)self->fencLumaStride;
    const LookKey key((const void *)self->fencplane, (const void *)ref->lowresPlane[0]);
    const bool batched = g_hip_look >= 3;
    const uint32_t fsum = low_sum(self->fencplane, stride, lines), rsum = low_sum4(ref->lowresPlane, ref->lumaStride, lines);
    LookEntry &e = g_look[key];
    g_look_frames++;
    if (batched && e.valid && e.fsum == fsum && e.rsum == rsum)
    {
        g_look_hits++;
        e.used = true;
        return &e;
    }
    std::vector<x265_hip_lookahead_job> jobs;
    std::vector<LookEntry *> dst;
    x265_hip_lookahead_job j;
    memset(&j, 0, sizeof(j));
    j.merange = (int16_t)merange;
    j.method = (uint8_t)self->searchMethod;
    j.subme = (uint8_t)self->subpelRefine;
    j.qp = (uint8_t)qp_of(self->m_cost);
    if (!g_qp_set[1][j.qp])
    {
        x265_hip_costtab_set_qp(g_ctx_low, j.qp);
        g_qp_set[1][j.qp] = true;
    }
    j.fenc_plane = device_low_src(self->fencplane, stride, lines);
    j.ref_plane = device_low_ref(ref->lowresPlane, ref->lumaStride, lines);
    jobs.push_back(j);
    e.fsum = fsum, e.rsum = rsum, e.valid = true, e.used = true;
    dst.push_back(&e);
    std::map<const pixel *, Lowres *>::iterator me = g_live.find(self->fencplane);
    if (batched && me != g_live.end())
    {
        const int num = me->second->frameNum, reach = me->second->bframes + 1;
        for (std::map<const pixel *, Lowres *>::iterator it = g_live.begin(); it != g_live.end() && jobs.size() < 11; ++it)
        {
            Lowres *l = it->second;
            if (l == me->second || l->lowresPlane[0] == ref->lowresPlane[0] || l->frameNum < 0 || abs(l->frameNum - num) > reach)
                continue;
            LookEntry &o = g_look[LookKey((const void *)self->fencplane, (const void *)l->lowresPlane[0])];
            const uint32_t osum = low_sum4(l->lowresPlane, l->lumaStride, lines);
            if (o.valid && o.fsum == fsum && o.rsum == osum)
                continue;
            if (o.valid && !o.used)
                g_look_unused++;
            j.ref_plane = device_low_ref(l->lowresPlane, l->lumaStride, lines);
            jobs.push_back(j);
            o.fsum = fsum, o.rsum = osum, o.valid = true, o.used = false;
            dst.push_back(&o);
        }
    }
    std::vector<x265_hip_me_result> out((size_t)W * H * jobs.size());
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (x265_hip_lookahead_batch(g_ctx_low, jobs.data(), (int)jobs.size(), W, H, out.data()) != X265_HIP_OK)
    {
        fprintf(stderr, "lookahead_batch: %s\n", x265_hip_last_error());
        abort();
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    g_look_seconds += (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    g_look_launches++;
    for (size_t k = 0; k < jobs.size(); k++)
    {
        dst[k]->res.resize((size_t)W * H); /* sized once: readers keep pointers into it */
        memcpy(dst[k]->res.data(), out.data() + k * (size_t)W * H, (size_t)W * H * sizeof(x265_hip_me_result));
    }
    return &e;
}

int look_answer(MotionEstimate *self, ReferencePlanes *ref, const MV &mvmin, const MV &mvmax, const MV &qmvp, int numc, const MV *mvc,
                int merange, MV &out)
{
    const int stride = (int)self->fencLumaStride;
    /* Lookahead::Lookahead, slicetype.cpp:62-63; Lowres::create, lowres.cpp:33-48 */
    const int W = ((g_width / 2) + 7) >> 3, H = ((g_height / 2) + 7) >> 3, lines = ((g_height / 2) + 7) / 8 * 8;
    const int cux = (int)(self->blockOffset % stride) / 8, cuy = (int)(self->blockOffset / stride) / 8;
    const LookKey key((const void *)self->fencplane, (const void *)ref->lowresPlane[0]);
    /* A pair's results are looked up under the lock once per thread and search: the vector a pair's results live in
     * is never reallocated (its size is the picture's), and a pair is searched again only after the encoder has
     * finished the estimate that used the previous results. */
    static thread_local LookKey t_key;
    static thread_local const x265_hip_me_result *t_res = nullptr;
    if (cux == W - 1 && cuy == H - 1)
    {
        std::lock_guard<std::mutex> g(g_mu);
        t_res = look_trigger(self, ref, merange, W, H, lines)->res.data();
        t_key = key;
    }
    else if (!t_res || t_key != key)
    {
        std::lock_guard<std::mutex> g(g_mu);
        std::map<LookKey, LookEntry>::iterator it = g_look.find(key);
        if (it == g_look.end() || !it->second.valid)
        {
            fprintf(stderr, "enc_capture: lowres call for CU (%d, %d) before the frame's first CU\n", cux, cuy);
            abort();
        }
        t_key = key;
        t_res = it->second.res.data();
    }
    const x265_hip_me_result r = t_res[(size_t)cuy * W + cux];
    g_look_calls++;
    if (g_hip_look & 1)
    {
        MV own;
        const int cost = real_me(self, ref, mvmin, mvmax, qmvp, numc, mvc, merange, own);
        if (cost != r.cost || own.x != r.mvx || own.y != r.mvy)
        {
            std::lock_guard<std::mutex> g(g_mu);
            g_look_mismatch++;
        }
    }
    out.x = r.mvx;
    out.y = r.mvy;
    return r.cost;
}
} // namespace
#endif

#ifdef WITH_HIP
/* X265_USE_HIP_PLANES=1: the planes the encoder produces for the search are produced again on the device, compared
 * with the reference's and written over them: the four lookahead planes of every Lowres::init (lowres.cpp:129-175)
 * by x265_hip_lowres_init, and every weighted reference plane MotionReference::applyWeight completes
 * (reference.cpp:78-121) by x265_hip_plane_weight. */
namespace {
int g_hip_planes = 0; /* 1: whole planes when complete, 2: rows as applyWeight adds them */
x265_hip_ctx *g_ctx_planes = nullptr;
long g_li_planes = 0, g_li_mismatch = 0, g_aw_planes = 0, g_aw_mismatch = 0, g_aw_partial = 0, g_aw_batches = 0, g_aw_batch_mismatch = 0;
std::vector<uint8_t> g_pl_buf;

/* rows -MY .. h + MY - 1, columns -MX .. w + MX - 1 of a padded buffer: what the reference writes */
bool take_region(uint8_t *host_base, const uint8_t *gpu_base, int stride, int w, int h)
{
    bool diff = false;
    for (int y = 0; y < h + 2 * MY; y++)
    {
        diff |= memcmp(host_base + (size_t)y * stride, gpu_base + (size_t)y * stride, w + 2 * MX) != 0;
        memcpy(host_base + (size_t)y * stride, gpu_base + (size_t)y * stride, w + 2 * MX);
    }
    return diff;
}

void planes_fail(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, x265_hip_last_error());
    abort();
}
}

void wrap_li(Lowres *self, TComPicYuv *orig, int poc, int type)
{
    real_li(self, orig, poc, type);
    if (g_hip_look >= 3)
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_live[self->lowresPlane[0]] = self;
    }
    if (!g_hip_planes)
        return;
    std::lock_guard<std::mutex> g(g_mu);
    const int stride = orig->getStride(), rows = (g_height + 63) / 64 * 64;
    if (orig->getLumaMarginX() != MX || orig->getLumaMarginY() != MY)
        abort();
    x265_hip_lowres_geometry geo;
    if (x265_hip_plane_upload(g_ctx_planes, 0, orig->getLumaAddr() - (intptr_t)MY * stride - MX, (size_t)stride * (rows + 2 * MY), stride,
                              MY * stride + MX) != X265_HIP_OK ||
        x265_hip_lowres_geometry_of(g_width, g_height, MX, MY, &geo) != X265_HIP_OK ||
        x265_hip_lowres_init(g_ctx_planes, 0, g_width, g_height, MX, MY, 1) != X265_HIP_OK)
        planes_fail("lowres_init");
    if (geo.stride != self->lumaStride || geo.width != self->width || geo.lines != self->lines)
    {
        fprintf(stderr, "lowres geometry differs: %d %d %d vs %d %d %d\n", geo.stride, geo.width, geo.lines, (int)self->lumaStride,
                self->width, self->lines);
        abort();
    }
    g_pl_buf.resize(geo.bytes);
    for (int k = 0; k < 4; k++)
    {
        if (x265_hip_plane_download(g_ctx_planes, 1 + k, g_pl_buf.data(), geo.bytes) != X265_HIP_OK)
            planes_fail("plane_download");
        g_li_planes++;
        g_li_mismatch += take_region(self->buffer[k], g_pl_buf.data(), geo.stride, geo.width, geo.lines);
    }
}

void wrap_aw(MotionReference *self, int rows, int numRows)
{
    const int before = self->m_numWeightedRows;
    real_aw(self, rows, numRows);
    if (g_hip_planes == 2)
    {
        const int after = self->m_numWeightedRows;
        if (!self->isWeighted || after == before)
            return;
        std::lock_guard<std::mutex> g(g_mu);
        if (after == numRows && before < numRows - 1)
        {
            g_aw_partial++; /* (see below) */
            return;
        }
        TComPicYuv *pic = self->m_reconPic;
        const int stride = pic->getStride(), prow = (g_height + 63) / 64 * 64;
        const size_t bytes = (size_t)stride * (prow + 2 * MY);
        const int row0 = before * 64, nrows = after == numRows ? g_height - row0 : (after - before) * 64;
        /* (the whole buffer goes up, rows that are still being reconstructed included: only rows row0 .. are read) */
        if (x265_hip_plane_upload(g_ctx_planes, 8, pic->getLumaAddr() - (intptr_t)MY * stride - MX, bytes, stride, MY * stride + MX) != X265_HIP_OK ||
            x265_hip_plane_weight_rows(g_ctx_planes, 8, 9, g_width, g_height, MX, MY, self->weight, self->offset, self->shift, row0, nrows) != X265_HIP_OK)
            planes_fail("plane_weight_rows");
        g_pl_buf.resize(bytes);
        if (x265_hip_plane_download(g_ctx_planes, 9, g_pl_buf.data(), bytes) != X265_HIP_OK)
            planes_fail("plane_download");
        /* what this call of the reference wrote: the rows with their side margins, the rows above the picture with
         * the first batch, the rows below it with the last (reference.cpp:95-119) */
        const int y0 = before == 0 ? 0 : MY + row0, y1 = after == numRows ? g_height + 2 * MY : MY + row0 + nrows;
        bool diff = false;
        for (int y = y0; y < y1; y++)
        {
            diff |= memcmp(self->m_weightBuffer + (size_t)y * stride, g_pl_buf.data() + (size_t)y * stride, g_width + 2 * MX) != 0;
            memcpy(self->m_weightBuffer + (size_t)y * stride, g_pl_buf.data() + (size_t)y * stride, g_width + 2 * MX);
        }
        g_aw_batches++;
        g_aw_batch_mismatch += diff;
        return;
    }
    if (!g_hip_planes || !self->isWeighted || self->m_numWeightedRows != numRows || before == numRows)
        return; /* only the call that completes the plane */
    std::lock_guard<std::mutex> g(g_mu);
    if (before < numRows - 1)
    {
        /* reference.cpp:88-90 takes the height of the last CU row alone whenever rows == numRows: a call that
         * completes more than one row at once (pictures of two CU rows or fewer) leaves the rest of the reference's
         * buffer as allocated, so there is nothing to compare with */
        g_aw_partial++;
        return;
    }
    TComPicYuv *pic = self->m_reconPic;
    const int stride = pic->getStride(), prow = (g_height + 63) / 64 * 64;
    const size_t bytes = (size_t)stride * (prow + 2 * MY);
    if (x265_hip_plane_upload(g_ctx_planes, 8, pic->getLumaAddr() - (intptr_t)MY * stride - MX, bytes, stride, MY * stride + MX) != X265_HIP_OK ||
        x265_hip_plane_weight(g_ctx_planes, 8, 9, g_width, g_height, MX, MY, self->weight, self->offset, self->shift) != X265_HIP_OK)
        planes_fail("plane_weight");
    g_pl_buf.resize(bytes);
    if (x265_hip_plane_download(g_ctx_planes, 9, g_pl_buf.data(), bytes) != X265_HIP_OK)
        planes_fail("plane_download");
    g_aw_planes++;
    g_aw_mismatch += take_region(self->m_weightBuffer, g_pl_buf.data(), stride, g_width, g_height);
}

/* X265_USE_HIP_MC=1: the luma block of every TComPrediction::motionCompensation call (TComPrediction.cpp:252-297) is
 * predicted again by x265_hip_predict_batch, compared with what the reference wrote and then written over it, so
 * the encoder goes on with the GPU's pixels.  With X265_CAPTURE_WEIGHTB=1 (--weightb) the blocks of B slices go
 * through xWeightedPredictionBi (TComPrediction.cpp:347-368) and so through the weighted forms of the request. */
namespace {
bool g_hip_mc = false;
long g_mc_calls = 0, g_mc_weighted = 0, g_mc_bi = 0, g_mc_wbi = 0, g_mc_mismatch = 0;
const int MC_DST = 60; /* plane id of the destination picture, outside device_plane's slots */
std::vector<uint8_t> g_mc_buf;
}

void wrap_mc(TComPrediction *self, TComDataCU *cu, TComYuv *predYuv, int list, int partIdx, bool bLuma, bool bChroma)
{
    real_mc(self, cu, predYuv, list, partIdx, bLuma, bChroma);
    if (!g_hip_mc || !bLuma)
        return;
    std::lock_guard<std::mutex> g(g_mu);
    int w, h;
    uint32_t partAddr;
    cu->getPartIndexAndSize(partIdx, partAddr, w, h);
    TComSlice *slice = cu->getSlice();
    if (slice->isInterP())
        list = REF_PIC_LIST_0;
    int lists[2] = { -1, -1 };
    bool weighted = false, weighted_bi = false;
    if (list != REF_PIC_LIST_X)
    {
        lists[0] = list;
        weighted = slice->getPPS()->getUseWP();
    }
    else
    {
        const bool wpbi = slice->getPPS()->getWPBiPred();
        const int r0 = cu->getCUMvField(REF_PIC_LIST_0)->getRefIdx(partAddr), r1 = cu->getCUMvField(REF_PIC_LIST_1)->getRefIdx(partAddr);
        if (r0 >= 0 && r1 >= 0)
        {
            /* xCheckIdenticalMotion, TComPrediction.cpp:232-250: never identical in a slice with weighted_bipred */
            const bool same = !wpbi && slice->getRefPic(REF_PIC_LIST_0, r0)->getPOC() == slice->getRefPic(REF_PIC_LIST_1, r1)->getPOC() &&
                cu->getCUMvField(REF_PIC_LIST_0)->getMv(partAddr) == cu->getCUMvField(REF_PIC_LIST_1)->getMv(partAddr);
            lists[0] = 0;
            if (!same)
                lists[1] = 1;
            weighted_bi = wpbi; /* xPredInterBi :347-350 */
        }
        else
        {
            lists[0] = r0 >= 0 ? 0 : 1;
            weighted = wpbi; /* xPredInterBi :356-368: xWeightedPredictionBi with one list = addWeightUni */
        }
    }
    const int rows = (g_height + 63) / 64 * 64;
    x265_hip_pred_req q;
    memset(&q, 0, sizeof(q));
    q.dst_plane = MC_DST;
    q.ref1_plane = -1;
    q.pu_w = (uint8_t)w, q.pu_h = (uint8_t)h;
    int stride = 0;
    const pixel *base0 = nullptr;
    for (int k = 0; k < 2 && lists[k] >= 0; k++)
    {
        const int refIdx = cu->getCUMvField(lists[k])->getRefIdx(partAddr);
        MV mv = cu->getCUMvField(lists[k])->getMv(partAddr);
        cu->clipMv(mv);
        TComPicYuv *rp = slice->getRefPic(lists[k], refIdx)->getPicYuvRec();
        stride = rp->getStride();
        const intptr_t off = rp->getLumaAddr(cu->getAddr(), cu->getZorderIdxInCU() + partAddr) - rp->getLumaAddr();
        q.pu_x = (int16_t)(off % stride), q.pu_y = (int16_t)(off / stride);
        int id = device_plane(rp->getLumaAddr(), stride, rows);
        if (k == 0)
        {
            q.ref0_plane = id, q.qmv0_x = mv.x, q.qmv0_y = mv.y;
            base0 = rp->getLumaAddr() - (intptr_t)MY * stride - MX;
            if (weighted)
            {
                wpScalingParam *wp;
                slice->getWpScaling(lists[k], refIdx, wp);
                q.ref1_plane = X265_HIP_PRED_WEIGHTED;
                q.qmv1_x = (int16_t)wp[0].inputWeight, q.qmv1_y = (int16_t)wp[0].inputOffset;
                q.log2_denom = (uint8_t)wp[0].log2WeightDenom;
            }
        }
        else
        {
            if (g_slots[q.ref0_plane].base != base0) /* evicted by the second upload: cannot happen with 60 slots in use by a handful of pictures, but check */
                abort();
            q.ref1_plane = id, q.qmv1_x = mv.x, q.qmv1_y = mv.y;
            if (weighted_bi)
            {
                wpScalingParam *wp0, *wp1;
                slice->getWpScaling(REF_PIC_LIST_0, cu->getCUMvField(REF_PIC_LIST_0)->getRefIdx(partAddr), wp0);
                slice->getWpScaling(REF_PIC_LIST_1, cu->getCUMvField(REF_PIC_LIST_1)->getRefIdx(partAddr), wp1);
                q.weighted_bi = 1;
                q.wp_w0 = (int16_t)wp0[0].inputWeight, q.wp_o0 = (int16_t)wp0[0].inputOffset;
                q.wp_w1 = (int16_t)wp1[0].inputWeight, q.wp_o1 = (int16_t)wp1[0].inputOffset;
                q.log2_denom = (uint8_t)wp0[0].log2WeightDenom;
            }
        }
    }
    const size_t plane_bytes = (size_t)stride * (rows + 2 * MY);
    if (g_mc_buf.size() != plane_bytes)
    {
        g_mc_buf.assign(plane_bytes, 0);
        if (x265_hip_plane_upload(g_ctx, MC_DST, g_mc_buf.data(), plane_bytes, stride, MY * stride + MX) != X265_HIP_OK)
            abort();
    }
    const size_t upto = (size_t)MY * stride + MX + (size_t)(q.pu_y + h) * stride;
    if (x265_hip_predict_batch(g_ctx, &q, 1) != X265_HIP_OK || x265_hip_plane_download(g_ctx, MC_DST, g_mc_buf.data(), upto) != X265_HIP_OK)
    {
        fprintf(stderr, "predict_batch: %s\n", x265_hip_last_error());
        abort();
    }
    const uint8_t *gpu = g_mc_buf.data() + (size_t)MY * stride + MX + (size_t)q.pu_y * stride + q.pu_x;
    pixel *dst = predYuv->getLumaAddr(partAddr);
    const int dstride = predYuv->getStride();
    bool diff = false;
    for (int y = 0; y < h; y++)
    {
        diff |= memcmp(dst + y * dstride, gpu + (size_t)y * stride, w) != 0;
        memcpy(dst + y * dstride, gpu + (size_t)y * stride, w);
    }
    g_mc_calls++;
    g_mc_weighted += weighted;
    g_mc_bi += lists[1] >= 0;
    g_mc_wbi += weighted_bi;
    g_mc_mismatch += diff;
    if (g_mc_calls % 200000 == 0)
        fprintf(stderr, "enc_capture: %ld motionCompensation calls answered so far\n", g_mc_calls);
}
#endif

int wrap_fm(TComDataCU *self, uint32_t partIdx, uint32_t partAddr, int picList, int refIdx, AMVPInfo *info, MV *mvc)
{
    const int n = real_fm(self, partIdx, partAddr, picList, refIdx, info, mvc);
    AmvpNote &a = t_amvp;
    a.valid = true;
    a.list = picList, a.ref = refIdx, a.part = (int)partIdx;
    a.cu_x = (int)self->getCUPelX();
    a.cu_y = (int)self->getCUPelY();
    a.num = info->m_num < 2 ? info->m_num : 2;
    a.cand[0] = info->m_mvCand[0];
    a.cand[1] = info->m_mvCand[1];
    TComPicYuv *rec = self->getSlice()->getRefPic(picList, refIdx)->getPicYuvRec();
    a.recon = rec->getLumaAddr();
    a.recon_stride = rec->getStride();
    return n;
}

int wrap_me(MotionEstimate *self, ReferencePlanes *ref, const MV &mvmin, const MV &mvmax, const MV &qmvp, int numc,
            const MV *mvc, int merange, MV &out)
{
#ifdef WITH_HIP
    if (g_hip_look && ref->isLowres)
        return look_answer(self, ref, mvmin, mvmax, qmvp, numc, mvc, merange, out);
#endif
    int cost = real_me(self, ref, mvmin, mvmax, qmvp, numc, mvc, merange, out);
    std::lock_guard<std::mutex> g(g_mu);
#ifdef WITH_HIP
    if (g_hip_me && numc <= X265_HIP_MAX_MVC)
    {
        const bool lowref = ref->isLowres;
        const int stride = (int)self->fencLumaStride, rows = (g_height + 63) / 64 * 64;
        x265_hip_ctx *ctx = lowref ? g_ctx_low : g_ctx;
        x265_hip_me_job j;
        memset(&j, 0, sizeof(j));
        if (lowref)
        {
            const int lines = ((g_height / 2) + 7) / 8 * 8; /* lowres.cpp:33-48 */
            j.fenc_plane = device_low_src(self->fencplane, stride, lines);
            j.ref_plane = device_low_ref(ref->lowresPlane, ref->lumaStride, lines);
        }
        else
        {
            j.fenc_plane = device_plane(self->fencplane, stride, rows);
            j.ref_plane = device_plane(ref->fpelPlane, ref->lumaStride, rows);
            if (g_slots[j.fenc_plane].base != self->fencplane - (intptr_t)MY * stride - MX) /* evicted by the second upload */
                j.fenc_plane = device_plane(self->fencplane, stride, rows);
        }
        j.pu_x = (int16_t)(self->blockOffset % stride);
        j.pu_y = (int16_t)(self->blockOffset / stride);
        j.pu_w = (uint8_t)self->blockwidth;
        j.pu_h = (uint8_t)self->blockheight;
        j.method = (uint8_t)(self->searchMethod | (lowref ? X265_HIP_LOWRES_REF : 0));
        j.subme = (uint8_t)self->subpelRefine;
        j.mvmin_x = mvmin.x, j.mvmin_y = mvmin.y, j.mvmax_x = mvmax.x, j.mvmax_y = mvmax.y;
        j.qmvp_x = qmvp.x, j.qmvp_y = qmvp.y;
        j.merange = (int16_t)merange;
        j.num_mvc = (uint8_t)numc;
        for (int i = 0; i < numc; i++)
            j.mvc[2 * i] = mvc[i].x, j.mvc[2 * i + 1] = mvc[i].y;
        j.qp = (uint8_t)qp_of(self->m_cost);
        if (!g_qp_set[lowref][j.qp])
        {
            x265_hip_costtab_set_qp(ctx, j.qp);
            g_qp_set[lowref][j.qp] = true;
        }
        x265_hip_me_result r;
        if (x265_hip_me_batch(ctx, &j, 1, &r) != X265_HIP_OK)
        {
            fprintf(stderr, "me_batch: %s\n", x265_hip_last_error());
            abort();
        }
        (lowref ? g_hip_low_calls : g_hip_calls)++;
        if ((g_hip_low_calls + g_hip_calls) % 200000 == 0)
            fprintf(stderr, "enc_capture: %ld + %ld calls answered so far\n", g_hip_calls, g_hip_low_calls);
        if (r.cost != cost || r.mvx != out.x || r.mvy != out.y)
            (lowref ? g_hip_low_mismatch : g_hip_mismatch)++;
        /* the encoder goes on with the GPU's answer */
        cost = r.cost;
        out.x = r.mvx;
        out.y = r.mvy;
    }
#endif
    const bool low = ref->isLowres;
    if (!low && g_seen_full++ < g_first_full)
    {
        if (t_pu_active && t_amvp.valid)
            t_pu_calls.push_back(PuCall{ -1, t_amvp.list, t_amvp.ref, t_amvp.part });
        return cost;
    }
    if ((low ? g_n_low >= g_max_low : g_n_full >= g_max_full) || numc > 12)
    {
        if (!low && t_pu_active && t_amvp.valid)
            t_pu_calls.push_back(PuCall{ -1, t_amvp.list, t_amvp.ref, t_amvp.part });
        return cost;
    }
    CallRec r;
    memset(&r, 0, sizeof(r));
    const int stride = (int)self->fencLumaStride;
    if (low)
    {
        /* Lowres::create: width / 2 and height / 2, lines rounded up to 8 (lowres.cpp:33-48) */
        const int lines = ((g_height / 2) + 7) / 8 * 8;
        r.fenc_plane = snapshot(self->fencplane, stride, lines);
        r.ref_plane = snapshot_lowres(ref->lowresPlane, ref->lumaStride, lines);
        g_n_low++;
    }
    else
    {
        const int rows = (g_height + 63) / 64 * 64;
        r.fenc_plane = snapshot(self->fencplane, stride, rows);
        r.ref_plane = snapshot(ref->fpelPlane, ref->lumaStride, rows);
        g_n_full++;
    }
    r.pu_y = (int16_t)(self->blockOffset / stride);
    r.pu_x = (int16_t)(self->blockOffset % stride);
    r.pu_w = (uint8_t)self->blockwidth;
    r.pu_h = (uint8_t)self->blockheight;
    r.method = (uint8_t)self->searchMethod;
    r.subme = (uint8_t)self->subpelRefine;
    r.mvmin_x = mvmin.x, r.mvmin_y = mvmin.y, r.mvmax_x = mvmax.x, r.mvmax_y = mvmax.y;
    r.qmvp_x = qmvp.x, r.qmvp_y = qmvp.y;
    r.merange = (int16_t)merange;
    r.num_mvc = (uint8_t)numc;
    for (int i = 0; i < numc; i++)
        r.mvc[2 * i] = mvc[i].x, r.mvc[2 * i + 1] = mvc[i].y;
    r.qp = (uint8_t)qp_of(self->m_cost);
    r.lowres = low;
    r.cost = cost;
    r.mvx = out.x, r.mvy = out.y;
    g_calls.push_back(r);
    if (!low && t_pu_active && t_amvp.valid)
        t_pu_calls.push_back(PuCall{ (int)g_calls.size() - 1, t_amvp.list, t_amvp.ref, t_amvp.part });
    AmvpRec ar;
    memset(&ar, 0, sizeof(ar));
    if (!low && t_amvp.valid)
    {
        const AmvpNote &a = t_amvp;
        ar.pred_plane = snapshot(a.recon, a.recon_stride, (g_height + 63) / 64 * 64);
        ar.cu_x = (int16_t)a.cu_x, ar.cu_y = (int16_t)a.cu_y;
        ar.num_amvp = (uint8_t)a.num;
        for (int i = 0; i < 2; i++)
            ar.amvp[2 * i] = a.cand[i].x, ar.amvp[2 * i + 1] = a.cand[i].y;
    }
    g_amvp.push_back(ar);
    return cost;
}

#ifdef WITH_HIP
namespace {
bool g_hip_merge = false;
long g_mg_calls = 0, g_mg_mismatch = 0, g_mg_bi = 0, g_mg_weighted = 0;
}
#endif

static void merge_probe(TEncSearch *self, TComDataCU *cu, int puIdx)
{
    bool record = false;
#ifdef WITH_HIP
    const bool answer = g_hip_merge;
#else
    const bool answer = false;
#endif
    {
        std::lock_guard<std::mutex> g(g_mu);
        /* recorded while the full-resolution motionEstimate calls are (the same pictures, so the same planes) */
        if (!answer && g_seen_full >= g_first_full && g_n_full > 0 && g_n_full < g_max_full)
            record = true;
    }
    if (!record && !answer)
        return;
    /* the call as predInterSearch makes it (TEncSearch.cpp:2256-2269); it leaves each candidate's motion in the CU's
     * fields at the unit's first entry (:2204-2207), which is put back */
    uint32_t partAddr;
    int w, h;
    cu->getPartIndexAndSize(puIdx, partAddr, w, h);
    MV keep_mv[2];
    char keep_ref[2];
    for (int l = 0; l < 2; l++)
    {
        keep_mv[l] = cu->getCUMvField(l)->m_mv[partAddr];
        keep_ref[l] = cu->getCUMvField(l)->m_refIdx[partAddr];
    }
    {
        TComPicYuv *src = cu->getSlice()->getPic()->getPicYuvOrg();
        self->m_me.setSourcePU((int)(src->getLumaAddr(cu->getAddr(), cu->getZorderIdxInCU() + partAddr) - src->getLumaAddr()), w, h);
    }
    MergeData m;
    m.absPartIdx = partAddr;
    m.width = w;
    m.height = h;
    const uint32_t cost = self->xMergeEstimation(cu, puIdx, m);
    for (int l = 0; l < 2; l++)
    {
        cu->getCUMvField(l)->m_mv[partAddr] = keep_mv[l];
        cu->getCUMvField(l)->m_refIdx[partAddr] = keep_ref[l];
    }
    std::lock_guard<std::mutex> g(g_mu);
    TComSlice *slice = cu->getSlice();
    TComPicYuv *fenc = slice->getPic()->getPicYuvOrg();
    const int stride = fenc->getStride(), rows = (g_height + 63) / 64 * 64;
    const intptr_t off = fenc->getLumaAddr(cu->getAddr(), cu->getZorderIdxInCU() + m.absPartIdx) - fenc->getLumaAddr();
    MergeRec r;
    memset(&r, 0, sizeof(r));
    auto plane_of = [&](const pixel *origin, int st) {
#ifdef WITH_HIP
        if (answer)
            return device_plane(origin, st, rows);
#endif
        return snapshot(origin, st, rows);
    };
    r.fenc_plane = plane_of(fenc->getLumaAddr(), stride);
    r.pu_x = (int16_t)(off % stride), r.pu_y = (int16_t)(off / stride);
    r.pu_w = (uint8_t)m.width, r.pu_h = (uint8_t)m.height;
    r.num_cand = (uint8_t)m.maxNumMergeCand;
    r.weighted = slice->isInterP() ? (slice->getPPS()->getUseWP() ? 1 : 0) : (slice->getPPS()->getWPBiPred() ? 2 : 0);
    r.cu_x = (int16_t)cu->getCUPelX(), r.cu_y = (int16_t)cu->getCUPelY();
    r.mvy_limit = self->m_cfg->m_totalFrameThreads > 1 ? (int16_t)((self->m_cfg->m_param->searchRange + 1) * 4) : 32767;
    r.lambda_sad = (uint32_t)self->m_rdCost->m_lambdaSAD;
    for (uint32_t c = 0; c < m.maxNumMergeCand && c < 5; c++)
    {
        r.cand_dir[c] = m.interDirNeighbours[c];
        for (int l = 0; l < 2; l++)
        {
            const TComMvField &f = m.mvFieldNeighbours[c][l];
            r.cand[c].mv[2 * l] = f.mv.x, r.cand[c].mv[2 * l + 1] = f.mv.y;
            r.cand[c].ref_plane[l] = -1;
            if (f.refIdx >= 0 && !(l == 1 && slice->isInterP()))
            {
                TComPicYuv *rp = slice->getRefPic(l, f.refIdx)->getPicYuvRec();
                r.cand[c].ref_plane[l] = plane_of(rp->getLumaAddr(), rp->getStride());
                wpScalingParam *wp;
                slice->getWpScaling(l, f.refIdx, wp);
                r.cand[c].wp_w[l] = (int16_t)wp[0].inputWeight, r.cand[c].wp_o[l] = (int16_t)wp[0].inputOffset;
                r.cand[c].log2_denom[l] = (uint8_t)wp[0].log2WeightDenom;
            }
        }
    }
    r.out_cost = cost, r.out_bits = m.bits;
    r.out_index = (uint8_t)m.index, r.out_dir = (uint8_t)m.interDir;
#ifdef WITH_HIP
    if (answer)
    {
        static_assert(sizeof(x265_hip_merge_pu) == 164, "MergeRec starts with an x265_hip_merge_pu");
        x265_hip_merge_pu pu;
        memcpy(&pu, &r, sizeof(pu));
        x265_hip_merge_result out;
        if (x265_hip_merge_estimate_batch(g_ctx, &pu, 1, g_width, g_height, &out) != X265_HIP_OK)
        {
            fprintf(stderr, "merge_estimate_batch: %s\n", x265_hip_last_error());
            abort();
        }
        g_mg_calls++;
        g_mg_weighted += r.weighted != 0;
        if (cost != 0xFFFFFFFFu)
            g_mg_bi += m.interDir == 3;
        if (out.cost != cost || (cost != 0xFFFFFFFFu && (out.bits != m.bits || out.index != m.index || out.inter_dir != m.interDir)))
            g_mg_mismatch++;
        return;
    }
#endif
    g_merges.push_back(r);
}

bool wrap_pi(TEncSearch *self, TComDataCU *cu, TComYuv *predYuv, bool bMergeOnly, bool bChroma)
{
    t_pu_active = true;
    t_pu_calls.clear();
    const bool ok = real_pi(self, cu, predYuv, bMergeOnly, bChroma);
    t_pu_active = false;
    if (cu->getPartitionSize(0) != SIZE_2Nx2N)
        for (int part = 0; part < cu->getNumPartInter(); part++)
            merge_probe(self, cu, part);
    if (t_pu_calls.empty())
        return ok;
    const PartSize partSize = cu->getPartitionSize(0);
    const int numPart = cu->getNumPartInter();
    const bool isP = cu->getSlice()->isInterP();
    const int *numRefIdx = cu->getSlice()->getNumRefIdx();
    uint32_t lastMode = 0; /* TEncSearch.cpp:2242; set where a unit ends up uni- or bi-directional (:2449-2499) */
    std::lock_guard<std::mutex> g(g_mu);
    for (int part = 0; part < numPart; part++)
    {
        uint32_t partAddr;
        int w, h;
        cu->getPartIndexAndSize(part, partAddr, w, h);
        const bool merged = cu->getMergeFlag(partAddr);
        const int dir = cu->getInterDir(partAddr);
        PuRec p;
        memset(&p, 0, sizeof(p));
        p.complete = 1;
        for (const PuCall &c : t_pu_calls)
            if (c.part == part && p.n_calls < 8)
            {
                p.call[p.n_calls] = c.index;
                p.call_list[p.n_calls] = (uint8_t)c.list;
                p.call_ref[p.n_calls] = (uint8_t)c.ref;
                p.n_calls++;
                if (c.index < 0)
                    p.complete = 0;
            }
        if (p.n_calls)
        {
            uint32_t bits[3];
            self->xGetBlkBits(partSize, isP, part, lastMode, bits);
            p.is_p = isP;
            p.part_size = (uint8_t)partSize;
            p.part_idx = (uint8_t)part;
            p.num_ref[0] = (uint8_t)numRefIdx[0];
            p.num_ref[1] = (uint8_t)(isP ? 0 : numRefIdx[1]);
            p.bipred_restriction = cu->isBipredRestriction();
            for (int k = 0; k < 3; k++)
                p.list_sel_bits[k] = (uint8_t)bits[k];
            p.merge_flag = merged;
            p.inter_dir = (uint8_t)dir;
            p.lambda_sad = (uint32_t)self->m_rdCost->m_lambdaSAD;
            for (int l = 0; l < 2; l++)
            {
                const MV mv = cu->getCUMvField(l)->getMv(partAddr);
                p.mv[2 * l] = mv.x, p.mv[2 * l + 1] = mv.y;
                p.ref[l] = (int8_t)cu->getCUMvField(l)->getRefIdx(partAddr);
                p.mvp_idx[l] = (int8_t)cu->getMVPIdx(l, partAddr);
            }
            g_pus.push_back(p);
        }
        if (!merged && p.n_calls)
            lastMode = dir == 3 ? 2 : dir == 2 ? 1 : 0;
    }
    return ok;
}

/* --wpbi in.bin out.bin: TComWeightPrediction::xWeightedPredictionBi (TComWeightPrediction.cpp:609-618) of the reference on
 * given intermediates, for tests/golden/wpbi.npz.  in.bin: int32 n, then per case seven int32 (w, h, inputWeight and
 * inputOffset of list 0's entry, of list 1's, log2WeightDenom) and two 64x64 int16 blocks (the ShortYuv of each list);
 * out.bin: per case the 64x64 uint8 block the reference wrote (w x h of it).  The slice's weight table is set the way
 * weightAnalyse's result is stored (TComSlice::setWpScaling) and read back by the reference's own getWpScaling. */
static int wpbi_fixture(const char *in_path, const char *out_path)
{
    FILE *in = fopen(in_path, "rb"), *out = fopen(out_path, "wb");
    int32_t n = 0;
    if (!in || !out || fread(&n, 4, 1, in) != 1)
        return 1;
    TComPPS pps;
    pps.setWPBiPred(true);
    TComSlice slice;
    slice.setPPS(&pps);
    slice.setSliceType(B_SLICE);
    TComDataCU cu;
    cu.m_slice = &slice;
    ShortYuv s0, s1;
    TComYuv dst;
    if (!s0.create(64, 64, X265_CSP_I420) || !s1.create(64, 64, X265_CSP_I420) || !dst.create(64, 64, X265_CSP_I420))
        return 1;
    TComWeightPrediction wp;
    for (int i = 0; i < n; i++)
    {
        int32_t hd[7];
        if (fread(hd, 4, 7, in) != 7 || fread(s0.getLumaAddr(), 2, 64 * 64, in) != 64 * 64 || fread(s1.getLumaAddr(), 2, 64 * 64, in) != 64 * 64)
            return 1;
        if (s0.m_width != 64 || s1.m_width != 64 || dst.getStride() != 64)
            return 1;
        wpScalingParam tab[2][MAX_NUM_REF][3];
        memset(tab, 0, sizeof(tab));
        for (int l = 0; l < 2; l++)
            for (int c = 0; c < 3; c++)
            {
                tab[l][0][c].bPresentFlag = true;
                tab[l][0][c].log2WeightDenom = (uint32_t)hd[6];
                tab[l][0][c].inputWeight = hd[2 + 2 * l];
                tab[l][0][c].inputOffset = hd[3 + 2 * l];
            }
        slice.setWpScaling(tab);
        memset(dst.getLumaAddr(), 0, 64 * 64);
        wp.xWeightedPredictionBi(&cu, &s0, &s1, 0, 0, 0, hd[0], hd[1], &dst, true, false);
        fwrite(dst.getLumaAddr(), 1, 64 * 64, out);
    }
    fclose(in);
    fclose(out);
    s0.destroy();
    s1.destroy();
    dst.destroy();
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 4 && !strcmp(argv[1], "--wpbi"))
        return wpbi_fixture(argv[2], argv[3]);
    if (argc < 6)
    {
        fprintf(stderr, "usage: %s in.yuv width height frames out_dir [max_full [max_lowres [first_full [pool_threads [frame_threads]]]]]\n", argv[0]);
        return 2;
    }
    g_width = atoi(argv[2]);
    g_height = atoi(argv[3]);
    const int frames = atoi(argv[4]);
    const std::string dir = argv[5];
    if (argc > 6)
        g_max_full = atol(argv[6]);
    if (argc > 7)
        g_max_low = atol(argv[7]);
    if (argc > 8)
        g_first_full = atol(argv[8]);
    /* timing runs: thread counts as the CLI's --threads / --frame-threads, 0 = the encoder's default */
    const int pool_threads = argc > 9 ? atoi(argv[9]) : 4, frame_threads = argc > 10 ? atoi(argv[10]) : 1;
    FILE *in = fopen(argv[1], "rb");
    g_planes = fopen((dir + "/planes.bin").c_str(), "wb");
    if (!in || !g_planes)
    {
        perror("open");
        return 1;
    }

    /* BASELINE.json configs[0]: --preset medium; one frame thread so that every reference is complete (and its
     * borders extended) before the next picture is searched, which makes the snapshots replayable */
    x265_param *p = x265_param_alloc();
    x265_param_default_preset(p, "medium", NULL);
    p->sourceWidth = g_width;
    p->sourceHeight = g_height;
    p->fpsNum = 25;
    p->fpsDenom = 1;
    p->frameNumThreads = frame_threads;
    p->poolNumThreads = pool_threads;
    p->logLevel = X265_LOG_WARNING;
    if (getenv("X265_CAPTURE_SPARSE")) /* whole-sequence captures: a plane's version is told by about 2048 samples per call */
        g_hip_sparse = true;
    if (getenv("X265_CAPTURE_WEIGHTB")) /* --weightb: weighted prediction in B slices (param.cpp:610) */
        p->bEnableWeightedBiPred = 1;
    if (getenv("X265_CAPTURE_RECT")) /* --rect --amp (param.cpp:562-563): partitions other than 2Nx2N, so xMergeEstimation runs */
        p->bEnableRectInter = p->bEnableAMP = 1;
    x265_encoder *enc = x265_encoder_open(p);
    if (!enc)
    {
        fprintf(stderr, "x265_encoder_open failed\n");
        return 1;
    }
#ifdef WITH_HIP
    /* Drop-in demonstration: the reference encoder with the function-pointer table overwritten by
     * source/common/hip/hip-primitives.cpp.  A patched x265_setup_primitives() would call Setup_HIP_Primitives before
     * Setup_Alias_Primitives; here the table is already set up, so the aliases that function derives from the
     * replaced entries (primitives.cpp:95-113) are refreshed by hand. */
    if (getenv("X265_USE_HIP"))
    {
        Setup_HIP_Primitives(primitives, p, 0);
        static const int sq[5] = { LUMA_4x4, LUMA_8x8, LUMA_16x16, LUMA_32x32, LUMA_64x64 };
        for (int i = 0; i < 5; i++)
        {
            primitives.sad_square[i] = primitives.sad[sq[i]];
            primitives.sa8d[i] = primitives.sa8d_inter[sq[i]];
        }
        fprintf(stderr, "enc_capture: HIP primitive table installed\n");
    }
    if (getenv("X265_USE_HIP_PLANES"))
    {
        if (x265_hip_ctx_create(0, &g_ctx_planes) != X265_HIP_OK)
        {
            fprintf(stderr, "x265_hip_ctx_create: %s\n", x265_hip_last_error());
            return 1;
        }
        g_hip_planes = atoi(getenv("X265_USE_HIP_PLANES")) == 2 ? 2 : 1;
        fprintf(stderr, "enc_capture: lookahead planes and weighted reference planes made by x265_hip_lowres_init / x265_hip_plane_weight\n");
    }
    if (getenv("X265_USE_HIP_MC"))
    {
        if (!g_ctx && x265_hip_ctx_create(0, &g_ctx) != X265_HIP_OK)
        {
            fprintf(stderr, "x265_hip_ctx_create: %s\n", x265_hip_last_error());
            return 1;
        }
        g_hip_mc = g_hip_sparse = true;
        fprintf(stderr, "enc_capture: motionCompensation (luma) answered by x265_hip_predict_batch\n");
    }
    if (getenv("X265_USE_HIP_MERGE"))
    {
        if (!g_ctx && x265_hip_ctx_create(0, &g_ctx) != X265_HIP_OK)
        {
            fprintf(stderr, "x265_hip_ctx_create: %s\n", x265_hip_last_error());
            return 1;
        }
        g_hip_merge = g_hip_sparse = true;
        fprintf(stderr, "enc_capture: xMergeEstimation answered again by x265_hip_merge_estimate_batch\n");
    }
    if (getenv("X265_USE_HIP_LOOKAHEAD"))
    {
        if (x265_hip_ctx_create(0, &g_ctx_low) != X265_HIP_OK)
        {
            fprintf(stderr, "x265_hip_ctx_create: %s\n", x265_hip_last_error());
            return 1;
        }
        g_hip_look = atoi(getenv("X265_USE_HIP_LOOKAHEAD"));
        if (g_hip_look < 1 || g_hip_look > 4)
            g_hip_look = 1;
        g_hip_sparse = true;
        fprintf(stderr, "enc_capture: the lookahead's motion searches answered a frame at a time by x265_hip_lookahead_batch%s%s\n",
                g_hip_look >= 3 ? ", the pairs of a picture batched ahead of their use" : "",
                (g_hip_look & 1) ? "" : " (the reference's own search does not run)");
    }
    if (getenv("X265_USE_HIP_ME") && !g_hip_look)
    {
        if ((!g_ctx && x265_hip_ctx_create(0, &g_ctx) != X265_HIP_OK) || x265_hip_ctx_create(0, &g_ctx_low) != X265_HIP_OK)
        {
            fprintf(stderr, "x265_hip_ctx_create: %s\n", x265_hip_last_error());
            return 1;
        }
        g_hip_me = g_hip_sparse = true;
        fprintf(stderr, "enc_capture: motionEstimate answered by x265_hip_me_batch\n");
    }
#endif
    x265_picture pic;
    x265_picture_init(p, &pic);
    const size_t ysz = (size_t)g_width * g_height, csz = ysz / 4;
    std::vector<uint8_t> buf(ysz + 2 * csz);
    x265_nal *nal;
    uint32_t nnal;
    size_t bytes_out = 0;
    uint64_t hash = 1469598103934665603ull; /* FNV-1a of the bitstream, to compare runs */
    int fed = 0, got = 0;
    auto take = [&]() {
        for (uint32_t i = 0; i < nnal; i++)
        {
            bytes_out += nal[i].sizeBytes;
            for (uint32_t k = 0; k < nal[i].sizeBytes; k++)
                hash = (hash ^ nal[i].payload[k]) * 1099511628211ull;
        }
    };
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (int f = 0; f < frames; f++)
    {
        if (fread(buf.data(), 1, buf.size(), in) != buf.size())
            break;
        pic.planes[0] = buf.data();
        pic.planes[1] = buf.data() + ysz;
        pic.planes[2] = buf.data() + ysz + csz;
        pic.stride[0] = g_width;
        pic.stride[1] = pic.stride[2] = g_width / 2;
        pic.bitDepth = 8;
        pic.pts = f;
        fed++;
        if (x265_encoder_encode(enc, &nal, &nnal, &pic, NULL) > 0)
        {
            got++;
            take();
        }
    }
    /* flush: the first calls may return 0 while the lookahead is still handing pictures over (clips shorter than
     * the lookahead), so stop only when every picture fed has come out */
    for (int idle = 0; got < fed && idle < 1000;)
    {
        if (x265_encoder_encode(enc, &nal, &nnal, NULL, NULL) > 0)
        {
            got++;
            idle = 0;
            take();
        }
        else
            idle++;
    }
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    x265_encoder_close(enc);
    x265_param_free(p);
    x265_cleanup();
    fclose(g_planes);
    FILE *out = fopen((dir + "/calls.bin").c_str(), "wb");
    fwrite(g_calls.data(), sizeof(CallRec), g_calls.size(), out);
    fclose(out);
    out = fopen((dir + "/amvp.bin").c_str(), "wb");
    fwrite(g_amvp.data(), sizeof(AmvpRec), g_amvp.size(), out);
    fclose(out);
    out = fopen((dir + "/merge.bin").c_str(), "wb");
    fwrite(g_merges.data(), sizeof(MergeRec), g_merges.size(), out);
    fclose(out);
    out = fopen((dir + "/pu.bin").c_str(), "wb");
    fwrite(g_pus.data(), sizeof(PuRec), g_pus.size(), out);
    fclose(out);
    fprintf(stderr, "enc_capture: %d pictures in %.2f s, bitstream hash %016llx\n", got,
            (ts1.tv_sec - ts0.tv_sec) + (ts1.tv_nsec - ts0.tv_nsec) * 1e-9, (unsigned long long)hash);
#ifdef WITH_HIP
    if (g_hip_planes)
        fprintf(stderr, "enc_capture: %ld lookahead planes made on the GPU, %ld differed; %ld weighted reference planes made on the GPU, %ld differed (%ld the reference left incomplete)\n",
                g_li_planes, g_li_mismatch, g_aw_planes, g_aw_mismatch, g_aw_partial);
    if (g_hip_planes == 2)
        fprintf(stderr, "enc_capture: %ld batches of weighted reference rows made on the GPU as applyWeight added them, %ld differed\n", g_aw_batches,
                g_aw_batch_mismatch);
    if (g_hip_mc)
        fprintf(stderr, "enc_capture: %ld motionCompensation luma blocks predicted on the GPU (%ld weighted, %ld bi-directional), %ld differed from the reference's own pixels; %ld of the bi-directional ones weighted\n",
                g_mc_calls, g_mc_weighted, g_mc_bi, g_mc_mismatch, g_mc_wbi);
    if (g_hip_merge)
        fprintf(stderr, "enc_capture: %ld xMergeEstimation calls answered on the GPU (%ld in weighted slices, %ld chose a bi-directional candidate), %ld differed\n",
                g_mg_calls, g_mg_weighted, g_mg_bi, g_mg_mismatch);
    if (g_hip_look)
    {
        fprintf(stderr, "enc_capture: %ld lookahead frame searches asked for, %ld launches of x265_hip_lookahead_batch in %.3f s (%.3f ms each, upload and download included), %ld lowres motionEstimate calls answered from them, %ld differed from the reference's own answer%s\n",
                g_look_frames, g_look_launches, g_look_seconds, g_look_launches ? 1e3 * g_look_seconds / g_look_launches : 0.0,
                g_look_calls.load(), g_look_mismatch, (g_hip_look & 1) ? "" : " (not compared in this mode)");
        if (g_hip_look >= 3)
            fprintf(stderr, "enc_capture: %ld of the searches asked for had been made ahead of time; %ld searches made ahead were replaced unused\n",
                    g_look_hits, g_look_unused);
    }
    if (g_hip_me)
    {
        fprintf(stderr, "enc_capture: %ld motionEstimate calls answered on the GPU, %ld differed from the reference's own answer\n",
                g_hip_calls, g_hip_mismatch);
        fprintf(stderr, "enc_capture: %ld lookahead (lowres) motionEstimate calls answered on the GPU, %ld differed\n",
                g_hip_low_calls, g_hip_low_mismatch);
    }
#endif
    fprintf(stderr, "enc_capture: %zu bytes of bitstream, %ld full-resolution + %ld lowres motionEstimate calls recorded, %d planes\n",
            bytes_out, g_n_full, g_n_low, g_next_plane);
    return 0;
}
