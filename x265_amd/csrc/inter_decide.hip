// inter_decide.hip -- what TEncSearch::predInterSearch does with a prediction unit behind its (list, reference)
// iterations (reference Lib/TLibEncoder/TEncSearch.cpp:2342-2358, 2360-2439, 2449-2499), on the device:
//   * per iteration: bits = list selection + MVP index + reference index (getTUBits) + BitCost::bitcost(outmv), cost =
//     (motionEstimate's cost - mvcost(outmv)) + RDCost::getCost(bits), and xCheckBestMVP's second look at the AMVP index
//     (:2565-2598); the first cheapest reference of each list;
//   * the bidirectional candidate: SATD of the block against pixelavg_pp of the two lists' predictions (made by
//     bidir_cost_kernel from requests this file writes), its zero-vector form when both predictors lie inside the
//     picture's search area (:2400-2437), with xCheckBestMVP for each half;
//   * the choice between list 0, list 1 and both (the merge candidate stays with the caller).
// No pixel is read here: the two kernels are one thread per prediction unit, a few hundred bytes each; the pixels are
// motionEstimate's (launch_inter_search in front) and bidir_cost_kernel's (between the two).
//
// BitCost::bitcost (encoder/bitcost.h:48-52) adds two entries of a float table (s_bitsizes, bitcost.cpp:68-80) and 0.5f
// and truncates: the table is made on the host exactly as the reference makes it and uploaded as data
// (DeviceTables::bitsizes), the two float additions are done here in the reference's order with IEEE rounding.
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

__device__ __forceinline__ int mv_bits(const float *bs, int x, int y, int px, int py)
{
    // (differences beyond the table's 16383 quarter samples, which the reference reads out of bounds, take its last entry)
    const int ix = min((abs(x - px) << 1) + (x < px ? 1 : 0), 32768), iy = min((abs(y - py) << 1) + (y < py ? 1 : 0), 32768);
    const float v = __fadd_rn(__fadd_rn(bs[ix], bs[iy]), 0.5f);
    return (int)(uint16_t)(int)v;
}

// RDCost::getCost, encoder/rdcost.h:112-115
__device__ __forceinline__ uint32_t cost_of_bits(uint32_t bits, uint32_t lambda_sad) { return (uint32_t)(((uint64_t)bits * lambda_sad + 128) >> 8); }

// TEncSearch::xCheckBestMVP, TEncSearch.cpp:2565-2598 (AMVP_MAX_NUM_CANDS = 2, MVP_IDX_BITS = 1)
__device__ __forceinline__ void check_best_mvp(const float *bs, const int16_t cand[4], int mvx, int mvy, int &idx, uint32_t &bits, uint32_t &cost,
                                               uint32_t lambda_sad)
{
    const int other = idx ^ 1;
    const int orig = mv_bits(bs, mvx, mvy, cand[2 * idx], cand[2 * idx + 1]) + 1;
    const int alt = mv_bits(bs, mvx, mvy, cand[2 * other], cand[2 * other + 1]) + 1;
    if (alt < orig)
    {
        const uint32_t before = bits;
        idx = other;
        bits = before - (uint32_t)orig + (uint32_t)alt;
        cost = (cost - cost_of_bits(before, lambda_sad)) + cost_of_bits(bits, lambda_sad);
    }
}

__device__ __forceinline__ void clip_mv_dev(int &qx, int &qy, int cu_x, int cu_y, int pic_w, int pic_h) // TComDataCU::clipMv, TComDataCU.cpp:2502-2514
{
    const int xmax = (pic_w + 8 - cu_x - 1) << 2, xmin = (-64 - 8 - cu_x + 1) << 2;
    const int ymax = (pic_h + 8 - cu_y - 1) << 2, ymin = (-64 - 8 - cu_y + 1) << 2;
    qx = min(xmax, max(xmin, qx));
    qy = min(ymax, max(ymin, qy));
}

__device__ __forceinline__ int s16d(int v) { return (int)(int16_t)v; }

// the (list, reference) loop of a unit (:2303-2358) and, where the unit has a bidirectional candidate, the two requests
// bidir_cost_kernel answers for it: 2 u the two lists' predictions at their vectors, 2 u + 1 the coincident blocks of
// the two searched planes (the zero-vector form) -- or, where that form is not tried, the first request again, and for
// a unit without a bidirectional candidate the block against its own searched plane: every request of the list is one
// the kernel may run
__global__ __launch_bounds__(64) void inter_decide_lists_kernel(const DeviceTables *tabs, const x265_hip_inter_job *jobs, const x265_hip_inter_result *res,
                                                               const x265_hip_inter_pu *pus, int npu, int pic_w, int pic_h, x265_hip_inter_pu_result *out,
                                                               x265_hip_bidir_req *bireq, int any_bidir)
{
    const int u = blockIdx.x * 64 + threadIdx.x;
    if (u >= npu)
        return;
    const x265_hip_inter_pu p = pus[u];
    const float *bs = tabs->bitsizes;
    x265_hip_inter_pu_result o = {};
    o.list[0].cost = o.list[1].cost = o.bidir_cost = 0xffffffffu;
    int best_job[2] = { p.first_job, p.first_job };
    int k = p.first_job;
    for (int l = 0; l < 2; l++)
        for (int ref = 0; ref < p.num_ref[l]; ref++, k++)
        {
            const x265_hip_inter_job &q = jobs[k];
            const x265_hip_inter_result r = res[k];
            uint32_t bits = (uint32_t)p.list_sel_bits[l] + 1u + (uint32_t)(ref + (ref < p.num_ref[l] - 1 ? 1 : 0)); // MVP_IDX_BITS, getTUBits
            bits += (uint32_t)mv_bits(bs, r.mvx, r.mvy, r.mvp_x, r.mvp_y);
            const uint16_t *tab = tabs->costtab[q.qp];
            const int mvc = (int)(uint16_t)(tab_at(tab, r.mvp_x, r.mvx) + tab_at(tab, r.mvp_y, r.mvy)); // BitCost::mvcost, bitcost.h:45
            uint32_t cost = (uint32_t)(r.cost - mvc) + cost_of_bits(bits, p.lambda_sad);
            int idx = r.mvp_idx;
            check_best_mvp(bs, q.amvp, r.mvx, r.mvy, idx, bits, cost, p.lambda_sad);
            if (cost < o.list[l].cost)
            {
                x265_hip_inter_side &s = o.list[l];
                s.cost = cost;
                s.bits = (int32_t)bits;
                s.mvx = r.mvx, s.mvy = r.mvy;
                s.mvp_x = q.amvp[2 * idx], s.mvp_y = q.amvp[2 * idx + 1];
                s.ref = (int8_t)ref;
                s.mvp_idx = (int8_t)idx;
                best_job[l] = k;
            }
        }
    const x265_hip_inter_job &q0 = jobs[best_job[0]], &q1 = jobs[best_job[1]];
    x265_hip_bidir_req b0 = {}, b1;
    b0.fenc_plane = q0.fenc_plane;
    b0.pu_x = q0.pu_x, b0.pu_y = q0.pu_y, b0.pu_w = q0.pu_w, b0.pu_h = q0.pu_h;
    b0.metric = X265_HIP_SATD;
    const bool bidir = any_bidir && !p.no_bidir && o.list[0].cost != 0xffffffffu && o.list[1].cost != 0xffffffffu;
    if (!bidir)
    {
        b0.ref0_plane = b0.ref1_plane = q0.ref_plane;
        b1 = b0;
        o.inter_dir = o.list[0].cost <= o.list[1].cost ? 1 : 2;
    }
    else
    {
        o.bidir[0] = o.list[0];
        o.bidir[1] = o.list[1];
        o.pending = 1;
        b0.ref0_plane = q0.pred_plane, b0.ref1_plane = q1.pred_plane;
        b0.qmv0_x = o.list[0].mvx, b0.qmv0_y = o.list[0].mvy, b0.qmv1_x = o.list[1].mvx, b0.qmv1_y = o.list[1].mvy;
        b1 = b0;
        bool try_zero = (o.list[0].mvx | o.list[0].mvy | o.list[1].mvx | o.list[1].mvy) != 0;
        if (try_zero)
        {
            // xSetSearchRange around the zero vector with merange = the larger picture side (:2400-2412, 2600-2617)
            const int d = s16d(max(pic_w, pic_h) << 2);
            int px = 0, py = 0;
            clip_mv_dev(px, py, q0.cu_x, q0.cu_y, pic_w, pic_h);
            int lox = s16d(px - d), loy = s16d(py - d), hix = s16d(px + d), hiy = s16d(py + d);
            clip_mv_dev(lox, loy, q0.cu_x, q0.cu_y, pic_w, pic_h);
            clip_mv_dev(hix, hiy, q0.cu_x, q0.cu_y, pic_w, pic_h);
            int minx = lox >> 2, maxx = hix >> 2, miny = min(loy >> 2, (int)q0.ref_lag), maxy = min(hiy >> 2, (int)q0.ref_lag);
            maxy = s16d(maxy + 2);
            minx = s16d(minx << 2), miny = s16d(miny << 2), maxx = s16d(maxx << 2), maxy = s16d(maxy << 2);
            for (int l = 0; l < 2; l++)
                try_zero = try_zero && o.list[l].mvp_x >= minx && o.list[l].mvp_x <= maxx && o.list[l].mvp_y >= miny && o.list[l].mvp_y <= maxy;
        }
        o.zero_tried = try_zero;
        if (try_zero)
        {
            b1.ref0_plane = q0.ref_plane, b1.ref1_plane = q1.ref_plane;
            b1.qmv0_x = b1.qmv0_y = b1.qmv1_x = b1.qmv1_y = 0;
        }
    }
    out[u] = o;
    bireq[2 * u] = b0;
    bireq[2 * u + 1] = b1;
}

// the bidirectional candidate from the two SATDs (:2383-2437) and the choice (:2449-2499 without the merge candidate)
__global__ __launch_bounds__(64) void inter_decide_bidir_kernel(const DeviceTables *tabs, const x265_hip_inter_job *jobs, const x265_hip_inter_pu *pus, int npu,
                                                               const int32_t *satd, x265_hip_inter_pu_result *out)
{
    const int u = blockIdx.x * 64 + threadIdx.x;
    if (u >= npu)
        return;
    x265_hip_inter_pu_result o = out[u];
    if (!o.pending)
        return;
    const x265_hip_inter_pu p = pus[u];
    const float *bs = tabs->bitsizes;
    const int sel = (int)p.list_sel_bits[2] - ((int)p.list_sel_bits[0] + (int)p.list_sel_bits[1]);
    o.bidir_bits = o.list[0].bits + o.list[1].bits + sel;
    o.bidir_cost = (uint32_t)satd[2 * u] + cost_of_bits((uint32_t)o.bidir_bits, p.lambda_sad);
    if (o.zero_tried)
    {
        const x265_hip_inter_job *q[2] = { &jobs[p.first_job + o.list[0].ref], &jobs[p.first_job + p.num_ref[0] + o.list[1].ref] };
        uint32_t bits[2];
        int idx[2];
        for (int l = 0; l < 2; l++)
        {
            const x265_hip_inter_side &s = o.list[l];
            idx[l] = s.mvp_idx;
            bits[l] = (uint32_t)(s.bits - mv_bits(bs, s.mvx, s.mvy, s.mvp_x, s.mvp_y) + mv_bits(bs, 0, 0, s.mvp_x, s.mvp_y));
        }
        uint32_t cost = (uint32_t)satd[2 * u + 1] + cost_of_bits(bits[0], p.lambda_sad) + cost_of_bits(bits[1], p.lambda_sad);
        check_best_mvp(bs, q[0]->amvp, 0, 0, idx[0], bits[0], cost, p.lambda_sad);
        check_best_mvp(bs, q[1]->amvp, 0, 0, idx[1], bits[1], cost, p.lambda_sad);
        if (cost < o.bidir_cost)
        {
            for (int l = 0; l < 2; l++)
            {
                o.bidir[l].mvx = o.bidir[l].mvy = 0;
                o.bidir[l].mvp_x = q[l]->amvp[2 * idx[l]], o.bidir[l].mvp_y = q[l]->amvp[2 * idx[l] + 1];
                o.bidir[l].mvp_idx = (int8_t)idx[l];
                o.bidir[l].bits = (int32_t)bits[l];
            }
            o.bidir_cost = cost;
            o.bidir_bits = (int32_t)(bits[0] + bits[1]) + sel;
        }
    }
    o.pending = 0;
    o.inter_dir = (o.bidir_cost < o.list[0].cost && o.bidir_cost < o.list[1].cost) ? 3 : (o.list[0].cost <= o.list[1].cost ? 1 : 2);
    out[u] = o;
}

void launch_inter_decide_lists(const DeviceTables *tabs, const x265_hip_inter_job *jobs, const x265_hip_inter_result *res, const x265_hip_inter_pu *pus, int npu,
                               int pic_w, int pic_h, x265_hip_inter_pu_result *out, x265_hip_bidir_req *bireq, bool any_bidir, hipStream_t st)
{
    if (npu > 0)
        hipLaunchKernelGGL(inter_decide_lists_kernel, dim3((npu + 63) / 64), dim3(64), 0, st, tabs, jobs, res, pus, npu, pic_w, pic_h, out, bireq,
                           any_bidir ? 1 : 0);
}

void launch_inter_decide_bidir(const DeviceTables *tabs, const x265_hip_inter_job *jobs, const x265_hip_inter_pu *pus, int npu, const int32_t *satd,
                               x265_hip_inter_pu_result *out, hipStream_t st)
{
    if (npu > 0)
        hipLaunchKernelGGL(inter_decide_bidir_kernel, dim3((npu + 63) / 64), dim3(64), 0, st, tabs, jobs, pus, npu, satd, out);
}

} // namespace x265hip
