// TEncSearch::xMergeEstimation for lists of prediction units (x265_hip_merge_estimate_batch; reference
// source/Lib/TLibEncoder/TEncSearch.cpp:2160-2226).  Three launches on the context's stream: a thread per (unit, candidate)
// writes the candidate's prediction as motionCompensation would make it as a request (merge_make_req, me_device.h);
// mc_satd_kernel measures the requests (its items are dealt over the lanes whatever the blocks' sizes); a thread per unit
// adds the bits of the candidate index and keeps the first cheapest.
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

__global__ __launch_bounds__(256) void merge_requests_kernel(const x265_hip_merge_pu *pus, int n, int pic_w, int pic_h, x265_hip_pred_req *reqs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * X265_HIP_MRG_MAX_NUM_CANDS)
        return;
    x265_hip_pred_req r;
    merge_make_req(pus[i / X265_HIP_MRG_MAX_NUM_CANDS], i % X265_HIP_MRG_MAX_NUM_CANDS, pic_w, pic_h, r);
    reqs[i] = r;
}

// getTUBits (TEncSearch.h:94-97) and RDCost::getCost (rdcost.h:112-115); the comparison is unsigned as in the reference
__global__ __launch_bounds__(256) void merge_pick_kernel(const x265_hip_merge_pu *pus, int n, const int32_t *satd, x265_hip_merge_result *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const x265_hip_merge_pu &p = pus[i];
    x265_hip_merge_result r = {};
    r.cost = 0xFFFFFFFFu;
    for (int c = 0; c < X265_HIP_MRG_MAX_NUM_CANDS; c++)
    {
        r.satd[c] = -1;
        if (!merge_cand_used(p, c))
            continue;
        const uint32_t s = (uint32_t)satd[i * X265_HIP_MRG_MAX_NUM_CANDS + c];
        const uint32_t bits = (uint32_t)(c + (c < (int)p.num_cand - 1));
        const uint32_t cost = s + (uint32_t)(((uint64_t)bits * p.lambda_sad + 128) >> 8);
        r.satd[c] = (int32_t)s;
        if (cost < r.cost)
        {
            r.cost = cost;
            r.bits = bits;
            r.index = (uint8_t)c;
            r.inter_dir = (uint8_t)((p.cand[c].ref_plane[0] >= 0 ? 1 : 0) | (p.cand[c].ref_plane[1] >= 0 ? 2 : 0));
        }
    }
    out[i] = r;
}

void launch_merge_estimate(const DeviceTables *tabs, const x265_hip_merge_pu *pus, int n, int pic_w, int pic_h, x265_hip_pred_req *reqs,
                           int32_t *satd, x265_hip_merge_result *out, uint32_t *counters, int parity, hipStream_t st)
{
    if (n <= 0)
        return;
    const int nreq = n * X265_HIP_MRG_MAX_NUM_CANDS;
    hipLaunchKernelGGL(merge_requests_kernel, dim3((nreq + 255) / 256), dim3(256), 0, st, pus, n, pic_w, pic_h, reqs);
    launch_mc_satd(tabs, reqs, nreq, satd, counters, parity, st);
    hipLaunchKernelGGL(merge_pick_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pus, n, satd, out);
}

} // namespace x265hip
