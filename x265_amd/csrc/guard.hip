// guard.hip -- record guards for the request lists of the `_device` entry points (x265_hip_ctx_check_records mode 2).
//
// The compute kernels take their records on trust: a record that names a missing plane or reaches outside one makes a
// wave read outside a buffer.  With the guard on, a thread per record puts it through the checks the host-array entry
// point makes (validate_cost / validate_dist / validate_refine / validate_bidir / validate_ctu / validate_pred in api.hip, against the plane directory
// and the planes' byte spans in device memory) and writes the list the kernel will run: a record that passes as it is, a
// record that fails replaced by the same-sized block at the origin of a plane the host knows to be large enough, zero
// vectors.  The hot kernels stay as they are; behind them a second pass sets the results of the replaced records to -1.
// Refusals are counted in the context's two guard words like those of motionEstimate jobs (me_route_kernel).
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

__device__ __forceinline__ void guard_refuse(uint32_t *words, int i)
{
    atomicAdd(&words[0], 1u);
    atomicMin(&words[1], (uint32_t)i);
}

__global__ __launch_bounds__(256) void guard_cost_kernel(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int max_metric, int safe_plane,
                                                         x265_hip_cost_req *sane, uint8_t *bad, uint32_t *words)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    x265_hip_cost_req q = reqs[i];
    const bool ok = guard_block(tabs, q.fenc_plane, q.ref_plane, q.pu_w, q.pu_h) && q.metric <= max_metric &&
                    guard_inside(tabs, q.fenc_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h) &&
                    guard_displaced(tabs, q.ref_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv_x, q.qmv_y);
    if (!ok)
    {
        guard_refuse(words, i);
        if (!guard_partition(q.pu_w, q.pu_h))
            q.pu_w = q.pu_h = 8;
        q.fenc_plane = q.ref_plane = safe_plane;
        q.pu_x = q.pu_y = q.qmv_x = q.qmv_y = 0;
        q.metric = X265_HIP_SAD;
    }
    sane[i] = q;
    bad[i] = !ok;
}

__global__ __launch_bounds__(256) void guard_refine_kernel(const DeviceTables *tabs, const x265_hip_refine_req *reqs, int n, int safe_plane,
                                                           x265_hip_refine_req *sane, uint8_t *bad, uint32_t *words)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    x265_hip_refine_req q = reqs[i];
    const int ring = q.pattern & (X265_HIP_RING_QPEL | X265_HIP_RING_HPEL);
    bool ok = guard_block(tabs, q.fenc_plane, q.ref_plane, q.pu_w, q.pu_h) && q.metric <= X265_HIP_SATD &&
              (ring == X265_HIP_RING_QPEL || ring == X265_HIP_RING_HPEL) &&
              !(q.pattern & ~(X265_HIP_RING_QPEL | X265_HIP_RING_HPEL | X265_HIP_RING_8 | X265_HIP_RING_CENTRE));
    if (ok)
    {
        // (validate_refine: rows (cy - step) >> 2 - 3 .. + h + 5, columns ix - 3 .. ix + w + 4 per horizontal position)
        const int step = ring == X265_HIP_RING_HPEL ? 2 : 1;
        const long x0 = q.pu_x + ((q.qmv_x - step) >> 2), x1 = q.pu_x + ((q.qmv_x + step) >> 2) + q.pu_w;
        const long y0 = q.pu_y + ((q.qmv_y - step) >> 2), y1 = y0 + q.pu_h + 2;
        ok = guard_inside(tabs, q.fenc_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h) && guard_reach(tabs, q.ref_plane, x0, y0, x1, y1);
    }
    if (!ok)
    {
        guard_refuse(words, i);
        if (!guard_partition(q.pu_w, q.pu_h))
            q.pu_w = q.pu_h = 8;
        q.fenc_plane = q.ref_plane = safe_plane;
        q.pu_x = q.pu_y = q.qmv_x = q.qmv_y = 0;
        q.metric = X265_HIP_SAD;
        q.pattern = X265_HIP_RING_HPEL;
    }
    sane[i] = q;
    bad[i] = !ok;
}

__global__ __launch_bounds__(256) void guard_bidir_kernel(const DeviceTables *tabs, const x265_hip_bidir_req *reqs, int n, int safe_plane,
                                                          x265_hip_bidir_req *sane, uint8_t *bad, uint32_t *words)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    x265_hip_bidir_req q = reqs[i];
    const int flags = q.metric & (X265_HIP_LOWRES_REF | X265_HIP_MC_AVG);
    bool ok = guard_block(tabs, q.fenc_plane, q.ref0_plane, q.pu_w, q.pu_h) && guard_block(tabs, q.fenc_plane, q.ref1_plane, q.pu_w, q.pu_h) &&
              (q.metric & ~(X265_HIP_LOWRES_REF | X265_HIP_MC_AVG)) <= X265_HIP_SSE && flags != (X265_HIP_LOWRES_REF | X265_HIP_MC_AVG);
    if (ok && (q.metric & X265_HIP_LOWRES_REF))
    {
        ok = q.pu_w == 8 && q.pu_h == 8;
        for (int r = 0; r < 2 && ok; r++)
        {
            const int rp = r ? q.ref1_plane : q.ref0_plane;
            ok = rp + 3 < X265_HIP_MAX_PLANES;
            for (int k = 1; k < 4 && ok; k++)
                ok = tabs->planes[rp + k].origin && tabs->planes[rp + k].stride == tabs->planes[rp].stride &&
                     tabs->span_first[rp + k] == tabs->span_first[rp] && tabs->span_end[rp + k] == tabs->span_end[rp];
        }
    }
    ok = ok && guard_inside(tabs, q.fenc_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h) &&
         guard_displaced(tabs, q.ref0_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv0_x, q.qmv0_y) &&
         guard_displaced(tabs, q.ref1_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv1_x, q.qmv1_y);
    if (!ok)
    {
        guard_refuse(words, i);
        if (!guard_partition(q.pu_w, q.pu_h))
            q.pu_w = q.pu_h = 8;
        q.fenc_plane = q.ref0_plane = q.ref1_plane = safe_plane;
        q.pu_x = q.pu_y = q.qmv0_x = q.qmv0_y = q.qmv1_x = q.qmv1_y = 0;
        q.metric = X265_HIP_SAD;
    }
    sane[i] = q;
    bad[i] = !ok;
}

// x265_hip_fullsearch_device (validate_ctu): a CTU that fails searches the one position of the safe plane's origin
__global__ __launch_bounds__(256) void guard_ctu_kernel(const DeviceTables *tabs, const x265_hip_ctu_job *jobs, int n, int safe_plane, int safe_qp,
                                                        x265_hip_ctu_job *sane, uint8_t *bad, uint32_t *words)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    x265_hip_ctu_job j = jobs[i];
    const bool ok = guard_block(tabs, j.fenc_plane, j.ref_plane, 64, 64) && j.qp < X265_HIP_MAX_QP && tabs->costtab[j.qp] &&
                    j.mvmin_x <= j.mvmax_x && j.mvmin_y <= j.mvmax_y && j.mvmax_x - j.mvmin_x <= 131 && j.mvmax_y - j.mvmin_y <= 131 &&
                    guard_inside(tabs, j.fenc_plane, j.ctu_x, j.ctu_y, 64, 64) &&
                    guard_reach(tabs, j.ref_plane, (long)j.ctu_x + j.mvmin_x, (long)j.ctu_y + j.mvmin_y, (long)j.ctu_x + j.mvmax_x + 64,
                                (long)j.ctu_y + j.mvmax_y + 64);
    if (!ok)
    {
        guard_refuse(words, i);
        j.qp = (uint8_t)safe_qp;
        j.fenc_plane = j.ref_plane = safe_plane;
        j.ctu_x = j.ctu_y = j.mvmin_x = j.mvmin_y = j.mvmax_x = j.mvmax_y = j.qmvp_x = j.qmvp_y = 0;
    }
    sane[i] = j;
    bad[i] = !ok;
}

// x265_hip_mc_satd_batch_device (validate_pred as mc_satd uses it: nothing is written, the "destination" is the source)
__global__ __launch_bounds__(256) void guard_pred_kernel(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, int safe_plane,
                                                         x265_hip_pred_req *sane, uint8_t *bad, uint32_t *words)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    x265_hip_pred_req q = reqs[i];
    bool ok = guard_block(tabs, q.dst_plane, q.ref0_plane, q.pu_w, q.pu_h) &&
              (q.ref1_plane < 0 || guard_block(tabs, q.dst_plane, q.ref1_plane, q.pu_w, q.pu_h));
    if (ok && q.ref1_plane == X265_HIP_PRED_WEIGHTED)
        ok = q.log2_denom <= 7 && q.qmv1_x >= -128 && q.qmv1_x <= 255 && q.qmv1_y >= -128 && q.qmv1_y <= 127;
    if (ok && q.weighted_bi)
        ok = q.ref1_plane >= 0 && q.log2_denom <= 7 && q.wp_w0 >= -128 && q.wp_w0 <= 255 && q.wp_w1 >= -128 && q.wp_w1 <= 255 &&
             q.wp_o0 >= -128 && q.wp_o0 <= 127 && q.wp_o1 >= -128 && q.wp_o1 <= 127;
    ok = ok && guard_inside(tabs, q.dst_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h) &&
         guard_displaced(tabs, q.ref0_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv0_x, q.qmv0_y) &&
         (q.ref1_plane < 0 || guard_displaced(tabs, q.ref1_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv1_x, q.qmv1_y));
    if (!ok)
    {
        guard_refuse(words, i);
        const bool part = guard_partition(q.pu_w, q.pu_h);
        q = x265_hip_pred_req{};
        q.pu_w = part ? reqs[i].pu_w : 8;
        q.pu_h = part ? reqs[i].pu_h : 8;
        q.dst_plane = q.ref0_plane = safe_plane;
        q.ref1_plane = -1;
    }
    sane[i] = q;
    bad[i] = !ok;
}

// results of the replaced records: `per` 32-bit words each (1 cost, the 9 costs of a ring, or with `pairs` the 85
// (cost, mv) results of a CTU), costs set to -1 and vectors to zero
__global__ __launch_bounds__(256) void guard_fix_kernel(const uint8_t *bad, int n, int per, int pairs, int32_t *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && bad[i])
        for (int k = 0; k < per; k++)
            out[(size_t)i * per + k] = (pairs && (k & 1)) ? 0 : -1;
}

void launch_guard_cost(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int max_metric, int safe_plane, x265_hip_cost_req *sane, uint8_t *bad,
                       uint32_t *words, hipStream_t st)
{
    hipLaunchKernelGGL(guard_cost_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tabs, reqs, n, max_metric, safe_plane, sane, bad, words);
}

void launch_guard_refine(const DeviceTables *tabs, const x265_hip_refine_req *reqs, int n, int safe_plane, x265_hip_refine_req *sane, uint8_t *bad,
                         uint32_t *words, hipStream_t st)
{
    hipLaunchKernelGGL(guard_refine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tabs, reqs, n, safe_plane, sane, bad, words);
}

void launch_guard_bidir(const DeviceTables *tabs, const x265_hip_bidir_req *reqs, int n, int safe_plane, x265_hip_bidir_req *sane, uint8_t *bad,
                        uint32_t *words, hipStream_t st)
{
    hipLaunchKernelGGL(guard_bidir_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tabs, reqs, n, safe_plane, sane, bad, words);
}

void launch_guard_ctu(const DeviceTables *tabs, const x265_hip_ctu_job *jobs, int n, int safe_plane, int safe_qp, x265_hip_ctu_job *sane, uint8_t *bad,
                      uint32_t *words, hipStream_t st)
{
    hipLaunchKernelGGL(guard_ctu_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tabs, jobs, n, safe_plane, safe_qp, sane, bad, words);
}

void launch_guard_pred(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, int safe_plane, x265_hip_pred_req *sane, uint8_t *bad,
                       uint32_t *words, hipStream_t st)
{
    hipLaunchKernelGGL(guard_pred_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tabs, reqs, n, safe_plane, sane, bad, words);
}

void launch_guard_fix(const uint8_t *bad, int n, int per, int32_t *out, hipStream_t st, bool pairs)
{
    hipLaunchKernelGGL(guard_fix_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bad, n, per, pairs ? 1 : 0, out);
}

} // namespace x265hip
