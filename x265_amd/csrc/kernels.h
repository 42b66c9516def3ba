// kernels.h -- host-visible launchers of the kernels in kernels.hip
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/x265_hip.h"

namespace x265hip {

struct DeviceTables;

enum { PRIM_HPP = 0, PRIM_VPP = 1, PRIM_HPS = 2, PRIM_VSP = 3, PRIM_VPS = 4, PRIM_VSS = 5, PRIM_P2S = 6 };

struct PrimCmpArgs
{
    const uint8_t *fenc;
    const uint8_t *fref[4];
    int fstride, rstride;
    int w, h, nref, metric;
};

struct PrimFilterArgs
{
    const void *src; // uint8 for hpp/vpp/hps/vps/p2s, int16 for vsp/vss; points at output sample (0,0)'s source
    void *dst;       // uint8 for hpp/vpp/vsp, int16 for hps/vps/vss/p2s
    int sstride, dstride; // in elements
    int w, rows, coeff, kind;
};

struct LowresArgs
{
    const uint8_t *src;   // full-resolution picture sample (0,0)
    uint8_t *dst[4];      // first byte of the four padded lowres buffers (lowresPlane[0..3])
    int src_stride, dst_stride;
    int width, lines;     // lowres picture size
    int margin_x, margin_y;
};

struct PlaneBorderArgs
{
    const uint8_t *src; // picture sample (0,0) of the source plane
    uint8_t *dst;       // picture sample (0,0) of the destination plane (may equal src)
    int stride;         // both planes
    int width, height, margin_x, margin_y;
    int weighted, w0, round, shift, offset; // weight_pp arguments as applyWeight passes them
    int py0, pyn; // padded rows py0 .. py0 + pyn - 1 of the height + 2 margin_y are produced (all of them: 0, height + 2 margin_y)
};

void launch_lowres_init(const LowresArgs &a, hipStream_t st);
void launch_plane_border(const PlaneBorderArgs &a, hipStream_t st);
struct PrimAvgArgs
{
    const uint8_t *src0, *src1;
    uint8_t *dst;
    int sstride0, sstride1, dstride, w, h;
};

void launch_prim_avg(const PrimAvgArgs &a, hipStream_t st);

struct PrimAddAvgArgs
{
    const int16_t *src0, *src1;
    uint8_t *dst;
    int sstride0, sstride1, dstride, w, h;
};

void launch_prim_addavg(const PrimAddAvgArgs &a, hipStream_t st);
// What a motionEstimate launch needs besides its arguments.  `route`: me_route_words(n) 32-bit words of device scratch
// (the lists that split the jobs between the kernels, me_shared.h: made and used by the launch) whose first
// me_route_counter_words() words were zeroed once; `parity` flips with every launch that uses the buffer.  The launch
// is ordered on `stream`; with `side` (another stream) and the two events it runs its large-block kernel beside the
// other ones.
struct MeRoute
{
    uint32_t *route;
    int parity;
    hipStream_t stream, side;
    hipEvent_t fork, join;
    // record guard (x265_hip_ctx_check_records mode 2): two words, [0] counts the refused records of the launches since
    // the words were last read, [1] is the lowest index of one; null: records are taken on trust
    uint32_t *guard;
};
size_t me_route_words(int n);
size_t me_route_counter_words();
// whether a job goes to one of the specialised kernels; a launch whose MeRoute::route is null has no such job
// (host-array callers check) and runs me_kernel with one workgroup per job
bool me_job_is_routed(int w, int h, int method);
void launch_me(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, x265_hip_me_result *out, const MeRoute &r);
// the steps of predInterSearch in front of motionEstimate, then the searches of the jobs they produce.  `mejobs`: scratch of
// X265HIP_INTER_SCRATCH_BYTES(n) bytes (n motionEstimate jobs, the candidates' 2 n cost requests, their 2 n costs);
// `counters` / `parity`: the batch counters of the cost kernels
#define X265HIP_INTER_SCRATCH_BYTES(n) ((size_t)(n) * (sizeof(x265_hip_me_job) + 2 * sizeof(x265_hip_cost_req) + 2 * sizeof(int32_t)))
void launch_inter_search(const DeviceTables *tabs, const x265_hip_inter_job *jobs, int n, int pic_w, int pic_h, x265_hip_me_job *mejobs,
                         x265_hip_inter_result *out, const MeRoute &r, uint32_t *counters, int parity);
// guard.hip: record guards of the request lists (check_records mode 2).  `sane`: the list the kernel will run, `bad`: a
// byte per record, `words`: the context's two guard words; launch_guard_fix sets the results of the replaced records
// (`per` words each) to -1 behind the kernel
void launch_guard_cost(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int max_metric, int safe_plane, x265_hip_cost_req *sane, uint8_t *bad,
                       uint32_t *words, hipStream_t st);
void launch_guard_refine(const DeviceTables *tabs, const x265_hip_refine_req *reqs, int n, int safe_plane, x265_hip_refine_req *sane, uint8_t *bad,
                         uint32_t *words, hipStream_t st);
void launch_guard_bidir(const DeviceTables *tabs, const x265_hip_bidir_req *reqs, int n, int safe_plane, x265_hip_bidir_req *sane, uint8_t *bad,
                        uint32_t *words, hipStream_t st);
void launch_guard_ctu(const DeviceTables *tabs, const x265_hip_ctu_job *jobs, int n, int safe_plane, int safe_qp, x265_hip_ctu_job *sane, uint8_t *bad,
                      uint32_t *words, hipStream_t st);
void launch_guard_pred(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, int safe_plane, x265_hip_pred_req *sane, uint8_t *bad,
                       uint32_t *words, hipStream_t st);
void launch_guard_fix(const uint8_t *bad, int n, int per, int32_t *out, hipStream_t st, bool pairs = false);
// inter_decide.hip: predInterSearch behind its iterations (`bireq`: 2 npu requests for launch_bidir_cost, whose 2 npu
// costs are `satd`)
void launch_inter_decide_lists(const DeviceTables *tabs, const x265_hip_inter_job *jobs, const x265_hip_inter_result *res, const x265_hip_inter_pu *pus, int npu,
                               int pic_w, int pic_h, x265_hip_inter_pu_result *out, x265_hip_bidir_req *bireq, bool any_bidir, hipStream_t st);
void launch_inter_decide_bidir(const DeviceTables *tabs, const x265_hip_inter_job *jobs, const x265_hip_inter_pu *pus, int npu, const int32_t *satd,
                               x265_hip_inter_pu_result *out, hipStream_t st);
// `counters`: X265HIP_COST_COUNTER_BYTES of zeroed device memory owned by the context (two alternating sets of
// batch counters); `parity` flips with every launch that shares them, and such launches are ordered on one stream
#define X265HIP_COST_COUNTER_BYTES (2 * 64 * 64)
void launch_cost(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int32_t *out, uint32_t *counters, int parity,
                 hipStream_t st);
void launch_predict(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, uint32_t *counters, int parity, hipStream_t st);
void launch_mc_satd(const DeviceTables *tabs, const x265_hip_pred_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st);
// xMergeEstimation: reqs / satd are scratch for X265_HIP_MRG_MAX_NUM_CANDS * n requests and their costs
void launch_merge_estimate(const DeviceTables *tabs, const x265_hip_merge_pu *pus, int n, int pic_w, int pic_h, x265_hip_pred_req *reqs,
                           int32_t *satd, x265_hip_merge_result *out, uint32_t *counters, int parity, hipStream_t st);
void launch_dist(const DeviceTables *tabs, const x265_hip_cost_req *reqs, int n, int32_t *out, uint32_t *counters, int parity,
                 hipStream_t st);
void launch_bidir_cost(const DeviceTables *tabs, const x265_hip_bidir_req *reqs, int n, int32_t *out, uint32_t *counters, int parity,
                       hipStream_t st);
void launch_refine(const DeviceTables *tabs, const x265_hip_refine_req *reqs, int n, int32_t *out, uint32_t *counters, int parity,
                   hipStream_t st);
// `tickets`: 16 zeroed words per search; `status`: one zeroed word; `results` filled with 0xff bytes (all on `st`, before)
void launch_lookahead(const DeviceTables *tabs, const x265_hip_lookahead_job *jobs, int njobs, int w_cu, int h_cu, uint32_t *tickets,
                      x265_hip_me_result *results, int *status, hipStream_t st);
void launch_refine_sections(const DeviceTables *tabs, const x265_hip_refine_req *reqs, const x265_hip_refine_section *sections, int nsec,
                            int32_t *out, uint32_t *counters, int parity, hipStream_t st);
void launch_prim_cmp(const PrimCmpArgs &a, int32_t *out, hipStream_t st);
void launch_prim_filter(const PrimFilterArgs &a, hipStream_t st);
void launch_fullsearch2(const DeviceTables *tabs, const x265_hip_ctu_job *jobs, int n, x265_hip_me_result *out,
                        hipStream_t st);
int fullsearch2_setup();

} // namespace x265hip
