"""numpy views of the structs declared in include/x265_hip.h (keep the two in step)."""
import numpy as np

MAX_MVC = 12
MAX_PLANES = 512
NUM_PARTS = 25

# LumaPartitions order, source/common/primitives.h:38-46 of the reference (index order is ABI)
PARTS = [(4, 4), (8, 8), (8, 4), (4, 8), (16, 16), (16, 8), (8, 16), (16, 12), (12, 16), (16, 4), (4, 16),
         (32, 32), (32, 16), (16, 32), (32, 24), (24, 32), (32, 8), (8, 32), (64, 64), (64, 32), (32, 64),
         (64, 48), (48, 64), (64, 16), (16, 64)]

DIA, HEX, UMH, STAR, FULL = range(5)
SAD, SATD, SA8D, SSE = 0, 1, 2, 3
LOWRES_REF = 0x80  # OR-ed into ME_JOB.method / BIDIR_REQ.metric: the reference is a Lowres (four half-pel planes, 8x8 blocks)
MC_AVG = 0x40  # OR-ed into BIDIR_REQ.metric: addAvg of 14-bit intermediates instead of pixelavg_pp

ME_JOB = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"),
    ("pu_x", "<i2"), ("pu_y", "<i2"),
    ("pu_w", "u1"), ("pu_h", "u1"), ("method", "u1"), ("subme", "u1"),
    ("mvmin_x", "<i2"), ("mvmin_y", "<i2"), ("mvmax_x", "<i2"), ("mvmax_y", "<i2"),
    ("qmvp_x", "<i2"), ("qmvp_y", "<i2"),
    ("merange", "<i2"), ("qp", "u1"), ("num_mvc", "u1"),
    ("mvc", "<i2", (2 * MAX_MVC,)),
], align=False)
assert ME_JOB.itemsize == 80

ME_RESULT = np.dtype([("cost", "<i4"), ("mvx", "<i2"), ("mvy", "<i2")])
assert ME_RESULT.itemsize == 8

COST_REQ = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"),
    ("pu_x", "<i2"), ("pu_y", "<i2"),
    ("pu_w", "u1"), ("pu_h", "u1"), ("metric", "u1"), ("pad", "u1"),
    ("qmv_x", "<i2"), ("qmv_y", "<i2"),
])
assert COST_REQ.itemsize == 20

BIDIR_REQ = np.dtype([
    ("fenc_plane", "<i4"), ("ref0_plane", "<i4"), ("ref1_plane", "<i4"),
    ("pu_x", "<i2"), ("pu_y", "<i2"),
    ("pu_w", "u1"), ("pu_h", "u1"), ("metric", "u1"), ("pad", "u1"),
    ("qmv0_x", "<i2"), ("qmv0_y", "<i2"), ("qmv1_x", "<i2"), ("qmv1_y", "<i2"),
])
assert BIDIR_REQ.itemsize == 28

# x265_hip_pred_req: motion-compensated luma prediction into dst_plane (uni-directional when ref1_plane < 0;
# ref1_plane == PRED_WEIGHTED: weighted uni-directional, w0 in qmv1_x, offset in qmv1_y, log2_denom;
# weighted_bi != 0 with ref1_plane >= 0: weighted bi-prediction with wp_w0, wp_o0, wp_w1, wp_o1, log2_denom)
PRED_WEIGHTED = -2
PRED_REQ = np.dtype([
    ("dst_plane", "<i4"), ("ref0_plane", "<i4"), ("ref1_plane", "<i4"),
    ("pu_x", "<i2"), ("pu_y", "<i2"),
    ("pu_w", "u1"), ("pu_h", "u1"), ("log2_denom", "u1"), ("weighted_bi", "u1"),
    ("qmv0_x", "<i2"), ("qmv0_y", "<i2"), ("qmv1_x", "<i2"), ("qmv1_y", "<i2"),
    ("wp_w0", "<i2"), ("wp_o0", "<i2"), ("wp_w1", "<i2"), ("wp_o1", "<i2"),
])
assert PRED_REQ.itemsize == 36

# x265_hip_refine_req: one sub-pel ring around `qmv` (costs of centre + square1[i] * step); nine costs per request
RING_QPEL, RING_HPEL, RING_8, RING_CENTRE, RING_COSTS = 0x01, 0x02, 0x10, 0x20, 9
SQUARE1 = [(0, 0), (0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1)]  # motion.cpp:71-73
REFINE_REQ = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"),
    ("pu_x", "<i2"), ("pu_y", "<i2"),
    ("pu_w", "u1"), ("pu_h", "u1"), ("metric", "u1"), ("pattern", "u1"),
    ("qmv_x", "<i2"), ("qmv_y", "<i2"),
])
assert REFINE_REQ.itemsize == 20


def refine_as_cost_reqs(reqs):
    """the candidates of a list of refine requests as plain cost requests (what a ring is defined by): a
    (len(reqs), 9) array of cost requests and the mask of the entries each request's pattern asks for"""
    reqs = np.asarray(reqs, REFINE_REQ)
    n = len(reqs)
    out = np.zeros((n, RING_COSTS), COST_REQ)
    step = np.where(reqs["pattern"] & RING_HPEL, 2, 1).astype(np.int64)
    for f in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "metric"):
        out[f] = reqs[f][:, None]
    sq = np.array(SQUARE1, np.int64)
    out["qmv_x"] = (reqs["qmv_x"].astype(np.int64)[:, None] + sq[None, :, 0] * step[:, None]).astype(np.int16)
    out["qmv_y"] = (reqs["qmv_y"].astype(np.int64)[:, None] + sq[None, :, 1] * step[:, None]).astype(np.int16)
    k = np.arange(RING_COSTS)[None, :]
    ndirs = np.where(reqs["pattern"] & RING_8, 8, 4)[:, None]
    used = np.where(k == 0, (reqs["pattern"] & RING_CENTRE)[:, None] != 0, k <= ndirs)
    return out, used


# x265_hip_inter_job / x265_hip_inter_result: one (list, reference) iteration of predInterSearch
INTER_JOB = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"), ("pred_plane", "<i4"),
    ("pu_x", "<i2"), ("pu_y", "<i2"),
    ("pu_w", "u1"), ("pu_h", "u1"), ("method", "u1"), ("subme", "u1"),
    ("cu_x", "<i2"), ("cu_y", "<i2"),
    ("amvp", "<i2", (4,)),
    ("merange", "<i2"), ("ref_lag", "<i2"),
    ("num_amvp", "u1"), ("qp", "u1"), ("num_mvc", "u1"), ("pad", "u1"),
    ("mvc", "<i2", (2 * MAX_MVC,)),
])
assert INTER_JOB.itemsize == 88
INTER_RESULT = np.dtype([
    ("cost", "<i4"), ("mvx", "<i2"), ("mvy", "<i2"),
    ("amvp_sad", "<i4", (2,)),
    ("mvmin_x", "<i2"), ("mvmin_y", "<i2"), ("mvmax_x", "<i2"), ("mvmax_y", "<i2"),
    ("mvp_x", "<i2"), ("mvp_y", "<i2"),
    ("mvp_idx", "u1"), ("pad", "u1", (3,)),
])
assert INTER_RESULT.itemsize == 32
# x265_hip_inter_pu / x265_hip_inter_side / x265_hip_inter_pu_result: predInterSearch behind its iterations
INTER_PU = np.dtype([("first_job", "<i4"), ("num_ref", "u1", (2,)), ("list_sel_bits", "u1", (3,)), ("no_bidir", "u1"), ("pad", "u1", (2,)),
                     ("lambda_sad", "<u4")])
assert INTER_PU.itemsize == 16
INTER_SIDE = np.dtype([("cost", "<u4"), ("bits", "<i4"), ("mvx", "<i2"), ("mvy", "<i2"), ("mvp_x", "<i2"), ("mvp_y", "<i2"), ("ref", "i1"),
                       ("mvp_idx", "i1"), ("pad", "u1", (2,))])
assert INTER_SIDE.itemsize == 20
INTER_PU_RESULT = np.dtype([("list", INTER_SIDE, (2,)), ("bidir", INTER_SIDE, (2,)), ("bidir_cost", "<u4"), ("bidir_bits", "<i4"),
                            ("inter_dir", "u1"), ("zero_tried", "u1"), ("pending", "u1"), ("pad", "u1")])
assert INTER_PU_RESULT.itemsize == 92

# x265_hip_merge_pu / x265_hip_merge_result: TEncSearch::xMergeEstimation of a prediction unit
MRG_MAX_NUM_CANDS = 5
MERGE_CAND = np.dtype([("ref_plane", "<i4", (2,)), ("mv", "<i2", (4,)), ("wp_w", "<i2", (2,)), ("wp_o", "<i2", (2,)),
                       ("log2_denom", "u1", (2,)), ("pad", "u1", (2,))])
assert MERGE_CAND.itemsize == 28
MERGE_PU = np.dtype([("fenc_plane", "<i4"), ("pu_x", "<i2"), ("pu_y", "<i2"), ("pu_w", "u1"), ("pu_h", "u1"), ("num_cand", "u1"),
                     ("weighted", "u1"), ("cu_x", "<i2"), ("cu_y", "<i2"), ("mvy_limit", "<i2"), ("pad", "<u2"), ("lambda_sad", "<u4"),
                     ("cand", MERGE_CAND, (MRG_MAX_NUM_CANDS,))])
assert MERGE_PU.itemsize == 164
MERGE_RESULT = np.dtype([("cost", "<u4"), ("bits", "<u4"), ("index", "u1"), ("inter_dir", "u1"), ("pad", "u1", (2,)),
                         ("satd", "<i4", (MRG_MAX_NUM_CANDS,))])
assert MERGE_RESULT.itemsize == 32

REFINE_SECTION = np.dtype([("count", "<i4"), ("pu_w", "u1"), ("pu_h", "u1"), ("pad", "u1", (2,))])
assert REFINE_SECTION.itemsize == 8
MAX_SECTIONS = 4

# x265_hip_lookahead_job: the motion search of one (picture, reference) pair of a lookahead frame cost estimate
LOOKAHEAD_JOB = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"), ("merange", "<i2"),
    ("method", "u1"), ("subme", "u1"), ("qp", "u1"), ("pad", "u1", (3,)),
])
assert LOOKAHEAD_JOB.itemsize == 16
LOOKAHEAD_QP = 12  # X265_LOOKAHEAD_QP
