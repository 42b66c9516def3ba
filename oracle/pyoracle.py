"""ctypes front-end for the CPU checker libraries -- TEST INFRASTRUCTURE ONLY.

Two libraries export the same entry points under different prefixes:

* ``oracle/liboracle.so``        (prefix ``x265o_``): our plain-C restatement, travels to the GPU box.
* ``oracle/_ref/libx265ref.so``  (prefix ``x265r_``): the real reference C path built from
  /root/reference by ``oracle/Makefile`` (present only where it was built).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

MAX_MVC = 12

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


PRED_REQ = np.dtype([
    ("dst_plane", "<i4"), ("ref0_plane", "<i4"), ("ref1_plane", "<i4"),
    ("pu_x", "<i2"), ("pu_y", "<i2"),
    ("pu_w", "u1"), ("pu_h", "u1"), ("log2_denom", "u1"), ("weighted_bi", "u1"),
    ("qmv0_x", "<i2"), ("qmv0_y", "<i2"), ("qmv1_x", "<i2"), ("qmv1_y", "<i2"),
    ("wp_w0", "<i2"), ("wp_o0", "<i2"), ("wp_w1", "<i2"), ("wp_o1", "<i2"),
])
assert PRED_REQ.itemsize == 36

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
INTER_PU = np.dtype([("first_job", "<i4"), ("num_ref", "u1", (2,)), ("list_sel_bits", "u1", (3,)), ("no_bidir", "u1"), ("pad", "u1", (2,)),
                     ("lambda_sad", "<u4")])
assert INTER_PU.itemsize == 16
INTER_SIDE = np.dtype([("cost", "<u4"), ("bits", "<i4"), ("mvx", "<i2"), ("mvy", "<i2"), ("mvp_x", "<i2"), ("mvp_y", "<i2"), ("ref", "i1"),
                       ("mvp_idx", "i1"), ("pad", "u1", (2,))])
assert INTER_SIDE.itemsize == 20
INTER_PU_RESULT = np.dtype([("list", INTER_SIDE, (2,)), ("bidir", INTER_SIDE, (2,)), ("bidir_cost", "<u4"), ("bidir_bits", "<i4"),
                            ("inter_dir", "u1"), ("zero_tried", "u1"), ("pad", "u1", (2,))])
assert INTER_PU_RESULT.itemsize == 92


LOOKAHEAD_JOB = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"), ("merange", "<i2"),
    ("method", "u1"), ("subme", "u1"), ("qp", "u1"), ("pad", "u1", (3,)),
])
assert LOOKAHEAD_JOB.itemsize == 16


class _Plane(C.Structure):
    _fields_ = [("origin", C.c_void_p), ("stride", C.c_ssize_t)]


PARTS = [(4, 4), (8, 8), (8, 4), (4, 8), (16, 16), (16, 8), (8, 16), (16, 12), (12, 16), (16, 4), (4, 16),
         (32, 32), (32, 16), (16, 32), (32, 24), (24, 32), (32, 8), (8, 32), (64, 64), (64, 32), (32, 64),
         (64, 48), (48, 64), (64, 16), (16, 64)]


def _ptr(a, off=0):
    return C.c_void_p(a.ctypes.data + int(off))


class Checker:
    """One loaded checker library (oracle restatement or reference build)."""

    def __init__(self, path, prefix, kind):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.kind = kind
        self._tabs = {}

    def _f(self, name, restype=None):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        return fn

    # ---- primitives on raw numpy buffers (offsets in elements) -------------------------
    def sad(self, part, fenc, foff, fstride, fref, roff, rstride):
        return self._f("sad", C.c_int)(C.c_int(part), _ptr(fenc, foff), C.c_ssize_t(fstride), _ptr(fref, roff),
                                       C.c_ssize_t(rstride))

    def satd(self, part, fenc, foff, fstride, fref, roff, rstride):
        return self._f("satd", C.c_int)(C.c_int(part), _ptr(fenc, foff), C.c_ssize_t(fstride), _ptr(fref, roff),
                                        C.c_ssize_t(rstride))

    def sa8d_inter(self, part, fenc, foff, fstride, fref, roff, rstride):
        return self._f("sa8d_inter", C.c_int)(C.c_int(part), _ptr(fenc, foff), C.c_ssize_t(fstride), _ptr(fref, roff),
                                              C.c_ssize_t(rstride))

    def sse_pp(self, part, fenc, foff, fstride, fref, roff, rstride):
        return self._f("sse_pp", C.c_int)(C.c_int(part), _ptr(fenc, foff), C.c_ssize_t(fstride), _ptr(fref, roff),
                                          C.c_ssize_t(rstride))

    def sad_x3(self, part, fenc, foff, fref, roffs, rstride):
        res = np.zeros(3, np.int32)
        self._f("sad_x3")(C.c_int(part), _ptr(fenc, foff), _ptr(fref, roffs[0]), _ptr(fref, roffs[1]),
                          _ptr(fref, roffs[2]), C.c_ssize_t(rstride), _ptr(res))
        return res

    def sad_x4(self, part, fenc, foff, fref, roffs, rstride):
        res = np.zeros(4, np.int32)
        self._f("sad_x4")(C.c_int(part), _ptr(fenc, foff), _ptr(fref, roffs[0]), _ptr(fref, roffs[1]),
                          _ptr(fref, roffs[2]), _ptr(fref, roffs[3]), C.c_ssize_t(rstride), _ptr(res))
        return res

    def luma_hpp(self, part, src, soff, sstride, dst, doff, dstride, coeff):
        self._f("luma_hpp")(C.c_int(part), _ptr(src, soff), C.c_ssize_t(sstride), _ptr(dst, doff),
                            C.c_ssize_t(dstride), C.c_int(coeff))

    def luma_vpp(self, part, src, soff, sstride, dst, doff, dstride, coeff):
        self._f("luma_vpp")(C.c_int(part), _ptr(src, soff), C.c_ssize_t(sstride), _ptr(dst, doff),
                            C.c_ssize_t(dstride), C.c_int(coeff))

    def luma_hps(self, part, src, soff, sstride, dst16, doff, dstride, coeff, row_ext):
        self._f("luma_hps")(C.c_int(part), _ptr(src, soff), C.c_ssize_t(sstride), _ptr(dst16, 2 * doff),
                            C.c_ssize_t(dstride), C.c_int(coeff), C.c_int(row_ext))

    def luma_vsp(self, part, src16, soff, sstride, dst, doff, dstride, coeff):
        self._f("luma_vsp")(C.c_int(part), _ptr(src16, 2 * soff), C.c_ssize_t(sstride), _ptr(dst, doff),
                            C.c_ssize_t(dstride), C.c_int(coeff))

    def luma_copy_pp(self, part, dst, doff, dstride, src, soff, sstride):
        self._f("luma_copy_pp")(C.c_int(part), _ptr(dst, doff), C.c_ssize_t(dstride), _ptr(src, soff),
                                C.c_ssize_t(sstride))

    def pixelavg_pp(self, part, dst, doff, dstride, s0, o0, ss0, s1, o1, ss1):
        self._f("pixelavg_pp")(C.c_int(part), _ptr(dst, doff), C.c_ssize_t(dstride), _ptr(s0, o0), C.c_ssize_t(ss0),
                               _ptr(s1, o1), C.c_ssize_t(ss1))

    def weight_pp(self, src, soff, dst, doff, sstride, dstride, width, height, w0, rnd, shift, offset):
        self._f("weight_pp")(_ptr(src, soff), _ptr(dst, doff), C.c_ssize_t(sstride), C.c_ssize_t(dstride), C.c_int(width),
                             C.c_int(height), C.c_int(w0), C.c_int(rnd), C.c_int(shift), C.c_int(offset))

    def extend_row_border(self, buf, off, stride, width, height, margin_x):
        """extendCURowColBorder (ipfilter.cpp:56-74): extendPicBorder without rows above and below"""
        self._f("extend_pic_border")(_ptr(buf, off), C.c_ssize_t(stride), C.c_int(width), C.c_int(height), C.c_int(margin_x), C.c_int(0))

    def frame_init_lowres(self, src, soff, dsts, doff, sstride, dstride, width, height):
        self._f("frame_init_lowres")(_ptr(src, soff), _ptr(dsts[0], doff), _ptr(dsts[1], doff), _ptr(dsts[2], doff),
                                     _ptr(dsts[3], doff), C.c_ssize_t(sstride), C.c_ssize_t(dstride), C.c_int(width),
                                     C.c_int(height))

    def partition_from_sizes(self, w, h):
        return self._f("partition_from_sizes", C.c_int)(C.c_int(w), C.c_int(h))

    def bitcost_table(self, qp):
        if qp not in self._tabs:
            out = np.zeros(65536, np.uint16)
            self._f("bitcost_table")(C.c_int(qp), _ptr(out))
            self._tabs[qp] = out
        return self._tabs[qp]

    # ---- batched front-ends ------------------------------------------------------------
    @staticmethod
    def _planes(planes):
        """planes: list of (buffer ndarray uint8, origin_offset, stride)."""
        arr = (_Plane * len(planes))()
        for i, (buf, off, stride) in enumerate(planes):
            arr[i].origin = buf.ctypes.data + off
            arr[i].stride = stride
        return arr

    def me_batch(self, planes, costtabs, jobs, nthreads=1):
        """costtabs: dict qp -> uint16[65536] (centre +32768)."""
        jobs = np.ascontiguousarray(jobs, dtype=ME_JOB)
        out = np.zeros(len(jobs), ME_RESULT)
        tabs = (C.c_void_p * 256)()
        keep = []
        for qp, t in costtabs.items():
            t = np.ascontiguousarray(t, np.uint16)
            keep.append(t)
            tabs[qp] = t.ctypes.data
        self._f("me_batch")(self._planes(planes), tabs, _ptr(jobs), C.c_int(len(jobs)), _ptr(out), C.c_int(nthreads))
        return out

    def me_batch_ops(self, planes, costtabs, jobs, nthreads=1):
        """me_batch plus the work model of the searches (oracle only): dict of algorithmic lane-ops and candidate counts"""
        jobs = np.ascontiguousarray(jobs, dtype=ME_JOB)
        out = np.zeros(len(jobs), ME_RESULT)
        tabs = (C.c_void_p * 256)()
        keep = []
        for qp, t in costtabs.items():
            t = np.ascontiguousarray(t, np.uint16)
            keep.append(t)
            tabs[qp] = t.ctypes.data
        ops = np.zeros(5, np.float64)
        self._f("me_batch_ops")(self._planes(planes), tabs, _ptr(jobs), C.c_int(len(jobs)), _ptr(out), C.c_int(nthreads), _ptr(ops))
        return out, {"sad_ops": ops[0], "satd_ops": ops[1], "interp_ops": ops[2], "sad_candidates": ops[3],
                     "subpel_candidates": ops[4], "lane_ops": float(ops[:3].sum())}

    def inter_search_batch(self, planes, costtabs, jobs, pic_w, pic_h):
        """predInterSearch's (list, reference) iteration (oracle only): AMVP pick, search window, motionEstimate"""
        jobs = np.ascontiguousarray(jobs, dtype=INTER_JOB)
        out = np.zeros(len(jobs), INTER_RESULT)
        tabs = (C.c_void_p * 256)()
        keep = []
        for qp, t in costtabs.items():
            t = np.ascontiguousarray(t, np.uint16)
            keep.append(t)
            tabs[qp] = t.ctypes.data
        self._f("inter_search_batch")(self._planes(planes), tabs, _ptr(jobs), C.c_int(len(jobs)), C.c_int(pic_w), C.c_int(pic_h), _ptr(out))
        return out

    def mv_bits(self, x, y, px, py):
        """BitCost::bitcost of vector (x, y) against predictor (px, py)"""
        return int(self._f("mv_bits", C.c_int)(C.c_int(x), C.c_int(y), C.c_int(px), C.c_int(py)))

    def inter_decide_batch(self, planes, costtabs, jobs, results, pus, pic_w, pic_h):
        """predInterSearch behind its iterations (oracle only): bits, xCheckBestMVP, the bidirectional candidate, the choice"""
        jobs = np.ascontiguousarray(jobs, dtype=INTER_JOB)
        results = np.ascontiguousarray(results, dtype=INTER_RESULT)
        pus = np.ascontiguousarray(pus, dtype=INTER_PU)
        out = np.zeros(len(pus), INTER_PU_RESULT)
        tabs = (C.c_void_p * 256)()
        keep = []
        for qp, t in costtabs.items():
            t = np.ascontiguousarray(t, np.uint16)
            keep.append(t)
            tabs[qp] = t.ctypes.data
        self._f("inter_decide_batch")(self._planes(planes), tabs, _ptr(jobs), _ptr(results), _ptr(pus), C.c_int(len(pus)), C.c_int(pic_w),
                                      C.c_int(pic_h), _ptr(out))
        return out

    def lookahead_batch(self, planes, costtabs, jobs, width_in_cu, height_in_cu, nthreads=1):
        """whole-frame lookahead searches (oracle only): (n, height_in_cu, width_in_cu) results"""
        jobs = np.ascontiguousarray(jobs, dtype=LOOKAHEAD_JOB)
        out = np.zeros((len(jobs), height_in_cu, width_in_cu), ME_RESULT)
        tabs = (C.c_void_p * 256)()
        keep = []
        for qp, t in costtabs.items():
            t = np.ascontiguousarray(t, np.uint16)
            keep.append(t)
            tabs[qp] = t.ctypes.data
        self._f("lookahead_batch")(self._planes(planes), tabs, _ptr(jobs), C.c_int(len(jobs)), C.c_int(width_in_cu),
                                   C.c_int(height_in_cu), _ptr(out), C.c_int(nthreads))
        return out

    def cost_batch(self, planes, reqs, nthreads=1):
        reqs = np.ascontiguousarray(reqs, dtype=COST_REQ)
        out = np.zeros(len(reqs), np.int32)
        self._f("cost_batch")(self._planes(planes), _ptr(reqs), C.c_int(len(reqs)), _ptr(out), C.c_int(nthreads))
        return out

    def lowres_planes(self, src, src_origin, src_stride, width, height, margin_x=96, margin_y=80):
        """Lowres::create + Lowres::init restated on numpy buffers (common/lowres.cpp:30-69, 164-174):
        `width` x `height` is the full-resolution picture; returns four padded planes as
        (buffer, origin, stride) plus the rounded lowres (width, lines).  frame_init_lowres_core and
        extendPicBorder are the checker library's."""
        lw, ll = width // 2, height // 2
        stride = lw + 2 * margin_x
        if stride & 31:
            stride += 32 - (stride & 31)
        lw, ll = -(-lw // 8) * 8, -(-ll // 8) * 8
        origin = stride * margin_y + margin_x
        bufs = [np.zeros(stride * (ll + 2 * margin_y), np.uint8) for _ in range(4)]
        self._f("frame_init_lowres")(_ptr(src, src_origin), _ptr(bufs[0], origin), _ptr(bufs[1], origin),
                                     _ptr(bufs[2], origin), _ptr(bufs[3], origin), C.c_ssize_t(src_stride),
                                     C.c_ssize_t(stride), C.c_int(lw), C.c_int(ll))
        for b in bufs:
            self._f("extend_pic_border")(_ptr(b, origin), C.c_ssize_t(stride), C.c_int(lw), C.c_int(ll),
                                         C.c_int(margin_x), C.c_int(margin_y))
        return [(b, origin, stride) for b in bufs], (lw, ll)

    def weighted_plane(self, buf, origin, stride, width, height, w0, offset, log2denom, margin_x=96, margin_y=80):
        """MotionReference::init + applyWeight over the whole picture (encoder/reference.cpp:57-121): weight_pp
        on the padded width with the interpolation-depth correction, then the border extension.  Returns
        the weighted padded buffer (same geometry as the source)."""
        out = np.zeros_like(buf)
        rnd = (1 << (log2denom - 1)) if log2denom else 0
        padw = (width + 15) & ~15
        self._f("weight_pp")(_ptr(buf, origin), _ptr(out, origin), C.c_ssize_t(stride), C.c_ssize_t(stride),
                             C.c_int(padw), C.c_int(height), C.c_int(w0), C.c_int(rnd << 6), C.c_int(log2denom + 6),
                             C.c_int(offset))
        self._f("extend_pic_border")(_ptr(out, origin), C.c_ssize_t(stride), C.c_int(width), C.c_int(height),
                                     C.c_int(margin_x), C.c_int(margin_y))
        return out

    def bidir_cost_batch(self, planes, reqs):
        reqs = np.ascontiguousarray(reqs, dtype=BIDIR_REQ)
        out = np.zeros(len(reqs), np.int32)
        self._f("bidir_cost_batch")(self._planes(planes), _ptr(reqs), C.c_int(len(reqs)), _ptr(out))
        return out

    def predict_batch(self, planes, reqs):
        """writes the predictions into the buffers of the planes the requests name as destination"""
        reqs = np.ascontiguousarray(reqs, dtype=PRED_REQ)
        self._f("predict_batch")(self._planes(planes), _ptr(reqs), C.c_int(len(reqs)))

    def luma_p2s(self, src, soff, sstride, dst, doff, width, height):
        self._f("luma_p2s")(_ptr(src, soff), C.c_ssize_t(sstride), _ptr(dst, 2 * doff), C.c_int(width), C.c_int(height))

    def luma_vps(self, part, src, soff, sstride, dst, doff, dstride, coeff):
        self._f("luma_vps")(C.c_int(part), _ptr(src, soff), C.c_ssize_t(sstride), _ptr(dst, 2 * doff), C.c_ssize_t(dstride),
                            C.c_int(coeff))

    def luma_vss(self, part, src, soff, sstride, dst, doff, dstride, coeff):
        self._f("luma_vss")(C.c_int(part), _ptr(src, 2 * soff), C.c_ssize_t(sstride), _ptr(dst, 2 * doff), C.c_ssize_t(dstride),
                            C.c_int(coeff))

    def luma_hvpp(self, part, src, soff, sstride, dst, doff, dstride, idx_x, idx_y):
        self._f("luma_hvpp")(C.c_int(part), _ptr(src, soff), C.c_ssize_t(sstride), _ptr(dst, doff), C.c_ssize_t(dstride),
                             C.c_int(idx_x), C.c_int(idx_y))

    def luma_addavg(self, part, src0, o0, src1, o1, dst, doff, sstride0, sstride1, dstride):
        self._f("luma_addavg")(C.c_int(part), _ptr(src0, 2 * o0), _ptr(src1, 2 * o1), _ptr(dst, doff), C.c_ssize_t(sstride0),
                               C.c_ssize_t(sstride1), C.c_ssize_t(dstride))

    def mc_satd_batch(self, planes, reqs):
        """satd of the block of plane reqs.dst_plane (the source picture) against the prediction of each request"""
        reqs = np.ascontiguousarray(reqs, dtype=PRED_REQ)
        out = np.zeros(len(reqs), np.int32)
        self._f("mc_satd_batch")(self._planes(planes), _ptr(reqs), C.c_int(len(reqs)), _ptr(out))
        return out

    def merge_estimate_batch(self, planes, pus, pic_width, pic_height):
        pus = np.ascontiguousarray(pus, dtype=MERGE_PU)
        out = np.zeros(len(pus), MERGE_RESULT)
        self._f("merge_estimate_batch")(self._planes(planes), _ptr(pus), C.c_int(len(pus)), C.c_int(pic_width), C.c_int(pic_height), _ptr(out))
        return out

    def add_weight_bi(self, src0, src1, width, height, w0, o0, w1, o1, log2_denom):
        """xWeightedPredictionBi's luma block from the two lists' intermediates (64x64 int16 arrays): 64x64 uint8, zero
        outside width x height"""
        src0, src1 = np.ascontiguousarray(src0, np.int16), np.ascontiguousarray(src1, np.int16)
        dst = np.zeros((64, 64), np.uint8)
        self._f("add_weight_bi")(_ptr(src0), _ptr(src1), _ptr(dst), C.c_ssize_t(64), C.c_ssize_t(64), C.c_ssize_t(64), C.c_int(width),
                                 C.c_int(height), C.c_int(w0), C.c_int(o0), C.c_int(w1), C.c_int(o1), C.c_int(log2_denom))
        return dst

    def ctu_fullsearch(self, fenc, ref, costtab, ctu_x, ctu_y, mvmin, mvmax, qmvp):
        out = np.zeros(85, ME_RESULT)
        pl = self._planes([fenc, ref])
        costtab = np.ascontiguousarray(costtab, np.uint16)
        self._f("ctu_fullsearch")(C.byref(pl[0]), C.byref(pl[1]), _ptr(costtab), C.c_int(ctu_x), C.c_int(ctu_y),
                                  C.c_int(mvmin[0]), C.c_int(mvmin[1]), C.c_int(mvmax[0]), C.c_int(mvmax[1]),
                                  C.c_int(qmvp[0]), C.c_int(qmvp[1]), _ptr(out))
        return out


def oracle_path():
    return os.path.join(HERE, "liboracle.so")


def ref_path():
    return os.path.join(HERE, "_ref", "libx265ref.so")


def load_oracle():
    return Checker(oracle_path(), "x265o_", "port")


def have_ref():
    return os.path.exists(ref_path())


def load_ref():
    return Checker(ref_path(), "x265r_", "reference")
