"""ctypes binding of libx265_hip.so (the C ABI in include/x265_hip.h).

Host-side plumbing only: arrays go in and come out as numpy; device-resident entry points take raw
device addresses (e.g. ``torch.Tensor.data_ptr()``).  There is no fallback: if the shared library is
missing or no gfx950 device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

from .abi import BIDIR_REQ, COST_REQ, INTER_JOB, INTER_RESULT, LOOKAHEAD_JOB, ME_JOB, ME_RESULT, NUM_PARTS, PRED_REQ, REFINE_REQ, RING_COSTS

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libx265_hip.so")

CTU_JOB = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"),
    ("ctu_x", "<i2"), ("ctu_y", "<i2"),
    ("mvmin_x", "<i2"), ("mvmin_y", "<i2"), ("mvmax_x", "<i2"), ("mvmax_y", "<i2"),
    ("qmvp_x", "<i2"), ("qmvp_y", "<i2"),
    ("qp", "u1"), ("pad", "u1", (3,)),
])
assert CTU_JOB.itemsize == 28
PUS_PER_CTU = 85

EXPORTS = [
    "x265_hip_device_count", "x265_hip_ctx_create", "x265_hip_ctx_destroy", "x265_hip_last_error",
    "x265_hip_ctx_stream", "x265_hip_ctx_sync", "x265_hip_setup_primitives", "x265_hip_partition_from_sizes",
    "x265_hip_register_host_plane", "x265_hip_unregister_host_plane", "x265_hip_plane_upload",
    "x265_hip_plane_bind", "x265_hip_plane_release", "x265_hip_costtab_upload", "x265_hip_cost_batch",
    "x265_hip_me_batch", "x265_hip_me_batch_device", "x265_hip_cost_batch_device", "x265_hip_fullsearch",
    "x265_hip_fullsearch_device", "x265_hip_costtab_build", "x265_hip_costtab_set_qp",
    "x265_hip_plane_download", "x265_hip_lowres_geometry_of", "x265_hip_lowres_init",
    "x265_hip_plane_extend_border", "x265_hip_plane_weight",
    "x265_hip_bidir_cost_batch", "x265_hip_bidir_cost_batch_device",
    "x265_hip_dist_batch", "x265_hip_dist_batch_device", "x265_hip_predict_batch", "x265_hip_predict_batch_device",
    "x265_hip_cost_batch_device_noalloc", "x265_hip_refine_batch", "x265_hip_refine_batch_device",
    "x265_hip_inter_search_batch", "x265_hip_inter_search_batch_device",
    "x265_hip_lookahead_batch", "x265_hip_lookahead_batch_device", "x265_hip_lookahead_status", "x265_hip_ctx_check_records", "x265_hip_ctx_guard_status", "x265_hip_inter_decide_batch", "x265_hip_inter_decide_batch_device",
    "x265_hip_mc_satd_batch", "x265_hip_mc_satd_batch_device", "x265_hip_merge_estimate_batch", "x265_hip_merge_estimate_batch_device",
    "x265_hip_refine_sections_device",
    "x265_hip_host_alloc", "x265_hip_host_free", "x265_hip_plane_alloc", "x265_hip_plane_upload_async", "x265_hip_download_async",
    "x265_hip_copy_wait_compute", "x265_hip_compute_wait_copy", "x265_hip_copy_sync",
    "x265_hip_plane_extend_border_rows", "x265_hip_plane_weight_rows",
]


class X265HipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise X265HipError(f"{LIB_PATH} is missing: build it with `make -C x265_amd/csrc` "
                               "(or __graft_entry__.build()); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.x265_hip_last_error.restype = C.c_char_p
        _lib.x265_hip_ctx_stream.restype = C.c_void_p
    return _lib


def _check(rc):
    if rc != 0:
        raise X265HipError(f"x265_hip error {rc}: {load().x265_hip_last_error().decode()}")


def _p(a):
    return C.c_void_p(a.ctypes.data)


class Primitives(C.Structure):
    """x265_hip_primitives: the function-pointer table with the reference's signatures."""
    _cmp = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t)
    _x3 = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p)
    _x4 = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p)
    _pp = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int)
    _avg = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int)
    _wpp = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
    _down = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int)
    _ps = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int)
    _hv = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int)
    _p2s = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int, C.c_int)
    _aavg = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t)
    _hps = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int)
    _ext = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int)
    _fields_ = [
        ("sad", _cmp * NUM_PARTS), ("sad_x3", _x3 * NUM_PARTS), ("sad_x4", _x4 * NUM_PARTS),
        ("satd", _cmp * NUM_PARTS), ("luma_hpp", _pp * NUM_PARTS), ("luma_hps", _hps * NUM_PARTS),
        ("luma_vpp", _pp * NUM_PARTS), ("luma_vsp", _pp * NUM_PARTS),
        ("pixelavg_pp", _avg * NUM_PARTS), ("weight_pp", _wpp), ("frame_init_lowres_core", _down),
        ("sa8d_inter", _cmp * NUM_PARTS), ("sse_pp", _cmp * NUM_PARTS),
        ("luma_vps", _ps * NUM_PARTS), ("luma_vss", _ps * NUM_PARTS), ("luma_hvpp", _hv * NUM_PARTS), ("luma_p2s", _p2s),
        ("luma_addAvg", _aavg * NUM_PARTS), ("extendRowBorder", _ext),
    ]


def setup_primitives(device=0):
    t = Primitives()
    _check(load().x265_hip_setup_primitives(C.byref(t), C.c_int(device)))
    return t


def costtab_build(qp):
    """BitCost::setQP's table for qp (host computation inside the library)."""
    out = np.zeros(65536, np.uint16)
    _check(load().x265_hip_costtab_build(C.c_int(qp), _p(out)))
    return out


class LowresGeometry(C.Structure):
    _fields_ = [("width", C.c_int), ("lines", C.c_int), ("stride", C.c_int), ("origin_offset", C.c_int),
                ("bytes", C.c_size_t)]


def lowres_geometry(width, height, margin_x=96, margin_y=80):
    g = LowresGeometry()
    _check(load().x265_hip_lowres_geometry_of(C.c_int(width), C.c_int(height), C.c_int(margin_x), C.c_int(margin_y),
                                              C.byref(g)))
    return g


def device_count():
    return load().x265_hip_device_count()


class PinnedBuffer:
    """page-locked host memory from x265_hip_host_alloc, as a numpy view (`array`, uint8)"""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(load().x265_hip_host_alloc(C.c_size_t(int(nbytes)), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr))

    def free(self):
        if self.ptr:
            self.array = None
            load().x265_hip_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """x265_hip_ctx: one device, one stream, planes and cost tables resident in HBM."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(load().x265_hip_ctx_create(C.c_int(device), C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            load().x265_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return load().x265_hip_ctx_stream(self._h)

    def sync(self):
        _check(load().x265_hip_ctx_sync(self._h))

    def plane_upload(self, plane_id, buf, stride, origin):
        buf = np.ascontiguousarray(buf, np.uint8)
        _check(load().x265_hip_plane_upload(self._h, C.c_int(plane_id), _p(buf), C.c_size_t(buf.size), C.c_int(stride),
                                            C.c_int(origin)))

    def plane_bind(self, plane_id, dev_ptr, nbytes, stride, origin):
        _check(load().x265_hip_plane_bind(self._h, C.c_int(plane_id), C.c_void_p(dev_ptr), C.c_size_t(nbytes),
                                          C.c_int(stride), C.c_int(origin)))

    def plane_release(self, plane_id):
        _check(load().x265_hip_plane_release(self._h, C.c_int(plane_id)))

    # asynchronous transfers on the context's copy stream (include/x265_hip.h)
    def plane_alloc(self, plane_id, nbytes, stride, origin):
        _check(load().x265_hip_plane_alloc(self._h, C.c_int(plane_id), C.c_size_t(nbytes), C.c_int(stride), C.c_int(origin)))

    def plane_upload_async(self, plane_id, pinned):
        """`pinned`: a PinnedBuffer (or any object with .ptr / .nbytes) holding the whole plane"""
        _check(load().x265_hip_plane_upload_async(self._h, C.c_int(plane_id), C.c_void_p(pinned.ptr), C.c_size_t(pinned.nbytes)))

    def download_async(self, d_src, pinned, nbytes=None):
        _check(load().x265_hip_download_async(self._h, C.c_void_p(d_src), C.c_void_p(pinned.ptr),
                                              C.c_size_t(pinned.nbytes if nbytes is None else nbytes)))

    def copy_wait_compute(self):
        _check(load().x265_hip_copy_wait_compute(self._h))

    def compute_wait_copy(self):
        _check(load().x265_hip_compute_wait_copy(self._h))

    def copy_sync(self):
        _check(load().x265_hip_copy_sync(self._h))

    def plane_download(self, plane_id, nbytes):
        out = np.zeros(nbytes, np.uint8)
        _check(load().x265_hip_plane_download(self._h, C.c_int(plane_id), _p(out), C.c_size_t(nbytes)))
        return out

    def lowres_init(self, src_plane, width, height, dst_plane0, margin_x=96, margin_y=80):
        """Lowres planes dst_plane0 .. +3 from full-resolution plane src_plane (asynchronous)."""
        _check(load().x265_hip_lowres_init(self._h, C.c_int(src_plane), C.c_int(width), C.c_int(height),
                                           C.c_int(margin_x), C.c_int(margin_y), C.c_int(dst_plane0)))
        return lowres_geometry(width, height, margin_x, margin_y)

    def plane_extend_border(self, plane_id, width, height, margin_x=96, margin_y=80):
        _check(load().x265_hip_plane_extend_border(self._h, C.c_int(plane_id), C.c_int(width), C.c_int(height),
                                                   C.c_int(margin_x), C.c_int(margin_y)))

    def plane_weight(self, src_plane, dst_plane, width, height, w0, offset, log2_denom, margin_x=96, margin_y=80):
        _check(load().x265_hip_plane_weight(self._h, C.c_int(src_plane), C.c_int(dst_plane), C.c_int(width),
                                            C.c_int(height), C.c_int(margin_x), C.c_int(margin_y), C.c_int(w0),
                                            C.c_int(offset), C.c_int(log2_denom)))

    def plane_extend_border_rows(self, plane_id, width, height, row0, rows, margin_x=96, margin_y=80):
        _check(load().x265_hip_plane_extend_border_rows(self._h, C.c_int(plane_id), C.c_int(width), C.c_int(height),
                                                        C.c_int(margin_x), C.c_int(margin_y), C.c_int(row0), C.c_int(rows)))

    def plane_weight_rows(self, src_plane, dst_plane, width, height, w0, offset, log2_denom, row0, rows, margin_x=96, margin_y=80):
        _check(load().x265_hip_plane_weight_rows(self._h, C.c_int(src_plane), C.c_int(dst_plane), C.c_int(width),
                                                 C.c_int(height), C.c_int(margin_x), C.c_int(margin_y), C.c_int(w0),
                                                 C.c_int(offset), C.c_int(log2_denom), C.c_int(row0), C.c_int(rows)))

    def costtab_upload(self, qp, tab):
        tab = np.ascontiguousarray(tab, np.uint16)
        assert tab.size == 65536
        _check(load().x265_hip_costtab_upload(self._h, C.c_int(qp), _p(tab)))

    def set_qp(self, qp):
        _check(load().x265_hip_costtab_set_qp(self._h, C.c_int(qp)))

    def cost_batch(self, reqs):
        reqs = np.ascontiguousarray(reqs, COST_REQ)
        out = np.zeros(len(reqs), np.int32)
        _check(load().x265_hip_cost_batch(self._h, _p(reqs), C.c_int(len(reqs)), _p(out)))
        return out

    def inter_search_batch(self, jobs, pic_width, pic_height):
        jobs = np.ascontiguousarray(jobs, INTER_JOB)
        out = np.zeros(len(jobs), INTER_RESULT)
        _check(load().x265_hip_inter_search_batch(self._h, _p(jobs), C.c_int(len(jobs)), C.c_int(pic_width), C.c_int(pic_height), _p(out)))
        return out

    def inter_decide_batch(self, jobs, pus, pic_width, pic_height):
        """(iteration results, unit results): x265_hip_inter_search_batch for the jobs, then bits / xCheckBestMVP / the
        bidirectional candidate / the choice for every prediction unit"""
        from .abi import INTER_PU, INTER_PU_RESULT
        jobs, pus = np.ascontiguousarray(jobs, INTER_JOB), np.ascontiguousarray(pus, INTER_PU)
        res, out = np.zeros(len(jobs), INTER_RESULT), np.zeros(len(pus), INTER_PU_RESULT)
        _check(load().x265_hip_inter_decide_batch(self._h, _p(jobs), C.c_int(len(jobs)), _p(pus), C.c_int(len(pus)), C.c_int(pic_width),
                                                  C.c_int(pic_height), _p(res), _p(out)))
        return res, out

    def inter_decide_batch_device(self, d_jobs, njobs, d_pus, npu, pic_width, pic_height, d_job_results, d_results, any_bidir=True):
        _check(load().x265_hip_inter_decide_batch_device(self._h, C.c_void_p(d_jobs), C.c_int(njobs), C.c_void_p(d_pus), C.c_int(npu),
                                                         C.c_int(pic_width), C.c_int(pic_height), C.c_void_p(d_job_results),
                                                         C.c_void_p(d_results), C.c_int(1 if any_bidir else 0)))

    def mc_satd_batch(self, reqs):
        """satd of the block of plane reqs.dst_plane (the source picture) against the prediction of each request"""
        reqs = np.ascontiguousarray(reqs, PRED_REQ)
        out = np.zeros(len(reqs), np.int32)
        _check(load().x265_hip_mc_satd_batch(self._h, _p(reqs), C.c_int(len(reqs)), _p(out)))
        return out

    def mc_satd_batch_device(self, d_reqs, n, d_costs):
        _check(load().x265_hip_mc_satd_batch_device(self._h, C.c_void_p(d_reqs), C.c_int(n), C.c_void_p(d_costs)))

    def merge_estimate_batch(self, pus, pic_width, pic_height):
        """TEncSearch::xMergeEstimation of every prediction unit"""
        from .abi import MERGE_PU, MERGE_RESULT
        pus = np.ascontiguousarray(pus, MERGE_PU)
        out = np.zeros(len(pus), MERGE_RESULT)
        _check(load().x265_hip_merge_estimate_batch(self._h, _p(pus), C.c_int(len(pus)), C.c_int(pic_width), C.c_int(pic_height), _p(out)))
        return out

    def merge_estimate_batch_device(self, d_pus, n, pic_width, pic_height, d_results):
        _check(load().x265_hip_merge_estimate_batch_device(self._h, C.c_void_p(d_pus), C.c_int(n), C.c_int(pic_width), C.c_int(pic_height),
                                                           C.c_void_p(d_results)))

    def refine_sections_device(self, d_reqs, sections, d_costs9):
        """sections: [(count, pu_w, pu_h), ...] (at most abi.MAX_SECTIONS runs of one PU size each, in list order)"""
        from .abi import REFINE_SECTION
        sec = np.zeros(len(sections), REFINE_SECTION)
        for i, (cnt, w, h) in enumerate(sections):
            sec[i]["count"], sec[i]["pu_w"], sec[i]["pu_h"] = cnt, w, h
        _check(load().x265_hip_refine_sections_device(self._h, C.c_void_p(d_reqs), _p(sec), C.c_int(len(sec)), C.c_void_p(d_costs9)))

    def check_records(self, on=True):
        """record-check mode of the *_device entry points (debugging aid): False / 0 off, True / 1 copy the records back
        and check them on the host, 2 the guard on the device (x265_hip_me_batch_device)"""
        _check(load().x265_hip_ctx_check_records(self._h, C.c_int(int(on))))

    def guard_status(self):
        """(refused records since the last look, lowest index of one in its list): waits for the stream"""
        n, first = C.c_uint32(0), C.c_uint32(0)
        _check(load().x265_hip_ctx_guard_status(self._h, C.byref(n), C.byref(first)))
        return n.value, first.value

    def lookahead_status(self):
        """status word of the last lookahead launch (0: every row finished); waits for the stream"""
        v = C.c_int(0)
        _check(load().x265_hip_lookahead_status(self._h, C.byref(v)))
        return v.value

    def lookahead_batch(self, jobs, width_in_cu, height_in_cu):
        """(n, height_in_cu, width_in_cu) results: the whole-frame searches of the lookahead"""
        jobs = np.ascontiguousarray(jobs, LOOKAHEAD_JOB)
        out = np.zeros((len(jobs), height_in_cu, width_in_cu), ME_RESULT)
        _check(load().x265_hip_lookahead_batch(self._h, _p(jobs), C.c_int(len(jobs)), C.c_int(width_in_cu), C.c_int(height_in_cu), _p(out)))
        return out

    def lookahead_batch_device(self, d_jobs, n, width_in_cu, height_in_cu, d_results):
        _check(load().x265_hip_lookahead_batch_device(self._h, C.c_void_p(d_jobs), C.c_int(n), C.c_int(width_in_cu),
                                                      C.c_int(height_in_cu), C.c_void_p(d_results)))

    def inter_search_batch_device(self, d_jobs, n, pic_width, pic_height, d_results):
        _check(load().x265_hip_inter_search_batch_device(self._h, C.c_void_p(d_jobs), C.c_int(n), C.c_int(pic_width),
                                                         C.c_int(pic_height), C.c_void_p(d_results)))

    def refine_batch(self, reqs):
        reqs = np.ascontiguousarray(reqs, REFINE_REQ)
        out = np.zeros((len(reqs), RING_COSTS), np.int32)
        _check(load().x265_hip_refine_batch(self._h, _p(reqs), C.c_int(len(reqs)), _p(out)))
        return out

    def refine_batch_device(self, d_reqs, n, d_costs9):
        _check(load().x265_hip_refine_batch_device(self._h, C.c_void_p(d_reqs), C.c_int(n), C.c_void_p(d_costs9)))

    def dist_batch(self, reqs):
        reqs = np.ascontiguousarray(reqs, COST_REQ)
        out = np.zeros(len(reqs), np.int32)
        _check(load().x265_hip_dist_batch(self._h, _p(reqs), C.c_int(len(reqs)), _p(out)))
        return out

    def dist_batch_device(self, d_reqs, n, d_costs):
        _check(load().x265_hip_dist_batch_device(self._h, C.c_void_p(d_reqs), C.c_int(n), C.c_void_p(d_costs)))

    def predict_batch(self, reqs):
        reqs = np.ascontiguousarray(reqs, PRED_REQ)
        _check(load().x265_hip_predict_batch(self._h, _p(reqs), C.c_int(len(reqs))))

    def predict_batch_device(self, d_reqs, n):
        _check(load().x265_hip_predict_batch_device(self._h, C.c_void_p(d_reqs), C.c_int(n)))

    def bidir_cost_batch(self, reqs):
        reqs = np.ascontiguousarray(reqs, BIDIR_REQ)
        out = np.zeros(len(reqs), np.int32)
        _check(load().x265_hip_bidir_cost_batch(self._h, _p(reqs), C.c_int(len(reqs)), _p(out)))
        return out

    def bidir_cost_batch_device(self, d_reqs, n, d_costs):
        _check(load().x265_hip_bidir_cost_batch_device(self._h, C.c_void_p(d_reqs), C.c_int(n), C.c_void_p(d_costs)))

    def me_batch(self, jobs):
        jobs = np.ascontiguousarray(jobs, ME_JOB)
        out = np.zeros(len(jobs), ME_RESULT)
        _check(load().x265_hip_me_batch(self._h, _p(jobs), C.c_int(len(jobs)), _p(out)))
        return out

    def fullsearch(self, jobs):
        jobs = np.ascontiguousarray(jobs, CTU_JOB)
        out = np.zeros(len(jobs) * PUS_PER_CTU, ME_RESULT)
        _check(load().x265_hip_fullsearch(self._h, _p(jobs), C.c_int(len(jobs)), _p(out)))
        return out.reshape(len(jobs), PUS_PER_CTU)

    # device-resident variants: raw device addresses, asynchronous on self.stream
    def me_batch_device(self, d_jobs, n, d_results):
        _check(load().x265_hip_me_batch_device(self._h, C.c_void_p(d_jobs), C.c_int(n), C.c_void_p(d_results)))

    def cost_batch_device(self, d_reqs, n, d_costs):
        _check(load().x265_hip_cost_batch_device(self._h, C.c_void_p(d_reqs), C.c_int(n), C.c_void_p(d_costs)))

    def cost_batch_device_noalloc(self, d_reqs, n, d_costs):
        _check(load().x265_hip_cost_batch_device_noalloc(self._h, C.c_void_p(d_reqs), C.c_int(n), C.c_void_p(d_costs)))

    def fullsearch_device(self, d_jobs, n, d_results):
        _check(load().x265_hip_fullsearch_device(self._h, C.c_void_p(d_jobs), C.c_int(n), C.c_void_p(d_results)))
