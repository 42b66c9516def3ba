"""predInterSearch as a stage on one 4K picture: every square PU as a prediction unit with its (list, reference)
iterations -- a P picture (three references in list 0) and a B picture (two references in list 0, one in list 1, the
bidirectional candidate tried) -- through x265_hip_inter_decide_batch_device, beside x265_hip_me_batch_device on the same
searches with the predictor and window given (what the stage adds: AMVP pick, window, bits, xCheckBestMVP, bidir)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from x265_amd import abi, lib, synth
from x265_amd.hipmem import DeviceBuffer

W, H = (3840, 2160) if (len(sys.argv) < 2 or sys.argv[1] == "4k") else (1920, 1080)
seq = synth.Sequence(W, H, seed=0x265)
ctx = lib.Context(0)
for i in range(4):
    p = synth.Plane(seq.frame(i))
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
for q in range(28, 40):
    ctx.set_qp(q)


def timed(fn, reps=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        fn()
        ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def units(lists):
    """lists: [[reference planes of list 0], [of list 1]]; jobs unit-major"""
    refs = lists[0] + lists[1]
    mj = synth.me_jobs(W, H, 3, refs, [seq.global_motion(3, r) for r in refs], method=abi.HEX, subme=2, merange=57)
    n = len(mj) // len(refs)
    mj = mj.reshape(len(refs), n).T.reshape(-1)  # unit-major: a unit's iterations are consecutive
    ij = np.zeros(len(mj), abi.INTER_JOB)
    for k in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "method", "subme", "merange", "qp", "num_mvc", "mvc"):
        ij[k] = mj[k]
    ij["pred_plane"] = mj["ref_plane"]
    ij["cu_x"], ij["cu_y"] = mj["pu_x"], mj["pu_y"]  # 2Nx2N units: the CU is the block
    ij["amvp"][:, 0], ij["amvp"][:, 1] = mj["qmvp_x"], mj["qmvp_y"]
    ij["amvp"][:, 2], ij["amvp"][:, 3] = mj["qmvp_x"] + 3, mj["qmvp_y"] - 2
    ij["num_amvp"], ij["ref_lag"] = 2, H
    pu = np.zeros(n, abi.INTER_PU)
    pu["first_job"] = np.arange(n) * len(refs)
    pu["num_ref"] = (len(lists[0]), len(lists[1]))
    pu["list_sel_bits"] = (3, 3, 5) if lists[1] else (1, 3, 5)
    pu["no_bidir"] = 0 if lists[1] else 1
    pu["lambda_sad"] = 4771
    return mj, ij, pu


for name, lists in (("P picture, 3 references", [[2, 1, 0], []]), ("B picture, 2 + 1 references", [[2, 1], [0]])):
    mj, ij, pu = units(lists)
    d_m, o_m = DeviceBuffer.from_array(mj), DeviceBuffer(8 * len(mj))
    d_j, d_p = DeviceBuffer.from_array(ij), DeviceBuffer.from_array(pu)
    d_r, d_o = DeviceBuffer(len(ij) * abi.INTER_RESULT.itemsize), DeviceBuffer(len(pu) * abi.INTER_PU_RESULT.itemsize)
    t_me = timed(lambda: ctx.me_batch_device(d_m.ptr, len(mj), o_m.ptr))
    t_is = timed(lambda: ctx.inter_search_batch_device(d_j.ptr, len(ij), W, H, d_r.ptr))
    t_st = timed(lambda: ctx.inter_decide_batch_device(d_j.ptr, len(ij), d_p.ptr, len(pu), W, H, d_r.ptr, d_o.ptr, any_bidir=bool(lists[1])))
    out = d_o.download(abi.INTER_PU_RESULT)
    print(f"{name}: {len(pu)} units, {len(ij)} iterations: me_batch {t_me:.3f} ms, inter_search {t_is:.3f} ms, inter_decide {t_st:.3f} ms "
          f"= {len(pu) / t_st * 1e3:.3e} units/s; directions {np.bincount(out['inter_dir'], minlength=4)[1:]}, zero form tried {int(out['zero_tried'].sum())}", flush=True)
ctx.close()
