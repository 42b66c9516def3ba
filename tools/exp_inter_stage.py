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
    """lists: [[reference planes of list 0], [of list 1]]"""
    refs = lists[0] + lists[1]
    mj = synth.me_jobs(W, H, 3, refs, [seq.global_motion(3, r) for r in refs], method=abi.HEX, subme=2, merange=57)
    ij, pu = synth.inter_units(mj, len(refs), H, n_list1=len(lists[1]))
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
