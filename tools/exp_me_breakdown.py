"""Where me_kernel's time goes: the 4K ME-stage list (3 references) with the sub-pel stage cut back step by step and
with the candidate vectors removed.  argv: 4k | 1080p"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from x265_amd import abi, lib, synth
from x265_amd.hipmem import DeviceBuffer

if os.environ.get("X265_EXP_LIB"):  # experiments: another build of the library
    lib.LIB_PATH = os.path.abspath(os.environ["X265_EXP_LIB"])

W, H = (3840, 2160) if (len(sys.argv) < 2 or sys.argv[1] == "4k") else (1920, 1080)
seq = synth.Sequence(W, H, seed=0x265)
ctx = lib.Context(0)
for i in range(4):
    p = synth.Plane(seq.frame(i))
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
for q in range(28, 40):
    ctx.set_qp(q)
refs = [2, 1, 0]
mot = [seq.global_motion(3, r) for r in refs]


def timed(jobs, reps=10):
    d, o = DeviceBuffer.from_array(jobs), DeviceBuffer(8 * len(jobs))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        ctx.me_batch_device(d.ptr, len(jobs), o.ptr)
        ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.me_batch_device(d.ptr, len(jobs), o.ptr)
    ctx.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    d.free()
    o.free()
    return ms


base = synth.me_jobs(W, H, 3, refs, mot, method=abi.HEX, subme=2, merange=57)
print(f"{len(base)} calls; sizes: " + ", ".join(f"{s}x{s} {int((base['pu_w'] == s).sum())}" for s in (64, 32, 16, 8)))
for name, f in (("HEX subme 2 (the bench list)", lambda j: j),
                ("subme 1", lambda j: j.__setitem__("subme", 1)),
                ("subme 0", lambda j: j.__setitem__("subme", 0)),
                ("subme 2, no candidate vectors", lambda j: j.__setitem__("num_mvc", 0)),
                ("subme 2, full-pel predictor and candidates", lambda j: (j.__setitem__("qmvp_x", j["qmvp_x"] & ~3), j.__setitem__("qmvp_y", j["qmvp_y"] & ~3),
                                                                        j.__setitem__("mvc", j["mvc"] & ~3))),
                ("DIA subme 2", lambda j: j.__setitem__("method", abi.DIA))):
    j = base.copy()
    f(j)
    ms = timed(j)
    line = f"{name:46s} {ms:7.3f} ms  {len(j) / ms * 1e3:9.3e} calls/s"
    for s in (64, 32, 16, 8):
        js = j[j["pu_w"] == s]
        line += f"   {s}x{s} {timed(js):6.3f}"
    print(line, flush=True)
ctx.close()
