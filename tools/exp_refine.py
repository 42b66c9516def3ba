"""Experiment: the sub-pel SATD candidate set of a picture through x265_hip_cost_batch (one request per candidate)
and through x265_hip_refine_batch (one request per ring, interpolation shared in LDS).  argv: 4k | 1080p"""
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
planes = [synth.Plane(seq.frame(i)) for i in range(4)]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
refs = [2, 1, 0]
mot = [seq.global_motion(3, r) for r in refs]


def timed(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn()
        ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def run(label, crq, rrq, check=True):
    d_c, d_r = DeviceBuffer.from_array(crq), DeviceBuffer.from_array(rrq)
    o_c, o_r = DeviceBuffer(4 * len(crq)), DeviceBuffer(36 * len(rrq))
    ms_c = timed(lambda: ctx.cost_batch_device(d_c.ptr, len(crq), o_c.ptr))
    ms_r = timed(lambda: ctx.refine_batch_device(d_r.ptr, len(rrq), o_r.ptr))
    ops = synth.satd_lane_ops(crq)
    ncand = len(crq)
    print(f"{label:12s} {ncand:9d} candidates  cost_batch {ms_c:7.3f} ms {ncand / ms_c * 1e3:9.3e}/s   refine_batch {ms_r:7.3f} ms "
          f"{ncand / ms_r * 1e3:9.3e}/s  x{ms_c / ms_r:.2f}   algorithmic lane-ops/s {ops / ms_r * 1e3:.3e} "
          f"({ops / ms_r * 1e3 / 7.86e13:.2f} of 7.86e13)", flush=True)
    if not check:
        return
    # same numbers: candidate (ref, ring offset, pu) of the cost list = entry (ref, ring, pu)[k] of the ring list
    got = o_r.download(np.int32).reshape(-1, 9)
    want = o_c.download(np.int32)
    npu = len(rrq) // (2 * len(refs))
    g = got.reshape(len(refs), 2, npu, 9)[:, :, :, 1:]          # ref, ring, pu, k
    wnt = want.reshape(len(refs), 2, 8, npu).transpose(0, 1, 3, 2)  # ref, ring, k, pu -> ref, ring, pu, k
    assert np.array_equal(g, wnt), "refine_batch differs from cost_batch"


crq = synth.subpel_requests(W, H, 3, refs, mot)
rrq = synth.refine_requests(W, H, 3, refs, mot)
run("all sizes", crq, rrq)
o = np.argsort(-rrq["pu_w"].astype(int), kind="stable")
run("sorted", crq, rrq[o], check=False)
per_size = 0.0
for s in (64, 32, 16, 8):
    sel = rrq[rrq["pu_w"] == s]
    run(f"{s}x{s}", crq[crq["pu_w"] == s], sel)
    d_r, o_r = DeviceBuffer.from_array(sel), DeviceBuffer(36 * len(sel))
    per_size += timed(lambda: ctx.refine_batch_device(d_r.ptr, len(sel), o_r.ptr))
# the four sizes as one list of four runs, one launch (x265_hip_refine_sections_device)
srt = rrq[o]
d_r, o_r = DeviceBuffer.from_array(srt), DeviceBuffer(36 * len(srt))
secs = [(int((srt["pu_w"] == s).sum()), s, s) for s in (64, 32, 16, 8)]
ms = timed(lambda: ctx.refine_sections_device(d_r.ptr, secs, o_r.ptr))
ops = synth.satd_lane_ops(crq)
print(f"one launch per size: {per_size:.3f} ms   four runs in one launch: {ms:.3f} ms  {8 * len(srt) / ms * 1e3:9.3e}/s  "
      f"({ops / ms * 1e3 / 7.86e13:.3f} of 7.86e13)")
# same numbers as the single-run launches
one = DeviceBuffer(36 * len(srt))
ctx.refine_batch_device(d_r.ptr, len(srt), one.ptr)
ctx.sync()
assert np.array_equal(o_r.download(np.int32), one.download(np.int32)), "sections differ from the plain launch"
ctx.close()
