"""Does the re-fetch of reference pixels cost anything when the planes no longer fit the caches?  The sub-pel rings
(refine_kernel) and the motionEstimate replay (one routed launch) of one 4K picture against 3, 16 and 32 distinct
reference pictures: 32 references + the source are 310 MB of planes, more than the 256 MB Infinity Cache and 75 times the
L2 of an XCD.  Per-candidate / per-call rates should not move if the kernels are bound by arithmetic."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from x265_amd import abi, lib, synth
from x265_amd.hipmem import DeviceBuffer

W, H, NMAX = 3840, 2160, 32
seq = synth.Sequence(W, H, seed=0x265)
ctx = lib.Context(0)
for i in range(NMAX + 1):
    p = synth.Plane(seq.frame(i))
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
for q in range(28, 40):
    ctx.set_qp(q)
print(f"{NMAX + 1} planes of {synth.Plane.nbytes_for(W, H) / 1e6:.1f} MB = {(NMAX + 1) * synth.Plane.nbytes_for(W, H) / 1e6:.0f} MB", flush=True)


def timed(fn, reps=8):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        fn()
        ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


cur = NMAX
for nref in (3, 16, 32):
    refs = [cur - 1 - i for i in range(nref)]
    mot = [seq.global_motion(cur, r) for r in refs]
    rq = synth.refine_requests(W, H, cur, refs, mot)
    lists = [rq[rq["pu_w"] == s] for s in (64, 32, 16, 8)]
    d = DeviceBuffer.from_array(np.concatenate(lists))
    o = DeviceBuffer(36 * len(rq))
    sections = [(len(q), int(q["pu_w"][0]), int(q["pu_h"][0])) for q in lists]
    ms = timed(lambda: ctx.refine_sections_device(d.ptr, sections, o.ptr))
    cands = 8 * len(rq)
    print(f"rings of 8, {nref:2d} references: {ms:8.3f} ms  {cands / ms * 1e3:.4e} candidates/s  {ms / nref:.4f} ms per reference", flush=True)
    d.free()
    o.free()
    jobs = synth.me_jobs(W, H, cur, refs, mot, method=abi.HEX, subme=2, merange=57)
    d, o = DeviceBuffer.from_array(jobs), DeviceBuffer(8 * len(jobs))
    ms = timed(lambda: ctx.me_batch_device(d.ptr, len(jobs), o.ptr))
    print(f"motionEstimate, {nref:2d} references: {ms:8.3f} ms  {len(jobs) / ms * 1e3:.4e} calls/s  {ms / nref:.4f} ms per reference", flush=True)
    d.free()
    o.free()
ctx.close()
