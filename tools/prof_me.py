"""Small driver for counter profiling: one ME-stage frame (HEX, subme 2, 3 refs) and one full-search frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from x265_amd import abi, lib, synth
from bench import build_ctu_jobs
W, H = 1920, 1080
seq = synth.Sequence(W, H, seed=0x265)
planes = [synth.Plane(seq.frame(i)) for i in range(4)]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
for q in range(28, 40):
    ctx.set_qp(q)
refs = [2, 1, 0]
jobs = synth.me_jobs(W, H, 3, refs, [seq.global_motion(3, r) for r in refs], method=abi.HEX, subme=2, merange=57)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for _ in range(3):
    if which in ("me", "both"):
        ctx.me_batch(jobs)
    if which in ("fs", "both"):
        ctx.fullsearch(build_ctu_jobs(seq, 3, 2, 3, 2))
ctx.close()
