"""Small driver for counter profiling of the other presets' searches: the ME-stage list of one 1080p picture x 3 references
with the method and sub-pel level given (argv: method 0..4, subme 0..7), three launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x265_amd import lib, synth

W, H = 1920, 1080
method, subme = int(sys.argv[1]), int(sys.argv[2])
seq = synth.Sequence(W, H, seed=0x265)
ctx = lib.Context(0)
for i in range(4):
    p = synth.Plane(seq.frame(i))
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
for q in range(28, 40):
    ctx.set_qp(q)
refs = [2, 1, 0]
jobs = synth.me_jobs(W, H, 3, refs, [seq.global_motion(3, r) for r in refs], method=method, subme=subme, merange=57)
print(len(jobs), "calls, method", method, "subme", subme)
for _ in range(3):
    ctx.me_batch(jobs)
ctx.close()
