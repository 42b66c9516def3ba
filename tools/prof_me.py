"""Small driver for counter profiling: one ME-stage frame (HEX, subme 2, 3 refs) and one full-search frame; with
`all` also the sub-pel SATD candidates of the picture (cost_kernel), the same as sa8d (dist_kernel) and a
bi-directional prediction of the picture in 16x16 blocks (predict_kernel)."""
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
if which == "all":
    rq = synth.subpel_requests(W, H, 3, refs, [seq.global_motion(3, r) for r in refs])
    rq8 = rq.copy()
    rq8["metric"] = abi.SA8D
    dst = synth.Plane(np.zeros((H, W), np.uint8))
    ctx.plane_upload(8, dst.buf, dst.stride, dst.origin)
    xs, ys = np.meshgrid(np.arange(0, W - 15, 16), np.arange(0, H - 15, 16))
    pr = np.zeros(xs.size, abi.PRED_REQ)
    pr["dst_plane"], pr["ref0_plane"], pr["ref1_plane"] = 8, 2, 1
    pr["pu_x"], pr["pu_y"], pr["pu_w"], pr["pu_h"] = xs.ravel(), ys.ravel(), 16, 16
    pr["qmv0_x"], pr["qmv0_y"], pr["qmv1_x"], pr["qmv1_y"] = 13, 7, -6, 9
for _ in range(3):
    if which in ("me", "both", "all"):
        ctx.me_batch(jobs)
    if which in ("fs", "both", "all"):
        ctx.fullsearch(build_ctu_jobs(seq, 3, 2, 3, 2))
    if which == "all":
        ctx.cost_batch(rq)
        ctx.dist_batch(rq8)
        ctx.predict_batch(pr)
ctx.close()
