"""Experiment: SATD sub-pel candidate rate of cost_kernel per block size (1080p), optionally with another build
of the library (argv[1] = path of the .so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from x265_amd import abi, lib, synth

if len(sys.argv) > 1:
    lib.LIB_PATH = os.path.abspath(sys.argv[1])
W, H = 1920, 1080
seq = synth.Sequence(W, H, seed=0x265)
planes = [synth.Plane(seq.frame(i)) for i in range(4)]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
dev = torch.device("cuda", 0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
mot = [seq.global_motion(3, 2)]


def timeit(rq, label, reps=5, items=6e7):
    # replicate to a fixed amount of work (4x4 tiles) so that every class fills the chip many times over
    tiles = float((rq["pu_w"].astype(np.int64) * rq["pu_h"]).sum()) / 16
    rq = np.tile(rq, max(1, int(items / tiles)))
    d = torch.from_numpy(rq.view(np.uint8).copy()).to(dev)
    out = torch.empty(len(rq), dtype=torch.int32, device=dev)
    ctx.cost_batch_device(d.data_ptr(), len(rq), out.data_ptr())
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(reps):
        ctx.cost_batch_device(d.data_ptr(), len(rq), out.data_ptr())
    e1.record(ext)
    ctx.sync()
    ms = e0.elapsed_time(e1) / reps
    ops = synth.satd_lane_ops(rq)
    print(f"{label:28s} n={len(rq):8d} {ms:8.3f} ms {len(rq) / ms * 1e3:10.3e} cand/s  frac {ops / (ms * 1e-3) / 3.93e13:.3f}", flush=True)


rq = synth.subpel_requests(W, H, 3, [2], mot)
timeit(rq, "all sizes")
timeit(rq[np.argsort(-rq["pu_w"].astype(int), kind="stable")], "all sizes, sorted by size")
timeit(rq[np.random.default_rng(0).permutation(len(rq))], "all sizes, shuffled")
for s in (64, 32, 16, 8):
    timeit(rq[rq["pu_w"] == s], f"{s}x{s}")
# by fraction kind, 16x16
base = rq[rq["pu_w"] == 16]
for name, fx, fy in (("full", 0, 0), ("h", 2, 0), ("v", 0, 2), ("hv", 2, 2), ("hv q", 1, 3)):
    r = base.copy()
    r["qmv_x"] = (r["qmv_x"] & ~3) | fx
    r["qmv_y"] = (r["qmv_y"] & ~3) | fy
    timeit(r, f"16x16 {name}")
ctx.close()
