"""Experiment: A/B of two builds of the library on the SATD candidate list of one 4K / 1080p picture (3 references).
argv: lib_a lib_b; each is loaded in its own process? No: ctypes can hold both, the contexts are separate."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from x265_amd import abi, lib, synth

paths = [os.path.abspath(p) for p in sys.argv[1:3]]
dev = torch.device("cuda", 0)
for (W, H) in ((1920, 1080), (3840, 2160)):
    seq = synth.Sequence(W, H, seed=0x265)
    planes = [synth.Plane(seq.frame(i)) for i in range(4)]
    refs = [2, 1, 0]
    mot = [seq.global_motion(3, r) for r in refs]
    rq = synth.subpel_requests(W, H, 3, refs, mot)
    if W == 1920 and len(sys.argv) > 3:
        rq = np.tile(rq, 4)  # a list as long as the 4K one over 1080p planes
    d = torch.from_numpy(rq.view(np.uint8).copy()).to(dev)
    o = torch.empty(len(rq), dtype=torch.int32, device=dev)
    ops = synth.satd_lane_ops(rq)
    ctxs = []
    for p in paths:
        lib._lib = None
        lib.LIB_PATH = p
        ctx = lib.Context(0)
        for i, pl in enumerate(planes):
            ctx.plane_upload(i, pl.buf, pl.stride, pl.origin)
        ctxs.append((p, lib._lib, ctx))
    for rnd in range(3):
        for p, L, ctx in ctxs:
            lib._lib = L
            ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
            ctx.cost_batch_device(d.data_ptr(), len(rq), o.data_ptr()); ctx.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for _ in range(10):
                ctx.cost_batch_device(d.data_ptr(), len(rq), o.data_ptr())
            e1.record(ext); ctx.sync()
            ms = e0.elapsed_time(e1) / 10
            print(f"{W}x{H} {os.path.basename(p):20s} {ms:8.3f} ms {len(rq)/ms*1e3:10.3e} cand/s frac {ops/(ms*1e-3)/3.93e13:.3f}", flush=True)
