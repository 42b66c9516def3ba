"""Experiment: A/B of several builds of the library on the ME-stage job list (HEX, subme 2, 3 references) of one
1080p and one 4K picture.  argv: paths of the builds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from x265_amd import abi, lib, synth

paths = [os.path.abspath(p) for p in sys.argv[1:]]
dev = torch.device("cuda", 0)
for (W, H) in ((1920, 1080), (3840, 2160)):
    seq = synth.Sequence(W, H, seed=0x265)
    planes = [synth.Plane(seq.frame(i)) for i in range(4)]
    refs = [2, 1, 0]
    mot = [seq.global_motion(3, r) for r in refs]
    jobs = synth.me_jobs(W, H, 3, refs, mot, method=abi.HEX, subme=2, merange=57)
    d = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    o = torch.empty(len(jobs) * 8, dtype=torch.uint8, device=dev)
    ctxs = []
    for p in paths:
        lib._lib = None
        lib.LIB_PATH = p
        ctx = lib.Context(0)
        for i, pl in enumerate(planes):
            ctx.plane_upload(i, pl.buf, pl.stride, pl.origin)
        for q in range(20, 45):
            ctx.set_qp(q)
        ctxs.append((p, lib._lib, ctx))
    outs = []
    for rnd in range(3):
        for p, L, ctx in ctxs:
            lib._lib = L
            ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
            ctx.me_batch_device(d.data_ptr(), len(jobs), o.data_ptr()); ctx.sync()
            if rnd == 0:
                outs.append(o.cpu().numpy().copy())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for _ in range(10):
                ctx.me_batch_device(d.data_ptr(), len(jobs), o.data_ptr())
            e1.record(ext); ctx.sync()
            ms = e0.elapsed_time(e1) / 10
            print(f"{W}x{H} {os.path.basename(p):20s} {ms:8.3f} ms/picture  {1e3/ms:8.1f} pictures/s", flush=True)
    assert all(np.array_equal(outs[0], x) for x in outs), "builds disagree"
