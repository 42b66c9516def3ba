"""Experiment: where does me_kernel time go?  Times subsets of the 1080p ME-stage job list."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from x265_amd import abi, lib, synth

W, H = 1920, 1080
seq = synth.Sequence(W, H, seed=0x265)
planes = [synth.Plane(seq.frame(i)) for i in range(4)]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
for q in range(28, 40):
    ctx.set_qp(q)
dev = torch.device("cuda", 0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
refs = [2, 1, 0]
mot = [seq.global_motion(3, r) for r in refs]


def timeit(jobs, label, reps=5):
    d = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    out = torch.empty(len(jobs) * 8, dtype=torch.uint8, device=dev)
    ctx.me_batch_device(d.data_ptr(), len(jobs), out.data_ptr())
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(reps):
        ctx.me_batch_device(d.data_ptr(), len(jobs), out.data_ptr())
    e1.record(ext)
    ctx.sync()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:40s} n={len(jobs):7d}  {ms:8.3f} ms  {ms * 1e6 / len(jobs):8.1f} ns/job")


base = synth.me_jobs(W, H, 3, refs, mot, method=abi.HEX, subme=2, merange=57)
timeit(base, "all, HEX subme2")
timeit(base[np.argsort(-base["pu_w"].astype(int), kind="stable")], "all, big PUs first")
timeit(base[np.argsort(base["pu_w"].astype(int), kind="stable")], "all, small PUs first")
for s in (64, 32, 16, 8):
    timeit(base[base["pu_w"] == s], f"only {s}x{s}")
for sub in (0, 1, 2, 5, 7):
    j = base.copy(); j["subme"] = sub
    timeit(j, f"all, HEX subme{sub}")
for m, name in ((0, "DIA"), (2, "UMH"), (3, "STAR")):
    j = base.copy(); j["method"] = m
    timeit(j, f"all, {name} subme2")
j = base.copy(); j["num_mvc"] = 0
timeit(j, "all, HEX subme2, no mvc")
# empty-ish kernel: jobs whose cost is zero at the predictor? use tiny merange
j = base.copy(); j["merange"] = 1
timeit(j, "all, HEX subme2, merange 1")
# cost kernel probe
rng = np.random.default_rng(1)
n = 200000
reqs = np.zeros(n, abi.COST_REQ)
for size in (8, 64):
    reqs["pu_w"] = reqs["pu_h"] = size
    reqs["pu_x"] = rng.integers(0, W - 64, n); reqs["pu_y"] = rng.integers(0, H - 64, n)
    reqs["fenc_plane"], reqs["ref_plane"] = 3, 2
    for metric, fr in ((0, 0), (1, 0), (1, 2), (1, 3)):
        reqs["metric"] = metric
        reqs["qmv_x"] = 4 * rng.integers(-16, 16, n) + (fr & 1) * 2; reqs["qmv_y"] = 4 * rng.integers(-16, 16, n) + (fr >> 1) * 2
        d = torch.from_numpy(reqs.view(np.uint8).copy()).to(dev)
        out = torch.empty(n, dtype=torch.int32, device=dev)
        ctx.cost_batch_device(d.data_ptr(), n, out.data_ptr()); ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ext)
        for _ in range(3):
            ctx.cost_batch_device(d.data_ptr(), n, out.data_ptr())
        e1.record(ext); ctx.sync()
        ms = e0.elapsed_time(e1) / 3
        print(f"cost_kernel {size}x{size} metric {metric} frac {fr}: {ms:8.3f} ms {ms*1e6/n:8.1f} ns/req")
