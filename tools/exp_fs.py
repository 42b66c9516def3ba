"""Experiment: fixed vs per-step cost of fullsearch2_kernel (varies the window of the 1080p job list)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from x265_amd import lib, synth

W, H = 1920, 1080
seq = synth.Sequence(W, H, seed=0x265)
planes = [synth.Plane(seq.frame(i)) for i in range(2)]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
ctx.set_qp(30)
dev = torch.device("cuda", 0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
base = bench.build_ctu_jobs(seq, 1, 0, 1, 0)


def timeit(jobs, label, reps=10):
    d = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    out = torch.empty(len(jobs) * lib.PUS_PER_CTU * 8, dtype=torch.uint8, device=dev)
    ctx.fullsearch_device(d.data_ptr(), len(jobs), out.data_ptr())
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(reps):
        ctx.fullsearch_device(d.data_ptr(), len(jobs), out.data_ptr())
    e1.record(ext)
    ctx.sync()
    ms = e0.elapsed_time(e1) / reps
    nx = jobs["mvmax_x"].astype(int) - jobs["mvmin_x"] + 1
    ny = jobs["mvmax_y"].astype(int) - jobs["mvmin_y"] + 1
    print(f"{label:34s} n={len(jobs):5d} nx~{nx.mean():6.1f} ny~{ny.mean():6.1f}  {ms:7.4f} ms  {ms * 1e3 / len(jobs) * 256:7.2f} us/CTU-slot")


timeit(base, "1080p +-64")
j = base.copy(); j["mvmax_x"] = np.maximum(j["mvmin_x"], j["mvmax_x"] - 1); timeit(j, "nx - 1 (no tail column)")
j = base.copy(); j["mvmax_y"] = j["mvmin_y"] + (j["mvmax_y"] - j["mvmin_y"]) // 2; timeit(j, "ny / 2")
j = base.copy(); j["mvmax_y"] = j["mvmin_y"]; timeit(j, "ny = 1")
j = base.copy(); j["mvmax_y"] = j["mvmin_y"]; j["mvmax_x"] = np.maximum(j["mvmin_x"], j["mvmax_x"] - 1); timeit(j, "ny = 1, no tail")
timeit(base[:256], "256 CTUs (one per CU)")
timeit(base[:512 - 2], "510")
timeit(np.concatenate([base, base[:2]]), "512 CTUs")
timeit(np.concatenate([base, base, base, base[:2 * 3 + 2]]), "2048 CTUs")
