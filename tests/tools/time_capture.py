"""Time x265_hip_me_batch_device on the full-resolution calls of an encoder capture (argv[1]); jobs resident in HBM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from x265_amd import abi, lib
if os.environ.get("X265_EXP_LIB"):  # experiments: another build of the library
    lib.LIB_PATH = os.path.abspath(os.environ["X265_EXP_LIB"])
sys.argv = [sys.argv[0], sys.argv[1], "none"]
src = open(os.path.join(ROOT, "tests", "tools", "replay_capture.py")).read().split("t0 = time.time()")[0]
ns = {"__file__": os.path.join(ROOT, "tests", "tools", "replay_capture.py")}
exec(compile(src, "replay_capture_head", "exec"), ns)
jobs, planes, calls = ns["jobs"], ns["planes"], ns["calls"]
assert len(planes) <= 64
ctx = lib.Context(0)
for i, (b, o, s) in enumerate(planes):
    ctx.plane_upload(i, b, s, o)
for q in np.unique(calls["qp"]):
    ctx.set_qp(int(q))
dev = torch.device("cuda", 0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
d = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
o = torch.empty(len(jobs) * 8, dtype=torch.uint8, device=dev)
for _ in range(60):
    ctx.me_batch_device(d.data_ptr(), len(jobs), o.data_ptr())
ctx.sync()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(ext)
for _ in range(20):
    ctx.me_batch_device(d.data_ptr(), len(jobs), o.data_ptr())
e1.record(ext)
ctx.sync()
ms = e0.elapsed_time(e1) / 20
got = o.cpu().numpy().view(abi.ME_RESULT)
bad = int(((got["cost"] != calls["cost"]) | (got["mvx"] != calls["mvx"]) | (got["mvy"] != calls["mvy"])).sum())
sizes, cnt = np.unique(jobs["pu_w"], return_counts=True)
print(f"{len(jobs)} encoder calls: {ms:.3f} ms = {len(jobs) / ms * 1e3:.3e} calls/s, {bad} mismatches; PU widths {dict(zip(sizes.tolist(), cnt.tolist()))}, "
      f"mean candidate vectors {jobs['num_mvc'].mean():.2f}")
# the reference's own MotionEstimate on the same calls, 16 host threads (oracle/_ref: checker library, timing only)
import time
from oracle import pyoracle
if pyoracle.have_ref():
    R = pyoracle.load_ref()
    tabs = {int(q): R.bitcost_table(int(q)) for q in np.unique(calls["qp"])}
    R.me_batch(planes, tabs, jobs[:2000], nthreads=16)
    t0 = time.perf_counter()
    want = R.me_batch(planes, tabs, jobs, nthreads=16)
    dt = time.perf_counter() - t0
    same = int(((want["cost"] == calls["cost"]) & (want["mvx"] == calls["mvx"]) & (want["mvy"] == calls["mvy"])).sum())
    print(f"reference MotionEstimate, 16 threads (C primitives, no asm): {dt * 1e3:.1f} ms = {len(jobs) / dt:.3e} calls/s; {same} of {len(jobs)} reproduce the capture")
