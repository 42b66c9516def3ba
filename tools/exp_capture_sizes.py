"""me_kernel on the reference encoder's own calls (a capture directory made by tools/capture_1080p.sh), whole list and
per PU size; jobs resident in HBM.  X265_EXP_LIB selects another build of the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from x265_amd import abi, lib
if os.environ.get("X265_EXP_LIB"):
    lib.LIB_PATH = os.path.abspath(os.environ["X265_EXP_LIB"])
sys.argv = [sys.argv[0], sys.argv[1], "none"]
src = open(os.path.join(ROOT, "tests", "tools", "replay_capture.py")).read().split("t0 = time.time()")[0]
ns = {"__file__": os.path.join(ROOT, "tests", "tools", "replay_capture.py")}
exec(compile(src, "replay_capture_head", "exec"), ns)
jobs, planes, calls = ns["jobs"], ns["planes"], ns["calls"]
ctx = lib.Context(0)
for i, (b, o, s) in enumerate(planes):
    ctx.plane_upload(i, b, s, o)
for q in np.unique(calls["qp"]):
    ctx.set_qp(int(q))
dev = torch.device("cuda", 0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)


def run(sel, label, reps=20):
    j, c = jobs[sel], calls[sel]
    d = torch.from_numpy(j.view(np.uint8).copy()).to(dev)
    o = torch.empty(len(j) * 8, dtype=torch.uint8, device=dev)
    for _ in range(40):
        ctx.me_batch_device(d.data_ptr(), len(j), o.data_ptr())
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(reps):
        ctx.me_batch_device(d.data_ptr(), len(j), o.data_ptr())
    e1.record(ext)
    ctx.sync()
    ms = e0.elapsed_time(e1) / reps
    got = o.cpu().numpy().view(abi.ME_RESULT)
    bad = int(((got["cost"] != c["cost"]) | (got["mvx"] != c["mvx"]) | (got["mvy"] != c["mvy"])).sum())
    print(f"{label:>10}: {len(j):7d} calls {ms:7.3f} ms = {len(j) / ms * 1e3:.3e} calls/s, {ms / len(j) * 1e6:6.2f} ns/call, {bad} mismatches", flush=True)
    return ms


everything = np.ones(len(jobs), bool)
run(everything, "all")
for s in (8, 16, 32, 64):
    run(jobs["pu_w"] == s, f"{s}x{s}")
order = np.argsort(-jobs["pu_w"].astype(int), kind="stable")
run(order, "big first")
run(everything, "all")
