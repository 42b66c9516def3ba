"""Replay a capture made by oracle/_ref/enc_capture (calls.bin + planes.bin in argv[1]) through the oracle (default)
or, with `gpu` as second argument, through x265_hip_me_batch; prints the number of calls whose (cost, mv) differ
from what the reference encoder got.  One-off verification tool for captures too large to commit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from x265_amd import abi

CALL = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"), ("pu_x", "<i2"), ("pu_y", "<i2"), ("pu_w", "u1"), ("pu_h", "u1"),
    ("method", "u1"), ("subme", "u1"), ("mvmin_x", "<i2"), ("mvmin_y", "<i2"), ("mvmax_x", "<i2"), ("mvmax_y", "<i2"),
    ("qmvp_x", "<i2"), ("qmvp_y", "<i2"), ("merange", "<i2"), ("num_mvc", "u1"), ("qp", "u1"), ("mvc", "<i2", (24,)),
    ("lowres", "u1"), ("pad", "u1", (3,)), ("cost", "<i4"), ("mvx", "<i2"), ("mvy", "<i2")])
d = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "oracle"
calls = np.fromfile(os.path.join(d, "calls.bin"), dtype=CALL)
raw = np.memmap(os.path.join(d, "planes.bin"), dtype=np.uint8, mode="r")
planes, pos = [], 0
while pos < len(raw):
    pid, stride, origin, nbytes = (int(v) for v in raw[pos:pos + 16].view("<i4"))
    planes.append((np.array(raw[pos + 16:pos + 16 + nbytes]), origin, stride))
    pos += 16 + nbytes
jobs = np.zeros(len(calls), abi.ME_JOB)
for k in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "subme", "mvmin_x", "mvmin_y", "mvmax_x", "mvmax_y",
          "qmvp_x", "qmvp_y", "merange", "num_mvc", "qp"):
    jobs[k] = calls[k]
jobs["method"] = calls["method"] | (calls["lowres"] * 0x80)
jobs["mvc"] = calls["mvc"]
print(f"{len(calls)} calls ({int(calls['lowres'].sum())} lookahead), {len(planes)} planes, QPs {np.unique(calls['qp']).tolist()}, "
      f"up to {int(calls['num_mvc'].max())} candidate vectors", flush=True)
t0 = time.time()
if mode == "gpu":
    from x265_amd import lib
    bad = 0
    # at most 64 planes are resident at a time: replay in runs of calls that touch a window of plane ids
    ctx = lib.Context(0)
    for q in np.unique(calls["qp"]):
        ctx.set_qp(int(q))
    top = np.maximum(jobs["fenc_plane"], jobs["ref_plane"] + 3 * calls["lowres"])
    lo = np.minimum(jobs["fenc_plane"], jobs["ref_plane"])
    start = 0
    while start < len(jobs):
        end = start
        first = int(lo[start])
        while end < len(jobs) and top[end] - min(first, int(lo[end])) < 60:
            first = min(first, int(lo[end]))
            end += 1
        ids = range(first, int(top[start:end].max()) + 1)
        for i in ids:
            ctx.plane_upload(i - first, planes[i][0], planes[i][2], planes[i][1])
        j = jobs[start:end].copy()
        j["fenc_plane"] -= first
        j["ref_plane"] -= first
        got = ctx.me_batch(j)
        c = calls[start:end]
        bad += int(((got["cost"] != c["cost"]) | (got["mvx"] != c["mvx"]) | (got["mvy"] != c["mvy"])).sum())
        start = end
    ctx.close()
else:
    from oracle import pyoracle
    O = pyoracle.load_oracle()
    tabs = {int(q): O.bitcost_table(int(q)) for q in np.unique(calls["qp"])}
    got = O.me_batch(planes, tabs, jobs, nthreads=8)
    bad = int(((got["cost"] != calls["cost"]) | (got["mvx"] != calls["mvx"]) | (got["mvy"] != calls["mvy"])).sum())
print(f"{mode}: {bad} mismatches of {len(calls)} calls ({time.time() - t0:.1f} s)")
sys.exit(1 if bad else 0)
