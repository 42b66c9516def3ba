"""Driver for counter profiling at 4K (BASELINE.json configs[3]): one P picture against its 3 references through every
leg bench.py times -- the sub-pel rings (refine_kernel, the four PU sizes as four runs of one launch), the same candidates as single requests
(cost_kernel), the motionEstimate replay (me_kernel), the +-64 full search (fullsearch2_kernel) -- and the lookahead's
whole-frame searches of the four pictures' half-resolution planes (lookahead_kernel, 12 pairs in one launch).
Three rounds; planes, lists and results stay on the device.  argv[1]: rounds (default 3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from x265_amd import abi, lib, runner as rn, synth
from x265_amd.hipmem import DeviceBuffer

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H = 3840, 2160
seq = synth.Sequence(W, H, seed=0x265)
ctx = lib.Context(0)
r = rn.SegmentRunner(ctx, seq, n_refs=3)
r.load((0, 4))
cur, refs = 3, [2, 1, 0]
rings = r.ring_requests(cur, refs)
satd = r.satd_requests(cur, refs)
me = r.me_jobs(cur, refs)
fs = r.ctu_jobs(cur, refs[0], 64)
ctx.set_qp(abi.LOOKAHEAD_QP)
for i in range(4):
    geo = ctx.lowres_init(i, W, H, 16 + 4 * i)
LW, LH = geo.width // 8, geo.lines // 8
pairs = [(a, b) for a in range(4) for b in range(4) if a != b]
lj = np.zeros(len(pairs), abi.LOOKAHEAD_JOB)
lj["fenc_plane"], lj["ref_plane"] = [16 + 4 * a for a, _ in pairs], [16 + 4 * b for _, b in pairs]
lj["merange"], lj["method"], lj["subme"], lj["qp"] = 16, abi.HEX, 1, abi.LOOKAHEAD_QP

d_rings = DeviceBuffer.from_array(np.concatenate(rings))
ring_sections = [(len(q), int(q["pu_w"][0]), int(q["pu_h"][0])) for q in rings]
o_ring = DeviceBuffer(36 * sum(len(q) for q in rings))
d_satd, o_satd = DeviceBuffer.from_array(satd), DeviceBuffer(4 * len(satd))
d_me, o_me = DeviceBuffer.from_array(me), DeviceBuffer(8 * len(me))
d_fs, o_fs = DeviceBuffer.from_array(fs), DeviceBuffer(8 * lib.PUS_PER_CTU * len(fs))
d_lj, o_lj = DeviceBuffer.from_array(lj), DeviceBuffer(8 * len(lj) * LW * LH)
for _ in range(rounds):
    ctx.refine_sections_device(d_rings.ptr, ring_sections, o_ring.ptr)
    ctx.cost_batch_device(d_satd.ptr, len(satd), o_satd.ptr)
    ctx.me_batch_device(d_me.ptr, len(me), o_me.ptr)
    ctx.fullsearch_device(d_fs.ptr, len(fs), o_fs.ptr)
    ctx.lookahead_batch_device(d_lj.ptr, len(lj), LW, LH, o_lj.ptr)
    ctx.sync()
print({"ring_requests": [len(q) for q in rings], "satd_requests": len(satd), "me_jobs": len(me), "ctus": len(fs),
                    "lookahead_pairs": len(lj), "lookahead_cus": [LW, LH], "plane_bytes": synth.Plane.nbytes_for(W, H)})
ctx.close()
