"""Small driver for counter profiling: the 4K sub-pel SATD candidate set through cost_kernel and refine_kernel
(argv[1]: PU size, 0 = all sizes; argv[2]: picture, 4k or 1080p)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from x265_amd import lib, synth
from x265_amd.hipmem import DeviceBuffer

size = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W, H = (1920, 1080) if len(sys.argv) > 2 and sys.argv[2] == "1080p" else (3840, 2160)
seq = synth.Sequence(W, H, seed=0x265)
planes = [synth.Plane(seq.frame(i)) for i in range(4)]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
refs = [2, 1, 0]
mot = [seq.global_motion(3, r) for r in refs]
crq = synth.subpel_requests(W, H, 3, refs, mot)
rrq = synth.refine_requests(W, H, 3, refs, mot)
if size:
    crq, rrq = crq[crq["pu_w"] == size], rrq[rrq["pu_w"] == size]
d_c, d_r = DeviceBuffer.from_array(crq), DeviceBuffer.from_array(rrq)
o_c, o_r = DeviceBuffer(4 * len(crq)), DeviceBuffer(36 * len(rrq))
for _ in range(3):
    ctx.cost_batch_device(d_c.ptr, len(crq), o_c.ptr)
    ctx.refine_batch_device(d_r.ptr, len(rrq), o_r.ptr)
ctx.sync()
print(len(crq), len(rrq))
ctx.close()
