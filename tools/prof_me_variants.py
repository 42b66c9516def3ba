"""Counter driver: me_kernel on the 8x8 calls of one 4K picture x 1 reference in several variants, one launch each, in a
fixed order (rocprofv3 --pmc ... -- python3 tools/prof_me_variants.py); prints the order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from x265_amd import abi, lib, synth
from x265_amd.hipmem import DeviceBuffer

W, H = 3840, 2160
seq = synth.Sequence(W, H, seed=0x265)
ctx = lib.Context(0)
for i in range(4):
    p = synth.Plane(seq.frame(i))
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
for q in range(28, 40):
    ctx.set_qp(q)
base = synth.me_jobs(W, H, 3, [2], [seq.global_motion(3, 2)], method=abi.HEX, subme=2, merange=57)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8
base = base[base["pu_w"] == size]


def v(name, f):
    j = base.copy()
    f(j)
    return name, j


def fullpel(j):
    j["qmvp_x"] &= ~3
    j["qmvp_y"] &= ~3
    j["mvc"] &= ~3


variants = [v("bench list (HEX, subme 2)", lambda j: None), v("subme 0", lambda j: j.__setitem__("subme", 0)),
            v("no candidate vectors", lambda j: j.__setitem__("num_mvc", 0)), v("full-pel predictor and candidates", fullpel),
            v("full-pel, no candidates, subme 0", lambda j: (fullpel(j), j.__setitem__("num_mvc", 0), j.__setitem__("subme", 0))),
            v("merange 1 (one hexagon step at most), full-pel, no candidates, subme 0",
              lambda j: (fullpel(j), j.__setitem__("num_mvc", 0), j.__setitem__("subme", 0), j.__setitem__("merange", 1)))]
out = DeviceBuffer(8 * len(base))
for name, j in variants:
    d = DeviceBuffer.from_array(j)
    ctx.me_batch_device(d.ptr, len(j), out.ptr)
    ctx.sync()
    print(len(j), name)
ctx.close()
