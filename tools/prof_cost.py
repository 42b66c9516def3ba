"""Small driver for counter profiling of cost_kernel: the 1080p SATD sub-pel candidate set of report_configs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from x265_amd import abi, lib, synth
W, H = 1920, 1080
seq = synth.Sequence(W, H, seed=0x265)
planes = [synth.Plane(seq.frame(i)) for i in range(4)]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
pus = synth.square_pus(W, H)
ring = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1)]
offs = [(2 * x, 2 * y) for x, y in ring] + ring
size = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if size:
    pus = pus[pus[:, 2] == size]
n = len(pus)
rq = np.zeros(n * len(offs), abi.COST_REQ)
for k, (ox, oy) in enumerate(offs):
    r = rq[k * n:(k + 1) * n]
    r["fenc_plane"], r["ref_plane"] = 3, 2
    r["pu_x"], r["pu_y"] = pus[:, 0], pus[:, 1]
    r["pu_w"] = r["pu_h"] = pus[:, 2]
    r["metric"] = abi.SATD
    r["qmv_x"], r["qmv_y"] = 12 + ox, 8 + oy
for _ in range(3):
    ctx.cost_batch(rq)
ctx.close()
