import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from x265_amd import lib, synth, runner as rn
from x265_amd.hipmem import DeviceBuffer
ctx = lib.Context(0)
seq = synth.Sequence(3840, 2160, seed=0x265)
r = rn.SegmentRunner(ctx, seq, n_refs=3)
r.load((0, 8))
def timed(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
for cur, refs in r.pairs():
    lists = r.ring_requests(cur, refs)
    ds = [(DeviceBuffer.from_array(rr), len(rr)) for rr in lists]
    nn = sum(c for _, c in ds); o = DeviceBuffer(36 * nn)
    def go():
        for d, c in ds:
            ctx.refine_batch_device(d.ptr, c, o.ptr)
    ms = timed(go)
    rr = np.concatenate(lists)
    print(cur, refs, len(rr), f"{ms:.3f} ms  {8*len(rr)/ms*1e3:.3e}/s", [seq.global_motion(cur, x) for x in refs], flush=True)
