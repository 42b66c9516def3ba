"""Time x265_hip_lookahead_batch_device at 1080p (120 x 68 CUs per search) for several batch sizes, and the oracle on
the host for comparison.  usage: python tools/time_lookahead.py [width height]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from x265_amd import abi, hipmem, lib, synth  # noqa: E402


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    seq = synth.Sequence(w, h, seed=0x265)
    ctx = lib.Context(0)
    npic = 8
    planes = [synth.Plane(seq.frame(i)) for i in range(npic)]
    for i, p in enumerate(planes):
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
        geo = ctx.lowres_init(i, p.w, p.h, 16 + 4 * i)
    ctx.set_qp(abi.LOOKAHEAD_QP)
    W, H = geo.width // 8, geo.lines // 8
    pairs = [(a, b) for a in range(npic) for b in range(npic) if a != b and abs(a - b) <= 4]
    print(f"{w}x{h}: {W} x {H} CUs per search, {len(pairs)} distinct pairs")
    for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        lj = np.zeros(n, abi.LOOKAHEAD_JOB)
        sel = [pairs[i % len(pairs)] for i in range(n)]
        lj["fenc_plane"], lj["ref_plane"] = [16 + 4 * a for a, _ in sel], [16 + 4 * b for _, b in sel]
        lj["merange"], lj["method"], lj["subme"], lj["qp"] = 16, abi.HEX, 1, abi.LOOKAHEAD_QP
        d_jobs = hipmem.DeviceBuffer.from_array(lj)
        d_out = hipmem.DeviceBuffer(n * W * H * 8)
        for _ in range(2):
            ctx.lookahead_batch_device(d_jobs.ptr, n, W, H, d_out.ptr)
        ctx.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.lookahead_batch_device(d_jobs.ptr, n, W, H, d_out.ptr)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"  {n:4d} searches per launch: {dt * 1e3:8.3f} ms  -> {n / dt:9.1f} searches/s, {n * W * H / dt / 1e6:7.2f} M CUs/s, "
              f"{dt / (W + 2 * (H - 1)) * 1e6:6.2f} us per chain step", flush=True)
        d_jobs.free()
        d_out.free()
    ctx.close()


if __name__ == "__main__":
    main()
