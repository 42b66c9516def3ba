"""BASELINE.md section 2 on the GPU box's host cores, reference code only (oracle/_ref, built from the reference's
own sources; C primitives, no x86 assembly: the image has no yasm):
  B1  the reference encoder end to end, --preset medium, 64 pictures of the 1080p synthetic clip
  B2  the ME stage of one picture through the reference MotionEstimate, 1 thread and all threads
  B3  per-primitive ns/call, single core
(B4, the full-search surface, is bench.py's cpu_baseline.)  Prints one JSON object."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import pyoracle
from x265_amd import abi, lib, synth

W, H, N = 1920, 1080, 64
cores = min(16, os.cpu_count() or 1)
out = {"cores": cores, "asm": False, "note": "reference sources compiled with g++ -O3, C primitives (no yasm in the image)"}
seq = synth.Sequence(W, H, seed=0x265)

# ---- B1 -------------------------------------------------------------------------------------------------------------
exe = os.path.join(ROOT, "oracle", "_ref", "enc_capture")
clip = "/tmp/x265_b1.yuv"
with open(clip, "wb") as f:
    for i in range(N):
        f.write(seq.frame(i).astype(np.uint8).tobytes())
        f.write(bytes([128]) * (W * H // 2))
print("clip written", file=sys.stderr, flush=True)
runs = []
for frame_threads in (0, 1):  # the encoder's default frame parallelism, and one frame at a time
    t0 = time.perf_counter()
    p = subprocess.Popen([exe, clip, str(W), str(H), str(N), "/tmp", "0", "0", "0", str(cores), str(frame_threads)],
                         stderr=subprocess.PIPE, text=True)
    shown = 0
    while p.poll() is None:
        time.sleep(0.05)
        if time.perf_counter() - t0 > shown + 20:
            shown += 20
            print(f"encoding ... {shown} s", file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    runs.append({"frame_threads": frame_threads or "default", "pool_threads": cores, "seconds": dt, "fps": N / dt,
                 "log": p.stderr.read().strip().splitlines()[-1]})
os.remove(clip)
out["B1_encoder_medium_1080p_64_pictures"] = runs

# ---- B2 -------------------------------------------------------------------------------------------------------------
R = pyoracle.load_ref()
planes = [synth.Plane(seq.frame(i)) for i in range(4)]
pl = [p.as_tuple() for p in planes]
refs = [2, 1, 0]
jobs = synth.me_jobs(W, H, 3, refs, [seq.global_motion(3, r) for r in refs], method=abi.HEX, subme=2, merange=57, seed=3)
tabs = {q: lib.costtab_build(q) for q in range(28, 40)}
b2 = {}
for n in (1, cores):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        R.me_batch(pl, tabs, jobs, nthreads=n)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    b2[f"{n}_threads"] = {"ms_per_picture": best * 1e3, "pictures_per_s": 1 / best, "calls_per_s": len(jobs) / best}
    print(f"B2 {n} threads done", file=sys.stderr, flush=True)
out["B2_me_stage_hex_subme2_3_refs"] = {"calls_per_picture": len(jobs), **b2}

# ---- B3 -------------------------------------------------------------------------------------------------------------
L = C.CDLL(pyoracle.ref_path())
L.x265r_time_primitive.restype = C.c_double
b3 = {}
for kind, name in enumerate(["sad", "sad_x3", "sad_x4", "satd", "luma_hpp", "luma_vpp", "luma_hps+luma_vsp"]):
    b3[name] = {}
    for size in (8, 16, 32, 64):
        part = abi.PARTS.index((size, size))
        ns = min(L.x265r_time_primitive(kind, part, 4000) for _ in range(3))
        npx = size * size * (3 if kind == 1 else 4 if kind == 2 else 1)
        b3[name][f"{size}x{size}"] = {"ns_per_call": ns, "pixels_per_s": npx / ns * 1e9}
out["B3_primitives_single_core"] = b3
print(json.dumps(out, indent=1))
