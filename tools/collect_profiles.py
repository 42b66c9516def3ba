"""Condense gpurun_out/prof/ (written by tools/make_profiles.sh on the GPU box) into profiles/<tag>_*."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"


def ours(name):
    return "x265hip::" in name


def find(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {SRC}")
    return hits[0]


def pmc(pattern):
    """per kernel and counter: mean over launches"""
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(find(pattern))):
        if ours(r["Kernel_Name"]):
            acc[(r["Kernel_Name"].split("(")[0].replace("x265hip::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


# 1. kernel stats of the bench run (rocprofv3 --kernel-trace --stats)
shutil.copy(find("stats/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, f"{tag}_bench_under_rocprof.json"))
# 2. SQ counters, our kernels only
sq, cnt = pmc("pmc_sq/**/*counter_collection.csv")
with open(os.path.join(DST, f"{tag}_pmc_sq.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for (k, c) in sorted(sq):
        f.write(f"{k},{c},{sq[(k, c)]:.0f},{cnt[(k, c)]}\n")
# 3. HBM traffic
fe, _ = pmc("pmc_fetch/**/*counter_collection.csv")
wr, _ = pmc("pmc_write/**/*counter_collection.csv")
traffic = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_me.py (one 1080p picture: "
                   "510 CTU full searches with one reference; 128700 motionEstimate calls, HEX subme 2, 3 references; the "
                   "picture's 2.06 M sub-pel SATD candidates through cost_kernel and as sa8d through dist_kernel; one "
                   "bi-directional prediction of the picture in 16x16 blocks), mean per "
                   "launch, KiB as reported.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane "
                   "streaming reads; fullsearch2_kernel stages with 16-B loads, so fetch_kib_corrected = 2 x raw for it; "
                   "me_kernel / cost_kernel read 4 B per lane (uncalibrated width): raw and 2 x raw are both given."}
for k in sorted({k for (k, _) in fe}):
    traffic[k] = {"fetch_kib_raw": fe[(k, "FETCH_SIZE")], "fetch_kib_corrected": 2 * fe[(k, "FETCH_SIZE")],
                  "write_kib": wr.get((k, "WRITE_SIZE"))}
json.dump(traffic, open(os.path.join(DST, f"{tag}_traffic.json"), "w"), indent=1)
# 4. the rest
for a, b in (("ubench_valu.txt", f"{tag}_ubench_valu.txt"), ("configs.json", f"{tag}_configs.json"), ("bench.json", f"{tag}_bench.json")):
    shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
print("wrote", sorted(p for p in os.listdir(DST) if p.startswith(tag)))
