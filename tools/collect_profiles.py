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
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def ours(name):
    return "x265hip::" in name


def short(name):
    """kernel name without namespace, return type, template arguments and parameters (the instantiations of
    me_team_kernel are one kernel here)"""
    n = name.split("(")[0].replace("x265hip::", "")
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


def find(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {SRC}")
    return hits[0]


def pmc(pattern):
    """per kernel and counter: the launches in dispatch order"""
    acc = collections.defaultdict(list)
    rows = list(csv.DictReader(open(find(pattern))))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        if ours(r["Kernel_Name"]):
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v)


# 1. kernel stats of the bench run (rocprofv3 --kernel-trace --stats) and of the 4K driver
shutil.copy(find("stats/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats.csv"))
shutil.copy(find("stats4k/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats_prof4k.csv"))
shutil.copy(find("stats_4k/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats_4k_only.csv"))
shutil.copy(os.path.join(SRC, "bench_4k_under_rocprof.json"), os.path.join(DST, f"{tag}_bench_4k_under_rocprof.json"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, f"{tag}_bench_under_rocprof.json"))
shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, f"{tag}_bench.json"))
# per-launch durations of the 4K driver by kernel; refine_kernel per PU size (dispatch order: 64, 32, 16, 8)
trace = list(csv.DictReader(open(find("stats4k/**/*kernel_trace.csv"))))
trace.sort(key=lambda r: int(r["Dispatch_Id"]))
dur = collections.defaultdict(list)
for r in trace:
    if ours(r["Kernel_Name"]):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
# 2. SQ counters, our kernels only: mean per launch; refine_kernel also per PU size
sq = pmc("pmc_sq/**/*counter_collection.csv")
with open(os.path.join(DST, f"{tag}_pmc_sq.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for (k, c) in sorted(sq):
        v = sq[(k, c)]
        f.write(f"{k},{c},{mean(v):.0f},{len(v)}\n")
# 3. HBM traffic (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in separate passes, KiB units; on gfx950 FETCH_SIZE counts
# half of what 16-byte-per-lane streaming reads move: raw and 2 x raw are both given, the corrected figure is the one
# quoted for kernels that stage with 16-byte loads)
fe = pmc("pmc_fetch/**/*counter_collection.csv")
wr = pmc("pmc_write/**/*counter_collection.csv")
import ast
run = next(ast.literal_eval(l) for l in open(os.path.join(SRC, "pmc_fetch.log")) if l.startswith("{"))
traffic = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_4k.py (one 4K P picture against "
                   "3 references: rings of 8 through refine_kernel (the four PU sizes as four runs of one launch), the same candidates as single requests through "
                   "cost_kernel, 516 k motionEstimate calls through me_kernel, 2040 CTU full searches, 12 lookahead frame searches of "
                   "240 x 135 CUs), mean per launch, bytes = KiB x 1024.  *_4k keys: bytes per bench step, as bench.py's `traffic` "
                   "fetch is given raw and x 2 (gfx950: FETCH_SIZE reports half "
                   "the bytes of 16-byte-per-lane reads; refine_kernel and fullsearch2_kernel read 16 bytes per lane, cost_kernel "
                   "and me_kernel mostly 4), hbm_bytes = corrected fetch + write for the 16-byte kernels, raw fetch + write otherwise",
           "workload": run}
wide = {"refine_kernel", "fullsearch2_kernel"}
for k in sorted({k for (k, _) in fe}):
    f_, w_ = fe[(k, "FETCH_SIZE")], wr.get((k, "WRITE_SIZE"), [0.0])
    per_step = 1
    fb, wb = mean(f_) * 1024 * per_step, mean(w_) * 1024 * per_step
    d = {"launches_per_step": per_step, "fetch_bytes_raw": fb, "fetch_bytes_x2": 2 * fb, "write_bytes": wb,
         "hbm_bytes": (2 * fb if k in wide else fb) + wb, "kernel_us_mean": mean(dur[k]) * per_step if k in dur else None}
    traffic[k + "_4k_detail"] = d
    traffic[k + "_4k"] = d["hbm_bytes"]
json.dump(traffic, open(os.path.join(DST, f"{tag}_traffic.json"), "w"), indent=1)
print("wrote", sorted(p for p in os.listdir(DST) if p.startswith(tag)))

# 4. the bench run's timed launches, leg by leg, from the kernel trace of the same run: the legs run one after the other
# (rings, single requests, motionEstimate, full search at 4K; then 1080p), so the last `steps` launches of a kernel in
# front of the next leg's first launch are the timed ones.  Compared with the HIP-event times
# bench.py printed in that run.
bt = list(csv.DictReader(open(find("stats/**/*kernel_trace.csv"))))
bt.sort(key=lambda r: int(r["Dispatch_Id"]))
# a motionEstimate launch is three kernels (me_route_kernel and me_big_kernel on the side stream beside me_kernel): one leg,
# timed by the union of their intervals
ME_FAMILY = {"me_route_kernel": "me_kernel", "me_big_kernel": "me_kernel", "me_team_kernel": "me_kernel", "me_rest_kernel": "me_kernel"}
seqk = [(ME_FAMILY.get(short(r["Kernel_Name"]), short(r["Kernel_Name"])), int(r["Start_Timestamp"]) * 1e-6, int(r["End_Timestamp"]) * 1e-6,
         short(r["Kernel_Name"])) for r in bt if ours(r["Kernel_Name"])]


def busy_ms(intervals):
    total, cur_s, cur_e = 0.0, None, None
    for s_, e_ in sorted(intervals):
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    return total + (cur_e - cur_s if cur_e is not None else 0.0)


line = json.loads(open(os.path.join(SRC, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
K = line["steps"]
# leg order of bench.py's three_legs: rings of 8, single requests, motionEstimate, full search, the subme-2 rings (two
# refine_kernel launches per step); the predInterSearch stage and the streamed sequence follow
order = [("refine_kernel", "refine_kernel", 1), ("cost_kernel", "cost_kernel", 1), ("me_kernel", "me_kernel", 5), ("fullsearch2_kernel", "fullsearch2_kernel", 1),
         ("refine_kernel_subme2", "refine_kernel", 2)]
# maximal runs of one kernel family; the legs are the first five consecutive runs of the expected families that are long
# enough to hold K timed steps (the set-up in front of them -- the integer searches and first rings that centre the
# subme-2 rings -- is a few launches of the same kernels)
runs, i0 = [], 0
for i_ in range(1, len(seqk) + 1):
    if i_ == len(seqk) or seqk[i_][0] != seqk[i0][0]:
        runs.append((seqk[i0][0], i0, i_))
        i0 = i_
first_run = next(r_ for r_ in range(len(runs) - len(order) + 1)
                 if all(runs[r_ + q][0] == order[q][1] and runs[r_ + q][2] - runs[r_ + q][1] >= K * min(order[q][2], 4) for q in range(len(order))))
legs, pos = {}, runs[first_run][1]
for i, (leg_name, k, per) in enumerate(order):
    nxt = order[i + 1][1] if i + 1 < len(order) else "inter_requests_kernel"  # (the predInterSearch stage's leg follows)
    start = pos
    while pos < len(seqk) and seqk[pos][0] == k:
        pos += 1
    block = [(s_, e_) for _, s_, e_, _ in seqk[start:pos]]
    if k == "me_kernel":
        # a launch is me_route_kernel and the three or four kernels behind it (the class of 8x8 cells only for long
        # lists): the timed steps start at the K-th last me_route_kernel
        routes = [i_ for i_, q in enumerate(seqk[start:pos]) if q[3] == "me_route_kernel"]
        timed = block[routes[-K]:]
        per = len(timed) / K
    else:
        timed = block[-K * per:]
    legs[leg_name] = {"kernels_in_leg": len(block), "timed_kernels": len(timed), "kernels_per_step": per, "trace_ms_per_step": busy_ms(timed) / K}
    assert seqk[pos][0] == nxt, (leg_name, seqk[pos][0])
legs["refine_kernel"]["bench_kernel_ms"] = line["roofline"]["kernel_ms"]
legs["cost_kernel"]["bench_kernel_ms"] = line["satd_single_requests"]["roofline"]["kernel_ms"]
legs["me_kernel"]["bench_kernel_ms"] = line["me_stage"]["kernel_ms_per_frame"]
legs["fullsearch2_kernel"]["bench_kernel_ms"] = line["fullsearch"]["roofline"]["kernel_ms"]
legs["refine_kernel_subme2"]["bench_kernel_ms"] = line["satd_rings_subme2"]["roofline"]["kernel_ms"]
for k in legs:
    legs[k]["ratio_events_over_trace"] = legs[k]["bench_kernel_ms"] / legs[k]["trace_ms_per_step"]
json.dump({"note": "4K legs of `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline`: kernel time per step from the "
                   "trace's own timestamps (timed launches only) beside the HIP-event time bench.py printed in the same run "
                   "(events bracket the launches, so they also hold the gaps between them)", "steps": K, "legs": legs},
          open(os.path.join(DST, f"{tag}_bench_legs_from_trace.json"), "w"), indent=1)
print(json.dumps(legs, indent=1))
