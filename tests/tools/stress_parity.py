"""One-off stress run: many random requests through every batched entry point against the oracle (16 threads).
Prints the number of mismatches per family; exits non-zero if there is any."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from conftest import random_bidir_reqs, random_cost_reqs, random_inter_units, random_lowres_jobs, random_me_jobs, random_pred_reqs  # noqa: E402
from oracle import pyoracle
from x265_amd import abi, lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
O = pyoracle.load_oracle()
seq = synth.Sequence(832, 480, seed=11, n_patches=5)
planes = [synth.Plane(seq.frame(i)) for i in range(3)]
flat = synth.Plane(np.full((480, 832), 77, np.uint8))
noise = synth.Plane(np.random.default_rng(1).integers(0, 256, (480, 832), dtype=np.uint8))
planes += [flat, noise]
pl = [p.as_tuple() for p in planes]
ctx = lib.Context(0)
for i, p in enumerate(planes):
    ctx.plane_upload(i, p.buf, p.stride, p.origin)
tabs = {q: O.bitcost_table(q) for q in range(0, 52)}
for q, t in tabs.items():
    ctx.costtab_upload(q, t)
W, H = 832, 480
bad_total = 0


def report(name, got, want, recs):
    global bad_total
    bad = np.nonzero(got != want)[0]
    bad_total += len(bad)
    print(f"{name:34s} n={len(recs):7d} mismatches={len(bad)}", flush=True)
    if len(bad):
        print("   first:", recs[bad[0]], got[bad[0]], want[bad[0]], flush=True)


rng = np.random.default_rng(2024)
for method in range(5):
    n = N if method < 4 else 200
    mer = (4, 8, 16, 32, 57) if method < 4 else (2, 5)
    jobs = []
    for subme in range(8):
        for planes_sel in ((1, (0, 2)), (1, (3,)), (4, (0, 4)), (3, (3,))):  # natural, flat ref, noise, flat vs flat
            j = random_me_jobs(rng, max(1, n // 32), method, subme, W, H, abi.ME_JOB, synth.search_range, abi.PARTS,
                               meranges=mer, planes=planes_sel)
            j["qp"] = rng.integers(0, 52, len(j))
            jobs.append(j)
    jobs = np.concatenate(jobs)
    t0 = time.time()
    want = O.me_batch(pl, tabs, jobs, nthreads=16)
    got = ctx.me_batch(jobs)
    report(f"me_batch method {method} ({time.time() - t0:.1f} s)", got, want, jobs)

low, size = [], None
for p in planes[:3]:
    four, size = O.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)
    low += four
for i, p in enumerate(planes[:3]):
    ctx.lowres_init(i, p.w, p.h, 8 + 4 * i)
jobs = random_lowres_jobs(rng, N, size[0], size[1], abi.ME_JOB, fenc_plane=4, ref_planes=(0, 8))
want = O.me_batch(low, tabs, jobs, nthreads=16)
gj = jobs.copy()
gj["fenc_plane"] += 8
gj["ref_plane"] += 8
report("me_batch lowres", ctx.me_batch(gj), want, jobs)

reqs = np.zeros(4 * N, abi.COST_REQ)
for r in reqs:
    w, h = abi.PARTS[int(rng.integers(0, 25))]
    r["fenc_plane"], r["ref_plane"] = int(rng.choice([1, 4])), int(rng.choice([0, 2, 3, 4]))
    r["pu_x"], r["pu_y"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
    r["pu_w"], r["pu_h"], r["metric"] = w, h, int(rng.integers(0, 2))
    r["qmv_x"], r["qmv_y"] = int(rng.integers(-200, 200)), int(rng.integers(-200, 200))
report("cost_batch", ctx.cost_batch(reqs), O.cost_batch(pl, reqs, nthreads=16), reqs)

br = random_bidir_reqs(rng, 2 * N, W, H, abi.BIDIR_REQ, abi.PARTS, planes=(1, 0, 2, 4), metrics=(0, 1, 2, 3), flags=(0, abi.MC_AVG))
report("bidir_cost_batch", ctx.bidir_cost_batch(br), O.bidir_cost_batch(pl, br), br)
from conftest import random_lowres_bidir_reqs
lbr = np.concatenate([random_lowres_bidir_reqs(rng, N // 2, size[0], size[1], abi.BIDIR_REQ, metric=m) for m in (0, 1)])
want = O.bidir_cost_batch(low, lbr)
glbr = lbr.copy()
for k in ("fenc_plane", "ref0_plane", "ref1_plane"):
    glbr[k] += 8
report("bidir_cost_batch lowres", ctx.bidir_cost_batch(glbr), want, lbr)
# mode-decision distortions: all four comparisons, natural / flat / noise planes
dr = random_cost_reqs(rng, 4 * N, W, H, abi.COST_REQ, abi.PARTS, metrics=(0, 1, 2, 2, 3, 3), planes=(1, (0, 2, 3, 4)))
dr["fenc_plane"] = rng.choice([1, 4], len(dr))
report("dist_batch", ctx.dist_batch(dr), O.cost_batch(pl, dr, nthreads=16), dr)
# motion-compensated prediction: batches of non-overlapping blocks into plane 5, the whole plane compared
dst = np.full_like(planes[0].buf, 0x5A)
ctx.plane_upload(5, dst, planes[0].stride, planes[0].origin)
npred, nbad = 0, 0
for _ in range(max(10, N // 200)):
    pr = random_pred_reqs(rng, W, H, abi.PRED_REQ, abi.PARTS, dst_plane=5, refs=(0, 2, 4), weighted_bi=True)
    ctx.predict_batch(pr)
    O.predict_batch(pl + [(dst, planes[0].origin, planes[0].stride)], pr)
    nbad += int((ctx.plane_download(5, dst.size) != dst).sum())
    npred += len(pr)
bad_total += nbad
print(f"{'predict_batch (bytes of the plane)':34s} n={npred:7d} mismatches={nbad}", flush=True)
# CTU full search: random windows (every width / height up to 132), predictors and QPs; natural, flat, noise
from concurrent.futures import ThreadPoolExecutor
nctu = max(50, N // 400)
cj = np.zeros(nctu, lib.CTU_JOB)
for j in cj:
    nx, ny = int(rng.integers(1, 133)), int(rng.integers(1, 133))
    if rng.integers(3) == 0:
        nx, ny = 129, 129
    j["fenc_plane"], j["ref_plane"] = int(rng.choice([1, 4, 3])), int(rng.choice([0, 2, 3, 4]))
    j["ctu_x"], j["ctu_y"] = 64 * int(rng.integers(1, W // 64 - 1)), 64 * int(rng.integers(1, H // 64 - 1))
    lox, loy = -int(rng.integers(0, min(nx, 65))), -int(rng.integers(0, min(ny, 65)))
    lox, loy = max(lox, -(nx - 1)), max(loy, -(ny - 1))
    lox, loy = max(lox, 64 - (nx - 1)) if nx - 1 + lox > 64 else lox, max(loy, 64 - (ny - 1)) if ny - 1 + loy > 64 else loy
    j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"] = lox, loy, lox + nx - 1, loy + ny - 1
    j["qmvp_x"], j["qmvp_y"] = int(rng.integers(-300, 300)), int(rng.integers(-300, 300))
    j["qp"] = int(rng.integers(0, 52))
got = ctx.fullsearch(cj)


def one(k):
    j = cj[k]
    return O.ctu_fullsearch(pl[int(j["fenc_plane"])], pl[int(j["ref_plane"])], tabs[int(j["qp"])], int(j["ctu_x"]), int(j["ctu_y"]),
                            (int(j["mvmin_x"]), int(j["mvmin_y"])), (int(j["mvmax_x"]), int(j["mvmax_y"])),
                            (int(j["qmvp_x"]), int(j["qmvp_y"])))


with ThreadPoolExecutor(16) as ex:
    want = np.stack(list(ex.map(one, range(nctu))))
report("fullsearch (CTUs x 85 PUs)", got.reshape(-1), want.reshape(-1).view(got.dtype), np.repeat(cj, 85))
# ---- round 2 entry points -----------------------------------------------------------------------------------------
# sub-pel rings: every partition, both rings, 4 / 8 points, with / without centre, sad / satd, any centre fraction;
# as a mixed list (sampled batches) and as runs of one size (batches cut per run; x265_hip_refine_sections_device)
def ring_reqs(n, parts):
    r = np.zeros(n, abi.REFINE_REQ)
    pi = rng.integers(0, len(parts), n)
    wh = np.array(parts)[pi]
    r["pu_w"], r["pu_h"] = wh[:, 0], wh[:, 1]
    r["fenc_plane"], r["ref_plane"] = rng.choice([1, 4], n), rng.choice([0, 2, 3, 4], n)
    r["pu_x"], r["pu_y"] = rng.integers(0, W - wh[:, 0] + 1), rng.integers(0, H - wh[:, 1] + 1)
    r["metric"] = rng.integers(0, 2, n)
    r["pattern"] = rng.choice([abi.RING_QPEL, abi.RING_HPEL], n) | (abi.RING_8 * rng.integers(0, 2, n)) | (abi.RING_CENTRE * rng.integers(0, 2, n))
    frac = rng.integers(0, 4, n) > 0
    r["qmv_x"] = np.where(frac, rng.integers(-200, 200, n), 4 * rng.integers(-50, 50, n))
    r["qmv_y"] = np.where(rng.integers(0, 4, n) > 0, rng.integers(-200, 200, n), 4 * rng.integers(-50, 50, n))
    return r


def ring_want(r):
    creqs, used = abi.refine_as_cost_reqs(r)
    w = O.cost_batch(pl, creqs.reshape(-1), nthreads=16).reshape(-1, abi.RING_COSTS)
    return np.where(used, w, 0x7fffffff)


rr = ring_reqs(N, abi.PARTS[1:])
got, want = ctx.refine_batch(rr), ring_want(rr)
report("refine_batch mixed sizes (rings)", (got != want).any(axis=1), np.zeros(len(rr), bool), rr)
from x265_amd.hipmem import DeviceBuffer
runs = [(64, 64, N // 40), (32, 32, N // 10), (16, 16, N // 3), (8, 8, N)]
rs = np.concatenate([ring_reqs(n, [(w, h)]) for w, h, n in runs])
d, o = DeviceBuffer.from_array(rs), DeviceBuffer(36 * len(rs))
ctx.refine_sections_device(d.ptr, [(n, w, h) for w, h, n in runs], o.ptr)
ctx.sync()
got, want = o.download(np.int32).reshape(-1, 9), ring_want(rs)
report("refine_sections_device (rings)", (got != want).any(axis=1), np.zeros(len(rs), bool), rs)

# predInterSearch's (list, reference) iteration: AMVP pick + search window + motionEstimate, all methods but FULL
ij = np.zeros(N, abi.INTER_JOB)
for j in ij:
    w, h = abi.PARTS[int(rng.integers(1, 25))]
    cu = max(8, 1 << int(np.ceil(np.log2(max(w, h)))))
    cx, cy = int(rng.integers(0, (W - cu) // 8 + 1)) * 8, int(rng.integers(0, (H - cu) // 8 + 1)) * 8
    j["fenc_plane"], j["ref_plane"] = int(rng.choice([1, 4])), int(rng.choice([0, 2, 3, 4]))
    j["pred_plane"] = j["ref_plane"] if rng.integers(3) else int(rng.choice([0, 2]))
    j["cu_x"], j["cu_y"] = cx, cy
    j["pu_x"], j["pu_y"] = cx + int(rng.integers(0, (cu - w) // 4 + 1)) * 4, cy + int(rng.integers(0, (cu - h) // 4 + 1)) * 4
    j["pu_w"], j["pu_h"] = w, h
    j["method"], j["subme"] = int(rng.integers(0, 4)), int(rng.integers(0, 8))
    j["merange"] = int(rng.choice([4, 8, 16, 57]))
    j["ref_lag"] = H if rng.integers(3) else int(rng.choice([8, 16, 57]))
    j["qp"] = int(rng.integers(0, 52))
    j["num_amvp"] = 1 + int(rng.integers(0, 4) > 0)
    kind = int(rng.integers(4))
    base = (0, 0) if kind == 0 else (int(rng.integers(-40, 40)), int(rng.integers(-40, 40))) if kind == 1 else \
        (int(rng.integers(-4000, 4000)), int(rng.integers(-2500, 2500)))
    j["amvp"][0], j["amvp"][1] = base
    j["amvp"][2], j["amvp"][3] = (base[0] + int(rng.integers(-12, 13)), base[1] + int(rng.integers(-12, 13))) if rng.integers(3) else base
    nm = int(rng.integers(0, 6))
    j["num_mvc"] = nm
    for c in range(nm):
        j["mvc"][2 * c] = base[0] + int(rng.integers(-30, 30)) if rng.integers(3) else 0
        j["mvc"][2 * c + 1] = base[1] + int(rng.integers(-30, 30)) if rng.integers(3) else 0
got = ctx.inter_search_batch(ij, W, H)
want = O.inter_search_batch(pl, tabs, ij, W, H)
report("inter_search_batch", got, want.view(got.dtype), ij)

# the rest of predInterSearch per prediction unit (round 3): bits, xCheckBestMVP, the bidirectional candidate, the choice
uj, up = random_inter_units(rng, max(1, N // 4), W, H, abi.INTER_JOB, abi.INTER_PU, abi.PARTS, qps=(0, 52))
res, got = ctx.inter_decide_batch(uj, up, W, H)
wres = O.inter_search_batch(pl, tabs, uj, W, H)
want = O.inter_decide_batch(pl, tabs, uj, wres, up, W, H)
report("inter_decide_batch (iterations)", res, wres.view(res.dtype), uj)
report("inter_decide_batch (units)", np.array([g.tobytes() for g in got]), np.array([w_.tobytes() for w_ in want]), up)

# xMergeEstimation per prediction unit and its measurement on single requests (round 3); the source picture is plane 1
from conftest import random_merge_pus
mp = random_merge_pus(rng, max(1, N // 2), W, H, abi.MERGE_PU, abi.PARTS, fenc_plane=1, refs=(0, 2, 3, 4))
got, want = ctx.merge_estimate_batch(mp, W, H), O.merge_estimate_batch(pl, mp, W, H)
report("merge_estimate_batch", np.array([g.tobytes() for g in got]), np.array([w_.tobytes() for w_ in want]), mp)
mr = np.concatenate([random_pred_reqs(rng, W, H, abi.PRED_REQ, abi.PARTS, dst_plane=int(rng.choice([1, 4])), refs=(0, 2, 3), weighted_bi=True)
                     for _ in range(max(4, N // 400))])
report("mc_satd_batch", ctx.mc_satd_batch(mr), O.mc_satd_batch(pl, mr), mr)

# one long diamond / hexagon list (round 3: lists of 262,144 calls and more search their 32-wide blocks in teams of 8x8
# cells as well), through the device-array entry point with the record guard on
from x265_amd.hipmem import DeviceBuffer
jobs = np.concatenate([random_me_jobs(rng, max(300000, N) // 16, m, sm, W, H, abi.ME_JOB, synth.search_range, abi.PARTS, meranges=(4, 8, 16, 57))
                       for m in (0, 1) for sm in range(8)])
jobs["qp"] = rng.integers(0, 52, len(jobs))
t0 = time.time()
want = O.me_batch(pl, tabs, jobs, nthreads=16)
d_j, d_o = DeviceBuffer.from_array(jobs), DeviceBuffer(8 * len(jobs))
ctx.check_records(2)
ctx.me_batch_device(d_j.ptr, len(jobs), d_o.ptr)
assert ctx.guard_status() == (0, 0xFFFFFFFF)
ctx.check_records(0)
report(f"me_batch_device, one list, guard on ({time.time() - t0:.1f} s)", d_o.download(abi.ME_RESULT), want, jobs)

# whole-frame lookahead searches: every ordered pair of the three natural pictures' lowres planes, a few settings
lj = []
for a in range(3):
    for b in range(3):
        if a != b:
            for (mer, method, subme, qp) in ((16, abi.HEX, 1, 12), (8, abi.DIA, 2, 30), (32, abi.UMH, 1, 12), (16, abi.STAR, 3, 40)):
                lj.append((4 * a, 4 * b, mer, method, subme, qp))
lk = np.zeros(len(lj), abi.LOOKAHEAD_JOB)
for k, (f, r, mer, method, subme, qp) in enumerate(lj):
    lk[k]["fenc_plane"], lk[k]["ref_plane"], lk[k]["merange"] = f, r, mer
    lk[k]["method"], lk[k]["subme"], lk[k]["qp"] = method, subme, qp
LW, LH = size[0] // 8, size[1] // 8
want = O.lookahead_batch(low, tabs, lk, LW, LH, nthreads=16)
gk = lk.copy()
gk["fenc_plane"] += 8
gk["ref_plane"] += 8
got = ctx.lookahead_batch(gk, LW, LH)
report(f"lookahead_batch ({len(lk)} x {LW * LH} CUs)", got.reshape(-1), want.reshape(-1).view(got.dtype), np.repeat(lk, LW * LH))
ctx.close()
print("TOTAL MISMATCHES", bad_total)
sys.exit(1 if bad_total else 0)
