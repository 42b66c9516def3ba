import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    if not os.path.exists(pyoracle.oracle_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return pyoracle.load_oracle()


@pytest.fixture(scope="session")
def small_seq():
    """three 416x240 padded planes of a synthetic sequence (plane ids 0, 1, 2)"""
    from x265_amd import synth
    seq = synth.Sequence(416, 240, seed=7, n_patches=3)
    return [synth.Plane(seq.frame(i)) for i in range(3)]


@pytest.fixture(scope="session")
def costtabs(oracle):
    return {q: oracle.bitcost_table(q) for q in range(20, 45)}


@pytest.fixture(scope="session")
def gpu_ctx(small_seq, costtabs):
    from x265_amd import lib
    ctx = lib.Context(0)
    for i, p in enumerate(small_seq):
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
    for q, t in costtabs.items():
        ctx.costtab_upload(q, t)
    yield ctx
    ctx.close()


def random_me_jobs(rng, n, method, subme, W, H, dtype, search_range, parts, meranges=(4, 8, 16, 57), planes=(1, (0, 2))):
    jobs = np.zeros(n, dtype)
    for i in range(n):
        part = int(rng.integers(1, 25))
        w, h = parts[part]
        x = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        j = jobs[i]
        j["fenc_plane"] = planes[0]
        j["ref_plane"] = int(rng.choice(planes[1]))
        j["pu_x"], j["pu_y"], j["pu_w"], j["pu_h"] = x, y, w, h
        j["method"], j["subme"] = method, subme
        mer = int(rng.choice(meranges))
        j["merange"] = mer
        j["qp"] = int(rng.integers(20, 45))
        mode = rng.integers(4)
        if mode == 0:
            px, py = 0, 0
        elif mode == 1:
            px, py = int(rng.integers(-40, 40)), int(rng.integers(-40, 40))
        elif mode == 2:
            px, py = int(rng.integers(-400, 400)), int(rng.integers(-300, 300))
        else:
            px, py = 4 * int(rng.integers(-8, 8)), 4 * int(rng.integers(-8, 8))
        j["qmvp_x"], j["qmvp_y"] = px, py
        # clipMv is relative to the block searched (TComDataCU.cpp:2502-2514), which keeps every candidate inside
        # the reference's margins; ranges taken relative to a CTU origin would let blocks low in the CTU leave them
        lox, loy, hix, hiy = search_range(np.int64(px), np.int64(py), mer, x, y, W, H)
        j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"] = lox, loy, hix, hiy
        nm = int(rng.integers(0, 6))
        j["num_mvc"] = nm
        for c in range(nm):
            j["mvc"][2 * c] = px + int(rng.integers(-30, 30)) if rng.integers(3) else 0
            j["mvc"][2 * c + 1] = py + int(rng.integers(-30, 30)) if rng.integers(3) else 0
    return jobs


def random_lowres_jobs(rng, n, lw, ll, dtype, fenc_plane=4, ref_planes=(0, 8), methods=(0, 1, 2, 3), qps=(12, 20, 33)):
    """motionEstimate calls against a Lowres reference the way the lookahead issues them (8x8 CUs, bounds of
    slicetype.cpp:1539-1543), with every search method and sub-pel level"""
    wcu, hcu = lw // 8, ll // 8
    jobs = np.zeros(n, dtype)
    for j in jobs:
        cux, cuy = int(rng.integers(0, wcu)), int(rng.integers(0, hcu))
        j["fenc_plane"], j["ref_plane"] = fenc_plane, int(rng.choice(ref_planes))
        j["pu_x"], j["pu_y"], j["pu_w"], j["pu_h"] = 8 * cux, 8 * cuy, 8, 8
        j["method"] = int(rng.choice(methods)) | 0x80
        j["subme"] = int(rng.integers(0, 8))
        j["mvmin_x"], j["mvmin_y"] = -cux * 8 - 8, -cuy * 8 - 8
        j["mvmax_x"], j["mvmax_y"] = (wcu - cux - 1) * 8 + 8, (hcu - cuy - 1) * 8 + 8
        j["qmvp_x"], j["qmvp_y"] = int(rng.integers(-24, 24)), int(rng.integers(-24, 24))
        j["merange"] = int(rng.choice([4, 16]))
        j["qp"] = int(rng.choice(qps))
        nm = int(rng.integers(0, 5))
        j["num_mvc"] = nm
        for c in range(nm):
            j["mvc"][2 * c] = int(rng.integers(-30, 30)) if rng.integers(3) else 0
            j["mvc"][2 * c + 1] = int(rng.integers(-30, 30)) if rng.integers(3) else 0
    return jobs


def random_bidir_reqs(rng, n, W, H, dtype, parts, planes=(1, 0, 2), metrics=(0, 1), flags=(0,)):
    """blocks of every partition against the average of two predictions at random quarter-pel vectors"""
    reqs = np.zeros(n, dtype)
    for r in reqs:
        w, h = parts[int(rng.integers(0, 25))]
        r["fenc_plane"], r["ref0_plane"], r["ref1_plane"] = planes[0], planes[1], int(rng.choice(planes[1:]))
        r["pu_x"], r["pu_y"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        r["pu_w"], r["pu_h"], r["metric"] = w, h, int(rng.choice(metrics)) | int(rng.choice(flags))
        for k in ("qmv0_x", "qmv0_y", "qmv1_x", "qmv1_y"):
            r[k] = int(rng.integers(-90, 90)) if rng.integers(4) else 4 * int(rng.integers(-20, 20))
    return reqs


def random_cost_reqs(rng, n, W, H, dtype, parts, metrics=(0, 1), planes=(1, (0, 2))):
    """blocks of every partition at random quarter-pel vectors with a metric drawn from `metrics`"""
    reqs = np.zeros(n, dtype)
    for r in reqs:
        w, h = parts[int(rng.integers(0, 25))]
        r["fenc_plane"], r["ref_plane"] = planes[0], int(rng.choice(planes[1]))
        r["pu_x"], r["pu_y"], r["pu_w"], r["pu_h"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h
        r["metric"] = int(rng.choice(metrics))
        r["qmv_x"] = int(rng.integers(-90, 90)) if rng.integers(4) else 4 * int(rng.integers(-20, 20))
        r["qmv_y"] = int(rng.integers(-90, 90)) if rng.integers(4) else 4 * int(rng.integers(-20, 20))
    return reqs


def random_pred_reqs(rng, W, H, dtype, parts, dst_plane=5, refs=(0, 2), cell=64, weighted_bi=False):
    """one prediction request per 64x64 cell of the picture (destination blocks of a batch must not overlap):
    every partition, uni- and bi-directional and weighted uni-directional (ref1_plane -2: w0 in qmv1_x, offset in
    qmv1_y), all fractions; with `weighted_bi` half of the bi-directional ones are weighted (xWeightedPredictionBi)"""
    cells = [(cx, cy) for cy in range(0, H - cell + 1, cell) for cx in range(0, W - cell + 1, cell)]
    reqs = np.zeros(len(cells), dtype)
    for r, (cx, cy) in zip(reqs, cells):
        w, h = parts[int(rng.integers(0, 25))]
        r["dst_plane"], r["ref0_plane"] = dst_plane, int(rng.choice(refs))
        r["ref1_plane"] = int(rng.choice(refs)) if rng.integers(3) else -1
        r["pu_x"], r["pu_y"] = cx + int(rng.integers(0, cell - w + 1)), cy + int(rng.integers(0, cell - h + 1))
        r["pu_w"], r["pu_h"] = w, h
        for k in ("qmv0_x", "qmv0_y", "qmv1_x", "qmv1_y"):
            r[k] = int(rng.integers(-90, 90)) if rng.integers(4) else 4 * int(rng.integers(-20, 20))
        if r["ref1_plane"] < 0 and rng.integers(2):
            # motionCompensation's getUseWP branch; unit weights (what an unweighted reference of such a slice
            # gets), the weights x265's analysis produces (denominator 7, weight near 128) and the whole range
            kind = int(rng.integers(3))
            denom = 7 if kind == 1 else int(rng.integers(0, 8))
            r["ref1_plane"], r["log2_denom"] = -2, denom
            r["qmv1_x"] = 1 << denom if kind == 0 else int(rng.integers(96, 160)) if kind == 1 else int(rng.integers(-128, 256))
            r["qmv1_y"] = 0 if kind == 0 else int(rng.integers(-20, 20)) if kind == 1 else int(rng.integers(-128, 128))
        elif weighted_bi and r["ref1_plane"] >= 0 and rng.integers(2):
            # xPredInterBi's getWPBiPred branch: unit weights, weights near the unit ones, the whole range
            kind = int(rng.integers(3))
            denom = 6 if kind == 1 else int(rng.integers(0, 8))
            r["weighted_bi"], r["log2_denom"] = 1, denom
            for w_, o_ in (("wp_w0", "wp_o0"), ("wp_w1", "wp_o1")):
                r[w_] = 1 << denom if kind == 0 else int(rng.integers(40, 90)) if kind == 1 else int(rng.integers(-128, 256))
                r[o_] = 0 if kind == 0 else int(rng.integers(-20, 20)) if kind == 1 else int(rng.integers(-128, 128))
    return reqs


def random_lowres_bidir_reqs(rng, n, lw, ll, dtype, fenc_plane=4, ref_planes=(0, 8), metric=1):
    """the lookahead's bidirectional cost (slicetype.cpp:1585-1603): 8x8 CUs of a lowres picture against two Lowres
    references at quarter-pel vectors inside the lookahead's search bounds; metric carries the lowres flag"""
    wcu, hcu = lw // 8, ll // 8
    reqs = np.zeros(n, dtype)
    for r in reqs:
        cux, cuy = int(rng.integers(0, wcu)), int(rng.integers(0, hcu))
        r["fenc_plane"], r["ref0_plane"], r["ref1_plane"] = fenc_plane, int(rng.choice(ref_planes)), int(rng.choice(ref_planes))
        r["pu_x"], r["pu_y"], r["pu_w"], r["pu_h"] = 8 * cux, 8 * cuy, 8, 8
        r["metric"] = metric | 0x80
        lox, loy = 4 * (-cux * 8 - 8), 4 * (-cuy * 8 - 8)
        hix, hiy = 4 * ((wcu - cux - 1) * 8 + 8), 4 * ((hcu - cuy - 1) * 8 + 8)
        for kx, ky in (("qmv0_x", "qmv0_y"), ("qmv1_x", "qmv1_y")):
            r[kx] = int(np.clip(rng.integers(-70, 70), lox, hix)) if rng.integers(5) else 0
            r[ky] = int(np.clip(rng.integers(-70, 70), loy, hiy)) if rng.integers(5) else 0
    return reqs


def random_inter_units(rng, n_units, W, H, job_dtype, pu_dtype, parts, qps=(20, 45)):
    """prediction units for x265_hip_inter_decide_batch: (jobs, units); a unit's iterations are jobs of one block, one to
    three references in list 0 and none to two in list 1"""
    sel_rows = [(3, 3, 5), (1, 3, 5), (0, 0, 3), (5, 7, 7), (7, 5, 7), (6, 6, 6), (0, 2, 3), (5, 5, 7), (3, 0, 0)]
    jobs, pus = [], []
    for u in range(n_units):
        w, h = parts[int(rng.integers(1, 25))]
        cu = max(w, h) if max(w, h) in (8, 16, 32, 64) else 64
        cx = int(rng.integers(0, (W - cu) // 8 + 1)) * 8
        cy = int(rng.integers(0, (H - cu) // 8 + 1)) * 8
        px, py = cx + int(rng.integers(0, (cu - w) // 4 + 1)) * 4, cy + int(rng.integers(0, (cu - h) // 4 + 1)) * 4
        n0, n1 = int(rng.integers(1, 4)), int(rng.integers(0, 3))
        p = np.zeros(1, pu_dtype)[0]
        p["first_job"], p["num_ref"] = len(jobs), (n0, n1)
        p["list_sel_bits"] = sel_rows[int(rng.integers(len(sel_rows)))]
        p["no_bidir"] = int(n1 == 0 or rng.integers(4) == 0)
        p["lambda_sad"] = int(rng.choice([64, 1024, 4771, 9000, 40000]))
        pus.append(p)
        method, subme, qp = int(rng.integers(0, 2)), int(rng.integers(0, 8)), int(rng.integers(qps[0], qps[1]))
        for k in range(n0 + n1):
            j = np.zeros(1, job_dtype)[0]
            j["fenc_plane"], j["ref_plane"] = 1, int(rng.choice([0, 2]))
            j["pred_plane"] = j["ref_plane"] if rng.integers(3) else int(rng.choice([0, 2]))
            j["cu_x"], j["cu_y"], j["pu_x"], j["pu_y"], j["pu_w"], j["pu_h"] = cx, cy, px, py, w, h
            j["method"], j["subme"], j["qp"] = method, subme, qp
            j["merange"], j["ref_lag"] = int(rng.choice([4, 8, 16])), H if rng.integers(3) else 16
            j["num_amvp"] = 1 + int(rng.integers(0, 4) > 0)
            kind = int(rng.integers(4))
            base = (0, 0) if kind == 0 else (int(rng.integers(-40, 40)), int(rng.integers(-40, 40))) if kind < 3 else \
                (int(rng.integers(-2500, 2500)), int(rng.integers(-1500, 1500)))
            j["amvp"][0], j["amvp"][1] = base
            j["amvp"][2], j["amvp"][3] = (base[0] + int(rng.integers(-12, 13)), base[1] + int(rng.integers(-12, 13))) if rng.integers(4) else (0, 0)
            jobs.append(j)
    return np.array(jobs, job_dtype), np.array(pus, pu_dtype)


def wpbi_fixture():
    """[(params, src (2, 64, 64) int16, dst (64, 64) uint8)] of tests/golden/wpbi.npz"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wpbi.npz"))
    out, so, do = [], 0, 0
    for hd in g["params"]:
        w, h = int(hd[0]), int(hd[1])
        src, dst = np.zeros((2, 64, 64), np.int16), np.zeros((64, 64), np.uint8)
        src[:, :h, :w] = g["src"][so:so + 2 * w * h].reshape(2, h, w)
        dst[:h, :w] = g["dst"][do:do + w * h].reshape(h, w)
        so, do = so + 2 * w * h, do + w * h
        out.append(([int(x) for x in hd], src, dst))
    return out


def random_merge_pus(rng, n, W, H, dtype, parts, fenc_plane=1, refs=(0, 2)):
    """prediction units with merge candidates of every kind xMergeEstimation meets: one list or both, the same plane and
    vector on both lists (xCheckIdenticalMotion), slices without / with weighted prediction (P) / with weighted
    bi-prediction (B), vectors past the picture edge (clipMv), candidates at the frame threads' limit (passed over), one
    to five candidates"""
    pus = np.zeros(n, dtype)
    for p in pus:
        w, h = parts[int(rng.integers(1, 25))]  # (no 4x4 inter partition)
        cu = 8
        while cu < max(w, h):
            cu *= 2
        p["fenc_plane"] = fenc_plane
        p["pu_w"], p["pu_h"] = w, h
        # the unit is the first or the second partition of a CU on the CU grid (clipMv keeps the block within reach of the
        # padding only for a unit inside its CU)
        p["cu_x"], p["cu_y"] = cu * int(rng.integers(0, W // cu)), cu * int(rng.integers(0, H // cu))
        p["pu_x"], p["pu_y"] = int(p["cu_x"]) + (cu - w) * int(rng.integers(0, 2)), int(p["cu_y"]) + (cu - h) * int(rng.integers(0, 2))
        p["num_cand"] = int(rng.integers(1, 6))
        p["weighted"] = int(rng.integers(0, 3))
        p["mvy_limit"] = 32767 if rng.integers(3) else 4 * 58
        p["lambda_sad"] = int(rng.integers(200, 12000))
        for c in p["cand"][:int(p["num_cand"])]:
            kind = int(rng.integers(0, 4)) if p["weighted"] != 1 else 0  # 0 list 0, 1 list 1, 2 both, 3 identical
            far = rng.integers(8) == 0
            for l in range(2):
                c["mv"][2 * l] = int(rng.integers(-4 * W, 4 * W)) if far else int(rng.integers(-120, 120))
                c["mv"][2 * l + 1] = int(rng.integers(-4 * H, 4 * H)) if far else int(rng.integers(-120, 260 if p["mvy_limit"] < 32767 else 120))
                c["wp_w"][l], c["wp_o"][l] = int(rng.integers(-128, 256)) if rng.integers(4) == 0 else int(rng.integers(20, 90)), int(rng.integers(-30, 30))
                c["log2_denom"][l] = int(rng.integers(0, 8))
            c["ref_plane"] = [int(rng.choice(refs)) if kind in (0, 2, 3) else -1, int(rng.choice(refs)) if kind in (1, 2, 3) else -1]
            if kind == 3:
                c["ref_plane"][1] = c["ref_plane"][0]
                c["mv"][2:] = c["mv"][:2]
    return pus
