"""CPU differential test: oracle restatement vs the real reference build, random inputs.
Skipped where oracle/_ref/libx265ref.so is absent (it is built only where /root/reference exists)."""
import numpy as np
import pytest

from conftest import random_bidir_reqs, random_lowres_bidir_reqs, random_lowres_jobs, random_me_jobs, random_pred_reqs
from oracle import pyoracle

pytestmark = pytest.mark.skipif(not pyoracle.have_ref(), reason="reference build not present")


@pytest.fixture(scope="module")
def ref():
    return pyoracle.load_ref()


def test_pixelcmp_random(oracle, ref):
    rng = np.random.default_rng(1)
    n = 64 * 200 + 4000
    bufs = [rng.integers(0, 256, n, dtype=np.uint8), np.zeros(n, np.uint8), np.full(n, 255, np.uint8)]
    for part in range(25):
        for _ in range(12):
            a, b = bufs[rng.integers(3)], bufs[rng.integers(3)]
            fo, ro = int(rng.integers(0, 16)) * 16, int(rng.integers(0, 3000))
            for rs in (64, 59, 100):
                assert oracle.sad(part, a, fo, 64, b, ro, rs) == ref.sad(part, a, fo, 64, b, ro, rs)
                assert oracle.satd(part, a, fo, 64, b, ro, rs) == ref.satd(part, a, fo, 64, b, ro, rs)
                assert oracle.sa8d_inter(part, a, fo, 64, b, ro, rs) == ref.sa8d_inter(part, a, fo, 64, b, ro, rs)
                assert oracle.sse_pp(part, a, fo, 64, b, ro, rs) == ref.sse_pp(part, a, fo, 64, b, ro, rs)
                assert np.array_equal(oracle.sad_x4(part, a, fo, b, [ro, ro + 1, ro + 2, ro + 3], rs),
                                      ref.sad_x4(part, a, fo, b, [ro, ro + 1, ro + 2, ro + 3], rs))


def test_bitcost_all_qp(oracle, ref):
    for q in range(70):
        assert np.array_equal(oracle.bitcost_table(q), ref.bitcost_table(q))


@pytest.mark.parametrize("method", range(5))
def test_motion_estimate_random(oracle, ref, small_seq, costtabs, method):
    from x265_amd import synth
    rng = np.random.default_rng(40 + method)
    planes = [p.as_tuple() for p in small_seq]
    W, H = small_seq[0].w, small_seq[0].h
    for subme in range(8):
        n, mer = (150, (4, 8, 16, 57)) if method != 4 else (8, (3, 6))
        jobs = random_me_jobs(rng, n, method, subme, W, H, pyoracle.ME_JOB, synth.search_range, pyoracle.PARTS, meranges=mer)
        a, b = oracle.me_batch(planes, costtabs, jobs), ref.me_batch(planes, costtabs, jobs)
        assert np.array_equal(a, b), (method, subme)


def test_motion_estimate_flat_and_extreme_content(oracle, ref):
    """ties everywhere (flat pictures) and saturated differences exercise the tie-break order"""
    from x265_amd import synth
    rng = np.random.default_rng(9)
    W, H = 192, 128
    pics = [np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8), np.zeros((H, W), np.uint8)]
    pics[2][::2, ::2] = 255
    planes = [synth.Plane(p) for p in pics]
    pl = [p.as_tuple() for p in planes]
    tabs = {q: oracle.bitcost_table(q) for q in range(20, 45)}
    for method in range(4):
        for fenc, refp in ((0, 0), (0, 1), (2, 0), (2, 2)):
            jobs = random_me_jobs(rng, 60, method, int(rng.integers(0, 8)), W, H, pyoracle.ME_JOB, synth.search_range,
                                  pyoracle.PARTS, planes=(fenc, (refp,)))
            assert np.array_equal(oracle.me_batch(pl, tabs, jobs), ref.me_batch(pl, tabs, jobs))


def test_lowres_planes_and_search(oracle, ref, small_seq):
    """frame_init_lowres_core + extendPicBorder, and motionEstimate against a Lowres (ref->isLowres)"""
    low_o, low_r, size = [], [], None
    for p in small_seq:
        a, size = oracle.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)
        b, _ = ref.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)
        for (x, _, _), (y, _, _) in zip(a, b):
            assert np.array_equal(x, y)
        low_o += a
        low_r += b
    rng = np.random.default_rng(5)
    for metric in (0, 1):  # the lookahead's bidirectional cost: lowresMC of two references, pixelavg_pp, satd (sad)
        br = random_lowres_bidir_reqs(rng, 1500, size[0], size[1], pyoracle.BIDIR_REQ, metric=metric)
        a, b = oracle.bidir_cost_batch(low_o, br), ref.bidir_cost_batch(low_r, br)
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, (br[bad[0]], a[bad[0]], b[bad[0]])
    jobs = random_lowres_jobs(rng, 1500, size[0], size[1], pyoracle.ME_JOB)
    tabs = {q: oracle.bitcost_table(q) for q in (12, 20, 33)}
    a, b = oracle.me_batch(low_o, tabs, jobs), ref.me_batch(low_r, tabs, jobs)
    bad = np.nonzero(a != b)[0]
    assert len(bad) == 0, (jobs[bad[0]], a[bad[0]], b[bad[0]])


def test_weighted_plane_and_border(oracle, ref, small_seq):
    """weight_pp with applyWeight's corrected arguments + extendPicBorder, several weights / offsets / denominators"""
    p = small_seq[0]
    for (w0, off, den) in ((64, 0, 6), (70, -3, 6), (37, 10, 5), (1, 0, 0), (200, -100, 7), (128, 40, 7)):
        a = oracle.weighted_plane(p.buf, p.origin, p.stride, p.w, p.h, w0, off, den)
        b = ref.weighted_plane(p.buf, p.origin, p.stride, p.w, p.h, w0, off, den)
        assert np.array_equal(a, b), (w0, off, den)


def test_bidir_cost(oracle, ref, small_seq):
    """two xPredInterLumaBlk predictions, pixelavg_pp, sad / satd: all partitions, all fraction pairs"""
    rng = np.random.default_rng(12)
    planes = [p.as_tuple() for p in small_seq]
    reqs = random_bidir_reqs(rng, 1500, small_seq[0].w, small_seq[0].h, pyoracle.BIDIR_REQ, pyoracle.PARTS)
    a, b = oracle.bidir_cost_batch(planes, reqs), ref.bidir_cost_batch(planes, reqs)
    bad = np.nonzero(a != b)[0]
    assert len(bad) == 0, (reqs[bad[0]], a[bad[0]], b[bad[0]])


def test_pixelavg_weight_lowres_primitives(oracle, ref):
    """pixelavg_pp[25], weight_pp and frame_init_lowres_core called the way the function-pointer table is"""
    rng = np.random.default_rng(21)
    a, b = rng.integers(0, 256, 200 * 200, dtype=np.uint8), rng.integers(0, 256, 200 * 200, dtype=np.uint8)
    for part in range(25):
        d1, d2 = np.full(200 * 200, 0x5A, np.uint8), np.full(200 * 200, 0x5A, np.uint8)
        oracle.pixelavg_pp(part, d1, 3, 71, a, 5, 83, b, 9, 64)
        ref.pixelavg_pp(part, d2, 3, 71, a, 5, 83, b, 9, 64)
        assert np.array_equal(d1, d2), part
    for (w, h, w0, rnd, sh, off) in ((64, 48, 70, 2048, 12, -3), (16, 16, 37, 1024, 11, 10), (128, 33, 1, 0, 6, 0)):
        d1, d2 = np.full(200 * 200, 0x5A, np.uint8), np.full(200 * 200, 0x5A, np.uint8)
        oracle.weight_pp(a, 7, d1, 11, 150, 160, w, h, w0, rnd, sh, off)
        ref.weight_pp(a, 7, d2, 11, 150, 160, w, h, w0, rnd, sh, off)
        assert np.array_equal(d1, d2), (w, h)
    for (w, h, mx, stride, off) in ((416, 64, 80, 576, 80 + 3 * 576), (33, 5, 7, 61, 9), (4, 1, 1, 6, 1)):  # extendRowBorder
        d1 = rng.integers(0, 256, 200 * 200, dtype=np.uint8)
        d2 = d1.copy()
        oracle.extend_row_border(d1, off, stride, w, h, mx)
        ref.extend_row_border(d2, off, stride, w, h, mx)
        assert np.array_equal(d1, d2), (w, h, mx)
    for (w, h) in ((48, 40), (17, 9), (80, 24)):
        o1 = [np.full(200 * 200, 0x5A, np.uint8) for _ in range(4)]
        o2 = [np.full(200 * 200, 0x5A, np.uint8) for _ in range(4)]
        oracle.frame_init_lowres(a, 3, o1, 5, 190, 97, w, h)
        ref.frame_init_lowres(a, 3, o2, 5, 190, 97, w, h)
        for x, y in zip(o1, o2):
            assert np.array_equal(x, y), (w, h)


def test_sa8d_extreme_blocks(oracle, ref):
    """patterns that put all the energy of an 8x8 difference into few Hadamard coefficients (the 16-bit lanes of the
    reference's packed arithmetic are fullest there)"""
    n = 64 * 80
    yy, xx = np.mgrid[0:80, 0:64]
    pats = [np.zeros((80, 64)), np.full((80, 64), 255), (xx & 1) * 255, (yy & 1) * 255, ((xx ^ yy) & 1) * 255,
            ((xx >> 1) & 1) * 255, ((yy >> 2) & 1) * 255, ((xx >> 2 ^ yy >> 1) & 1) * 255]
    pats = [p.astype(np.uint8).reshape(-1) for p in pats]
    for part in range(25):
        for a in pats:
            for b in pats:
                assert oracle.sa8d_inter(part, a, 0, 64, b, 0, 64) == ref.sa8d_inter(part, a, 0, 64, b, 0, 64)
                assert oracle.sse_pp(part, a, 0, 64, b, 0, 64) == ref.sse_pp(part, a, 0, 64, b, 0, 64)


def test_cost_requests_all_metrics(oracle, ref, small_seq):
    """subpelCompare with each of the four comparisons (sad, satd, sa8d_inter, sse_pp), all partitions and fractions"""
    rng = np.random.default_rng(33)
    planes = [p.as_tuple() for p in small_seq]
    W, H = small_seq[0].w, small_seq[0].h
    reqs = np.zeros(2000, pyoracle.COST_REQ)
    for r in reqs:
        w, h = pyoracle.PARTS[int(rng.integers(0, 25))]
        r["fenc_plane"], r["ref_plane"] = 1, int(rng.choice([0, 2]))
        r["pu_x"], r["pu_y"], r["pu_w"], r["pu_h"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h
        r["qmv_x"], r["qmv_y"], r["metric"] = int(rng.integers(-90, 90)), int(rng.integers(-90, 90)), int(rng.integers(0, 4))
    a, b = oracle.cost_batch(planes, reqs), ref.cost_batch(planes, reqs)
    bad = np.nonzero(a != b)[0]
    assert len(bad) == 0, (reqs[bad[0]], a[bad[0]], b[bad[0]])


def test_mc_primitives(oracle, ref):
    """luma_p2s, luma_vps, luma_vss, luma_hvpp, luma_addAvg: whole destination buffers compared"""
    rng = np.random.default_rng(44)
    src = rng.integers(0, 256, 200 * 200, dtype=np.uint8)
    s16a = rng.integers(-12000, 14000, 200 * 200).astype(np.int16)
    s16b = rng.integers(-12000, 14000, 200 * 200).astype(np.int16)
    for (w, h) in ((4, 4), (16, 12), (64, 64), (24, 32)):
        d1, d2 = np.full(80 * 80, 77, np.int16), np.full(80 * 80, 77, np.int16)
        oracle.luma_p2s(src, 811, 150, d1, 5, w, h)
        ref.luma_p2s(src, 811, 150, d2, 5, w, h)
        assert np.array_equal(d1, d2)
    for part in range(25):
        for coeff in range(4):
            d1, d2 = np.full(200 * 200, 77, np.int16), np.full(200 * 200, 77, np.int16)
            oracle.luma_vps(part, src, 1807, 150, d1, 9, 71, coeff)
            ref.luma_vps(part, src, 1807, 150, d2, 9, 71, coeff)
            assert np.array_equal(d1, d2), (part, coeff)
            oracle.luma_vss(part, s16a, 1807, 150, d1, 9, 71, coeff)
            ref.luma_vss(part, s16a, 1807, 150, d2, 9, 71, coeff)
            assert np.array_equal(d1, d2), (part, coeff)
            p1, p2 = np.full(200 * 200, 0x5A, np.uint8), np.full(200 * 200, 0x5A, np.uint8)
            oracle.luma_hvpp(part, src, 1807, 150, p1, 9, 71, coeff, 3 - coeff)
            ref.luma_hvpp(part, src, 1807, 150, p2, 9, 71, coeff, 3 - coeff)
            assert np.array_equal(p1, p2), (part, coeff)
        p1, p2 = np.full(200 * 200, 0x5A, np.uint8), np.full(200 * 200, 0x5A, np.uint8)
        oracle.luma_addavg(part, s16a, 3, s16b, 11, p1, 9, 64, 70, 81)
        ref.luma_addavg(part, s16a, 3, s16b, 11, p2, 9, 64, 70, 81)
        assert np.array_equal(p1, p2), part


def test_predict_batch(oracle, ref, small_seq):
    """xPredInterUni / xPredInterBi for luma into a destination plane: every partition and fraction pair"""
    rng = np.random.default_rng(45)
    p0 = small_seq[0]
    planes = [p.as_tuple() for p in small_seq]
    d1, d2 = np.full_like(p0.buf, 0x5A), np.full_like(p0.buf, 0x5A)
    for _ in range(60):
        reqs = random_pred_reqs(rng, p0.w, p0.h, pyoracle.PRED_REQ, pyoracle.PARTS, dst_plane=3)
        oracle.predict_batch(planes + [(d1, p0.origin, p0.stride)], reqs)
        ref.predict_batch(planes + [(d2, p0.origin, p0.stride)], reqs)
        assert np.array_equal(d1, d2)
    assert (d1 != 0x5A).any()


def test_bidir_cost_mc_average_and_all_metrics(oracle, ref, small_seq):
    """bidirectional requests with the motion-compensation average (luma_addAvg of 14-bit intermediates: the
    xMergeEstimation measurement) and with sa8d / sse beside sad / satd"""
    rng = np.random.default_rng(47)
    planes = [p.as_tuple() for p in small_seq]
    reqs = random_bidir_reqs(rng, 2500, small_seq[0].w, small_seq[0].h, pyoracle.BIDIR_REQ, pyoracle.PARTS,
                             metrics=(0, 1, 2, 3), flags=(0, 0x40))
    a, b = oracle.bidir_cost_batch(planes, reqs), ref.bidir_cost_batch(planes, reqs)
    bad = np.nonzero(a != b)[0]
    assert len(bad) == 0, (reqs[bad[0]], a[bad[0]], b[bad[0]])
    assert (reqs["metric"] & 0x40).any() and (reqs["metric"] & 3 == 2).any()


def test_mv_bits_match_the_reference(oracle, ref):
    """BitCost::bitcost restated (x265o_mv_bits) against the reference's own, over differences of every size and sign"""
    rng = np.random.default_rng(48)
    for _ in range(4000):
        s = int(rng.choice([3, 40, 400, 4000, 16000]))
        x, y, px, py = (int(v) for v in rng.integers(-s, s + 1, 4))
        if abs(x - px) > 16383 or abs(y - py) > 16383:
            continue
        assert oracle.mv_bits(x, y, px, py) == ref.mv_bits(x, y, px, py), (x, y, px, py)
