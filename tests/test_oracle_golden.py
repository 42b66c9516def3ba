"""CPU tests: the oracle restatement against fixtures generated from the real reference
(tests/golden/*.npz, made by oracle/gen_golden.py from oracle/_ref/libx265ref.so)."""
import os
import zlib

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8).tobytes())


def check_prims(impl, g):
    """impl: object with the pyoracle.Checker primitive interface"""
    bufs = g["bufs"]
    for k, (part, ia, ib, ro) in enumerate(g["cases"]):
        part, ro = int(part), int(ro)
        a, b = bufs[ia], bufs[ib]
        for si, rs in enumerate((64, 59)):
            assert impl.sad(part, a, 0, 64, b, ro, rs) == g["sad"][k, si]
            assert impl.satd(part, a, 0, 64, b, ro, rs) == g["satd"][k, si]
        assert np.array_equal(impl.sad_x3(part, a, 0, b, [ro, ro + 1, ro + 2], 59), g["x3"][k])
        assert np.array_equal(impl.sad_x4(part, a, 0, b, [ro, ro + 1, ro + 2, ro + 3], 59), g["x4"][k])
    src, s16 = g["src"], g["s16"]
    for (part, coeff, ss, ds), want in zip(g["fcases"], g["fcrc"]):
        part, coeff, ss, ds = int(part), int(coeff), int(ss), int(ds)
        so = 6 * ss + 6
        got = []
        for name in ("luma_hpp", "luma_vpp"):
            d = np.full(200 * 200, 0x5A, np.uint8)
            getattr(impl, name)(part, src, so, ss, d, 0, ds, coeff)
            got.append(crc(d))
        for ext in (0, 1):
            d = np.full(200 * 200, 0x5A5A, np.int16)
            impl.luma_hps(part, src, so, ss, d, 0, ds, coeff, ext)
            got.append(crc(d))
        d = np.full(200 * 200, 0x5A, np.uint8)
        impl.luma_vsp(part, s16, so, ss, d, 0, ds, coeff)
        got.append(crc(d))
        assert got == [int(v) for v in want], (part, coeff)


def golden_planes():
    from x265_amd import synth
    seq = synth.Sequence(416, 240, seed=7, n_patches=3)
    return [synth.Plane(seq.frame(i)) for i in range(3)]


def test_primitives_match_reference_fixtures(oracle):
    check_prims(oracle, np.load(os.path.join(GOLD, "prims.npz")))


def test_extremes_from_survey(oracle):
    """known answers recorded from the reference in SURVEY.md appendix B"""
    z, f = np.zeros(64 * 64, np.uint8), np.full(64 * 64, 255, np.uint8)
    assert oracle.sad(18, z, 0, 64, f, 0, 64) == 1044480
    assert oracle.satd(18, z, 0, 64, f, 0, 64) == 522240
    t = oracle.bitcost_table(30)

    def mvcost(x, y):
        return (int(t[32768 + x]) + int(t[32768 + y])) & 0xffff
    assert (mvcost(0, 0), mvcost(4, 0), mvcost(-4, 0), mvcost(100, -37), mvcost(32767, 32767)) == (12, 57, 57, 218, 508)


def test_partition_table(oracle):
    from oracle import pyoracle
    for p, (w, h) in enumerate(pyoracle.PARTS):
        assert oracle.partition_from_sizes(w, h) == p
    assert oracle.partition_from_sizes(12, 12) == -1


def test_bitcost_tables(oracle):
    g = np.load(os.path.join(GOLD, "bitcost.npz"))
    for q in range(70):
        t = oracle.bitcost_table(q)
        assert crc(t) == int(g["crc"][q]), q
        assert np.array_equal(t[32768 + g["probes"]], g["vals"][q])
        assert np.array_equal(t[32768 - g["probes"]], g["vals"][q])  # symmetric in the sign of the difference


def test_synthetic_generator_is_pinned():
    g = np.load(os.path.join(GOLD, "me.npz"))
    assert [crc(p.buf) for p in golden_planes()] == [int(v) for v in g["plane_crc"]]


def test_motion_estimate_matches_reference_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "me.npz"))
    planes = [p.as_tuple() for p in golden_planes()]
    tabs = {q: oracle.bitcost_table(q) for q in range(20, 45)}
    got = oracle.me_batch(planes, tabs, g["jobs"])
    want = g["results"].view(got.dtype)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (g["jobs"][bad[0]], got[bad[0]], want[bad[0]])
    # the threaded driver gives the same answers
    assert np.array_equal(oracle.me_batch(planes, tabs, g["jobs"], nthreads=4), got)


def test_ctu_fullsearch_matches_reference_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "me.npz"))
    planes = [p.as_tuple() for p in golden_planes()]
    for c, want in zip(g["ctu_cfg"], g["ctu_res"]):
        c = [int(v) for v in c]
        got = oracle.ctu_fullsearch(planes[1], planes[0], oracle.bitcost_table(c[8]), c[0], c[1], (c[2], c[3]),
                                    (c[4], c[5]), (c[6], c[7]))
        assert np.array_equal(got, want.view(got.dtype))


def test_ctu_fullsearch_cost_is_sad_plus_mvcost(oracle):
    """the per-CTU search result re-derived from an explicit cost request (composition rule in DESIGN.md)"""
    from oracle import pyoracle
    planes = [p.as_tuple() for p in golden_planes()]
    tab = oracle.bitcost_table(30)
    res = oracle.ctu_fullsearch(planes[1], planes[0], tab, 64, 64, (-6, -6), (6, 6), (0, 0))
    req = np.zeros(1, pyoracle.COST_REQ)
    r = req[0]
    r["fenc_plane"], r["ref_plane"], r["pu_x"], r["pu_y"], r["pu_w"], r["pu_h"] = 1, 0, 64, 64, 64, 64
    r["qmv_x"], r["qmv_y"] = res[0]["mvx"], res[0]["mvy"]
    sad = int(oracle.cost_batch(planes, req)[0])
    mvc = (int(tab[32768 + int(res[0]["mvx"])]) + int(tab[32768 + int(res[0]["mvy"])])) & 0xffff
    assert sad + mvc == int(res[0]["cost"])


def golden_lowres(impl):
    """the 12 lookahead planes of the golden pictures, made by `impl`: plane ids 4 p + k"""
    low, size = [], None
    for p in golden_planes():
        four, size = impl.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)
        low += four
    return low, size


def test_lowres_planes_and_search_match_reference_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "lowres.npz"))
    low, size = golden_lowres(oracle)
    assert list(size) == [int(v) for v in g["size"]]
    assert [crc(b) for (b, _, _) in low] == [int(v) for v in g["plane_crc"]]
    tabs = {q: oracle.bitcost_table(q) for q in (12, 20, 33)}
    got = oracle.me_batch(low, tabs, g["jobs"])
    want = g["results"].view(got.dtype)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (g["jobs"][bad[0]], got[bad[0]], want[bad[0]])


def test_weighted_planes_match_reference_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "lowres.npz"))
    p = golden_planes()[0]
    got = [crc(oracle.weighted_plane(p.buf, p.origin, p.stride, p.w, p.h, int(a), int(b), int(c))) for a, b, c in g["weight_cases"]]
    assert got == [int(v) for v in g["weight_crc"]]


def test_bidir_costs_match_reference_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "lowres.npz"))
    got = oracle.bidir_cost_batch([p.as_tuple() for p in golden_planes()], g["bidir_reqs"])
    assert np.array_equal(got, g["bidir_costs"])


def test_distortions_match_reference_fixtures(oracle):
    """dist.npz: sa8d_inter / sse_pp / satd / sad of predictions, computed by the reference build"""
    g = np.load(os.path.join(GOLD, "dist.npz"))
    got = oracle.cost_batch([p.as_tuple() for p in golden_planes()], g["reqs"])
    assert set(g["reqs"]["metric"]) == {0, 1, 2, 3}
    assert np.array_equal(got, g["costs"])


def test_predictions_match_reference_fixtures(oracle):
    """pred.npz: CRC32 of the destination plane after each batch of uni- / bi-directional predictions, by the
    reference build"""
    import zlib
    from oracle import pyoracle
    g = np.load(os.path.join(GOLD, "pred.npz"))
    planes = golden_planes()
    p0 = planes[0]
    dst = np.full_like(p0.buf, 0x5A)
    for reqs, want in zip(g["reqs"], g["crc"]):
        oracle.predict_batch([p.as_tuple() for p in planes] + [(dst, p0.origin, p0.stride)], reqs.view(pyoracle.PRED_REQ))
        assert zlib.crc32(dst.tobytes()) & 0xffffffff == int(want)


def test_weighted_bi_prediction_matches_reference_fixture(oracle):
    """wpbi.npz: TComWeightPrediction::xWeightedPredictionBi of the reference (getWpScaling + addWeightBi, run through
    oracle/_ref/enc_capture --wpbi) on 75 blocks of intermediates: every partition, unit / typical / extreme weights"""
    from conftest import wpbi_fixture
    cases = wpbi_fixture()
    assert len(cases) == 75 and any(c[0][2] < 0 for c in cases) and any(c[0][6] == 0 for c in cases)
    for hd, src, dst in cases:
        assert np.array_equal(oracle.add_weight_bi(src[0], src[1], *hd), dst), hd


def merge_fixture():
    g = np.load(os.path.join(GOLD, "encoder_merge.npz"))
    planes, pos = [], 0
    for nbytes, origin, stride in g["plane_geo"]:
        planes.append((g["plane_bytes"][pos:pos + nbytes].copy(), int(origin), int(stride)))
        pos += int(nbytes)
    return g, planes


def check_merge(res, want):
    for k in ("cost", "bits", "index", "inter_dir"):
        assert np.array_equal(res[k], want[k]), k


def test_merge_estimation_matches_the_reference_encoder(oracle):
    """encoder_merge.npz: 930 TEncSearch::xMergeEstimation calls of a reference encoder run with rectangular and
    asymmetric partitions (oracle/gen_golden.py gen_encoder_merge): weighted P slices, weighted B slices, candidates of
    one list and of both; cost, bits, index and direction as the reference left them"""
    g, planes = merge_fixture()
    pus, want = g["merge_pus"], g["merge_out"]
    assert len(pus) > 900 and (pus["weighted"] == 1).sum() > 300 and (pus["weighted"] == 2).sum() > 100 and (want["inter_dir"] == 3).sum() > 30
    W, H = (int(v) for v in g["pic_size"])
    check_merge(oracle.merge_estimate_batch(planes, pus, W, H), want)


def test_lowres_bidir_costs_match_reference_fixtures(oracle):
    """lowres_bidir.npz: the lookahead's bidirectional cost (lowresMC x 2, pixelavg_pp, satd) by the reference build"""
    g = np.load(os.path.join(GOLD, "lowres_bidir.npz"))
    low = []
    for p in golden_planes():
        four, _ = oracle.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)
        low += four
    assert np.array_equal(oracle.bidir_cost_batch(low, g["reqs"]), g["costs"])


def test_bidir_mc_costs_match_reference_fixtures(oracle):
    """bidir_mc.npz: bidirectional blocks with the motion-compensation average and all four comparisons, by the reference"""
    g = np.load(os.path.join(GOLD, "bidir_mc.npz"))
    got = oracle.bidir_cost_batch([p.as_tuple() for p in golden_planes()], g["reqs"])
    assert np.array_equal(got, g["costs"])


def encoder_fixture():
    """encoder_me.npz: motionEstimate calls intercepted inside the reference encoder (--preset medium, 416x240 synthetic
    clip; oracle/enc_capture.cpp): planes as (buffer, origin, stride) tuples, jobs, results"""
    g = np.load(os.path.join(GOLD, "encoder_me.npz"))
    planes, pos = [], 0
    for nbytes, origin, stride in g["plane_geo"]:
        planes.append((g["plane_bytes"][pos:pos + nbytes].copy(), int(origin), int(stride)))
        pos += int(nbytes)
    return planes, g["jobs"], g["results"]


def inter_fixture():
    """the predInterSearch iterations of encoder_me.npz: x265_hip_inter_job records made from what the reference
    encoder's TComDataCU::fillMvpCand returned in front of each full-resolution call, and what the reference then did:
    the predictor it picked (motionEstimate's qmvp), xSetSearchRange's window, cost and vector"""
    g = np.load(os.path.join(GOLD, "encoder_me.npz"))
    planes, jobs, res = encoder_fixture()
    sel = g["inter_call"]
    return planes, g["inter_jobs"], jobs[sel], res[sel], int(g["pic_size"][0]), int(g["pic_size"][1])


def check_inter(out, mej, res):
    assert np.array_equal(out["mvp_x"], mej["qmvp_x"]) and np.array_equal(out["mvp_y"], mej["qmvp_y"])
    for k in ("mvmin_x", "mvmin_y", "mvmax_x", "mvmax_y"):
        assert np.array_equal(out[k], mej[k]), k
    for k in ("cost", "mvx", "mvy"):
        assert np.array_equal(out[k], res[k]), k


def test_inter_search_iterations_match_the_reference_encoder(oracle):
    """AMVP pick by least SAD, xSetSearchRange and motionEstimate restated (x265o_inter_search_batch) against the
    reference encoder's own choices: 2400 iterations, about a quarter of which pick the second candidate"""
    from oracle import pyoracle
    planes, ij, mej, res, W, H = inter_fixture()
    ij = ij.view(pyoracle.INTER_JOB)
    assert len(ij) == 2400 and (ij["amvp"][:, 0] != ij["amvp"][:, 2]).sum() > 1000
    tabs = {int(q): oracle.bitcost_table(int(q)) for q in np.unique(ij["qp"])}
    out = oracle.inter_search_batch(planes, tabs, ij, W, H)
    assert 200 < (out["mvp_idx"] == 1).sum() < 2200
    check_inter(out, mej, res)


def pu_fixture():
    """encoder_pu.npz: the prediction units of a reference encoder run (oracle/gen_golden.py gen_encoder_pu): planes, the
    units' iterations as x265_hip_inter_job records with motionEstimate's own results, the x265_hip_inter_pu records,
    and what predInterSearch left in the CU for each unit"""
    g = np.load(os.path.join(GOLD, "encoder_pu.npz"))
    planes, pos = [], 0
    for nbytes, origin, stride in g["plane_geo"]:
        planes.append((g["plane_bytes"][pos:pos + nbytes], int(origin), int(stride)))
        pos += nbytes
    return planes, g["inter_jobs"], g["me_results"], g["pus"], g["final"], int(g["pic_size"][0]), int(g["pic_size"][1])


def check_pu_decisions(out, pus, final):
    """direction, vectors, references and AMVP indices of every unit as the encoder's (no unit of the run merged: the
    medium preset's units are all 2Nx2N, which have no merge candidate here, TEncSearch.cpp:2264)"""
    assert not final["merge_flag"].any()
    assert np.array_equal(out["inter_dir"], final["inter_dir"])
    n = {1: 0, 2: 0, 3: 0}
    for o, f in zip(out, final):
        d = int(f["inter_dir"])
        n[d] += 1
        for l in (0, 1):
            if not d & (1 << l):
                continue
            s = o["bidir"][l] if d == 3 else o["list"][l]
            assert (s["mvx"], s["mvy"]) == tuple(f["mv"][2 * l:2 * l + 2]), (o, f)
            assert s["mvp_idx"] == f["mvp_idx"][l] and o["list"][l]["ref"] == f["ref"][l], (o, f)
    return n


def test_pu_decisions_match_the_reference_encoder(oracle):
    """x265o_inter_decide_batch against what TEncSearch::predInterSearch left in the CU: 1981 prediction units of the
    first P picture and the B pictures behind it; xCheckBestMVP changes the AMVP index of some, the bidirectional
    candidate wins some, and its zero-vector form is tried"""
    from oracle import pyoracle
    planes, ij, mres, pus, final, W, H = pu_fixture()
    ij, pus = ij.view(pyoracle.INTER_JOB), pus.view(pyoracle.INTER_PU)
    tabs = {int(q): oracle.bitcost_table(int(q)) for q in np.unique(ij["qp"])}
    res = oracle.inter_search_batch(planes, tabs, ij, W, H)
    for k in ("cost", "mvx", "mvy"):
        assert np.array_equal(res[k], mres[k]), k
    out = oracle.inter_decide_batch(planes, tabs, ij, res, pus, W, H)
    n = check_pu_decisions(out, pus, final)
    assert len(pus) == 1981 and n[3] > 100 and n[2] > 100 and out["zero_tried"].sum() > 50
    # xCheckBestMVP moved the index away from the candidate the SAD pick chose
    first = res["mvp_idx"][pus["first_job"]]
    assert (out["list"]["mvp_idx"][:, 0] != first)[pus["num_ref"][:, 0] == 1].sum() > 10


def lookahead_fixture():
    """the frame cost estimates encoder_me.npz holds completely: (planes, lookahead jobs, width and height in CUs, what
    the reference encoder's own motionEstimate calls returned for every CU, shaped (n, H, W))"""
    from oracle import pyoracle
    g = np.load(os.path.join(GOLD, "encoder_me.npz"))
    planes, jobs, res = encoder_fixture()
    jobs = jobs.view(pyoracle.ME_JOB)
    W, H = (int(g["pic_size"][0]) // 2 + 7) // 8, (int(g["pic_size"][1]) // 2 + 7) // 8
    low = np.nonzero(jobs["method"] & 0x80)[0]
    pairs = {}
    for i in low:
        pairs.setdefault((int(jobs["fenc_plane"][i]), int(jobs["ref_plane"][i])), []).append(i)
    lj, want = [], []
    for (f, r), idx in pairs.items():
        idx = np.array(idx)
        if len(idx) != W * H:
            continue
        j = jobs[idx]
        assert (j["method"] == 0x81).all() and (j["subme"] == 1).all() and (j["merange"] == 16).all() and (j["qp"] == 12).all()
        w = np.zeros((H, W), res.dtype)
        w[j["pu_y"] // 8, j["pu_x"] // 8] = res[idx]
        lj.append((f, r))
        want.append(w)
    out = np.zeros(len(lj), pyoracle.LOOKAHEAD_JOB)
    out["fenc_plane"], out["ref_plane"] = [p[0] for p in lj], [p[1] for p in lj]
    out["merange"], out["method"], out["subme"], out["qp"] = 16, 1, 1, 12
    return planes, out, W, H, np.stack(want)


def test_lookahead_frames_match_the_reference_encoder(oracle):
    """estimateCUCost's search loop restated (x265o_lookahead_batch: bounds, candidates from the CUs already done, median
    predictor, motionEstimate) against the frame cost estimates the capture holds completely: every CU's cost and
    vector as the reference encoder's own calls returned them"""
    planes, lj, W, H, want = lookahead_fixture()
    assert len(lj) == 3 and (W, H) == (26, 15)
    got = oracle.lookahead_batch(planes, {12: oracle.bitcost_table(12)}, lj, W, H)
    assert np.array_equal(got, want.view(got.dtype))
    assert len(np.unique(got["mvx"])) > 5


def test_encoder_calls_match_reference_fixture(oracle):
    """what the reference encoder itself asked of motionEstimate and got back: AMVP predictors and candidate lists,
    real search bounds and QPs, reconstructed references, the lookahead's lowres searches"""
    from oracle import pyoracle
    planes, jobs, want = encoder_fixture()
    jobs = jobs.view(pyoracle.ME_JOB)
    assert (jobs["method"] & 0x80).any() and not (jobs["method"] & 0x80).all() and jobs["num_mvc"].max() >= 5
    tabs = {int(q): oracle.bitcost_table(int(q)) for q in np.unique(jobs["qp"])}
    got = oracle.me_batch(planes, tabs, jobs, nthreads=8)
    want = want.view(got.dtype)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (jobs[bad[0]], got[bad[0]], want[bad[0]])
