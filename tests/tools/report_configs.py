"""Throughput of BASELINE configs[1..3] at 1080p and 4K on one MI355X (JSON on stdout).

  configs[1]  integer-pel full search, +-64, 85 square PUs per CTU, 1 reference     (x265_hip_fullsearch)
  configs[2]  motionEstimate replay, HEX + subme 2, merange 57, 3 references        (x265_hip_me_batch)
  configs[3]  the same at 4K, plus the SATD-candidate rate of the sub-pel stage:     (x265_hip_cost_batch)
              every square PU x 3 references x 16 candidates (half-pel ring + quarter-pel ring around the
              integer vector), SATD after 8-tap interpolation

Everything is resident in HBM before the timed region; times are HIP events on the context's stream.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from x265_amd import abi, lib, synth

LANE_PEAK = bench.VALU_PEAK  # 7.86e13 32-bit VALU lane-ops/s (MI355X_MICROARCH.md; DESIGN.md 4: the kernels' own forms issue at half of it)
PX_PEAK = bench.SAD_PX_PEAK  # 1.57e14 pixel-ops/s


def timed(ext, ctx, fn, reps):
    # warm up for 30 ms of work: the clocks ramp under sustained load, a single call measures the ramp
    t0 = time.perf_counter()
    while True:
        fn()
        ctx.sync()
        if time.perf_counter() - t0 > 0.03:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(reps):
        fn()
    e1.record(ext)
    ctx.sync()
    return e0.elapsed_time(e1) / reps


def run(w, h, tag):
    dev = torch.device("cuda", 0)
    seq = synth.Sequence(w, h, seed=0x265)
    planes = [synth.Plane(seq.frame(i)) for i in range(4)]
    ctx = lib.Context(0)
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
    for i, p in enumerate(planes):
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
    for q in range(28, 40):
        ctx.set_qp(q)
    res = {"picture": f"{w}x{h}"}

    fj = bench.build_ctu_jobs(seq, 3, 2, 3, 2, size=(w, h))
    vp = bench.valid_pus_per_ctu(w, h)
    npos = (fj["mvmax_x"].astype(np.int64) - fj["mvmin_x"] + 1) * (fj["mvmax_y"].astype(np.int64) - fj["mvmin_y"] + 1)
    d = torch.from_numpy(fj.view(np.uint8).copy()).to(dev)
    o = torch.empty(len(fj) * lib.PUS_PER_CTU * 8, dtype=torch.uint8, device=dev)
    ms = timed(ext, ctx, lambda: ctx.fullsearch_device(d.data_ptr(), len(fj), o.data_ptr()), 20)
    res["fullsearch"] = {"ctus": len(fj), "ms_per_picture_per_ref": ms, "candidates_per_s": float((npos * vp).sum()) / ms * 1e3,
                         "pixel_ops_per_s": float(npos.sum()) * 4096 / ms * 1e3,
                         "frac_of_sad_roof": float(npos.sum()) * 4096 / ms * 1e3 / PX_PEAK}

    refs = [2, 1, 0]
    mot = [seq.global_motion(3, r) for r in refs]
    for name, method, subme in (("hex_subme2", abi.HEX, 2), ("umh_subme5", abi.UMH, 5), ("star_subme7", abi.STAR, 7)):
        mj = synth.me_jobs(w, h, 3, refs, mot, method=method, subme=subme, merange=57)
        d = torch.from_numpy(mj.view(np.uint8).copy()).to(dev)
        o = torch.empty(len(mj) * 8, dtype=torch.uint8, device=dev)
        ms = timed(ext, ctx, lambda: ctx.me_batch_device(d.data_ptr(), len(mj), o.data_ptr()), 10)
        res["me_stage_" + name] = {"calls": len(mj), "references": 3, "ms_per_picture": ms, "pictures_per_s": 1e3 / ms,
                                   "calls_per_s": len(mj) / ms * 1e3}
        # roofline by the checker's op model of what these searches evaluate (SURVEY.md 8(d) lane-ops)
        from oracle import pyoracle
        tabs = {q: lib.costtab_build(q) for q in range(28, 40)}
        _, model = pyoracle.load_oracle().me_batch_ops([p.as_tuple() for p in planes], tabs, mj, nthreads=16)
        res["me_stage_" + name]["roofline"] = {
            "bound": "valu", "algorithmic_lane_ops": model["lane_ops"], "achieved": model["lane_ops"] / (ms * 1e-3) / 1e12,
            "peak": LANE_PEAK / 1e12, "unit": "Tlane-op/s", "frac": model["lane_ops"] / (ms * 1e-3) / LANE_PEAK,
            "model": {k: model[k] for k in ("sad_ops", "satd_ops", "interp_ops", "sad_candidates", "subpel_candidates")}}

    # lookahead: lowres planes of four pictures, then every 8x8 CU of one picture against
    # two Lowres references (HEX, subme 1, merange 16, QP 12: the lookahead's settings, slicetype.h:73-77)
    ctx.set_qp(12)
    geo = lib.lowres_geometry(w, h)

    def all_lowres():
        for i in range(4):
            ctx.lowres_init(i, w, h, 8 + 4 * i)
    ms = timed(ext, ctx, all_lowres, 20) / 4
    alg = geo.width * geo.lines * 8  # 4 source bytes read and 4 plane bytes written per lowres sample
    wrote = (geo.width + 192) * (geo.lines + 160) * 4 + geo.width * geo.lines * 4  # with the margins it also writes
    res["lowres_init"] = {"lowres_picture": f"{geo.width}x{geo.lines}", "ms_per_picture": ms,
                          "algorithmic_bytes": alg, "algorithmic_gb_per_s": alg / ms / 1e6, "frac_of_hbm_peak": alg / ms / 1e6 / 8000.0,
                          "bytes_incl_margins": wrote, "gb_per_s_incl_margins": wrote / ms / 1e6}
    # bidirectional candidates: every square PU, predictions from references 2 and 1 at
    # quarter-pel vectors (both diagonal: the most expensive case), SATD against their average
    pus = synth.square_pus(w, h)
    br = np.zeros(len(pus), abi.BIDIR_REQ)
    br["fenc_plane"], br["ref0_plane"], br["ref1_plane"] = 3, 2, 1
    br["pu_x"], br["pu_y"] = pus[:, 0], pus[:, 1]
    br["pu_w"] = br["pu_h"] = pus[:, 2]
    br["metric"] = abi.SATD
    br["qmv0_x"], br["qmv0_y"] = 4 * mot[0][0] + 1, 4 * mot[0][1] + 2
    br["qmv1_x"], br["qmv1_y"] = 4 * mot[1][0] - 1, 4 * mot[1][1] + 1
    d = torch.from_numpy(br.view(np.uint8).copy()).to(dev)
    o = torch.empty(len(br), dtype=torch.int32, device=dev)
    ms = timed(ext, ctx, lambda: ctx.bidir_cost_batch_device(d.data_ptr(), len(br), o.data_ptr()), 10)
    res["bidir_candidates"] = {"requests": len(br), "ms": ms, "candidates_per_s": len(br) / ms * 1e3,
                               "pixels_per_s": float((br["pu_w"].astype(np.int64) * br["pu_h"]).sum()) / ms * 1e3}

    # reference-plane production: weighted copy + border extension of one picture
    ms = timed(ext, ctx, lambda: ctx.plane_weight(0, 40, w, h, 70, -3, 6), 20)
    alg = w * h + (w + 192) * (h + 160)  # picture read once, padded plane written once
    res["plane_weight"] = {"ms_per_picture": ms, "algorithmic_bytes": alg, "algorithmic_gb_per_s": alg / ms / 1e6,
                           "frac_of_hbm_peak": alg / ms / 1e6 / 8000.0}
    ctx.plane_release(40)
    lmot = [(mx // 2, my // 2) for (mx, my) in (seq.global_motion(3, 2), seq.global_motion(3, 1))]
    lj = np.concatenate([synth.lookahead_jobs(geo.width, geo.lines, 8 + 4 * 3, 8 + 4 * r, m)
                         for r, m in zip((2, 1), lmot)])
    d = torch.from_numpy(lj.view(np.uint8).copy()).to(dev)
    o = torch.empty(len(lj) * 8, dtype=torch.uint8, device=dev)
    ms = timed(ext, ctx, lambda: ctx.me_batch_device(d.data_ptr(), len(lj), o.data_ptr()), 10)
    res["lookahead_me"] = {"calls": len(lj), "references": 2, "ms_per_picture": ms, "pictures_per_s": 1e3 / ms,
                           "calls_per_s": len(lj) / ms * 1e3}
    if os.environ.get("X265_REPORT_CPU", "1") == "1":
        import time
        from oracle import pyoracle  # checker library, timed here as the CPU baseline only
        chk = pyoracle.load_ref() if pyoracle.have_ref() else pyoracle.load_oracle()
        t0 = time.perf_counter()
        low = []
        for p in planes:
            low += chk.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)[0]
        t_init = (time.perf_counter() - t0) / 4 * 1e3
        cj = lj.copy()
        cj["fenc_plane"] = 12
        cj["ref_plane"] = np.where(lj["ref_plane"] == 8 + 4 * 2, 8, 4)
        tabs = {12: lib.costtab_build(12)}
        chk.me_batch(low, tabs, cj[:1000], nthreads=16)
        t0 = time.perf_counter()
        chk.me_batch(low, tabs, cj, nthreads=16)
        t_me = (time.perf_counter() - t0) * 1e3
        res["lookahead_cpu"] = {"kind": chk.kind, "lowres_init_ms_per_picture_1_thread": t_init, "me_ms_per_picture_16_threads": t_me,
                                "me_pictures_per_s": 1e3 / t_me}

    # host-array entry points (job upload, kernel, result download, all synchronous): the PCIe-inclusive rate
    import time
    mj = synth.me_jobs(w, h, 3, refs, mot, method=abi.HEX, subme=2, merange=57)
    ctx.me_batch(mj)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.me_batch(mj)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    res["host_array_me_batch"] = {"calls": len(mj), "ms_per_picture": ms, "pictures_per_s": 1e3 / ms}
    ctx.fullsearch(fj)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.fullsearch(fj)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    res["host_array_fullsearch"] = {"ctus": len(fj), "ms_per_picture_per_ref": ms,
                                    "candidates_per_s": float((npos * vp).sum()) / ms * 1e3}

    # the sub-pel stage as rings (x265_hip_refine_batch), one launch per PU size, device-resident and through the
    # host-array entry point (request upload, kernel, cost download, synchronous: the PCIe-inclusive rate)
    rr = synth.refine_requests(w, h, 3, refs, mot)
    by_size = [rr[rr["pu_w"] == sz] for sz in (64, 32, 16, 8)]
    dr = [torch.from_numpy(q.view(np.uint8).copy()).to(dev) for q in by_size]
    orr = torch.empty(len(rr) * abi.RING_COSTS, dtype=torch.int32, device=dev)

    d_all = torch.from_numpy(np.concatenate(by_size).view(np.uint8).copy()).to(dev)
    secs = [(len(q), int(q["pu_w"][0]), int(q["pu_h"][0])) for q in by_size]

    def rings():  # the four sizes as four runs of one list, one launch
        ctx.refine_sections_device(d_all.data_ptr(), secs, orr.data_ptr())
    ms = timed(ext, ctx, rings, 10)
    res["satd_rings"] = {"ring_requests": len(rr), "candidates": 8 * len(rr), "ms": ms, "candidates_per_s": 8 * len(rr) / ms * 1e3}
    h_all = np.concatenate(by_size)
    ctx.refine_batch(h_all)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.refine_batch(h_all)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    res["host_array_refine_batch"] = {"ring_requests": len(rr), "ms_per_picture": ms, "candidates_per_s": 8 * len(rr) / ms * 1e3,
                                      "bytes_up": 20 * len(rr), "bytes_down": 36 * len(rr)}

    rq = synth.subpel_requests(w, h, 3, refs, mot)
    res["satd_rings"]["frac_of_valu_peak"] = synth.satd_lane_ops(rq) / (res["satd_rings"]["ms"] * 1e-3) / LANE_PEAK
    d = torch.from_numpy(rq.view(np.uint8).copy()).to(dev)
    o = torch.empty(len(rq), dtype=torch.int32, device=dev)
    ms = timed(ext, ctx, lambda: ctx.cost_batch_device(d.data_ptr(), len(rq), o.data_ptr()), 5)
    ops = synth.satd_lane_ops(rq)
    res["satd_candidates"] = {"requests": len(rq), "ms": ms, "candidates_per_s": len(rq) / ms * 1e3,
                              "pixels_per_s": float((rq["pu_w"].astype(np.int64) * rq["pu_h"]).sum()) / ms * 1e3,
                              "algorithmic_lane_ops_per_s": ops / ms * 1e3, "frac_of_valu_peak": ops / ms * 1e3 / LANE_PEAK}
    for size in (8, 16, 32, 64):
        sel = rq[rq["pu_w"] == size]
        d = torch.from_numpy(sel.view(np.uint8).copy()).to(dev)
        ms = timed(ext, ctx, lambda: ctx.cost_batch_device(d.data_ptr(), len(sel), o.data_ptr()), 5)
        ops = synth.satd_lane_ops(sel)
        res["satd_candidates"][f"{size}x{size}"] = {"requests": len(sel), "ms": ms, "candidates_per_s": len(sel) / ms * 1e3,
                                                    "frac_of_valu_peak": ops / ms * 1e3 / LANE_PEAK}
    # mode-decision distortions of the same predictions: x265_hip_dist_batch
    res["dist_candidates"] = {}
    for name, metric in (("sa8d", abi.SA8D), ("sse", abi.SSE)):
        sel = rq.copy()
        sel["metric"] = metric
        d = torch.from_numpy(sel.view(np.uint8).copy()).to(dev)
        ms = timed(ext, ctx, lambda: ctx.dist_batch_device(d.data_ptr(), len(sel), o.data_ptr()), 5)
        ops = synth.satd_lane_ops(sel)
        res["dist_candidates"][name] = {"requests": len(sel), "ms": ms, "candidates_per_s": len(sel) / ms * 1e3,
                                        "frac_of_valu_peak": ops / ms * 1e3 / LANE_PEAK}
    # motion-compensated prediction of a whole picture: x265_hip_predict_batch, the picture
    # tiled by PUs of one size, quarter-pel diagonal vectors; destination = plane 8
    dstp = synth.Plane(np.zeros((h, w), np.uint8))
    ctx.plane_upload(8, dstp.buf, dstp.stride, dstp.origin)
    res["predict"] = {}
    for size in (8, 16, 64):
        xs, ys = np.meshgrid(np.arange(0, w - size + 1, size), np.arange(0, h - size + 1, size))
        pr = np.zeros(xs.size, abi.PRED_REQ)
        pr["dst_plane"], pr["ref0_plane"], pr["ref1_plane"] = 8, 2, -1
        pr["pu_x"], pr["pu_y"], pr["pu_w"], pr["pu_h"] = xs.ravel(), ys.ravel(), size, size
        pr["qmv0_x"], pr["qmv0_y"], pr["qmv1_x"], pr["qmv1_y"] = 4 * mot[0][0] + 1, 4 * mot[0][1] + 3, 4 * mot[1][0] + 2, 4 * mot[1][1] + 1
        for name, ref1 in (("uni", -1), ("bi", 1)):
            pr["ref1_plane"] = ref1
            d = torch.from_numpy(pr.view(np.uint8).copy()).to(dev)
            ms = timed(ext, ctx, lambda: ctx.predict_batch_device(d.data_ptr(), len(pr)), 10)
            px = float(len(pr)) * size * size
            nref = 2 if ref1 >= 0 else 1
            # interpolation lane-ops as in satd_lane_ops (diagonal) per list, + 4 per pixel for addAvg
            ops = px * (nref * (8 * (size + 7) / size + 8 + 3) + (4 if nref == 2 else 0))
            res["predict"][f"{name}_{size}x{size}"] = {"requests": len(pr), "us_per_picture": ms * 1e3, "pixels_per_s": px / ms * 1e3,
                                                       "frac_of_valu_peak": ops / ms * 1e3 / LANE_PEAK,
                                                       "hbm_GBps": (nref + 1) * px / ms * 1e3 / 1e9}
    ctx.close()
    return res


if __name__ == "__main__":
    out = {"1080p": run(1920, 1080, "1080p"), "4k": run(3840, 2160, "4k")}
    print(json.dumps(out, indent=1))
