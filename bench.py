#!/usr/bin/env python3
"""bench.py -- ME cost path throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): batched integer-pel full search, 1080p, 64x64 CTUs, +-64 range.
One step = one picture searched against one reference: every 64x64 CTU, every square PU
(64/32/16/8) wholly inside the picture, every integer position of the window that
xSetSearchRange gives around the CTU's predictor, SAD + mvcost, first minimum in raster order.
`value` counts (PU, position) cost evaluations per second with planes, job lists and results resident
in HBM.  Two more timed legs (reported in extra keys) run on the same pictures: the full motionEstimate
replay (HEX, subme 2, merange 57, 3 references: configs[2]) and the SATD sub-pel candidate set (every
square PU x 3 references x 16 half-/quarter-pel candidates through 8-tap interpolation + SATD).

Unless --no-4k is given the same three legs are also run on one 4K picture (configs[3]) and reported under
`uhd` (kernel times, per GPU; not part of `value`).

For N > 1 the driver launches one rank per GPU; ranks take disjoint GOP segments of the
sequence (no collective on the data path) and the job-wide rate is units / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
RANGE = 64
PX_PEAK = 1.34e14     # measured v_qsad_pk_u16_u8 / v_sad_u8 pixel rate, profiles/r01_ubench_valu.txt
LANE_PEAK = 3.93e13   # integer VALU lane-ops/s: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz (measured 3.65e13)
HBM_PEAK = 8000.0     # GB/s


def build_ctu_jobs(seq, cur, ref, fenc_plane, ref_plane, qp=30, size=None):
    from x265_amd import lib, synth
    W, H = size or (globals()["W"], globals()["H"])
    ncx, ncy = synth.ctu_grid(W, H)
    dx, dy = seq.global_motion(cur, ref)
    jobs = np.zeros(ncx * ncy, lib.CTU_JOB)
    k = 0
    for cy in range(ncy):
        for cx in range(ncx):
            j = jobs[k]
            k += 1
            x, y = cx * 64, cy * 64
            lox, loy, hix, hiy = synth.search_range(np.int64(4 * dx), np.int64(4 * dy), RANGE, x, y, W, H)
            j["fenc_plane"], j["ref_plane"] = fenc_plane, ref_plane
            j["ctu_x"], j["ctu_y"] = x, y
            j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"] = lox, loy, hix, hiy
            j["qmvp_x"], j["qmvp_y"] = 4 * dx, 4 * dy
            j["qp"] = qp
    return jobs


def valid_pus_per_ctu(size=None):
    """number of square PUs wholly inside the picture, per CTU in raster order"""
    from x265_amd import synth
    W, H = size or (globals()["W"], globals()["H"])
    ncx, ncy = synth.ctu_grid(W, H)
    out = np.zeros(ncx * ncy, np.int64)
    for cy in range(ncy):
        for cx in range(ncx):
            n = 0
            for s in (64, 32, 16, 8):
                nx = min(64, W - cx * 64) // s
                ny = min(64, H - cy * 64) // s
                n += nx * ny
            out[cy * ncx + cx] = n
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=100)  # the clocks ramp under sustained load: tens of ms
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true", help="skip the extra 4K legs (reported under the key uhd)")
    ap.add_argument("--frames", type=int, default=4, help="pictures per rank kept resident (steps cycle over them)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # X265_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then
    # share devices); the driver's runs use RCCL, one rank per GPU
    backend = os.environ.get("X265_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from x265_amd import abi, lib, synth

    # ---- resident input: this rank's GOP segment -----------------------------------------------
    nfr = args.frames
    seq = synth.Sequence(W, H, seed=0x265)
    first = rank * nfr  # disjoint segments of the same synthetic sequence
    planes = [synth.Plane(seq.frame(first + i)) for i in range(nfr + 3)]
    ctx = lib.Context(local)
    for i, p in enumerate(planes):
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
    for qp in range(28, 40):
        ctx.set_qp(qp)

    # step s searches picture (3 + s % nfr) against the previous picture
    vp = valid_pus_per_ctu()
    fs_jobs, fs_units, fs_pos = [], [], []
    for i in range(nfr):
        cur = 3 + i
        j = build_ctu_jobs(seq, first + cur, first + cur - 1, cur, cur - 1)
        npos = (j["mvmax_x"].astype(np.int64) - j["mvmin_x"] + 1) * (j["mvmax_y"].astype(np.int64) - j["mvmin_y"] + 1)
        fs_units.append(int((npos * vp).sum()))
        fs_pos.append(int(npos.sum()))
        fs_jobs.append(torch.from_numpy(j.view(np.uint8).copy()).to(dev))
    n_ctu = len(vp)
    fs_out = torch.empty(n_ctu * lib.PUS_PER_CTU * 8, dtype=torch.uint8, device=dev)

    me_jobs, me_n = [], []
    for i in range(nfr):
        cur = 3 + i
        refs = [cur - 1, cur - 2, cur - 3]
        mot = [seq.global_motion(first + cur, first + r) for r in refs]
        j = synth.me_jobs(W, H, cur, refs, mot, method=abi.HEX, subme=2, merange=57, seed=first + cur)
        me_n.append(len(j))
        me_jobs.append(torch.from_numpy(j.view(np.uint8).copy()).to(dev))
    me_out = torch.empty(max(me_n) * 8, dtype=torch.uint8, device=dev)

    sp_reqs, sp_n, sp_ops = [], [], []
    for i in range(nfr):
        cur = 3 + i
        refs = [cur - 1, cur - 2, cur - 3]
        rq = synth.subpel_requests(W, H, cur, refs, [seq.global_motion(first + cur, first + r) for r in refs])
        sp_n.append(len(rq))
        sp_ops.append(synth.satd_lane_ops(rq))
        sp_reqs.append(torch.from_numpy(rq.view(np.uint8).copy()).to(dev))
    sp_out = torch.empty(max(sp_n), dtype=torch.int32, device=dev)

    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_fs(step):
        i = step % nfr
        ctx.fullsearch_device(fs_jobs[i].data_ptr(), n_ctu, fs_out.data_ptr())
        return fs_units[i], fs_pos[i]

    def run_me(step):
        i = step % nfr
        ctx.me_batch_device(me_jobs[i].data_ptr(), me_n[i], me_out.data_ptr())
        return me_n[i]

    def run_sp(step):
        i = step % nfr
        ctx.cost_batch_device(sp_reqs[i].data_ptr(), sp_n[i], sp_out.data_ptr())
        return sp_n[i], sp_ops[i]

    def timed(fn, steps, warmup):
        for s in range(warmup):
            fn(s)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(ext)
        acc = []
        for s in range(steps):
            acc.append(fn(s))
        e1.record(ext)
        barrier()
        dt = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, ev_ms, acc

    dt_fs, ev_fs, acc_fs = timed(run_fs, args.steps, args.warmup)
    units = sum(a[0] for a in acc_fs)
    positions = sum(a[1] for a in acc_fs)
    dt_me, ev_me, acc_me = timed(run_me, args.steps, args.warmup)
    me_calls = sum(acc_me)
    dt_sp, ev_sp, acc_sp = timed(run_sp, args.steps, args.warmup)
    sp_cands = sum(a[0] for a in acc_sp)
    sp_lane_ops = sum(a[1] for a in acc_sp)

    # ---- the same three legs on one 4K picture (BASELINE configs[3]); extra information, outside `value` ----
    uhd = None
    if not args.no_4k:
        UW, UH = 3840, 2160
        useq = synth.Sequence(UW, UH, seed=0x265)
        ubase = 16  # plane ids 16 .. 19
        for i in range(4):
            p = synth.Plane(useq.frame(first + i))
            ctx.plane_upload(ubase + i, p.buf, p.stride, p.origin)
        urefs = [2, 1, 0]
        umot = [useq.global_motion(first + 3, first + r) for r in urefs]
        uvp = valid_pus_per_ctu((UW, UH))
        uj = build_ctu_jobs(useq, first + 3, first + 2, ubase + 3, ubase + 2, size=(UW, UH))
        unpos = (uj["mvmax_x"].astype(np.int64) - uj["mvmin_x"] + 1) * (uj["mvmax_y"].astype(np.int64) - uj["mvmin_y"] + 1)
        umj = synth.me_jobs(UW, UH, 3, urefs, umot, method=abi.HEX, subme=2, merange=57, seed=first + 3)
        urq = synth.subpel_requests(UW, UH, 3, urefs, umot)
        for a in (umj, urq):
            a["fenc_plane"] += ubase
            a["ref_plane"] += ubase
        d_fs, d_me, d_sp = (torch.from_numpy(a.view(np.uint8).copy()).to(dev) for a in (uj, umj, urq))
        o_fs = torch.empty(len(uj) * lib.PUS_PER_CTU * 8, dtype=torch.uint8, device=dev)
        o_me = torch.empty(len(umj) * 8, dtype=torch.uint8, device=dev)
        o_sp = torch.empty(len(urq), dtype=torch.int32, device=dev)

        def leg(fn, reps=20):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.05:  # warm-up by time: the clocks ramp under sustained load
                fn()
                ctx.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for _ in range(reps):
                fn()
            e1.record(ext)
            ctx.sync()
            return e0.elapsed_time(e1) / reps

        ms_fs = leg(lambda: ctx.fullsearch_device(d_fs.data_ptr(), len(uj), o_fs.data_ptr()))
        ms_me = leg(lambda: ctx.me_batch_device(d_me.data_ptr(), len(umj), o_me.data_ptr()))
        ms_sp = leg(lambda: ctx.cost_batch_device(d_sp.data_ptr(), len(urq), o_sp.data_ptr()))
        uops = synth.satd_lane_ops(urq)
        uhd = {"workload": "configs[3]: one 3840x2160 picture per step and GPU, same three legs, 3 references for the ME legs; "
                           "kernel time (HIP events), per GPU",
               "fullsearch": {"candidates_per_s": float((unpos * uvp).sum()) / ms_fs * 1e3, "ms_per_picture_per_ref": ms_fs,
                              "frac_of_sad_roof": float(unpos.sum()) * 4096 / (ms_fs * 1e-3) / PX_PEAK},
               "me_stage": {"frames_per_s": 1e3 / ms_me, "ms_per_frame": ms_me, "me_calls_per_s": len(umj) / ms_me * 1e3},
               "satd_candidates": {"candidates_per_s": len(urq) / ms_sp * 1e3, "ms_per_frame": ms_sp,
                                   "frac_of_valu_peak": uops / (ms_sp * 1e-3) / LANE_PEAK}}
        for i in range(4):
            ctx.plane_release(ubase + i)

    # ---- CPU baseline: the reference's own C primitives (oracle/_ref) when present, else the port ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import pyoracle  # checker libraries: used for this baseline leg only
        chk = pyoracle.load_ref() if pyoracle.have_ref() else pyoracle.load_oracle()
        cores = min(16, os.cpu_count() or 1)
        j = build_ctu_jobs(seq, first + 3, first + 2, 3, 2)
        tab = lib.costtab_build(30)
        pl = [p.as_tuple() for p in planes]

        def one(k):
            jj = j[k]
            chk.ctu_fullsearch(pl[3], pl[2], tab, int(jj["ctu_x"]), int(jj["ctu_y"]),
                               (int(jj["mvmin_x"]), int(jj["mvmin_y"])), (int(jj["mvmax_x"]), int(jj["mvmax_y"])),
                               (int(jj["qmvp_x"]), int(jj["qmvp_y"])))
            npos = (int(jj["mvmax_x"]) - int(jj["mvmin_x"]) + 1) * (int(jj["mvmax_y"]) - int(jj["mvmin_y"]) + 1)
            return npos * int(vp[k])
        # whole pictures of the same workload, repeated until about 10 s of wall time have passed
        t0 = time.perf_counter()
        cu, passes = 0, 0
        with ThreadPoolExecutor(cores) as ex:
            while time.perf_counter() - t0 < 10.0:
                cu += sum(ex.map(one, range(n_ctu)))
                passes += 1
        cdt = time.perf_counter() - t0
        cpu = {"value": cu / cdt, "unit": "candidates/s", "cores": cores, "kind": chk.kind,
               "sample": f"{passes} pass(es) over the {n_ctu} CTUs of one 1080p picture, same windows; "
                         f"{'reference C sad primitives, -O3 (no asm: no yasm in the image)' if chk.kind == 'reference' else 'C restatement'}"
                         f", one thread per CTU, {cdt:.1f} s"}
        # the same for the ME stage: the job list of one picture through the reference MotionEstimate
        mj = synth.me_jobs(W, H, 3, [2, 1, 0], [seq.global_motion(first + 3, first + r) for r in (2, 1, 0)],
                           method=abi.HEX, subme=2, merange=57, seed=first + 3)
        tabs = {q: lib.costtab_build(q) for q in range(28, 40)}
        t0 = time.perf_counter()
        mp = 0
        while time.perf_counter() - t0 < 8.0:
            chk.me_batch(pl, tabs, mj, nthreads=cores)
            mp += 1
        mdt = time.perf_counter() - t0
        cpu_me = {"frames_per_s": mp / mdt, "me_calls_per_s": mp * len(mj) / mdt, "cores": cores, "kind": chk.kind,
                  "sample": f"{mp} pass(es) over one picture's {len(mj)} motionEstimate calls, {mdt:.1f} s"}

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_final_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE)
            t = json.load(open(tpath))["fullsearch2_kernel"]
            traffic = (t["fetch_kib_corrected"] + t["write_kib"]) * 1024
        total_units = units * world
        k_ms = ev_fs / args.steps
        px_ops = positions / args.steps * 4096  # 64x64 pixel-ops per position per CTU with full 8x8 reuse
        alg_bytes = n_ctu * (4096 + (64 + 2 * RANGE) ** 2 + 85 * 8 + 28)
        line = {
            "metric": "SATD ME candidates/sec + ME-stage frames/sec, bit-exact: value = integer-pel search candidates/s (SAD, configs[1]); SATD candidates/s and ME-stage frames/s in satd_candidates / me_stage",
            "value": total_units / dt_fs, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_fs / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: 1080p, 64x64 CTUs, +-64 integer full search, 85 square PUs per CTU, 1 reference",
                       "ctus_per_step": n_ctu, "gop_sharding": f"{world} segment(s) of {nfr} pictures"},
            "me_stage": {"workload": "configs[2]: motionEstimate replay, HEX, subme 2, merange 57, 3 references, "
                                     "square PUs 64..8",
                         "frames_per_s": args.steps * world / dt_me, "me_calls_per_s": me_calls * world / dt_me,
                         "ms_per_frame": dt_me / args.steps * 1e3, "kernel_ms_per_frame": ev_me / args.steps},
            "satd_candidates": {"workload": "sub-pel stage of the same pictures: square PUs 64..8 x 3 references x 16 half-/"
                                            "quarter-pel candidates, 8-tap interpolation + SATD (x265_hip_cost_batch)",
                                "candidates_per_s": sp_cands * world / dt_sp, "ms_per_frame": dt_sp / args.steps * 1e3,
                                "roofline": {"bound": "valu", "achieved": sp_lane_ops / (ev_sp * 1e-3) / 1e12,
                                             "peak": LANE_PEAK / 1e12, "unit": "Tlane-op/s",
                                             "frac": sp_lane_ops / (ev_sp * 1e-3) / LANE_PEAK, "kernel": "cost_kernel"}},
            "roofline": {"bound": "valu", "achieved": px_ops / (k_ms * 1e-3) / 1e12, "peak": PX_PEAK / 1e12,
                         "unit": "Tpixel-op/s", "frac": px_ops / (k_ms * 1e-3) / PX_PEAK,
                         "kernel": "fullsearch2_kernel", "kernel_ms": k_ms, "traffic": traffic,
                         "algorithmic_bytes": alg_bytes,
                         "hbm": {"achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                                 "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK}},
        }
        if uhd:
            line["uhd"] = uhd
        if cpu:
            line["cpu_baseline"] = cpu
            line["me_stage"]["cpu_baseline"] = cpu_me
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
