#!/usr/bin/env python3
"""bench.py -- ME cost path throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[3] / configs[4]): a synthetic 3840x2160 sequence cut into closed GOP segments
(x265_amd/shard.py), segment s on rank s mod N, every rank holding the planes of its own segments in HBM
(x265_amd/runner.py); nothing crosses ranks on the data path.  One step = the sub-pel SATD candidate stage of one P
picture of the rank's segments: every square PU (64/32/16/8) x every reference (the up to 3 previous pictures of the
same segment) x 16 half-/quarter-pel candidates, each an 8-tap interpolation + SATD (MotionEstimate::subpelCompare +
satd, motion.cpp:1126-1164), submitted as rings of 8 in one launch (x265_hip_refine_sections_device).  `value` = SATD candidates per second over all ranks with planes, request lists and
results resident in HBM.

Further timed legs on the same pictures, each with its own roofline, in extra keys: `me_stage` (the full motionEstimate
replay, HEX / subme 2 / merange 57: ME-stage pictures per second) and `fullsearch` (configs[1]'s integer SAD search,
+-64, against the nearest reference).  `hd` repeats the three legs on 1080p pictures (kernel times, per GPU).
`cpu_baseline` times the reference's own subpelCompare + satd (oracle/_ref, C primitives; no yasm in the image, so no
x86 assembly) on the host cores over the same request list.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RANGE = 64
# Peaks (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 at 2.4 GHz; tools/ubench_valu.hip, profiles/r02_ubench_valu.txt)
VALU_PEAK = 7.86e13        # 32-bit lane-ops/s: a wave64 instruction every 2 cycles per SIMD (v_add_u32, v_fma_f32 ...)
VALU_ISSUE_4CYC = 3.93e13  # the forms these kernels are made of (packed 16-bit, v_dot4 / v_dot2, v_perm, v_sad_u8, DPP) issue every 4
SAD_PX_PEAK = 1.57e14      # CUs x 64 lanes x 2.4 GHz x 4 pixels per lane-op (v_qsad_pk_u16_u8 measured: 1.53e14)
HBM_PEAK = 8000.0          # GB/s


def valid_pus_per_ctu(w, h):
    """number of square PUs wholly inside the picture, per CTU in raster order"""
    ncx, ncy = -(-w // 64), -(-h // 64)
    cx, cy = np.meshgrid(np.arange(ncx), np.arange(ncy), indexing="xy")
    out = np.zeros(ncx * ncy, np.int64)
    for s in (64, 32, 16, 8):
        out += ((np.minimum(64, w - cx * 64) // s) * (np.minimum(64, h - cy * 64) // s)).reshape(-1)
    return out


def build_ctu_jobs(seq, cur, ref, fenc_plane, ref_plane, qp=30, size=None):
    """CTU jobs of the integer full search of picture `cur` against `ref` (kept for the tests that import it)"""
    from x265_amd import lib, synth
    W, H = size or (seq.w, seq.h)
    ncx, ncy = synth.ctu_grid(W, H)
    dx, dy = seq.global_motion(cur, ref)
    cx, cy = np.meshgrid(np.arange(ncx) * 64, np.arange(ncy) * 64, indexing="xy")
    cx, cy = cx.reshape(-1), cy.reshape(-1)
    jobs = np.zeros(ncx * ncy, lib.CTU_JOB)
    lox, loy, hix, hiy = synth.search_range(np.int64(4 * dx), np.int64(4 * dy), RANGE, cx, cy, W, H)
    jobs["fenc_plane"], jobs["ref_plane"] = fenc_plane, ref_plane
    jobs["ctu_x"], jobs["ctu_y"] = cx, cy
    jobs["mvmin_x"], jobs["mvmin_y"], jobs["mvmax_x"], jobs["mvmax_y"] = lox, loy, hix, hiy
    jobs["qmvp_x"], jobs["qmvp_y"], jobs["qp"] = 4 * dx, 4 * dy, qp
    return jobs


class Picture:
    """device-resident work lists of one P picture"""

    def __init__(self, torch, dev, runner, cur, refs, vp):
        from x265_amd import lib, synth
        self.cur, self.refs = cur, refs
        rq = runner.satd_requests(cur, refs)
        self.n_satd, self.satd_ops = len(rq), synth.satd_lane_ops(rq)
        self.d_satd = torch.from_numpy(rq.view(np.uint8).copy()).to(dev)
        rings = runner.ring_requests(cur, refs)  # one list per PU size: 64, 32, 16, 8
        assert 8 * sum(len(r) for r in rings) == len(rq)
        self.n_ring = sum(len(r) for r in rings)
        # the four lists back to back: one list of four runs, submitted in one launch
        self.d_rings = torch.from_numpy(np.concatenate(rings).view(np.uint8).copy()).to(dev)
        self.ring_sections = [(len(r), int(r["pu_w"][0]), int(r["pu_h"][0])) for r in rings if len(r)]
        # the medium preset's own sub-pel stage (motion.cpp:51: 1 + 4 half-pel candidates, 4 quarter-pel) around the integer
        # search's winners: two ring lists, each four runs by PU size
        from x265_amd import abi
        self.s2 = []
        self.n_s2 = self.n_s2_rings = 0
        self.s2_ops = 0.0
        for rings in runner.subme2_rings(cur, refs):
            creqs, used = abi.refine_as_cost_reqs(np.concatenate(rings))
            self.n_s2 += int(used.sum())
            self.n_s2_rings += len(creqs)
            self.s2_ops += synth.satd_lane_ops(creqs[used])
            self.s2.append((torch.from_numpy(np.concatenate(rings).view(np.uint8).copy()).to(dev),
                            [(len(r), int(r["pu_w"][0]), int(r["pu_h"][0])) for r in rings if len(r)]))
        self.me_host = runner.me_jobs(cur, refs)
        self.n_me = len(self.me_host)
        self.d_me = torch.from_numpy(self.me_host.view(np.uint8).copy()).to(dev)
        cj = runner.ctu_jobs(cur, refs[0], RANGE)
        npos = (cj["mvmax_x"].astype(np.int64) - cj["mvmin_x"] + 1) * (cj["mvmax_y"].astype(np.int64) - cj["mvmin_y"] + 1)
        self.n_ctu, self.fs_units, self.fs_pos = len(cj), int((npos * vp).sum()), int(npos.sum())
        self.d_fs = torch.from_numpy(cj.view(np.uint8).copy()).to(dev)
        self.plane_bytes = (1 + len(refs)) * runner.seq_plane_bytes
        self.satd_host = rq if cur == runner.keep_host_for else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps per leg; every leg also warms up for 60 ms (clock ramp)")
    ap.add_argument("--gop", type=int, default=8, help="pictures per closed GOP segment")
    ap.add_argument("--segments-per-rank", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1024, help="pictures of the streamed sequence (key `sequence`, strong scaling)")
    ap.add_argument("--seq-gop", type=int, default=32, help="pictures per closed GOP segment of the streamed sequence")
    ap.add_argument("--no-sequence", action="store_true", help="skip the streamed sequence (key `sequence`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hd", action="store_true", help="skip the 1080p legs (key hd)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # X265_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # devices); the driver's runs use RCCL, one rank per GPU
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

    from x265_amd import abi, lib, runner as rn, shard, synth

    ctx = lib.Context(local)
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def leg(fn, steps, warmup):
        """fn(step) launches one step and returns its unit counts; exactly `steps` steps inside the bracket"""
        for s in range(warmup):
            fn(s)
        t0 = time.perf_counter()
        s = warmup
        while time.perf_counter() - t0 < 0.06:  # the clocks ramp under sustained load
            fn(s)
            s += 1
            ctx.sync()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(ext)
        acc = [fn(s) for s in range(steps)]
        e1.record(ext)
        barrier()
        dt = time.perf_counter() - t0
        return max_over_ranks(dt), e0.elapsed_time(e1), acc

    def load_pictures(seq, gop, spr, n_refs=3):
        """this rank's segments of a world * spr * gop picture sequence, all resident; one Picture per P picture"""
        n_frames = world * spr * gop
        vp = valid_pus_per_ctu(seq.w, seq.h)
        pics, runners = [], []
        for i, seg in enumerate(shard.segments_for_rank(n_frames, gop, rank, world)):
            r = rn.SegmentRunner(ctx, seq, n_refs=n_refs, plane_base=i * gop)
            r.load(seg)
            r.seq_plane_bytes = synth.Plane.nbytes_for(seq.w, seq.h)
            r.keep_host_for = seg[0] + min(gop - 1, n_refs) if (rank == 0 and i == 0) else -1
            runners.append(r)
            for cur, refs in r.pairs():
                pics.append(Picture(torch, dev, r, cur, refs, vp))
        return pics, runners, n_frames

    def three_legs(pics, steps, warmup):
        n = len(pics)
        o_satd = torch.empty(max(p.n_satd for p in pics), dtype=torch.int32, device=dev)
        o_ring = torch.empty(max(p.n_ring for p in pics) * abi.RING_COSTS, dtype=torch.int32, device=dev)
        o_me = torch.empty(max(p.n_me for p in pics) * 8, dtype=torch.uint8, device=dev)
        o_fs = torch.empty(max(p.n_ctu for p in pics) * lib.PUS_PER_CTU * 8, dtype=torch.uint8, device=dev)

        def run_satd(s):
            p = pics[s % n]
            ctx.cost_batch_device(p.d_satd.data_ptr(), p.n_satd, o_satd.data_ptr())
            return p

        def run_ring(s):
            p = pics[s % n]
            ctx.refine_sections_device(p.d_rings.data_ptr(), p.ring_sections, o_ring.data_ptr())
            return p

        def run_me(s):
            p = pics[s % n]
            ctx.me_batch_device(p.d_me.data_ptr(), p.n_me, o_me.data_ptr())
            return p

        def run_s2(s):
            p = pics[s % n]
            for d, sections in p.s2:
                ctx.refine_sections_device(d.data_ptr(), sections, o_ring.data_ptr())
            return p

        def run_fs(s):
            p = pics[s % n]
            ctx.fullsearch_device(p.d_fs.data_ptr(), p.n_ctu, o_fs.data_ptr())
            return p

        return leg(run_ring, steps, warmup), leg(run_satd, steps, warmup), leg(run_me, steps, warmup), leg(run_fs, steps, warmup), leg(run_s2, steps, warmup)

    # ---- configs[3] / configs[4]: 4K, GOP-sharded ----------------------------------------------------------------
    UW, UH = 3840, 2160
    seq = synth.Sequence(UW, UH, seed=0x265)
    pics, runners, n_frames = load_pictures(seq, args.gop, args.segments_per_rank)
    (dt_sp, ev_sp, acc_sp), (dt_c1, ev_c1, acc_c1), (dt_me, ev_me, acc_me), (dt_fs, ev_fs, acc_fs), (dt_s2, ev_s2, acc_s2) = \
        three_legs(pics, args.steps, args.warmup)

    # ---- predInterSearch as a stage: the ME-stage list of a picture with all its references as
    # prediction units -- AMVP pick, window, search per iteration, then bits / xCheckBestMVP / choice per unit (P picture:
    # every reference in list 0)
    p3 = max(pics, key=lambda p: len(p.refs))
    ij, pu = synth.inter_units(p3.me_host, len(p3.refs), UH)
    d_ij, d_pu = torch.from_numpy(ij.view(np.uint8).copy()).to(dev), torch.from_numpy(pu.view(np.uint8).copy()).to(dev)
    o_ij = torch.empty(len(ij) * abi.INTER_RESULT.itemsize, dtype=torch.uint8, device=dev)
    o_pu = torch.empty(len(pu) * abi.INTER_PU_RESULT.itemsize, dtype=torch.uint8, device=dev)

    def run_inter(s):
        ctx.inter_decide_batch_device(d_ij.data_ptr(), len(ij), d_pu.data_ptr(), len(pu), UW, UH, o_ij.data_ptr(), o_pu.data_ptr(), any_bidir=False)
    dt_in, ev_in, _ = leg(run_inter, max(4, args.steps // 4), 2)
    inter_leg = {"workload": f"x265_hip_inter_decide_batch: the {len(pu)} square PUs of one 4K P picture as prediction units with {len(p3.refs)} "
                             "iterations each (two AMVP candidates; HEX, subme 2, merange 57): AMVP pick, search window and motionEstimate per "
                             "iteration, bits / xCheckBestMVP / best reference / choice per unit",
                 "units_per_s": len(pu) * max(4, args.steps // 4) * world / dt_in, "iterations_per_unit": len(p3.refs),
                 "kernel_ms_per_frame": ev_in / max(4, args.steps // 4)}
    del d_ij, d_pu, o_ij, o_pu

    # ---- configs[4] as stated: a --frames picture 4K sequence in closed GOPs of --seq-gop, cut over the ranks (strong
    # scaling), streamed: the pictures of the next segment arrive from pinned host memory and the results of finished
    # pictures leave while the current segment is searched (runner.SegmentStreamer); the bracket holds the transfers.
    # One segment's worth of synthetic pictures is cycled (every segment has the same content), so that generating
    # pictures and job lists is not what is timed.
    seq_leg = None
    if not args.no_sequence:
        from x265_amd.hipmem import DeviceBuffer
        for r in runners:
            r.release()
        sg = args.seq_gop
        my_segments = shard.segments_for_rank(args.frames, sg, rank, world)
        probe = synth.Plane(seq.frame(0))
        geo = (probe.buf.nbytes, probe.stride, probe.origin)
        pinned = []
        for i in range(sg):
            b = lib.PinnedBuffer(geo[0])
            b.array[:] = synth.Plane(seq.frame(i)).buf.reshape(-1)
            pinned.append(b)
        lists = {}
        for pos, refs in shard.reference_pairs((0, sg), 3):
            j = synth.me_jobs(UW, UH, pos, refs, [seq.global_motion(pos, r) for r in refs], method=abi.HEX, subme=2, merange=57, seed=pos)
            lists[(pos, 0)] = (DeviceBuffer.from_array(j), len(j))
            j["fenc_plane"] += sg
            j["ref_plane"] += sg
            lists[(pos, 1)] = (DeviceBuffer.from_array(j), len(j))
        n_out = max(n for _, n in lists.values())
        st = rn.SegmentStreamer(ctx, geo, sg, n_out)

        def seq_jobs(cur, seg, s):
            b, n = lists[(cur - seg[0], s)]
            return b.ptr, n

        def seq_frame(pic):
            return pinned[pic % sg]
        st.run(my_segments[:2], seq_frame, seq_jobs)  # warm-up: both plane sets, the clocks
        st.uploaded_bytes = st.downloaded_bytes = 0
        barrier()
        t0 = time.perf_counter()
        st.run(my_segments, seq_frame, seq_jobs)
        barrier()
        dt_seq = max_over_ranks(time.perf_counter() - t0)
        # the same launches with the planes resident and nothing transferred (set 0 holds a whole segment)
        o_res = DeviceBuffer(8 * n_out)
        barrier()
        t0 = time.perf_counter()
        for seg in my_segments:
            for cur, _ in shard.reference_pairs(seg, 3):
                b, n = lists[(cur - seg[0], 0)]
                ctx.me_batch_device(b.ptr, n, o_res.ptr)
        barrier()
        dt_res = max_over_ranks(time.perf_counter() - t0)
        my_pics = sum(b - a for a, b in my_segments)
        seq_leg = {"workload": f"configs[4]: {args.frames} pictures of 3840x2160 in closed GOPs of {sg}, segment s on rank s mod {world} (strong "
                               "scaling), ME stage of every P picture (HEX, subme 2, merange 57, up to 3 references of its segment); planes "
                               "uploaded from pinned host memory and results downloaded inside the timed region, overlapped with the searches "
                               "(two plane sets, copy stream; x265_amd/runner.py SegmentStreamer)",
                   "frames": args.frames, "gop": sg, "segments_this_rank": len(my_segments),
                   "pictures_per_s": args.frames / dt_seq, "seconds": dt_seq,
                   "h2d_GBps_per_gpu": st.uploaded_bytes / dt_seq / 1e9, "d2h_GBps_per_gpu": st.downloaded_bytes / dt_seq / 1e9,
                   "resident_pictures_per_s": args.frames / dt_res,
                   "transfer_overhead": dt_seq / dt_res - 1.0,
                   "note": "pictures_per_s counts every picture of the sequence (the first of a segment is uploaded, not searched)"}
        assert my_pics * geo[0] == st.uploaded_bytes
        st.close()
        o_res.free()
        for b, _ in lists.values():
            b.free()
        for b in pinned:
            b.free()

    hd = None
    if not args.no_hd:
        hseq = synth.Sequence(1920, 1080, seed=0x265)
        for r in runners:
            r.release()  # (nothing to do when the sequence leg has released them)
        hpics, hrunners, _ = load_pictures(hseq, 4, 1)
        (_, h_sp, a_sp), (_, h_c1, a_c1), (_, h_me, a_me), (_, h_fs, a_fs), (_, h_s2, a_s2) = three_legs(hpics, args.steps, 0)
        hd = {"workload": "the same three legs on 1920x1080 pictures (configs[1], configs[2]): one closed segment of 4 pictures per "
                          "GPU, kernel time (HIP events), per GPU",
              "satd_candidates_per_s": sum(p.n_satd for p in a_sp) / h_sp * 1e3,
              "satd_candidates_per_s_single_requests": sum(p.n_satd for p in a_c1) / h_c1 * 1e3,
              "satd_frac_of_valu_peak": sum(p.satd_ops for p in a_sp) / (h_sp * 1e-3) / VALU_PEAK,
              "satd_candidates_per_s_subme2_rings": sum(p.n_s2 for p in a_s2) / h_s2 * 1e3,
              "satd_subme2_frac_of_valu_peak": sum(p.s2_ops for p in a_s2) / (h_s2 * 1e-3) / VALU_PEAK,
              "me_frames_per_s": len(a_me) / h_me * 1e3, "me_calls_per_s": sum(p.n_me for p in a_me) / h_me * 1e3,
              "fullsearch_candidates_per_s": sum(p.fs_units for p in a_fs) / h_fs * 1e3,
              "fullsearch_frac_of_sad_peak": sum(p.fs_pos for p in a_fs) * 4096 / (h_fs * 1e-3) / SAD_PX_PEAK}
        # the lookahead's whole-frame searches on the half-resolution planes of the same four pictures (120 x 68 CUs):
        # 96 (picture, reference) pairs per launch -- what a slice-type decision over a 20-picture window asks for --
        # and one pair alone (a chain of 120 + 2 x 67 dependent searches)
        ctx.set_qp(abi.LOOKAHEAD_QP)
        for i in range(4):
            geo = ctx.lowres_init(i, 1920, 1080, 16 + 4 * i)
        lw, lh = geo.width // 8, geo.lines // 8
        lpairs = [(a, b) for a in range(4) for b in range(4) if a != b] * 8
        lj = np.zeros(len(lpairs), abi.LOOKAHEAD_JOB)
        lj["fenc_plane"], lj["ref_plane"] = [16 + 4 * a for a, _ in lpairs], [16 + 4 * b for _, b in lpairs]
        lj["merange"], lj["method"], lj["subme"], lj["qp"] = 16, abi.HEX, 1, abi.LOOKAHEAD_QP
        d_lj = torch.from_numpy(lj.view(np.uint8).copy()).to(dev)
        o_lj = torch.empty(len(lj) * lw * lh * 8, dtype=torch.uint8, device=dev)

        def run_look(s):
            ctx.lookahead_batch_device(d_lj.data_ptr(), len(lj), lw, lh, o_lj.data_ptr())

        def run_look1(s):
            ctx.lookahead_batch_device(d_lj.data_ptr(), 1, lw, lh, o_lj.data_ptr())
        _, h_lk, _ = leg(run_look, args.steps, 0)
        _, h_lk1, _ = leg(run_look1, args.steps, 0)
        hd["lookahead"] = {"workload": "x265_hip_lookahead_batch: HEX / subme 1 / merange 16 searches of every 8x8 CU of a 960x544 "
                                       "lowres picture in the reference's dependency order, 96 (picture, reference) pairs per launch",
                           "searches_per_s": len(lj) * args.steps / h_lk * 1e3, "cus_per_s": len(lj) * lw * lh * args.steps / h_lk * 1e3,
                           "ms_per_launch": h_lk / args.steps, "single_search_ms": h_lk1 / args.steps,
                           "chain_steps": lw + 2 * (lh - 1)}
        for i in range(16):
            ctx.plane_release(16 + i)
        for r in hrunners:
            r.release()

    # ---- CPU baselines and work models (rank 0, N = 1): the reference's own code where it was built --------------
    cpu = cpu_me = me_model = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle  # checker libraries: used for this baseline leg only
        chk = pyoracle.load_ref() if pyoracle.have_ref() else pyoracle.load_oracle()
        cores = min(16, os.cpu_count() or 1)
        p0 = next(p for p in pics if p.satd_host is not None)
        r0 = runners[0]
        first = r0_first = shard.segments_for_rank(n_frames, args.gop, 0, 1)[0][0]
        # host copies of the planes this picture reads, indexed by plane id
        host_planes = {}
        for pic in [p0.cur] + list(p0.refs):
            host_planes[r0.plane_base + (pic - first)] = synth.Plane(seq.frame(pic)).as_tuple()
        pl = [host_planes.get(i, host_planes[min(host_planes)]) for i in range(max(host_planes) + 1)]
        rq = p0.satd_host
        t0 = time.perf_counter()
        passes = 0
        while time.perf_counter() - t0 < 10.0:
            chk.cost_batch(pl, rq, nthreads=cores)
            passes += 1
        cdt = time.perf_counter() - t0
        what = "reference subpelCompare + satd, C primitives -O3 (no asm: no yasm in the image)" if chk.kind == "reference" else "C restatement"
        cpu = {"value": passes * len(rq) / cdt, "unit": "SATD candidates/s", "cores": cores, "kind": chk.kind,
               "sample": f"{passes} pass(es) over the {len(rq)} requests of one 4K P picture with {len(p0.refs)} references; {what}; {cdt:.1f} s"}
        tabs = {q: lib.costtab_build(q) for q in range(28, 40)}
        t0 = time.perf_counter()
        mp = 0
        while time.perf_counter() - t0 < 8.0:
            chk.me_batch(pl, tabs, p0.me_host, nthreads=cores)
            mp += 1
        mdt = time.perf_counter() - t0
        cpu_me = {"frames_per_s": mp / mdt, "me_calls_per_s": mp * p0.n_me / mdt, "cores": cores, "kind": chk.kind,
                  "sample": f"{mp} pass(es) over one 4K picture's {p0.n_me} motionEstimate calls, {mdt:.1f} s"}
        # what the searches of that picture evaluate, priced at SURVEY.md 8(d)'s algorithmic lane-ops (oracle work model)
        _, me_model = pyoracle.load_oracle().me_batch_ops(pl, tabs, p0.me_host, nthreads=cores)
        me_model["calls"] = p0.n_me
        if not (args.no_hd and args.no_sequence):  # the later legs released the 4K planes: put this picture's back
            for pid, (buf, origin, stride) in host_planes.items():
                ctx.plane_upload(pid, buf, stride, origin)
        o_model = torch.empty(p0.n_me * 8, dtype=torch.uint8, device=dev)

        def run_model(s):
            ctx.me_batch_device(p0.d_me.data_ptr(), p0.n_me, o_model.data_ptr())
            return p0
        _, ev_model, _ = leg(run_model, 20, 0)
        me_model["kernel_ms"] = ev_model / 20

    if rank == 0:
        def profile_json(name):
            path = os.path.join(ROOT, "profiles", name)
            return json.load(open(path)) if os.path.exists(path) else {}

        traffic = profile_json("r03_traffic.json") or profile_json("r02_traffic.json")
        K = args.steps
        cands = sum(p.n_satd for p in acc_sp)
        ops = sum(p.satd_ops for p in acc_sp)
        k_ms = ev_sp / K
        # algorithmic bytes of a launch: the planes it reads once (source + references) + 20 B per ring request + 4 B per candidate cost
        alg_bytes = sum(p.plane_bytes + 20 * p.n_ring + 36 * p.n_ring for p in acc_sp) / K
        t_cost = traffic.get("refine_kernel_4k")
        line = {
            "metric": "SATD ME candidates/sec + ME-stage frames/sec @1080p/4K, 1-8 GPU, bit-exact",
            "value": cands * world / dt_sp, "unit": "SATD candidates/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt_sp / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[3]/[4]: 3840x2160, closed GOP segments sharded over the ranks, sub-pel stage of one P picture "
                                   "per step: square PUs 64..8 x up to 3 references of the segment x 16 half-/quarter-pel candidates, "
                                   "8-tap interpolation + SATD, asked for as rings of 8 around the integer vector, the PU sizes as four runs "
                                   "of one list (x265_hip_refine_sections_device, one launch per step)",
                       "sequence_frames": n_frames, "gop": args.gop, "segments_per_rank": args.segments_per_rank,
                       "p_pictures_per_rank": len(pics), "candidates_per_step": cands / K},
            "roofline": {"bound": "valu", "kernel": "refine_kernel", "kernel_ms": k_ms, "launches_per_step": 1,
                         "achieved": ops / K / (k_ms * 1e-3) / 1e12, "peak": VALU_PEAK / 1e12, "unit": "Tlane-op/s",
                         "frac": ops / K / (k_ms * 1e-3) / VALU_PEAK,
                         "algorithmic_lane_ops": ops / K,
                         "issue_ceiling": {"peak": VALU_ISSUE_4CYC / 1e12, "frac": ops / K / (k_ms * 1e-3) / VALU_ISSUE_4CYC,
                                           "note": "measured issue rate of the 4-cycle integer forms the kernel is made of"},
                         "algorithmic_bytes": alg_bytes, "traffic": t_cost,
                         "hbm": {"achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                                 "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK}},
            "satd_single_requests": {"workload": "the same candidates one request each (x265_hip_cost_batch: nothing shared between candidates)",
                                     "candidates_per_s": sum(p.n_satd for p in acc_c1) * world / dt_c1, "ms_per_frame": dt_c1 / K * 1e3,
                                     "roofline": {"bound": "valu", "kernel": "cost_kernel", "kernel_ms": ev_c1 / K,
                                                  "achieved": sum(p.satd_ops for p in acc_c1) / (ev_c1 * 1e-3) / 1e12, "peak": VALU_PEAK / 1e12,
                                                  "unit": "Tlane-op/s", "frac": sum(p.satd_ops for p in acc_c1) / (ev_c1 * 1e-3) / VALU_PEAK,
                                                  "traffic": traffic.get("cost_kernel_4k")}},
            "satd_rings_subme2": {"workload": "what the medium preset's sub-pel stage asks for (workload[2], motion.cpp:51): per square PU and "
                                              "reference a 4-point half-pel ring with the centre, then a 4-point quarter-pel ring, around "
                                              "the winners of the integer search of that PU (x265_hip_fullsearch, +-64) and of the first "
                                              "ring: 9 candidates per (PU, reference), two launches per picture (x265_hip_refine_sections_device)",
                                  "candidates_per_s": sum(p.n_s2 for p in acc_s2) * world / dt_s2, "ms_per_frame": dt_s2 / K * 1e3,
                                  "rings_per_frame": sum(p.n_s2_rings for p in acc_s2) / K,
                                  "roofline": {"bound": "valu", "kernel": "refine_kernel", "kernel_ms": ev_s2 / K, "launches_per_step": 2,
                                               "achieved": sum(p.s2_ops for p in acc_s2) / (ev_s2 * 1e-3) / 1e12, "peak": VALU_PEAK / 1e12,
                                               "unit": "Tlane-op/s", "frac": sum(p.s2_ops for p in acc_s2) / (ev_s2 * 1e-3) / VALU_PEAK,
                                               "algorithmic_lane_ops": sum(p.s2_ops for p in acc_s2) / K}},
            "me_stage": {"workload": "motionEstimate replay of the same pictures: HEX, subme 2, merange 57, square PUs 64..8 x the "
                                     "segment's references (x265_hip_me_batch)",
                         "frames_per_s": K * world / dt_me, "me_calls_per_s": sum(p.n_me for p in acc_me) * world / dt_me,
                         "ms_per_frame": dt_me / K * 1e3, "kernel_ms_per_frame": ev_me / K},
            "fullsearch": {"workload": "configs[1] at 4K: +-64 integer SAD search of every CTU (85 square PUs) against the nearest reference",
                           "candidates_per_s": sum(p.fs_units for p in acc_fs) * world / dt_fs, "ms_per_frame": dt_fs / K * 1e3,
                           "roofline": {"bound": "valu", "kernel": "fullsearch2_kernel", "kernel_ms": ev_fs / K,
                                        "achieved": sum(p.fs_pos for p in acc_fs) * 4096 / (ev_fs * 1e-3) / 1e12, "peak": SAD_PX_PEAK / 1e12,
                                        "unit": "Tpixel-op/s", "frac": sum(p.fs_pos for p in acc_fs) * 4096 / (ev_fs * 1e-3) / SAD_PX_PEAK,
                                        "traffic": traffic.get("fullsearch2_kernel_4k")}},
        }
        if me_model:
            ms = me_model["kernel_ms"]  # that picture's list timed on its own
            line["me_stage"]["roofline"] = {
                "bound": "valu", "kernel": "me_team_kernel + me_rest_kernel + me_big_kernel (one routed launch)", "kernel_ms": ms, "algorithmic_lane_ops": me_model["lane_ops"],
                "achieved": me_model["lane_ops"] / (ms * 1e-3) / 1e12, "peak": VALU_PEAK / 1e12, "unit": "Tlane-op/s",
                "frac": me_model["lane_ops"] / (ms * 1e-3) / VALU_PEAK,
                "issue_ceiling": {"peak": VALU_ISSUE_4CYC / 1e12, "frac": me_model["lane_ops"] / (ms * 1e-3) / VALU_ISSUE_4CYC},
                "model": {k: me_model[k] for k in ("sad_ops", "satd_ops", "interp_ops", "sad_candidates", "subpel_candidates", "calls")},
                "traffic": (sum(traffic.get(k + "_4k", 0) for k in ("me_kernel", "me_big_kernel", "me_route_kernel", "me_team_kernel", "me_rest_kernel")) or None)}
        line["inter_stage"] = inter_leg
        if seq_leg:
            line["sequence"] = seq_leg
        if hd:
            line["hd"] = hd
        if cpu:
            line["cpu_baseline"] = cpu
            line["vs_cpu_baseline"] = line["value"] / cpu["value"]
            line["me_stage"]["cpu_baseline"] = cpu_me
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
