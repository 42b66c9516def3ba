"""Sequence runner: the ME stage of closed GOP segments, one rank per GPU (BASELINE.json configs[4]).

A sequence is cut into closed segments (`shard.gop_segments`; the reference's closed-GOP settings are
`--no-open-gop --keyint <gop>`, source/common/param.cpp:118-120); segment s belongs to rank s mod world
(`shard.segments_for_rank`).  A rank uploads the pictures of one of its segments, runs every P picture of the
segment against the up to `n_refs` previous pictures of the same segment (`shard.reference_pairs`) and releases the
planes when the segment is done.  Nothing crosses ranks: no collective on the data path.

Plane ids are segment-local: picture p of a segment starting at `first` lives in plane `base + (p - first)`, so the
job lists of a segment do not depend on which rank or in which order it is processed (sharded == unsharded results).
"""
import numpy as np

from . import abi, shard, synth


class SegmentRunner:
    def __init__(self, ctx, seq, n_refs=3, method=abi.HEX, subme=2, merange=57, qps=range(28, 40), plane_base=0):
        self.ctx, self.seq = ctx, seq
        self.w, self.h = seq.w, seq.h
        self.n_refs, self.method, self.subme, self.merange = n_refs, method, subme, merange
        self.plane_base = plane_base
        self.segment = None
        for q in qps:
            ctx.set_qp(q)

    # ---- planes ---------------------------------------------------------------------------------------------
    def plane_of(self, pic):
        first, last = self.segment
        assert first <= pic < last, "picture outside the resident segment"
        return self.plane_base + (pic - first)

    def load(self, segment, keep_host=False):
        """upload the pictures of `segment` (first, last_exclusive); returns the host planes if asked for (tests)"""
        self.release()
        first, last = segment
        if last - first > abi.MAX_PLANES - self.plane_base:
            raise ValueError("segment longer than the plane directory")
        self.segment = segment
        host = []
        for pic in range(first, last):
            p = synth.Plane(self.seq.frame(pic))
            self.ctx.plane_upload(self.plane_of(pic), p.buf, p.stride, p.origin)
            if keep_host:
                host.append(p)
        return host

    def release(self):
        if self.segment is not None:
            first, last = self.segment
            for pic in range(first, last):
                self.ctx.plane_release(self.plane_base + (pic - first))
            self.segment = None

    # ---- work lists of one P picture ------------------------------------------------------------------------------
    def pairs(self):
        """(current, [references]) of the resident segment's P pictures"""
        return shard.reference_pairs(self.segment, self.n_refs)

    def _motions(self, cur, refs):
        return [self.seq.global_motion(cur, r) for r in refs]

    def me_jobs(self, cur, refs):
        return synth.me_jobs(self.w, self.h, self.plane_of(cur), [self.plane_of(r) for r in refs], self._motions(cur, refs),
                             method=self.method, subme=self.subme, merange=self.merange, seed=cur)

    def satd_requests(self, cur, refs):
        return synth.subpel_requests(self.w, self.h, self.plane_of(cur), [self.plane_of(r) for r in refs], self._motions(cur, refs))

    def ring_requests(self, cur, refs):
        """the same candidates as satd_requests() as ring requests (x265_hip_refine_batch: a half-pel and a quarter-pel
        ring of 8 per PU and reference), as an encoder that works depth by depth produces them: one list per PU size
        (64, 32, 16, 8), each submitted on its own"""
        rq = synth.refine_requests(self.w, self.h, self.plane_of(cur), [self.plane_of(r) for r in refs], self._motions(cur, refs))
        return [rq[rq["pu_w"] == s] for s in (64, 32, 16, 8)]

    def ctu_jobs(self, cur, ref, search_range=64, qp=30):
        """integer full search of every CTU of `cur` against `ref` (configs[1] at this picture size)"""
        from .lib import CTU_JOB
        ncx, ncy = synth.ctu_grid(self.w, self.h)
        dx, dy = self.seq.global_motion(cur, ref)
        cx, cy = np.meshgrid(np.arange(ncx) * 64, np.arange(ncy) * 64, indexing="xy")
        cx, cy = cx.reshape(-1), cy.reshape(-1)
        jobs = np.zeros(ncx * ncy, CTU_JOB)
        lox, loy, hix, hiy = synth.search_range(np.int64(4 * dx), np.int64(4 * dy), search_range, cx, cy, self.w, self.h)
        jobs["fenc_plane"], jobs["ref_plane"] = self.plane_of(cur), self.plane_of(ref)
        jobs["ctu_x"], jobs["ctu_y"] = cx, cy
        jobs["mvmin_x"], jobs["mvmin_y"], jobs["mvmax_x"], jobs["mvmax_y"] = lox, loy, hix, hiy
        jobs["qmvp_x"], jobs["qmvp_y"], jobs["qp"] = 4 * dx, 4 * dy, qp
        return jobs

    # ---- the ME stage of a segment through the host-array entry point (tests, small runs) -------------------------
    def run_segment(self, segment):
        """{cur: results of its motionEstimate calls} for every P picture of `segment`"""
        self.load(segment)
        out = {}
        for cur, refs in self.pairs():
            out[cur] = self.ctx.me_batch(self.me_jobs(cur, refs))
        self.release()
        return out


def run_sequence(ctx, seq, n_frames, gop, rank=0, world=1, **kw):
    """ME stage of this rank's segments of an n_frames sequence: {cur: results}"""
    runner = SegmentRunner(ctx, seq, **kw)
    out = {}
    for seg in shard.segments_for_rank(n_frames, gop, rank, world):
        out.update(runner.run_segment(seg))
    return out
