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

    def subme2_rings(self, cur, refs, search_range=64):
        """the medium preset's sub-pel stage of picture `cur` as two ring lists (synth.subme2_hpel_rings / _qpel_rings),
        centred on the winners of the integer search of every PU (x265_hip_fullsearch, +-search_range) and, for the
        second ring, of the first: ([half-pel lists per PU size], [quarter-pel lists per PU size]).  Runs the
        searches and the first rings on the device (set-up, not timed by anyone)."""
        mask = synth.ctu_pu_mask(self.w, self.h)
        winners = []
        for r in refs:
            fs = self.ctx.fullsearch(self.ctu_jobs(cur, r, search_range)).reshape(-1)[mask]
            winners.append(np.stack([fs["mvx"], fs["mvy"]], axis=1).astype(np.int64))
        hp = synth.subme2_hpel_rings(self.w, self.h, self.plane_of(cur), [self.plane_of(r) for r in refs], winners)
        qp = synth.subme2_qpel_rings(hp, self.ctx.refine_batch(hp))
        return [hp[hp["pu_w"] == s] for s in (64, 32, 16, 8)], [qp[qp["pu_w"] == s] for s in (64, 32, 16, 8)]

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


class SegmentStreamer:
    """The ME stage of a rank's closed GOP segments with the transfers overlapped (BASELINE.json configs[4]).

    Two sets of `gop` planes alternate: while the searches of segment k read one set on the context's stream, the
    pictures of segment k + 1 arrive in the other over the context's copy stream from pinned host memory, and the
    results of every finished picture leave over it (x265_hip_plane_upload_async / x265_hip_download_async and the
    two fences, include/x265_hip.h).  Per picture of segment k, in host program order:

        compute_wait_copy()                       # this search starts after the transfers queued one picture ago
        copy_wait_compute()                       # transfers from here on start after the kernels queued so far ...
        download_async(results of the picture before)        # ... so this one follows that picture's search
        plane_upload_async(one picture of segment k + 1)      # its plane set was last read by segment k - 1, long over
        me_batch_device(this picture)

    so one picture's transfers run beside that picture's search.  Plane ids are segment-local and the two sets
    are addressed by the same ids (set 1 = ids + gop), so a picture's job list exists once per set.

    `frames(pic)` -> PinnedBuffer with the padded plane of picture `pic`; `jobs(cur, segment, set)` -> (device
    address, count) of that picture's job list for a plane set; both are the caller's (bench.py cycles one segment's
    worth of synthetic pictures so that generating them is not what is timed, the tests use distinct pictures).
    """

    def __init__(self, ctx, plane_geometry, gop, n_out_max, n_refs=3):
        from . import lib
        from .hipmem import DeviceBuffer
        self.ctx, self.gop, self.n_refs = ctx, gop, n_refs
        self.nbytes, self.stride, self.origin = plane_geometry
        for i in range(2 * gop):
            ctx.plane_alloc(i, self.nbytes, self.stride, self.origin)
        # results: two device arrays and two pinned arrays in turn
        self.d_out = [DeviceBuffer(8 * n_out_max) for _ in range(2)]
        self.h_out = [lib.PinnedBuffer(8 * n_out_max) for _ in range(2)]
        self.uploaded_bytes = self.downloaded_bytes = 0

    def close(self):
        self.ctx.sync()
        self.ctx.copy_sync()
        for b in self.d_out:
            b.free()
        for b in self.h_out:
            b.free()
        for i in range(2 * self.gop):
            self.ctx.plane_release(i)

    def run(self, segments, frames, jobs, on_result=None):
        """`segments`: this rank's [(first, last_exclusive)]; `frames(pic)` -> PinnedBuffer; `jobs(cur, segment, set)`
        -> (device address, count); on_result(pic, ME_RESULT array) for every P picture, called once the picture's
        results are on the host (the array is only valid during the call)"""
        ctx = self.ctx
        if not segments:
            return
        pending = None  # (pic, slot, count): searched, download not yet queued
        queued = None   # download queued, not yet handed to the host
        n_launch = 0

        def upload(pic, seg, s):
            ctx.plane_upload_async(s * self.gop + (pic - seg[0]), frames(pic))
            self.uploaded_bytes += self.nbytes

        def hand_over(done):
            if done is not None and on_result is not None:
                ctx.copy_sync()  # the copy stream has passed the slot's download
                pic, slot, n = done
                on_result(pic, self.h_out[slot].array[: 8 * n].view(abi.ME_RESULT))

        def transfers():
            """one picture's worth of transfers behind the kernels queued so far"""
            nonlocal pending, queued
            ctx.copy_wait_compute()
            hand_over(queued)
            queued = None
            if pending is not None:
                pic, slot, n = pending
                ctx.download_async(self.d_out[slot].ptr, self.h_out[slot], 8 * n)
                self.downloaded_bytes += 8 * n
                queued, pending = pending, None

        for pic in range(*segments[0]):
            upload(pic, segments[0], 0)
        for k, seg in enumerate(segments):
            s = k & 1
            nxt = segments[k + 1] if k + 1 < len(segments) else None
            nxt_pics = list(range(*nxt)) if nxt else []
            pairs = shard.reference_pairs(seg, self.n_refs)
            for j, (cur, refs) in enumerate(pairs):
                d_jobs, n = jobs(cur, seg, s)
                # this launch starts after the transfers queued one launch ago and earlier: its pictures have arrived (they
                # were queued during the segment before), and the results that last used its output slot have left
                ctx.compute_wait_copy()
                transfers()
                # the next segment's pictures, spread over this segment's launches (its plane set was last read by the
                # segment before this one)
                for _ in range(-(-len(nxt_pics) // (len(pairs) - j))):
                    upload(nxt_pics.pop(0), nxt, s ^ 1)
                slot = n_launch & 1
                ctx.me_batch_device(d_jobs, n, self.d_out[slot].ptr)
                pending = (cur, slot, n)
                n_launch += 1
            for pic in nxt_pics:  # (a segment without P pictures has no launches to spread them over)
                upload(pic, nxt, s ^ 1)
        transfers()
        hand_over(queued)
        ctx.sync()
        ctx.copy_sync()


def run_sequence(ctx, seq, n_frames, gop, rank=0, world=1, **kw):
    """ME stage of this rank's segments of an n_frames sequence: {cur: results}"""
    runner = SegmentRunner(ctx, seq, **kw)
    out = {}
    for seg in shard.segments_for_rank(n_frames, gop, rank, world):
        out.update(runner.run_segment(seg))
    return out
