"""BASELINE configs[4] on one GPU: the GOP-sharded sequence runner (x265_amd/runner.py).  A short 3840x2160 sequence is
cut into closed segments; two "ranks" (two contexts on device 0, as two processes would hold them) each run their own
segments, and together they must give exactly what one rank running every segment gives, and what the oracle gives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, FRAMES, GOP = 3840, 2160, 10, 4  # segments (0,4) (4,8) (8,10): a full GOP, a full GOP, a short tail


@pytest.fixture(scope="module")
def seq4k():
    from x265_amd import synth
    return synth.Sequence(W, H, seed=0x4C0)


def test_sharded_equals_unsharded_and_the_oracle(seq4k, oracle):
    from x265_amd import lib, runner, shard
    one = lib.Context(0)
    whole = runner.run_sequence(one, seq4k, FRAMES, GOP, rank=0, world=1, n_refs=3)
    one.close()
    # P pictures: every picture but the first of each segment
    assert sorted(whole) == [1, 2, 3, 5, 6, 7, 9]
    parts = {}
    ranks = [lib.Context(0), lib.Context(0)]
    for r, ctx in enumerate(ranks):
        mine = runner.run_sequence(ctx, seq4k, FRAMES, GOP, rank=r, world=2, n_refs=3)
        assert not set(mine) & set(parts)  # disjoint shards
        parts.update(mine)
    for ctx in ranks:
        ctx.close()
    assert sorted(parts) == sorted(whole)
    for cur in whole:
        assert np.array_equal(parts[cur], whole[cur]), cur
    # picture 7 has three references (6, 5, 4), picture 9 one (8): both against the oracle
    ctx = lib.Context(0)
    run = runner.SegmentRunner(ctx, seq4k, n_refs=3)
    tabs = {q: lib.costtab_build(q) for q in range(28, 40)}
    for seg, cur in (((4, 8), 7), ((8, 10), 9)):
        host = run.load(seg, keep_host=True)
        refs = dict(run.pairs())[cur]
        assert refs == list(range(cur - 1, max(seg[0], cur - 3) - 1, -1))
        jobs = run.me_jobs(cur, refs)
        assert len(jobs) == 172020 * len(refs)
        want = oracle.me_batch([p.as_tuple() for p in host], tabs, jobs, nthreads=16)
        assert np.array_equal(whole[cur], want), cur
        # and the segment's sub-pel SATD list of that picture (what bench.py times), sampled
        reqs = run.satd_requests(cur, refs)[::37].copy()
        assert np.array_equal(ctx.cost_batch(reqs), oracle.cost_batch([p.as_tuple() for p in host], reqs, nthreads=16))
    run.release()
    ctx.close()


def test_a_64_picture_segment_with_its_lookahead_planes_fits_the_plane_directory():
    """a 64-picture GOP with four lookahead planes per picture is 320 planes (the directory held 64 in round 1)"""
    from x265_amd import abi, lib, synth
    seq = synth.Sequence(416, 240, seed=11, n_patches=2)
    ctx = lib.Context(0)
    ctx.set_qp(12)  # X265_LOOKAHEAD_QP
    geo = None
    for i in range(64):
        p = synth.Plane(seq.frame(i % 5))
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
        geo = ctx.lowres_init(i, 416, 240, 64 + 4 * i)
    assert 64 + 4 * 63 + 3 < abi.MAX_PLANES
    # a lookahead search of the last picture against the first, through the highest plane ids
    jobs = synth.lookahead_jobs(geo.width, geo.lines, 64 + 4 * 63, 64, (0, 0))
    res = ctx.me_batch(jobs)
    assert len(res) == len(jobs) and (res["cost"] >= 0).all()
    ctx.close()


def test_streamed_sequence_of_144_pictures_sharded_equals_unsharded_and_the_oracle(oracle):
    """configs[4]'s runner with the transfers overlapped (runner.SegmentStreamer: double-buffered plane sets, uploads of
    the next segment and downloads of finished results on the copy stream beside the searches): 144 distinct 416x240
    pictures in closed GOPs of 16 -- as one rank, as two ranks (two contexts on device 0), and through the
    synchronous runner must all give the same results; sampled pictures equal the oracle."""
    from x265_amd import abi, lib, runner, shard, synth
    from x265_amd.hipmem import DeviceBuffer
    w, h, frames, gop = 416, 240, 144, 16
    seq = synth.Sequence(w, h, seed=0x51, n_patches=3)
    geo = (synth.Plane.nbytes_for(w, h),) + synth.Plane(seq.frame(0)).as_tuple()[2:0:-1]  # bytes, stride, origin
    host_planes, pinned = {}, {}

    def frame(pic):
        if pic not in pinned:
            p = synth.Plane(seq.frame(pic))
            host_planes[pic] = p
            b = lib.PinnedBuffer(p.buf.nbytes)
            b.array[:] = p.buf.reshape(-1)
            pinned[pic] = b
        return pinned[pic]

    def stream(rank, world):
        ctx = lib.Context(0)
        for q in range(28, 40):
            ctx.set_qp(q)
        lists = {}

        def jobs(cur, seg, s):
            refs = dict(shard.reference_pairs(seg, 3))[cur]
            j = synth.me_jobs(w, h, s * gop + cur - seg[0], [s * gop + r - seg[0] for r in refs], [seq.global_motion(cur, r) for r in refs],
                              method=abi.HEX, subme=2, merange=57, seed=cur)
            lists[cur] = DeviceBuffer.from_array(j)
            return lists[cur].ptr, len(j)
        st = runner.SegmentStreamer(ctx, geo, gop, 3 * len(synth.square_pus(w, h, (64, 32, 16, 8))))
        out = {}
        st.run(shard.segments_for_rank(frames, gop, rank, world), frame, jobs, on_result=lambda pic, res: out.__setitem__(pic, res.copy()))
        assert st.uploaded_bytes == geo[0] * sum(b - a for a, b in shard.segments_for_rank(frames, gop, rank, world))
        st.close()
        ctx.close()
        return out

    whole = stream(0, 1)
    assert sorted(whole) == [p for p in range(frames) if p % gop]
    parts = {}
    for r in range(2):
        mine = stream(r, 2)
        assert not set(mine) & set(parts)
        parts.update(mine)
    assert sorted(parts) == sorted(whole)
    for cur in whole:
        assert np.array_equal(parts[cur], whole[cur]), cur
    # the synchronous runner on the same sequence (plane ids of set 0)
    ctx = lib.Context(0)
    sync = runner.run_sequence(ctx, seq, frames, gop, n_refs=3)
    ctx.close()
    for cur in whole:
        assert np.array_equal(sync[cur], whole[cur]), cur
    # sampled pictures against the oracle: one of an even segment (plane set 0), one of an odd one (set 1)
    tabs = {q: lib.costtab_build(q) for q in range(28, 40)}
    for cur in (37, 21, 143):
        seg = (cur // gop * gop, min(cur // gop * gop + gop, frames))
        refs = dict(shard.reference_pairs(seg, 3))[cur]
        j = synth.me_jobs(w, h, cur - seg[0], [r - seg[0] for r in refs], [seq.global_motion(cur, r) for r in refs], method=abi.HEX, subme=2,
                          merange=57, seed=cur)
        planes = [host_planes[p].as_tuple() for p in range(seg[0], seg[1])]
        assert np.array_equal(whole[cur], oracle.me_batch(planes, tabs, j, nthreads=16)), cur
    for b in pinned.values():
        b.free()
