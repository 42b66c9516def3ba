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
