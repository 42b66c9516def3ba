"""CPU tests of the multi-rank path: GOP sharding is a pure partition (no data-path collective);
the world_size-2 gloo test exercises the same barrier / max-over-ranks bracket bench.py uses."""
import os

import pytest


def test_segments_partition_the_sequence():
    from x265_amd import shard
    for n, gop, world in ((1024, 32, 8), (1000, 64, 4), (7, 3, 2), (0, 8, 2), (64, 64, 8)):
        seen = []
        for r in range(world):
            for a, b in shard.segments_for_rank(n, gop, r, world):
                assert 0 < b - a <= gop
                seen.extend(range(a, b))
        assert sorted(seen) == list(range(n))
        counts = [len(shard.segments_for_rank(n, gop, r, world)) for r in range(world)]
        assert max(counts) - min(counts) <= 1


def test_reference_pairs_stay_inside_the_segment():
    from x265_amd import shard
    pairs = shard.reference_pairs((32, 64), 3)
    assert pairs[0] == (33, [32]) and pairs[1] == (34, [33, 32]) and pairs[-1] == (63, [62, 61, 60])
    assert all(min(refs) >= 32 for _, refs in pairs)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from x265_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    segs = shard.segments_for_rank(100, 16, rank, world)
    frames = [f for a, b in segs for f in range(a, b)]
    gathered = [None] * world
    dist.all_gather_object(gathered, frames)  # test-only exchange: proves the shards are disjoint and complete
    dist.barrier()
    t = shard.max_over_ranks(1.0 + rank, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, gathered, t))


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, gathered, t in out:
        assert sorted(gathered[0] + gathered[1]) == list(range(100))
        assert not set(gathered[0]) & set(gathered[1])
        assert t == 2.0  # max over ranks of (1.0, 2.0)


def test_runner_plan_is_the_same_sharded_and_unsharded():
    """the sequence runner's plan (which picture is searched against which references, in which plane slots) is a
    function of the segment alone, so it does not depend on the number of ranks"""
    from x265_amd import shard
    def plan(rank, world):
        return [(seg, cur, refs) for seg in shard.segments_for_rank(40, 8, rank, world) for cur, refs in shard.reference_pairs(seg, 3)]
    whole = plan(0, 1)
    for world in (2, 4, 8):
        parts = [p for r in range(world) for p in plan(r, world)]
        assert sorted(parts) == sorted(whole)
    assert all(seg[0] <= min(refs) and max(refs) < cur < seg[1] for seg, cur, refs in whole)
