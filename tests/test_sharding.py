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


class _FakeStreams:
    """x265_hip_ctx as the streamed runner sees it: two in-order streams and two fences.  Records every operation with
    the operations it is ordered behind (its stream's previous one and, through the fences, the other stream's)."""

    def __init__(self):
        self.ops = []                  # (kind, payload, set of op indices that happen before it)
        self.last = {"copy": None, "compute": None}
        self.gate = {"copy": set(), "compute": set()}  # what the stream's next operations wait for (a fence)

    def _add(self, stream, kind, payload):
        before = set(self.gate[stream])
        if self.last[stream] is not None:
            before.add(self.last[stream])
        closure = set(before)
        for b in before:
            closure |= self.ops[b][2]
        self.ops.append((kind, payload, closure))
        self.last[stream] = len(self.ops) - 1

    def plane_alloc(self, *a):
        pass

    def plane_release(self, *a):
        pass

    def plane_upload_async(self, plane, buf):
        self._add("copy", "upload", plane)

    def download_async(self, dev, host, nbytes):
        self._add("copy", "download", (dev, host.tag))

    def me_batch_device(self, d_jobs, n, d_out):
        self._add("compute", "launch", (d_jobs, d_out))

    def compute_wait_copy(self):
        if self.last["copy"] is not None:
            self.gate["compute"].add(self.last["copy"])

    def copy_wait_compute(self):
        if self.last["compute"] is not None:
            self.gate["copy"].add(self.last["compute"])

    def sync(self):
        pass

    def copy_sync(self):
        pass


def test_streamed_runner_orders_transfers_and_searches(monkeypatch):
    """runner.SegmentStreamer against a model of the two streams: every search is ordered behind the last upload of every
    plane it reads and is not concurrent with any upload into them, every result download lies between the search that
    wrote the slot and the next search that writes it, and every P picture's results are downloaded exactly once"""
    from x265_amd import hipmem, lib, runner, shard

    class Buf:
        n = 0

        def __init__(self, nbytes):
            Buf.n += 1
            self.ptr = self.tag = 1000 + Buf.n
            self.array = bytearray(0)

        def free(self):
            pass

    monkeypatch.setattr(hipmem, "DeviceBuffer", Buf)
    monkeypatch.setattr(lib, "PinnedBuffer", Buf)
    for frames, gop, rank, world in ((70, 8, 0, 1), (70, 8, 1, 2), (33, 16, 0, 3), (5, 1, 0, 1), (9, 4, 1, 2)):
        ctx = _FakeStreams()
        st = runner.SegmentStreamer(ctx, (4096, 64, 0), gop, 100)
        segments = shard.segments_for_rank(frames, gop, rank, world)
        reads = {}

        def jobs(cur, seg, s, reads=reads):
            refs = dict(shard.reference_pairs(seg, 3))[cur]
            token = len(reads) + 1
            reads[token] = (cur, {s * gop + p - seg[0] for p in [cur] + refs})
            return token, 10

        st.run(segments, lambda pic: Buf(1), jobs)
        ops = ctx.ops
        launches = [i for i, o in enumerate(ops) if o[0] == "launch"]
        assert [reads[ops[i][1][0]][0] for i in launches] == [cur for seg in segments for cur, _ in shard.reference_pairs(seg, 3)]
        uploads = [i for i, o in enumerate(ops) if o[0] == "upload"]
        assert len(uploads) == sum(b - a for a, b in segments)
        for i in launches:
            planes = reads[ops[i][1][0]][1]
            for p in planes:
                ups = [u for u in uploads if ops[u][1] == p]
                assert any(u in ops[i][2] for u in ups), "a search reads a plane nothing was uploaded into before it"
                for u in ups:
                    assert u in ops[i][2] or i in ops[u][2], "an upload into a plane runs beside a search that reads it"
        downloads = [i for i, o in enumerate(ops) if o[0] == "download"]
        assert len(downloads) == len(launches)
        for k, i in enumerate(launches):
            slot = ops[i][1][1]
            mine = [d for d in downloads if ops[d][1][0] == slot and i in ops[d][2]]
            assert mine, "a search's results are never downloaded behind it"
            d = min(mine)
            later = [j for j in launches[k + 1:] if ops[j][1][1] == slot]
            if later:
                assert d in ops[later[0]][2], "a slot is written again before its results have left"
                assert not any(ops[x][1][0] == slot and i in ops[x][2] and later[0] not in ops[x][2] and x != d for x in downloads)
