"""BASELINE.json configs[4] at its stated size, once, on one GPU: a sequence of 1024 distinct 3840x2160 pictures in closed
GOPs of 32 through the streamed runner (x265_amd/runner.py SegmentStreamer: planes uploaded and results downloaded beside
the searches) as one rank, and as eight ranks -- eight contexts one after the other, each with the segments
shard.segments_for_rank gives it, which is what eight processes on eight GPUs would each do.  The results of every P
picture (516,060 motionEstimate calls with three references) are compared between the two by CRC, and sampled pictures
with the oracle.  The pictures and job lists are made by worker processes as the runner asks for them (nothing of that
size is kept).  argv: [frames [gop [world]]].  Run through gpurun from the repository root; the verdict goes to profiles/."""
import multiprocessing as mp
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

W, H = 3840, 2160
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
GOP = int(sys.argv[2]) if len(sys.argv) > 2 else 32
WORLD = int(sys.argv[3]) if len(sys.argv) > 3 else 8
SEED = 0x4C4
_seq = None


def _init():
    global _seq
    from x265_amd import synth
    _seq = synth.Sequence(W, H, seed=SEED)


def make(pic):
    """(picture, its padded plane, its job list with the plane ids of set 0 or None for the first picture of a segment)"""
    from x265_amd import abi, shard, synth
    seg = (pic // GOP * GOP, min(pic // GOP * GOP + GOP, FRAMES))
    plane = synth.Plane(_seq.frame(pic))
    refs = dict(shard.reference_pairs(seg, 3)).get(pic)
    jobs = None
    if refs:
        jobs = synth.me_jobs(W, H, pic - seg[0], [r - seg[0] for r in refs], [_seq.global_motion(pic, r) for r in refs], method=abi.HEX, subme=2,
                             merange=57, seed=pic)
    return pic, plane.buf, jobs


def main():
    pool = mp.get_context("spawn").Pool(14, initializer=_init)  # before anything touches the GPU
    from oracle import pyoracle
    from x265_amd import lib, runner, shard, synth
    from x265_amd.hipmem import DeviceBuffer
    _init()
    p0 = synth.Plane(_seq.frame(0))
    geo = (p0.buf.nbytes, p0.stride, p0.origin)
    n_out = 3 * len(synth.square_pus(W, H, (64, 32, 16, 8)))
    sample = sorted({GOP + 1, GOP * (FRAMES // GOP // 2) + 2, FRAMES - 1} - {p for p in range(0, FRAMES, GOP)})
    kept = {}

    def stream(rank, world):
        segs = shard.segments_for_rank(FRAMES, GOP, rank, world)
        order = [p for a, b in segs for p in range(a, b)]
        it = pool.imap(make, order, chunksize=1)
        ready, lists, crcs = {}, {}, {}
        ring = [lib.PinnedBuffer(geo[0]) for _ in range(3 * GOP)]
        n_frames = [0]
        ctx = lib.Context(0)
        for q in range(28, 40):
            ctx.set_qp(q)

        def want(pic):
            while pic not in ready:
                k, buf, jobs = next(it)
                ready[k] = [buf, jobs]
            return ready[pic]

        def frame(pic):
            item = want(pic)
            b = ring[n_frames[0] % len(ring)]  # its last use was three segments ago
            n_frames[0] += 1
            b.array[:] = item[0].reshape(-1)
            item[0] = None
            if item[1] is None:
                del ready[pic]
            return b

        def jobs(cur, seg, s):
            j = want(cur)[1]
            del ready[cur]
            j["fenc_plane"] += s * GOP
            j["ref_plane"] += s * GOP
            lists[cur] = DeviceBuffer.from_array(j)
            return lists[cur].ptr, len(j)

        def result(pic, res):
            crcs[pic] = zlib.crc32(res.tobytes())
            if pic in sample:
                kept[pic] = res.copy()
            lists.pop(pic).free()

        st = runner.SegmentStreamer(ctx, geo, GOP, n_out)
        t0 = time.time()
        st.run(segs, frame, jobs, on_result=result)
        dt = time.time() - t0
        st.close()
        for b in ring:
            b.free()
        ctx.close()
        print(f"rank {rank} of {world}: {len(segs)} segments, {len(order)} pictures, {len(crcs)} P pictures searched in {dt:.1f} s "
              f"(pictures and lists are made as they are asked for: this is not the runner's rate)", flush=True)
        return crcs

    t0 = time.time()
    whole = stream(0, 1)
    assert sorted(whole) == [p for p in range(FRAMES) if p % GOP], "every P picture once"
    parts = {}
    for r in range(WORLD):
        mine = stream(r, WORLD)
        assert not set(mine) & set(parts), "disjoint shards"
        parts.update(mine)
    pool.close()
    bad = [p for p in whole if parts.get(p) != whole[p]]
    print(f"{FRAMES} pictures {W}x{H}, GOP {GOP}: {len(whole)} P pictures of up to {n_out} motionEstimate calls (three references); one rank vs {WORLD} ranks: "
          f"{len(bad)} pictures differ, {len(set(whole) ^ set(parts))} missing", flush=True)
    # sampled pictures against the oracle
    O = pyoracle.load_oracle()
    tabs = {q: lib.costtab_build(q) for q in range(28, 40)}
    nbad = 0
    for cur in sample:
        seg = (cur // GOP * GOP, min(cur // GOP * GOP + GOP, FRAMES))
        refs = dict(shard.reference_pairs(seg, 3))[cur]
        planes = {p: synth.Plane(_seq.frame(p)).as_tuple() for p in [cur] + refs}
        pl = [planes.get(seg[0] + i, planes[cur]) for i in range(max(planes) - seg[0] + 1)]
        _, _, j = make(cur)
        wantr = O.me_batch(pl, tabs, j, nthreads=16)
        d = int(((wantr["cost"] != kept[cur]["cost"]) | (wantr["mvx"] != kept[cur]["mvx"]) | (wantr["mvy"] != kept[cur]["mvy"])).sum())
        nbad += d
        print(f"picture {cur} (references {refs}): {len(j)} calls against the oracle, {d} differ", flush=True)
    print(f"total {time.time() - t0:.0f} s; {'OK' if not bad and not nbad and set(whole) == set(parts) else 'FAILED'}", flush=True)
    sys.exit(0 if not bad and not nbad and set(whole) == set(parts) else 1)


if __name__ == "__main__":
    main()
