"""GOP sharding of a sequence over ranks (one process per GPU, no data-path collective).

A closed GOP segment (IDR-started) references nothing outside itself, so segment s of a sequence is
handed to rank s mod world.  Each rank uploads only its own pictures; results are indexed by
(segment, picture, job) and never exchanged on the data path.  The only cross-rank traffic is the
barrier and the max-over-ranks reduction bench.py uses for timing.
"""


def gop_segments(n_frames, gop):
    """[(first, last_exclusive)] closed segments of at most `gop` pictures"""
    if n_frames < 0 or gop <= 0:
        raise ValueError("n_frames >= 0 and gop > 0 required")
    return [(s, min(s + gop, n_frames)) for s in range(0, n_frames, gop)]


def segments_for_rank(n_frames, gop, rank, world):
    """segments owned by `rank`: round-robin, so ranks differ by at most one segment"""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    return [seg for i, seg in enumerate(gop_segments(n_frames, gop)) if i % world == rank]


def reference_pairs(segment, n_refs):
    """(current, [references]) for the P pictures of a closed segment: picture i uses up to n_refs
    previous pictures of the same segment (the first picture is intra and has no ME)"""
    first, last = segment
    return [(cur, list(range(cur - 1, max(first, cur - n_refs) - 1, -1))) for cur in range(first + 1, last)]


def max_over_ranks(value, dist=None, device=None):
    """the job-wide time of a step bracket: max over ranks (identity when not distributed)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
