"""GPU test of x265_hip_cost_batch_device_noalloc, the entry point that allocates nothing: its costs must equal
x265_hip_cost_batch_device's (which the parity tests pin to the oracle) on two calls in a row, the second with a
smaller input, and sizes beyond the limit must be refused.  Collected last (see the file name)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _requests(rng, n, W, H):
    from x265_amd import abi
    reqs = np.zeros(n, abi.COST_REQ)
    parts = rng.integers(1, 25, n)
    wh = np.array(abi.PARTS)[parts]
    reqs["pu_w"], reqs["pu_h"] = wh[:, 0], wh[:, 1]
    reqs["pu_x"] = (rng.random(n) * (W - wh[:, 0])).astype(int)
    reqs["pu_y"] = (rng.random(n) * (H - wh[:, 1])).astype(int)
    reqs["fenc_plane"], reqs["ref_plane"] = 1, rng.choice([0, 2], n)
    reqs["metric"] = rng.integers(0, 2, n)
    reqs["qmv_x"], reqs["qmv_y"] = rng.integers(-60, 60, n), rng.integers(-60, 60, n)
    return reqs


def test_noalloc_entry_point_matches_cost_batch_device(gpu_ctx, small_seq, oracle):
    from x265_amd import lib
    from x265_amd.hipmem import DeviceBuffer
    ctx = gpu_ctx
    rng = np.random.default_rng(77)
    # a second context on the same GPU leaves its counts in the shared array between the two calls
    other = lib.Context(0)
    p = small_seq[0]
    other.plane_upload(0, p.buf, p.stride, p.origin)
    other.plane_upload(1, p.buf, p.stride, p.origin)
    for n in (40000, 1500, 33):  # the second and third calls are smaller than the first
        reqs = _requests(rng, n, 416, 240)
        d_reqs = DeviceBuffer.from_array(reqs)
        d_want, d_got = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
        d_want.fill(0xEE)
        d_got.fill(0xDD)
        ctx.cost_batch_device(d_reqs.ptr, n, d_want.ptr)
        ctx.cost_batch_device_noalloc(d_reqs.ptr, n, d_got.ptr)
        ctx.sync()
        got, want = d_got.download(np.int32), d_want.download(np.int32)
        assert np.array_equal(got, want)
        # and both equal the oracle
        assert np.array_equal(got, oracle.cost_batch([q.as_tuple() for q in small_seq], reqs, nthreads=8))
        o_reqs = reqs[: n // 2 + 1].copy()
        o_reqs["fenc_plane"], o_reqs["ref_plane"] = 1, 0
        od, oo = DeviceBuffer.from_array(o_reqs), DeviceBuffer(4 * len(o_reqs))
        other.cost_batch_device_noalloc(od.ptr, len(o_reqs), oo.ptr)
        other.sync()
    other.close()


def test_noalloc_calls_of_two_contexts_overlap(gpu_ctx, small_seq):
    """Two contexts on one GPU call the entry point back to back with no synchronisation in between: each has a slot of
    the array of its own, so neither disturbs the other's batch counters (with one shared set of counters a call of one
    context cleared the counts of the other's kernel in flight, and that kernel then skipped batches silently)."""
    from x265_amd import lib
    from x265_amd.hipmem import DeviceBuffer
    rng = np.random.default_rng(78)
    a, b = gpu_ctx, lib.Context(0)
    for i, p in enumerate(small_seq[:3]):
        b.plane_upload(i, p.buf, p.stride, p.origin)
    n = 60000
    ra, rb = _requests(rng, n, 416, 240), _requests(rng, n, 416, 240)
    da, db = DeviceBuffer.from_array(ra), DeviceBuffer.from_array(rb)
    wa, wb = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    a.cost_batch_device(da.ptr, n, wa.ptr)
    b.cost_batch_device(db.ptr, n, wb.ptr)
    a.sync(), b.sync()
    want_a, want_b = wa.download(np.int32), wb.download(np.int32)
    for _ in range(6):
        ga, gb = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
        ga.fill(0xDD), gb.fill(0xDD)
        a.sync(), b.sync()
        for _ in range(3):  # several launches in flight on both streams
            a.cost_batch_device_noalloc(da.ptr, n, ga.ptr)
            b.cost_batch_device_noalloc(db.ptr, n, gb.ptr)
        a.sync(), b.sync()
        assert np.array_equal(ga.download(np.int32), want_a)
        assert np.array_equal(gb.download(np.int32), want_b)
        ga.free(), gb.free()
    b.close()


def test_noalloc_entry_point_rejects_sizes_beyond_the_limit(gpu_ctx):
    from x265_amd import lib
    with pytest.raises(lib.X265HipError, match="exceed the limit"):
        gpu_ctx.cost_batch_device_noalloc(0x1000, (1 << 24) + 1, 0x1000)
    gpu_ctx.cost_batch_device_noalloc(0, 0, 0)  # nothing to do: fine
