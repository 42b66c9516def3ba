"""GPU parity tests: the HIP path through the C ABI against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from conftest import random_bidir_reqs, random_lowres_jobs, random_me_jobs

pytestmark = pytest.mark.gpu


def _planes(small_seq):
    return [p.as_tuple() for p in small_seq]


def test_cost_batch_all_partitions_all_fractions(gpu_ctx, oracle, small_seq):
    from x265_amd import abi
    rng = np.random.default_rng(11)
    W, H = small_seq[0].w, small_seq[0].h
    reqs = []
    for part in range(25):
        w, h = abi.PARTS[part]
        for metric in (0, 1):
            for fx in range(4):
                for fy in range(4):
                    x = int(rng.integers(0, W - w + 1))
                    y = int(rng.integers(0, H - h + 1))
                    reqs.append((1, int(rng.choice([0, 2])), x, y, w, h, metric, 0,
                                 4 * int(rng.integers(-20, 20)) + fx, 4 * int(rng.integers(-20, 20)) + fy))
    reqs = np.array(reqs, dtype=abi.COST_REQ)
    got = gpu_ctx.cost_batch(reqs)
    want = oracle.cost_batch(_planes(small_seq), reqs)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,small_only", [(1, False), (15, False), (17, True), (1013, False), (1999, True)])
def test_cost_batch_ragged(gpu_ctx, oracle, small_seq, n, small_only):
    """request counts that do not fill the last wave, sizes mixed inside a wave (the prefix-sum path) and
    waves of small blocks only (the lane = 4 * request + tile path)"""
    from x265_amd import abi
    rng = np.random.default_rng(n)
    W, H = small_seq[0].w, small_seq[0].h
    parts = [p for p in range(25) if abi.PARTS[p][0] * abi.PARTS[p][1] <= 64] if small_only else list(range(25))
    reqs = np.zeros(n, abi.COST_REQ)
    for r in reqs:
        w, h = abi.PARTS[int(rng.choice(parts))]
        r["fenc_plane"], r["ref_plane"] = 1, int(rng.choice([0, 2]))
        r["pu_x"], r["pu_y"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        r["pu_w"], r["pu_h"], r["metric"] = w, h, int(rng.integers(0, 2))
        r["qmv_x"], r["qmv_y"] = int(rng.integers(-90, 90)), int(rng.integers(-90, 90))
    got = gpu_ctx.cost_batch(reqs)
    want = oracle.cost_batch(_planes(small_seq), reqs)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"request {bad[0]}: {reqs[bad[0]]} got {got[bad[0]]} want {want[bad[0]]}"


@pytest.mark.parametrize("sizes,n", [(((64, 64),), 4500), (((32, 32),), 9000), (((16, 16),), 9000), (((64, 64), (8, 8)), 9000)])
def test_cost_batch_lists_that_resize_their_batches(gpu_ctx, oracle, small_seq, sizes, n):
    """lists long enough for batches of several requests and made of large blocks: the kernel lowers the batch
    size from its sample of the list (cost_body: one size -> batches of one pass; clustered sizes, sorted here,
    -> by the mean), which must not change any result; cost_batch and dist_batch"""
    from x265_amd import abi
    rng = np.random.default_rng(n + len(sizes))
    W, H = small_seq[0].w, small_seq[0].h
    reqs = np.zeros(n, abi.COST_REQ)
    for i, r in enumerate(reqs):
        w, h = sizes[i * len(sizes) // n]
        r["fenc_plane"], r["ref_plane"] = 1, int(rng.choice([0, 2]))
        r["pu_x"], r["pu_y"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        r["pu_w"], r["pu_h"], r["metric"] = w, h, int(rng.integers(0, 2))
        r["qmv_x"], r["qmv_y"] = int(rng.integers(-90, 90)), int(rng.integers(-90, 90))
    want = oracle.cost_batch(_planes(small_seq), reqs)
    assert np.array_equal(gpu_ctx.cost_batch(reqs), want)
    reqs["metric"] = rng.integers(0, 4, n)
    assert np.array_equal(gpu_ctx.dist_batch(reqs), oracle.cost_batch(_planes(small_seq), reqs))


@pytest.mark.parametrize("method", [0, 1, 2, 3])
def test_me_batch_methods(gpu_ctx, oracle, small_seq, costtabs, method):
    from x265_amd import abi, synth
    rng = np.random.default_rng(100 + method)
    W, H = small_seq[0].w, small_seq[0].h
    for subme in range(8):
        jobs = random_me_jobs(rng, 120, method, subme, W, H, abi.ME_JOB, synth.search_range, abi.PARTS)
        got = gpu_ctx.me_batch(jobs)
        want = oracle.me_batch(_planes(small_seq), costtabs, jobs)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, f"method {method} subme {subme}: job {jobs[bad[0]]} got {got[bad[0]]} want {want[bad[0]]}"


def test_me_batch_full_search(gpu_ctx, oracle, small_seq, costtabs):
    from x265_amd import abi, synth
    rng = np.random.default_rng(5)
    W, H = small_seq[0].w, small_seq[0].h
    jobs = random_me_jobs(rng, 24, 4, 2, W, H, abi.ME_JOB, synth.search_range, abi.PARTS, meranges=(3, 6, 9))
    got = gpu_ctx.me_batch(jobs)
    want = oracle.me_batch(_planes(small_seq), costtabs, jobs)
    assert np.array_equal(got, want)


def test_fullsearch_ctu(gpu_ctx, oracle, small_seq, costtabs):
    from x265_amd import lib
    jobs = np.zeros(4, lib.CTU_JOB)
    cfg = [(64, 64, -16, -12, 15, 11, 3, -5, 30), (128, 64, -33, -20, 31, 22, -40, 17, 24),
           (0, 0, -8, -8, 8, 8, 0, 0, 38), (320, 128, -64, -64, 64, 64, 9, 2, 33)]
    for j, c in zip(jobs, cfg):
        j["fenc_plane"], j["ref_plane"] = 1, 0
        (j["ctu_x"], j["ctu_y"], j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"], j["qmvp_x"], j["qmvp_y"],
         j["qp"]) = c
    got = gpu_ctx.fullsearch(jobs)
    pl = _planes(small_seq)
    for k, c in enumerate(cfg):
        want = oracle.ctu_fullsearch(pl[1], pl[0], costtabs[c[8]], c[0], c[1], (c[2], c[3]), (c[4], c[5]), (c[6], c[7]))
        bad = np.nonzero(got[k] != want)[0]
        assert len(bad) == 0, f"ctu {k}: pu {bad[0]} got {got[k][bad[0]]} want {want[bad[0]]}"


def test_fullsearch_window_shapes(gpu_ctx, oracle, small_seq, costtabs):
    """window widths around the multiples of four (the extra-column pass), narrow and tall windows,
    the largest window the kernel takes (132 x 132), and a flat reference where every position ties on SAD"""
    from x265_amd import lib, synth
    shapes = [(nx, ny) for nx in (1, 2, 3, 4, 5, 6, 7, 8, 9, 127, 128, 129, 130, 131, 132) for ny in (1, 2, 31, 32, 33, 129)]
    shapes += [(129, ny) for ny in (63, 64, 65, 96, 97, 128, 130, 131, 132)] + [(132, 132)]
    rng = np.random.default_rng(77)
    flat = synth.Plane(np.full((240, 416), 93, np.uint8))
    gpu_ctx.plane_upload(5, flat.buf, flat.stride, flat.origin)
    pl = _planes(small_seq) + [None, None, flat.as_tuple()]
    jobs = np.zeros(len(shapes) + 6, lib.CTU_JOB)
    for k, j in enumerate(jobs):
        nx, ny = shapes[k] if k < len(shapes) else (129, 129)
        j["fenc_plane"], j["ref_plane"] = 1, (0 if k < len(shapes) else 5)
        j["ctu_x"], j["ctu_y"] = 64 * int(rng.integers(1, 5)), 64 * int(rng.integers(1, 2))
        lox, loy = -int(rng.integers(0, nx)) if nx < 66 else -(nx // 2), -int(rng.integers(0, ny)) if ny < 66 else -(ny // 2)
        j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"] = lox, loy, lox + nx - 1, loy + ny - 1
        j["qmvp_x"], j["qmvp_y"] = int(rng.integers(-60, 60)), int(rng.integers(-60, 60))
        j["qp"] = int(rng.integers(20, 45))
    got = gpu_ctx.fullsearch(jobs)
    for k, j in enumerate(jobs):
        want = oracle.ctu_fullsearch(pl[1], pl[int(j["ref_plane"])], costtabs[int(j["qp"])], int(j["ctu_x"]), int(j["ctu_y"]),
                                     (int(j["mvmin_x"]), int(j["mvmin_y"])), (int(j["mvmax_x"]), int(j["mvmax_y"])),
                                     (int(j["qmvp_x"]), int(j["qmvp_y"])))
        bad = np.nonzero(got[k] != want)[0]
        assert len(bad) == 0, f"ctu {k} {j}: pu {bad[0]} got {got[k][bad[0]]} want {want[bad[0]]}"
    gpu_ctx.plane_release(5)


def test_primitive_table(oracle):
    """the function-pointer table (host pointers), inputs shaped like the reference's TestBench:
    random / all-0 / all-255 buffers, fref stride 64 and FENC_STRIDE-5, offsets stepping by 32
    (test/pixelharness.cpp:152-171, 215-272); filters with random strides (ipfilterharness.cpp:344-488)"""
    import ctypes as C
    from x265_amd import lib
    t = lib.setup_primitives(0)
    rng = np.random.default_rng(2)
    size = 64 * 128 + 32 * 100
    bufs = [rng.integers(0, 256, size, dtype=np.uint8), np.zeros(size, np.uint8), np.full(size, 255, np.uint8)]

    def addr(a, off=0):
        return C.c_void_p(a.ctypes.data + off)

    for part in range(25):
        for it in range(6):
            a, b = bufs[it % 3], bufs[(it // 3 + it) % 3] if it else bufs[0]
            ro = 32 * it
            for rs in (64, 59):
                assert t.sad[part](addr(a), 64, addr(b, ro), rs) == oracle.sad(part, a, 0, 64, b, ro, rs)
                assert t.satd[part](addr(a), 64, addr(b, ro), rs) == oracle.satd(part, a, 0, 64, b, ro, rs)
            res = np.full(4, -1, np.int32)
            t.sad_x3[part](addr(a), addr(b, ro), addr(b, ro + 1), addr(b, ro + 2), 59, addr(res))
            assert np.array_equal(res[:3], oracle.sad_x3(part, a, 0, b, [ro, ro + 1, ro + 2], 59)) and res[3] == -1
            t.sad_x4[part](addr(a), addr(b, ro), addr(b, ro + 1), addr(b, ro + 2), addr(b, ro + 3), 59, addr(res))
            assert np.array_equal(res, oracle.sad_x4(part, a, 0, b, [ro, ro + 1, ro + 2, ro + 3], 59))

    src = rng.integers(0, 256, 200 * 200, dtype=np.uint8)
    s16 = rng.integers(-12000, 12000, 200 * 200).astype(np.int16)
    for part in range(25):
        for coeff in range(4):
            ss, ds = int(rng.integers(72, 100)), int(rng.integers(64, 164))
            so = 6 * ss + 6
            for name in ("luma_hpp", "luma_vpp"):
                d1 = np.full(200 * 200, 0x5A, np.uint8)
                d2 = d1.copy()
                getattr(t, name)[part](addr(src, so), ss, addr(d1), ds, coeff)
                getattr(oracle, name)(part, src, so, ss, d2, 0, ds, coeff)
                assert np.array_equal(d1, d2), (name, part, coeff)  # whole buffer: nothing outside w x h written
            for ext in (0, 1):
                d1 = np.full(200 * 200, 0x5A5A, np.int16)
                d2 = d1.copy()
                t.luma_hps[part](addr(src, so), ss, addr(d1), ds, coeff, ext)
                oracle.luma_hps(part, src, so, ss, d2, 0, ds, coeff, ext)
                assert np.array_equal(d1, d2), ("hps", part, coeff, ext)
            d1 = np.full(200 * 200, 0x5A, np.uint8)
            d2 = d1.copy()
            t.luma_vsp[part](addr(s16, 2 * so), ss, addr(d1), ds, coeff)
            oracle.luma_vsp(part, s16, so, ss, d2, 0, ds, coeff)
            assert np.array_equal(d1, d2), ("vsp", part, coeff)


def test_lowres_planes_and_search(oracle, small_seq, costtabs):
    """x265_hip_lowres_init against the oracle's frame_init_lowres_core + extendPicBorder (every byte of the
    four padded planes), then motionEstimate against those planes (ref->isLowres), all methods and sub-pel levels"""
    from x265_amd import abi, lib
    ctx = lib.Context(0)
    low = []
    for i, p in enumerate(small_seq):
        ctx.plane_upload(20 + i, p.buf, p.stride, p.origin)
        geo = ctx.lowres_init(20 + i, p.w, p.h, 4 * i)
        want, size = oracle.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)
        assert (geo.width, geo.lines, geo.stride, geo.origin_offset) == (size[0], size[1], want[0][2], want[0][1])
        for k in range(4):
            got = ctx.plane_download(4 * i + k, geo.bytes)
            bad = np.nonzero(got != want[k][0])[0]
            assert len(bad) == 0, f"picture {i} plane {k}: first difference at byte {bad[0]} (row {bad[0] // geo.stride}, col {bad[0] % geo.stride})"
        low += want
    for q in (12, 20, 33):
        ctx.costtab_upload(q, costtabs.get(q, oracle.bitcost_table(q)))
    rng = np.random.default_rng(6)
    jobs = random_lowres_jobs(rng, 2000, size[0], size[1], abi.ME_JOB)
    got = ctx.me_batch(jobs)
    want = oracle.me_batch(low, {q: oracle.bitcost_table(q) for q in (12, 20, 33)}, jobs)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"job {jobs[bad[0]]} got {got[bad[0]]} want {want[bad[0]]}"
    # a lowres job must be an 8x8 block
    j = jobs[:1].copy()
    j["pu_w"] = 16
    with pytest.raises(lib.X265HipError):
        ctx.me_batch(j)
    ctx.close()


def test_lookahead_whole_frame_search(oracle, small_seq, costtabs):
    """x265_hip_lookahead_batch against the oracle's restatement of estimateCUCost's search loop: every ordered pair of
    the three small pictures (lowres planes made on the device), the reference's settings and a few others, in one
    launch and one search at a time; every CU's (cost, mv) equal"""
    from x265_amd import abi, lib
    ctx = lib.Context(0)
    low = []
    for i, p in enumerate(small_seq):
        ctx.plane_upload(20 + i, p.buf, p.stride, p.origin)
        geo = ctx.lowres_init(20 + i, p.w, p.h, 4 * i)
        low += oracle.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)[0]
    for q in (12, 33):
        ctx.costtab_upload(q, oracle.bitcost_table(q))
    W, H = geo.width // 8, geo.lines // 8
    jobs = []
    for a in range(3):
        for b in range(3):
            if a != b:
                jobs.append((4 * a, 4 * b, 16, abi.HEX, 1, 12))
    jobs += [(0, 4, 16, abi.DIA, 0, 12), (4, 8, 8, abi.UMH, 1, 33), (8, 0, 32, abi.STAR, 2, 12), (0, 8, 16, abi.HEX, 5, 33),
             (0, 0, 16, abi.HEX, 1, 12)]
    lj = np.zeros(len(jobs), abi.LOOKAHEAD_JOB)
    for k, (f, r, merange, method, subme, qp) in enumerate(jobs):
        lj[k]["fenc_plane"], lj[k]["ref_plane"], lj[k]["merange"] = f, r, merange
        lj[k]["method"], lj[k]["subme"], lj[k]["qp"] = method, subme, qp
    want = oracle.lookahead_batch(low, {q: oracle.bitcost_table(q) for q in (12, 33)}, lj, W, H, nthreads=8)
    assert len(np.unique(want["mvx"])) > 20 and (want["cost"] > 0).any()
    got = ctx.lookahead_batch(lj, W, H)
    bad = np.argwhere(got != want.view(got.dtype))
    assert len(bad) == 0, (len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    for k in (0, 7, len(lj) - 1):
        one = ctx.lookahead_batch(lj[k:k + 1], W, H)
        assert np.array_equal(one[0], got[k])
    # a picture narrower than the planes: only the CUs asked for are searched, with bounds of that size
    small = ctx.lookahead_batch(lj[:2], 5, 3)
    assert np.array_equal(small, oracle.lookahead_batch(low, {12: oracle.bitcost_table(12)}, lj[:2], 5, 3).view(small.dtype))
    one_cu = ctx.lookahead_batch(lj[:1], 1, 1)
    assert np.array_equal(one_cu, oracle.lookahead_batch(low, {12: oracle.bitcost_table(12)}, lj[:1], 1, 1).view(one_cu.dtype))
    assert ctx.lookahead_batch(lj[:0], W, H).shape == (0, H, W)
    # the device-array form, whose caller reads the launch's status word through x265_hip_lookahead_status
    from x265_amd.hipmem import DeviceBuffer
    d_lj, d_res = DeviceBuffer.from_array(lj), DeviceBuffer(8 * len(lj) * W * H)
    ctx.lookahead_batch_device(d_lj.ptr, len(lj), W, H, d_res.ptr)
    assert ctx.lookahead_status() == 0
    assert np.array_equal(d_res.download(abi.ME_RESULT).reshape(len(lj), H, W), got)
    # refused: a picture larger than the planes, a reference without its three half-pel planes, no cost table
    with pytest.raises(lib.X265HipError, match="outside"):
        ctx.lookahead_batch(lj[:1], W + 30, H)
    bad_job = lj[:1].copy()
    bad_job["ref_plane"] = 20
    with pytest.raises(lib.X265HipError):
        ctx.lookahead_batch(bad_job, W, H)
    bad_job = lj[:1].copy()
    bad_job["qp"] = 40
    with pytest.raises(lib.X265HipError, match="cost table"):
        ctx.lookahead_batch(bad_job, W, H)
    ctx.close()


def test_plane_weight_and_border(oracle, small_seq):
    """x265_hip_plane_weight against weight_pp + extendPicBorder, and x265_hip_plane_extend_border in place on a
    plane whose margins were wiped"""
    from x265_amd import lib
    ctx = lib.Context(0)
    p = small_seq[0]
    ctx.plane_upload(0, p.buf, p.stride, p.origin)
    rows = np.arange(p.rows)[:, None]
    cols = np.arange(p.stride)[None, :]
    # what the reference writes in every row: columns -marginX .. width + marginX - 1 (bytes beyond, up to the
    # stride, are copies of unwritten memory there)
    written = ((cols < p.w + 192) & (rows < p.h + 160)).reshape(-1)
    for (w0, off, den) in ((64, 0, 6), (70, -3, 6), (37, 10, 5), (1, 0, 0), (200, -100, 7)):
        ctx.plane_weight(0, 1, p.w, p.h, w0, off, den)
        got = ctx.plane_download(1, p.buf.size)
        want = oracle.weighted_plane(p.buf, p.origin, p.stride, p.w, p.h, w0, off, den)
        bad = np.nonzero((got != want) & written)[0]
        assert len(bad) == 0, f"weight {(w0, off, den)}: byte {bad[0]} (row {bad[0] // p.stride}, col {bad[0] % p.stride}) got {got[bad[0]]} want {want[bad[0]]}"
    wiped = p.buf.reshape(p.rows, p.stride).copy()
    inside = (rows >= 80) & (rows < 80 + p.h) & (cols >= 96) & (cols < 96 + p.w)
    wiped[~inside] = 0x55
    ctx.plane_upload(2, wiped.reshape(-1), p.stride, p.origin)
    ctx.plane_extend_border(2, p.w, p.h)
    got = ctx.plane_download(2, p.buf.size)
    bad = np.nonzero((got != p.buf) & written)[0]
    assert len(bad) == 0, f"border: byte {bad[0]} (row {bad[0] // p.stride}, col {bad[0] % p.stride})"
    with pytest.raises(lib.X265HipError):
        ctx.plane_extend_border(2, p.w + 2, p.h)  # width must be a multiple of 4
    with pytest.raises(lib.X265HipError):
        ctx.plane_extend_border(2, p.w, p.h, margin_y=200)  # does not fit the buffer
    ctx.close()


def test_plane_rows_as_they_become_available(oracle, small_seq):
    """the row-range forms (MotionReference::applyWeight(rows, numRows), reference.cpp:78-121; the frame filter's
    extendRowBorder, framefilter.cpp:221): CTU row by CTU row they must build the plane the whole-picture forms build,
    and after every step the rows produced so far must already be final"""
    from x265_amd import lib
    ctx = lib.Context(0)
    p = small_seq[1]
    ctx.plane_upload(0, p.buf, p.stride, p.origin)
    rows = np.arange(p.rows)[:, None]
    cols = np.arange(p.stride)[None, :]
    written = ((cols < p.w + 192) & (rows < p.h + 160))
    ctu_rows = [(r, min(64, p.h - r)) for r in range(0, p.h, 64)]  # 64, 64, 64, 48
    for (w0, off, den) in ((70, -3, 6), (37, 10, 5)):
        want = oracle.weighted_plane(p.buf, p.origin, p.stride, p.w, p.h, w0, off, den).reshape(p.rows, p.stride)
        ctx.plane_weight(0, 1, p.w, p.h, 1, 0, 0)  # something else in the destination first
        for r0, n in ctu_rows:
            ctx.plane_weight_rows(0, 1, p.w, p.h, w0, off, den, r0, n)
            got = ctx.plane_download(1, p.buf.size).reshape(p.rows, p.stride)
            done = written & (rows < 80 + r0 + n) if r0 + n < p.h else written
            assert np.array_equal(got[done], want[done]), (w0, r0)
    wiped = p.buf.reshape(p.rows, p.stride).copy()
    inside = (rows >= 80) & (rows < 80 + p.h) & (cols >= 96) & (cols < 96 + p.w)
    wiped[~inside] = 0x55
    ctx.plane_upload(2, wiped.reshape(-1), p.stride, p.origin)
    for r0, n in ctu_rows:
        ctx.plane_extend_border_rows(2, p.w, p.h, r0, n)
        got = ctx.plane_download(2, p.buf.size).reshape(p.rows, p.stride)
        done = written & (rows < 80 + r0 + n) if r0 + n < p.h else written
        assert np.array_equal(got[done], p.buf.reshape(p.rows, p.stride)[done]), r0
        if r0 + n < p.h:  # rows not yet available are untouched
            assert (got[80 + r0 + n:, :96] == 0x55).all()
    with pytest.raises(lib.X265HipError):
        ctx.plane_extend_border_rows(2, p.w, p.h, 200, 64)  # past the last row
    ctx.close()


@pytest.mark.parametrize("n", [1, 33, 2500])
def test_bidir_cost_batch(gpu_ctx, oracle, small_seq, n):
    """x265_hip_bidir_cost_batch: block vs the rounded average of two interpolated predictions"""
    from x265_amd import abi
    rng = np.random.default_rng(200 + n)
    reqs = random_bidir_reqs(rng, n, small_seq[0].w, small_seq[0].h, abi.BIDIR_REQ, abi.PARTS)
    got = gpu_ctx.bidir_cost_batch(reqs)
    want = oracle.bidir_cost_batch(_planes(small_seq), reqs)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"request {bad[0]}: {reqs[bad[0]]} got {got[bad[0]]} want {want[bad[0]]}"


def test_primitive_table_plane_entries(oracle):
    """pixelavg_pp[25], weight_pp and frame_init_lowres_core of the function-pointer table (host pointers, odd
    strides and offsets, widths that are not multiples of four): whole destination buffers compared, so nothing
    outside the block may be written"""
    import ctypes as C
    from x265_amd import lib
    t = lib.setup_primitives(0)
    rng = np.random.default_rng(22)
    a, b = rng.integers(0, 256, 200 * 200, dtype=np.uint8), rng.integers(0, 256, 200 * 200, dtype=np.uint8)

    def addr(x, off=0):
        return C.c_void_p(x.ctypes.data + off)

    for part in range(25):
        d1, d2 = np.full(200 * 200, 0x5A, np.uint8), np.full(200 * 200, 0x5A, np.uint8)
        t.pixelavg_pp[part](addr(d1, 3), 71, addr(a, 5), 83, addr(b, 9), 64, 32)
        oracle.pixelavg_pp(part, d2, 3, 71, a, 5, 83, b, 9, 64)
        assert np.array_equal(d1, d2), part
    for (w, h, w0, rnd, sh, off) in ((64, 48, 70, 2048, 12, -3), (16, 16, 37, 1024, 11, 10), (128, 33, 1, 0, 6, 0), (30, 5, 90, 32, 6, 4)):
        d1, d2 = np.full(200 * 200, 0x5A, np.uint8), np.full(200 * 200, 0x5A, np.uint8)
        t.weight_pp(addr(a, 7), addr(d1, 11), 150, 160, w, h, w0, rnd, sh, off)
        oracle.weight_pp(a, 7, d2, 11, 150, 160, w, h, w0, rnd, sh, off)
        assert np.array_equal(d1, d2), (w, h)
    # extendRowBorder (primitives.h:259): a CTU row of luma and of chroma with the encoder's margins, and odd shapes
    for (w, h, mx, stride, off) in ((416, 64, 80, 576, 80 + 3 * 576), (208, 32, 40, 288, 40), (33, 5, 7, 61, 9), (4, 1, 1, 6, 1), (90, 17, 96, 300, 400)):
        d1 = rng.integers(0, 256, 200 * 200, dtype=np.uint8)
        d2 = d1.copy()
        t.extendRowBorder(addr(d1, off), stride, w, h, mx)
        oracle.extend_row_border(d2, off, stride, w, h, mx)
        assert np.array_equal(d1, d2), (w, h, mx)
    for (w, h) in ((48, 40), (17, 9), (80, 24), (4, 4)):
        o1 = [np.full(200 * 200, 0x5A, np.uint8) for _ in range(4)]
        o2 = [np.full(200 * 200, 0x5A, np.uint8) for _ in range(4)]
        t.frame_init_lowres_core(addr(a, 3), addr(o1[0], 5), addr(o1[1], 5), addr(o1[2], 5), addr(o1[3], 5), 190, 97, w, h)
        oracle.frame_init_lowres(a, 3, o2, 5, 190, 97, w, h)
        for k, (x, y) in enumerate(zip(o1, o2)):
            assert np.array_equal(x, y), (w, h, k)


def test_dist_batch_all_metrics(gpu_ctx, oracle, small_seq):
    """x265_hip_dist_batch: sad / satd / sa8d_inter / sse_pp of predictions, every partition and fraction; lists
    of one metric, mixed lists (sa8d blocks beside single-strip blocks inside a batch), and ragged lengths"""
    from conftest import random_cost_reqs
    from x265_amd import abi
    planes = [p.as_tuple() for p in small_seq]
    W, H = small_seq[0].w, small_seq[0].h
    rng = np.random.default_rng(61)
    for metrics, n in (((2,), 3000), ((3,), 1500), ((0, 1, 2, 3), 4000), ((2, 3), 33), ((2,), 1)):
        reqs = random_cost_reqs(rng, n, W, H, abi.COST_REQ, oracle_parts(), metrics=metrics)
        got, want = gpu_ctx.dist_batch(reqs), oracle.cost_batch(planes, reqs)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (metrics, reqs[bad[0]], got[bad[0]], want[bad[0]])
    # a list of 8x8 blocks only (two strips each: the batch layout without the item search), saturated differences
    reqs = random_cost_reqs(rng, 500, W, H, abi.COST_REQ, oracle_parts(), metrics=(2,))
    reqs["pu_w"] = reqs["pu_h"] = 8
    reqs["pu_x"] = np.minimum(reqs["pu_x"], W - 8)
    reqs["pu_y"] = np.minimum(reqs["pu_y"], H - 8)
    assert np.array_equal(gpu_ctx.dist_batch(reqs), oracle.cost_batch(planes, reqs))


def oracle_parts():
    from oracle import pyoracle
    return pyoracle.PARTS


def test_primitive_table_distortions(oracle):
    """sa8d_inter[25] and sse_pp[25] of the function-pointer table: random, flat and checkerboard content"""
    import ctypes as C
    from x265_amd import lib
    t = lib.setup_primitives(0)
    rng = np.random.default_rng(23)
    n = 64 * 90
    yy, xx = np.mgrid[0:90, 0:64]
    bufs = [rng.integers(0, 256, n, dtype=np.uint8), np.zeros(n, np.uint8), np.full(n, 255, np.uint8),
            (((xx ^ yy) & 1) * 255).astype(np.uint8).reshape(-1), (((xx >> 2) & 1) * 255).astype(np.uint8).reshape(-1)]
    for part in range(25):
        for _ in range(6):
            a, b = bufs[rng.integers(5)], bufs[rng.integers(5)]
            fo, ro = int(rng.integers(0, 16)) * 16, int(rng.integers(0, 1000))
            for rs in (64, 71):
                pa, pb = C.c_void_p(a.ctypes.data + fo), C.c_void_p(b.ctypes.data + ro)
                assert t.sa8d_inter[part](pa, 64, pb, rs) == oracle.sa8d_inter(part, a, fo, 64, b, ro, rs), part
                assert t.sse_pp[part](pa, 64, pb, rs) == oracle.sse_pp(part, a, fo, 64, b, ro, rs), part


def test_predict_batch(gpu_ctx, oracle, small_seq):
    """x265_hip_predict_batch against the oracle: whole destination buffers (nothing outside the blocks may change),
    every partition, uni- and bi-directional, weighted uni-directional and (every other batch) weighted bi-directional,
    all fraction pairs"""
    from conftest import random_pred_reqs
    from x265_amd import abi, lib
    rng = np.random.default_rng(62)
    p0 = small_seq[0]
    planes = [p.as_tuple() for p in small_seq]
    want = np.full_like(p0.buf, 0x5A)
    gpu_ctx.plane_upload(5, want, p0.stride, p0.origin)
    n_wbi = 0
    for it in range(40):
        reqs = random_pred_reqs(rng, p0.w, p0.h, abi.PRED_REQ, oracle_parts(), dst_plane=5, weighted_bi=bool(it & 1))
        n_wbi += int(reqs["weighted_bi"].sum())
        gpu_ctx.predict_batch(reqs)
        reqs_o = reqs.copy()
        reqs_o["dst_plane"] = 3
        oracle.predict_batch(planes + [(want, p0.origin, p0.stride)], reqs_o)
        got = gpu_ctx.plane_download(5, want.size)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (len(bad), divmod(int(bad[0]) - p0.origin, p0.stride))
    assert n_wbi > 50
    # weights outside what the slice header can carry, and a weighted pair with one reference, are refused
    bi = reqs[(reqs["ref1_plane"] >= 0)][:1].copy()
    bi["weighted_bi"], bi["wp_w0"], bi["wp_w1"], bi["log2_denom"] = 1, 64, 64, 6
    gpu_ctx.predict_batch(bi)
    for k, v in (("wp_w0", 256), ("wp_w1", -129), ("wp_o0", 128), ("wp_o1", -129), ("log2_denom", 8), ("ref1_plane", -1)):
        q = bi.copy()
        q[k] = v
        with pytest.raises(lib.X265HipError):
            gpu_ctx.predict_batch(q)
    gpu_ctx.plane_release(5)


def test_mc_satd_and_merge_estimation(gpu_ctx, oracle, small_seq):
    """x265_hip_mc_satd_batch (satd of the source block against motionCompensation's prediction, every form of the
    request) and x265_hip_merge_estimate_batch (xMergeEstimation) against the oracle, host-array and device forms"""
    from conftest import random_merge_pus, random_pred_reqs
    from x265_amd import abi, lib
    from x265_amd.hipmem import DeviceBuffer
    rng = np.random.default_rng(77)
    p0 = small_seq[0]
    planes = [p.as_tuple() for p in small_seq]
    for it in range(6):
        reqs = random_pred_reqs(rng, p0.w, p0.h, abi.PRED_REQ, oracle_parts(), dst_plane=1, refs=(0, 2), weighted_bi=True)
        assert np.array_equal(gpu_ctx.mc_satd_batch(reqs), oracle.mc_satd_batch(planes, reqs))
    pus = random_merge_pus(rng, 3000, p0.w, p0.h, abi.MERGE_PU, abi.PARTS)
    want = oracle.merge_estimate_batch(planes, pus, p0.w, p0.h)
    assert (want["cost"] == 0xFFFFFFFF).sum() > 5 and (want["satd"] < 0).sum() > 1000 and len(np.unique(want["index"])) == 5
    got = gpu_ctx.merge_estimate_batch(pus, p0.w, p0.h)
    assert np.array_equal(got, want)
    d_pus, d_out = DeviceBuffer.from_array(pus), DeviceBuffer(abi.MERGE_RESULT.itemsize * len(pus))
    gpu_ctx.merge_estimate_batch_device(d_pus.ptr, len(pus), p0.w, p0.h, d_out.ptr)
    gpu_ctx.sync()
    assert np.array_equal(d_out.download(abi.MERGE_RESULT), want)
    d_pus.free()
    d_out.free()
    # refused: no candidates, a candidate without a reference, a plane that was not uploaded
    for k, v in (("num_cand", 0), ("num_cand", 6), ("weighted", 3), ("fenc_plane", 300)):
        q = pus[:1].copy()
        q[k] = v
        with pytest.raises(lib.X265HipError):
            gpu_ctx.merge_estimate_batch(q, p0.w, p0.h)
    q = pus[:1].copy()
    q["cand"]["ref_plane"][0, 0] = [-1, -1]
    with pytest.raises(lib.X265HipError):
        gpu_ctx.merge_estimate_batch(q, p0.w, p0.h)
    q = pus[:1].copy()
    q["cand"]["ref_plane"][0, 0] = [300, -1]
    with pytest.raises(lib.X265HipError):
        gpu_ctx.merge_estimate_batch(q, p0.w, p0.h)
    assert len(gpu_ctx.merge_estimate_batch(pus[:0], p0.w, p0.h)) == 0
    # record-check mode 1 on the device forms: the records are copied back and refused like the host arrays
    gpu_ctx.check_records(1)
    try:
        q = pus[:4].copy()
        q["cand"]["ref_plane"][2, 0] = [300, -1]
        d_pus, d_out = DeviceBuffer.from_array(q), DeviceBuffer(abi.MERGE_RESULT.itemsize * 4)
        with pytest.raises(lib.X265HipError, match=r"merge_pu\[2\] candidate 0"):
            gpu_ctx.merge_estimate_batch_device(d_pus.ptr, 4, p0.w, p0.h, d_out.ptr)
        r = reqs[:4].copy()
        r["qmv0_y"][1] = 4 * (p0.h + 400)
        d_r = DeviceBuffer.from_array(r)
        with pytest.raises(lib.X265HipError, match=r"pred_req\[1\]"):
            gpu_ctx.mc_satd_batch_device(d_r.ptr, 4, d_out.ptr)
        d_good = DeviceBuffer.from_array(reqs[:4])
        gpu_ctx.mc_satd_batch_device(d_good.ptr, 4, d_out.ptr)
        gpu_ctx.sync()
        assert np.array_equal(d_out.download(np.int32)[:4], oracle.mc_satd_batch(planes, reqs[:4]))
    finally:
        gpu_ctx.check_records(False)


def test_primitive_table_mc_entries(oracle):
    """luma_vps / luma_vss / luma_hvpp / luma_p2s / luma_addAvg of the function-pointer table: host pointers, odd
    strides; whole destination buffers compared"""
    import ctypes as C
    from x265_amd import lib
    t = lib.setup_primitives(0)
    rng = np.random.default_rng(46)
    src = rng.integers(0, 256, 200 * 200, dtype=np.uint8)
    s16a = rng.integers(-12000, 14000, 200 * 200).astype(np.int16)
    s16b = rng.integers(-12000, 14000, 200 * 200).astype(np.int16)

    def addr(x, off=0):
        return C.c_void_p(x.ctypes.data + off * x.itemsize)

    for (w, h) in ((4, 4), (16, 12), (64, 64), (24, 32)):
        d1, d2 = np.full(80 * 80, 77, np.int16), np.full(80 * 80, 77, np.int16)
        t.luma_p2s(addr(src, 811), 150, addr(d1, 5), w, h)
        oracle.luma_p2s(src, 811, 150, d2, 5, w, h)
        assert np.array_equal(d1, d2), (w, h)
    for part in range(25):
        coeff = part % 4
        d1, d2 = np.full(200 * 200, 77, np.int16), np.full(200 * 200, 77, np.int16)
        t.luma_vps[part](addr(src, 1807), 150, addr(d1, 9), 71, coeff)
        oracle.luma_vps(part, src, 1807, 150, d2, 9, 71, coeff)
        assert np.array_equal(d1, d2), part
        t.luma_vss[part](addr(s16a, 1807), 150, addr(d1, 9), 71, coeff)
        oracle.luma_vss(part, s16a, 1807, 150, d2, 9, 71, coeff)
        assert np.array_equal(d1, d2), part
        p1, p2 = np.full(200 * 200, 0x5A, np.uint8), np.full(200 * 200, 0x5A, np.uint8)
        t.luma_hvpp[part](addr(src, 1807), 150, addr(p1, 9), 71, 1 + coeff % 3, 3 - coeff % 3)
        oracle.luma_hvpp(part, src, 1807, 150, p2, 9, 71, 1 + coeff % 3, 3 - coeff % 3)
        assert np.array_equal(p1, p2), part
        t.luma_addAvg[part](addr(s16a, 3), addr(s16b, 11), addr(p1, 9), 64, 70, 81)
        oracle.luma_addavg(part, s16a, 3, s16b, 11, p2, 9, 64, 70, 81)
        assert np.array_equal(p1, p2), part


def test_lowres_bidir_cost(oracle, small_seq):
    """the lookahead's bidirectional cost: lowres planes made on the device, random 8x8 CUs and vector pairs, sad and
    satd, a ragged count; mixed into one list with full-resolution bidirectional requests"""
    from conftest import random_bidir_reqs, random_lowres_bidir_reqs
    from x265_amd import abi, lib
    ctx = lib.Context(0)
    low = []
    for i, p in enumerate(small_seq):
        ctx.plane_upload(20 + i, p.buf, p.stride, p.origin)
        geo = ctx.lowres_init(20 + i, p.w, p.h, 4 * i)
        four, _ = oracle.lowres_planes(p.buf, p.origin, p.stride, p.w, p.h)
        low += four
    full = [p.as_tuple() for p in small_seq]
    rng = np.random.default_rng(63)
    for metric, n in ((1, 2000), (0, 777)):
        br = random_lowres_bidir_reqs(rng, n, geo.width, geo.lines, abi.BIDIR_REQ, metric=metric)
        got, want = ctx.bidir_cost_batch(br), oracle.bidir_cost_batch(low, br)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (br[bad[0]], got[bad[0]], want[bad[0]])
    # one list holding both kinds: planes 0..11 lowres, 20..22 full resolution
    a = random_lowres_bidir_reqs(rng, 300, geo.width, geo.lines, abi.BIDIR_REQ)
    b = random_bidir_reqs(rng, 300, small_seq[0].w, small_seq[0].h, abi.BIDIR_REQ, oracle_parts())
    want = np.concatenate([oracle.bidir_cost_batch(low, a), oracle.bidir_cost_batch(full, b)])
    for k in ("fenc_plane", "ref0_plane", "ref1_plane"):
        b[k] += 20
    mixed = np.concatenate([a, b])
    perm = rng.permutation(len(mixed))
    assert np.array_equal(ctx.bidir_cost_batch(mixed[perm]), want[perm])
    ctx.close()


def test_bidir_cost_mc_average_and_all_metrics(gpu_ctx, oracle, small_seq):
    """bidirectional requests: pixelavg_pp or the motion-compensation average, sad / satd / sa8d / sse, every partition
    and fraction pair, mixed in one list; a ragged count"""
    from conftest import random_bidir_reqs
    from x265_amd import abi
    rng = np.random.default_rng(64)
    planes = [p.as_tuple() for p in small_seq]
    for n in (5000, 37):
        reqs = random_bidir_reqs(rng, n, small_seq[0].w, small_seq[0].h, abi.BIDIR_REQ, oracle_parts(), metrics=(0, 1, 2, 3),
                                 flags=(0, 0x40))
        got, want = gpu_ctx.bidir_cost_batch(reqs), oracle.bidir_cost_batch(planes, reqs)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (reqs[bad[0]], got[bad[0]], want[bad[0]])


def test_concurrent_contexts_and_table_calls(oracle, small_seq, costtabs):
    """threading as the reference uses the path: several host threads, each with its own context,
    search at the same time while other threads call primitive-table entries (which share the default context);
    every result equals the oracle's"""
    import ctypes as C
    import threading
    from x265_amd import abi, lib, synth
    W, H = small_seq[0].w, small_seq[0].h
    planes = _planes(small_seq)
    table = lib.setup_primitives(0)
    errors = []

    def searcher(seed):
        try:
            rng = np.random.default_rng(seed)
            ctx = lib.Context(0)
            for i, p in enumerate(small_seq):
                ctx.plane_upload(i, p.buf, p.stride, p.origin)
            for q, t in costtabs.items():
                ctx.costtab_upload(q, t)
            for rnd in range(4):
                jobs = random_me_jobs(rng, 400, int(rng.integers(0, 4)), int(rng.integers(0, 8)), W, H, abi.ME_JOB,
                                      synth.search_range, abi.PARTS)
                got, want = ctx.me_batch(jobs), oracle.me_batch(planes, costtabs, jobs)
                if not np.array_equal(got, want):
                    errors.append(("me_batch", seed, rnd))
            ctx.close()
        except Exception as e:  # noqa: BLE001 - reported by the assertion below
            errors.append(("searcher", seed, repr(e)))

    def table_user(seed):
        try:
            rng = np.random.default_rng(seed)
            a = rng.integers(0, 256, 64 * 80, dtype=np.uint8)
            b = rng.integers(0, 256, 100 * 80, dtype=np.uint8)
            for _ in range(60):
                part = int(rng.integers(0, 25))
                got = table.satd[part](C.c_void_p(a.ctypes.data), 64, C.c_void_p(b.ctypes.data + 7), 100)
                if got != oracle.satd(part, a, 0, 64, b, 7, 100):
                    errors.append(("satd", seed, part))
        except Exception as e:  # noqa: BLE001
            errors.append(("table_user", seed, repr(e)))

    threads = [threading.Thread(target=searcher, args=(70 + k,)) for k in range(3)]
    threads += [threading.Thread(target=table_user, args=(80 + k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_device_resident_entry_points(gpu_ctx, oracle, small_seq, costtabs):
    """the `_device` forms (records and results in HBM, asynchronous on the context's stream) against the oracle:
    me_batch, cost_batch, dist_batch, bidir_cost_batch, predict_batch, fullsearch"""
    import ctypes as C
    from conftest import random_bidir_reqs, random_cost_reqs, random_pred_reqs
    from x265_amd import abi, lib, synth
    hip = C.CDLL("libamdhip64.so")  # plain HIP allocations: what a C host would hand over
    rng = np.random.default_rng(65)
    W, H = small_seq[0].w, small_seq[0].h
    planes = _planes(small_seq)
    held = []

    def dev_alloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 16))) == 0
        held.append(p)
        return p

    def to_dev(a):
        a = np.ascontiguousarray(a)
        p = dev_alloc(a.nbytes)
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0  # hipMemcpyHostToDevice
        return p.value

    def run(call, recs, out_dtype, n_out=None):
        d = to_dev(recs)
        n_out = len(recs) if n_out is None else n_out
        out = np.zeros(n_out, out_dtype)
        o = dev_alloc(out.nbytes)
        call(d, len(recs), o.value)
        gpu_ctx.sync()
        assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), o, C.c_size_t(out.nbytes), 2) == 0  # hipMemcpyDeviceToHost
        return out

    jobs = random_me_jobs(rng, 600, 1, 2, W, H, abi.ME_JOB, synth.search_range, abi.PARTS)
    assert np.array_equal(run(gpu_ctx.me_batch_device, jobs, abi.ME_RESULT), oracle.me_batch(planes, costtabs, jobs))
    reqs = random_cost_reqs(rng, 1500, W, H, abi.COST_REQ, abi.PARTS)
    assert np.array_equal(run(gpu_ctx.cost_batch_device, reqs, np.int32), oracle.cost_batch(planes, reqs))
    reqs = random_cost_reqs(rng, 1500, W, H, abi.COST_REQ, abi.PARTS, metrics=(0, 1, 2, 3))
    assert np.array_equal(run(gpu_ctx.dist_batch_device, reqs, np.int32), oracle.cost_batch(planes, reqs))
    br = random_bidir_reqs(rng, 1500, W, H, abi.BIDIR_REQ, abi.PARTS, metrics=(0, 1, 2, 3), flags=(0, abi.MC_AVG))
    assert np.array_equal(run(gpu_ctx.bidir_cost_batch_device, br, np.int32), oracle.bidir_cost_batch(planes, br))
    cj = np.zeros(3, lib.CTU_JOB)
    for k, j in enumerate(cj):
        j["fenc_plane"], j["ref_plane"], j["ctu_x"], j["ctu_y"], j["qp"] = 1, 0, 64 * (k + 1), 64, 30
        j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"] = -9 - k, -7, 8, 6 + k
        j["qmvp_x"], j["qmvp_y"] = 5 * k, -3
    got = run(gpu_ctx.fullsearch_device, cj, abi.ME_RESULT, n_out=3 * 85).reshape(3, 85)
    for k, j in enumerate(cj):
        want = oracle.ctu_fullsearch(planes[1], planes[0], costtabs[30], int(j["ctu_x"]), int(j["ctu_y"]),
                                     (int(j["mvmin_x"]), int(j["mvmin_y"])), (int(j["mvmax_x"]), int(j["mvmax_y"])),
                                     (int(j["qmvp_x"]), int(j["qmvp_y"])))
        assert np.array_equal(got[k], want.view(got.dtype)), k
    p0 = small_seq[0]
    want = np.full_like(p0.buf, 0x5A)
    gpu_ctx.plane_upload(5, want, p0.stride, p0.origin)
    pr = random_pred_reqs(rng, W, H, abi.PRED_REQ, abi.PARTS, dst_plane=5)
    gpu_ctx.predict_batch_device(to_dev(pr), len(pr))
    gpu_ctx.sync()
    pr["dst_plane"] = 3
    oracle.predict_batch(planes + [(want, p0.origin, p0.stride)], pr)
    assert np.array_equal(gpu_ctx.plane_download(5, want.size), want)
    gpu_ctx.plane_release(5)
    for p in held:
        hip.hipFree(p)


def _random_refine_reqs(rng, n, W, H, parts, planes=(1, (0, 2))):
    from x265_amd import abi
    reqs = np.zeros(n, abi.REFINE_REQ)
    for r in reqs:
        w, h = parts[int(rng.integers(0, 25))]
        r["fenc_plane"], r["ref_plane"] = planes[0], int(rng.choice(planes[1]))
        r["pu_x"], r["pu_y"], r["pu_w"], r["pu_h"] = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h
        r["metric"] = int(rng.integers(0, 2))
        r["pattern"] = int(rng.choice([abi.RING_QPEL, abi.RING_HPEL])) | (abi.RING_8 if rng.integers(2) else 0) | \
            (abi.RING_CENTRE if rng.integers(2) else 0)
        r["qmv_x"] = int(rng.integers(-90, 90)) if rng.integers(4) else 4 * int(rng.integers(-20, 20))
        r["qmv_y"] = int(rng.integers(-90, 90)) if rng.integers(4) else 4 * int(rng.integers(-20, 20))
    return reqs


def _check_refine(ctx, oracle, planes, reqs):
    from x265_amd import abi
    got = ctx.refine_batch(reqs)
    creqs, used = abi.refine_as_cost_reqs(reqs)
    want = oracle.cost_batch([p.as_tuple() for p in planes], creqs.reshape(-1), nthreads=8).reshape(-1, abi.RING_COSTS)
    want = np.where(used, want, 0x7fffffff)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, (len(bad), reqs[bad[0]], got[bad[0]], want[bad[0]])


def test_refine_rings_all_partitions_patterns_fractions(gpu_ctx, small_seq, oracle):
    """x265_hip_refine_batch: the shared-interpolation ring kernel equals nine independent subpelCompare + sad / satd
    for every partition, both rings, 4 / 8 directions, with and without the centre, at any centre fraction"""
    from x265_amd import abi
    rng = np.random.default_rng(2024)
    _check_refine(gpu_ctx, oracle, small_seq, _random_refine_reqs(rng, 6000, 416, 240, abi.PARTS))


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 63, 257])
def test_refine_ragged_counts(gpu_ctx, small_seq, oracle, n):
    from x265_amd import abi
    rng = np.random.default_rng(n)
    _check_refine(gpu_ctx, oracle, small_seq, _random_refine_reqs(rng, n, 416, 240, abi.PARTS))


@pytest.mark.parametrize("size", [8, 16, 32, 64])
def test_refine_lists_of_one_size(gpu_ctx, small_seq, oracle, size):
    """lists of a single PU size (the cell grouping packs four 8x8 PUs into a pass, a 64x64 PU is sixteen cells),
    every fraction of the centre, SATD, half-pel ring with centre then quarter-pel ring"""
    from x265_amd import abi
    reqs = []
    for pat in (abi.RING_HPEL | abi.RING_8 | abi.RING_CENTRE, abi.RING_QPEL | abi.RING_8, abi.RING_HPEL | abi.RING_CENTRE, abi.RING_QPEL):
        for y in range(0, 240 - size + 1, size):
            for x in range(0, 416 - size + 1, size):
                r = np.zeros((), abi.REFINE_REQ)
                r["fenc_plane"], r["ref_plane"], r["pu_x"], r["pu_y"], r["pu_w"], r["pu_h"] = 1, 0, x, y, size, size
                r["metric"], r["pattern"] = abi.SATD, pat
                r["qmv_x"], r["qmv_y"] = (x // size) % 9 - 4, (y // size) % 7 - 3
                reqs.append(r)
    _check_refine(gpu_ctx, oracle, small_seq, np.array(reqs, abi.REFINE_REQ))


@pytest.mark.parametrize("runs", [[(64, 64, 7), (32, 32, 33), (16, 16, 130), (8, 8, 517)], [(8, 8, 1), (64, 64, 1)], [(32, 24, 70), (16, 4, 300), (12, 16, 90)],
                                  [(16, 16, 2000)], [(64, 64, 3), (8, 8, 0), (8, 32, 65)], [(8, 8, 40), (16, 16, 40), (8, 8, 40), (16, 16, 40), (8, 8, 40)]])
def test_refine_lists_made_of_runs(gpu_ctx, small_seq, oracle, runs):
    """lists that are a few runs of one PU size each: through the host-array entry point (which finds the runs and cuts
    the batches per run; five runs fall back to the sampled launch) and through x265_hip_refine_sections_device with the
    runs given, both equal to the oracle"""
    from x265_amd import abi, lib
    from x265_amd.hipmem import DeviceBuffer
    rng = np.random.default_rng(len(runs) * 1000 + runs[0][2])
    parts = []
    for w, h, n in runs:
        r = _random_refine_reqs(rng, n, 416, 240, [(w, h)] * 25)
        parts.append(r)
    reqs = np.concatenate(parts)
    _check_refine(gpu_ctx, oracle, small_seq, reqs)
    if len(runs) <= abi.MAX_SECTIONS:
        want = gpu_ctx.refine_batch(reqs)
        d, o = DeviceBuffer.from_array(reqs), DeviceBuffer(36 * len(reqs))
        gpu_ctx.refine_sections_device(d.ptr, [(n, w, h) for w, h, n in runs], o.ptr)
        gpu_ctx.sync()
        assert np.array_equal(o.download(np.int32).reshape(-1, 9), want)
        # in record-check mode a section that does not hold what it promises is refused
        if len(runs) > 1 and runs[0][2] and runs[1][2]:
            gpu_ctx.check_records(True)
            try:
                lied = [(n, w, h) for w, h, n in runs]
                lied[0], lied[1] = (lied[0][0], lied[1][1], lied[1][2]), (lied[1][0], lied[0][1], lied[0][2])
                with pytest.raises(lib.X265HipError, match="in a section of"):
                    gpu_ctx.refine_sections_device(d.ptr, lied, o.ptr)
            finally:
                gpu_ctx.check_records(False)
    else:
        with pytest.raises(lib.X265HipError):
            gpu_ctx.refine_sections_device(0, [(n, w, h) for w, h, n in runs], 0)


def test_me_jobs_that_could_step_outside_a_tight_plane_are_refused(oracle):
    """the host-array entry point validates what a search can reach, per method and sub-pel level: the reference's
    hexagon search steps up to 3 samples outside [mvmin, mvmax] (2 + the square refine), sub-pel refinement up to 2
    more, and the 8-tap filters 4 below.  A plane whose bottom margin is one row short of that must be refused, one
    that has exactly the rows must be accepted and searched like the oracle does."""
    from x265_amd import abi, lib, synth
    seq = synth.Sequence(128, 128, seed=5, n_patches=1)
    full = synth.Plane(seq.frame(0))
    ctx = lib.Context(0)
    ctx.set_qp(30)
    job = np.zeros(1, abi.ME_JOB)
    job["fenc_plane"], job["ref_plane"] = 0, 1
    job["pu_x"], job["pu_y"], job["pu_w"], job["pu_h"] = 32, 64, 16, 16
    job["method"], job["subme"], job["merange"], job["qp"] = abi.HEX, 7, 16, 30
    job["mvmin_x"], job["mvmin_y"], job["mvmax_x"], job["mvmax_y"] = -8, -8, 8, 40
    # lowest row a HEX / subme 7 search can read: block bottom (64 + 16 - 1) + mvmax 40 + 3 (hexagon + square) +
    # ceil((2 * 2 + 2) / 4) = 2 (sub-pel) + 4 (filter) = row 128
    need_rows = synth.MARGIN_Y + 64 + 16 + 40 + 3 + 2 + 4
    ctx.plane_upload(0, full.buf, full.stride, full.origin)
    for rows, ok in ((need_rows, True), (need_rows - 1, False)):
        ctx.plane_upload(1, full.buf[:rows * full.stride], full.stride, full.origin)
        if ok:
            got = ctx.me_batch(job)
            want = oracle.me_batch([full.as_tuple(), full.as_tuple()], {30: lib.costtab_build(30)}, job)
            assert np.array_equal(got, want)
        else:
            with pytest.raises(lib.X265HipError, match="lie outside"):
                ctx.me_batch(job)
    # a diamond search with integer-only refinement reaches less: 1 + 1 + 4
    job["method"], job["subme"] = abi.DIA, 0
    ctx.plane_upload(1, full.buf[:(need_rows - 3) * full.stride], full.stride, full.origin)
    ctx.me_batch(job)
    ctx.close()


def test_inter_search_stage_vs_oracle(gpu_ctx, small_seq, costtabs, oracle):
    """predInterSearch's (list, reference) iteration on the device (AMVP pick, xSetSearchRange, motionEstimate) against
    the oracle: every partition, all search methods, one or two candidates (equal, near, far, outside the picture so
    that clipMv acts), a frame-parallel lag clamp, predictions from another plane than the one searched"""
    from x265_amd import abi
    rng = np.random.default_rng(99)
    W, H, n = 416, 240, 3000
    jobs = np.zeros(n, abi.INTER_JOB)
    for j in jobs:
        w, h = abi.PARTS[int(rng.integers(1, 25))]
        cu = max(w, h) if max(w, h) in (8, 16, 32, 64) else 64
        cu = 8 if cu < 8 else cu
        cx = int(rng.integers(0, (W - cu) // 8 + 1)) * 8
        cy = int(rng.integers(0, (H - cu) // 8 + 1)) * 8
        j["fenc_plane"], j["ref_plane"] = 1, int(rng.choice([0, 2]))
        j["pred_plane"] = j["ref_plane"] if rng.integers(3) else int(rng.choice([0, 2]))
        j["cu_x"], j["cu_y"] = cx, cy
        j["pu_x"], j["pu_y"] = cx + int(rng.integers(0, (cu - w) // 4 + 1)) * 4, cy + int(rng.integers(0, (cu - h) // 4 + 1)) * 4
        j["pu_w"], j["pu_h"] = w, h
        j["method"], j["subme"] = int(rng.integers(0, 4)), int(rng.integers(0, 8))
        j["merange"] = int(rng.choice([4, 8, 16, 57]))
        j["ref_lag"] = H if rng.integers(3) else int(rng.choice([8, 16, 57]))
        j["qp"] = int(rng.integers(20, 45))
        j["num_amvp"] = 1 + int(rng.integers(0, 4) > 0)
        kind = int(rng.integers(4))
        base = (0, 0) if kind == 0 else (int(rng.integers(-40, 40)), int(rng.integers(-40, 40))) if kind == 1 else \
            (int(rng.integers(-2500, 2500)), int(rng.integers(-1500, 1500)))
        j["amvp"][0], j["amvp"][1] = base
        if rng.integers(3):
            j["amvp"][2], j["amvp"][3] = base[0] + int(rng.integers(-12, 13)), base[1] + int(rng.integers(-12, 13))
        else:
            j["amvp"][2], j["amvp"][3] = base
        nm = int(rng.integers(0, 6))
        j["num_mvc"] = nm
        for c in range(nm):
            j["mvc"][2 * c] = base[0] + int(rng.integers(-30, 30)) if rng.integers(3) else 0
            j["mvc"][2 * c + 1] = base[1] + int(rng.integers(-30, 30)) if rng.integers(3) else 0
    got = gpu_ctx.inter_search_batch(jobs, W, H)
    want = oracle.inter_search_batch([p.as_tuple() for p in small_seq], costtabs, jobs, W, H)
    assert (got["mvp_idx"] == 1).sum() > 100
    bad = np.nonzero(got != want.view(got.dtype))[0]
    assert len(bad) == 0, (len(bad), jobs[bad[0]], got[bad[0]], want[bad[0]])
    for bad_job, match in ((("num_amvp", 0), "AMVP candidates"), (("pred_plane", 7), "not uploaded")):
        jb = jobs[:4].copy()
        jb[bad_job[0]][2] = bad_job[1]
        with pytest.raises(Exception, match=match):
            gpu_ctx.inter_search_batch(jb, W, H)


def test_inter_decide_stage_vs_oracle(gpu_ctx, small_seq, costtabs, oracle):
    """the rest of predInterSearch per prediction unit (x265_hip_inter_decide_batch) against the oracle on random units:
    one to three references in list 0, none to two in list 1, every partition, far and near AMVP candidates (so that
    xCheckBestMVP moves the index), units with the bidirectional candidate switched off, every xGetBlkBits row, small and
    large lambdas; every byte of the result records"""
    from x265_amd import abi
    from conftest import random_inter_units
    rng = np.random.default_rng(2360)
    W, H = 416, 240
    jobs, pus = random_inter_units(rng, 900, W, H, abi.INTER_JOB, abi.INTER_PU, abi.PARTS)
    res, got = gpu_ctx.inter_decide_batch(jobs, pus, W, H)
    planes = [p.as_tuple() for p in small_seq]
    wres = oracle.inter_search_batch(planes, costtabs, jobs, W, H)
    assert res.tobytes() == wres.tobytes()
    want = oracle.inter_decide_batch(planes, costtabs, jobs, wres, pus, W, H)
    bad = [u for u in range(len(pus)) if got[u].tobytes() != want[u].tobytes()]
    assert not bad, (len(bad), pus[bad[0]], got[bad[0]], want[bad[0]])
    assert (got["inter_dir"] == 3).sum() > 30 and (got["inter_dir"] == 2).sum() > 30 and got["zero_tried"].sum() > 30
    assert (got["list"]["mvp_idx"][:, 0] != res["mvp_idx"][pus["first_job"]])[pus["num_ref"][:, 0] == 1].sum() > 5


def test_record_check_mode_refuses_bad_device_records(gpu_ctx, small_seq):
    """x265_hip_ctx_check_records: with the mode on, the *_device entry points validate device-resident records like the
    host-array forms and return an error instead of launching (a record reaching outside its plane would otherwise
    fault the GPU); good lists still run and give the usual results"""
    from x265_amd import abi, lib, synth
    from x265_amd.hipmem import DeviceBuffer
    W, H = small_seq[0].w, small_seq[0].h
    rng = np.random.default_rng(77)
    jobs = random_me_jobs(rng, 200, abi.HEX, 2, W, H, abi.ME_JOB, synth.search_range, abi.PARTS)
    want = gpu_ctx.me_batch(jobs)
    d_out = DeviceBuffer(8 * len(jobs))
    gpu_ctx.check_records(True)
    try:
        d_jobs = DeviceBuffer.from_array(jobs)
        gpu_ctx.me_batch_device(d_jobs.ptr, len(jobs), d_out.ptr)
        gpu_ctx.sync()
        assert np.array_equal(d_out.download(abi.ME_RESULT), want)
        bad = jobs.copy()
        bad["mvmax_y"][17] = 30000  # a window that reaches far below the plane
        d_bad = DeviceBuffer.from_array(bad)
        with pytest.raises(lib.X265HipError, match=r"me_job\[17\]"):
            gpu_ctx.me_batch_device(d_bad.ptr, len(bad), d_out.ptr)
        rq = np.zeros(4, abi.COST_REQ)
        rq["fenc_plane"], rq["ref_plane"], rq["pu_w"], rq["pu_h"], rq["metric"] = 1, 0, 16, 16, abi.SATD
        rq["pu_x"], rq["pu_y"] = 32, 32
        rq["qmv_y"][2] = 4 * (H + 500)
        d_rq, d_c = DeviceBuffer.from_array(rq), DeviceBuffer(16)
        with pytest.raises(lib.X265HipError, match=r"cost_req\[2\]"):
            gpu_ctx.cost_batch_device(d_rq.ptr, 4, d_c.ptr)
        rr = np.zeros(2, abi.REFINE_REQ)
        rr["fenc_plane"], rr["ref_plane"], rr["pu_w"], rr["pu_h"], rr["metric"] = 1, 7, 8, 8, abi.SATD  # plane 7: not uploaded
        rr["pattern"] = abi.RING_HPEL
        d_rr, d_c9 = DeviceBuffer.from_array(rr), DeviceBuffer(72)
        with pytest.raises(lib.X265HipError, match="not uploaded"):
            gpu_ctx.refine_batch_device(d_rr.ptr, 2, d_c9.ptr)
    finally:
        gpu_ctx.check_records(False)


def test_record_guard_on_the_device_skips_bad_jobs(gpu_ctx, oracle, small_seq, costtabs):
    """x265_hip_ctx_check_records mode 2: me_route_kernel checks every job of a device-resident list (one thread per
    record, nothing copied back); bad records of every kind are searched by no kernel and come back as (cost -1, mv 0),
    the others give the usual results, the refusals are counted on the device; and the guard draws the line where the
    host-array entry point's check draws it"""
    from x265_amd import abi, lib, synth
    from x265_amd.hipmem import DeviceBuffer
    W, H = small_seq[0].w, small_seq[0].h
    rng = np.random.default_rng(99)
    jobs = np.concatenate([random_me_jobs(rng, 400, m, int(rng.integers(0, 8)), W, H, abi.ME_JOB, synth.search_range, abi.PARTS) for m in range(4)])
    rng.shuffle(jobs)
    want = oracle.me_batch(_planes(small_seq), costtabs, jobs)
    bad = jobs.copy()
    spoil = {
        5: lambda j: j.__setitem__("ref_plane", 511),          # not uploaded
        64: lambda j: j.__setitem__("fenc_plane", 65000),      # no such plane
        65: lambda j: j.__setitem__("pu_w", 20),               # not a partition
        300: lambda j: j.__setitem__("subme", 9),
        301: lambda j: j.__setitem__("num_mvc", abi.MAX_MVC + 1),
        700: lambda j: j.__setitem__("qp", 81),                # no table for it
        701: lambda j: j.__setitem__("mvmin_x", j["mvmax_x"] + 1),
        1023: lambda j: j.__setitem__("mvmax_y", 30000),       # reaches far below the plane
        1024: lambda j: j.__setitem__("mvmin_y", -30000),   # ... far above it
        1500: lambda j: j.__setitem__("pu_y", H + 400),        # the source block itself lies outside
        1599: lambda j: j.__setitem__("method", 9),
    }
    accepted = []
    for i, f in spoil.items():
        f(bad[i])
        try:  # the host-array form refuses each of them
            gpu_ctx.me_batch(bad[i:i + 1])
            accepted.append(i)
        except lib.X265HipError:
            pass
    assert not accepted
    d_jobs, d_out = DeviceBuffer.from_array(bad), DeviceBuffer(8 * len(bad))
    gpu_ctx.check_records(2)
    try:
        assert gpu_ctx.guard_status() == (0, 0xFFFFFFFF)
        gpu_ctx.me_batch_device(d_jobs.ptr, len(bad), d_out.ptr)
        with pytest.raises(lib.X265HipError, match=rf"refused {len(spoil)} .* index 5 "):
            gpu_ctx.sync()
        gpu_ctx.sync()  # reported once
        got = d_out.download(abi.ME_RESULT)
        ok = np.ones(len(bad), bool)
        ok[list(spoil)] = False
        assert np.array_equal(got[ok], want[ok])
        assert (got["cost"][~ok] == -1).all() and (got["mvx"][~ok] == 0).all() and (got["mvy"][~ok] == 0).all()
        # two launches, then one look
        gpu_ctx.me_batch_device(d_jobs.ptr, len(bad), d_out.ptr)
        gpu_ctx.me_batch_device(d_jobs.ptr + 64 * abi.ME_JOB.itemsize, len(bad) - 64, d_out.ptr)
        assert gpu_ctx.guard_status() == (2 * len(spoil) - 1, 0)  # (job 5 is not in the second list, job 64 heads it)
        assert gpu_ctx.guard_status() == (0, 0xFFFFFFFF)
        # a good list
        d_good = DeviceBuffer.from_array(jobs)
        gpu_ctx.me_batch_device(d_good.ptr, len(jobs), d_out.ptr)
        gpu_ctx.sync()
        assert np.array_equal(d_out.download(abi.ME_RESULT), want)
        # the line: a block near the lower right corner whose window grows past the plane's margins step by step
        edge = np.repeat(jobs[:1], 160)
        edge["pu_w"], edge["pu_h"], edge["pu_x"], edge["pu_y"] = 16, 16, W - 16, H - 16
        edge["method"], edge["subme"], edge["qmvp_x"], edge["qmvp_y"], edge["num_mvc"] = abi.HEX, 2, 0, 0, 0
        edge["mvmin_x"], edge["mvmin_y"] = -4, -4
        edge["mvmax_x"] = np.arange(160) // 2 + 20
        edge["mvmax_y"] = (np.arange(160) + 1) // 2 + 20
        host_ok = np.ones(160, bool)
        for i in range(160):
            try:
                gpu_ctx.me_batch(edge[i:i + 1])
            except lib.X265HipError:
                host_ok[i] = False
        assert host_ok[0] and not host_ok[-1]
        d_edge = DeviceBuffer.from_array(edge)
        gpu_ctx.me_batch_device(d_edge.ptr, 160, d_out.ptr)
        n_ref, first = gpu_ctx.guard_status()
        got = d_out.download(abi.ME_RESULT, 160)
        assert np.array_equal(got["cost"] != -1, host_ok) and n_ref == int((~host_ok).sum()) and first == int(np.argmin(host_ok))
    finally:
        gpu_ctx.check_records(False)


def test_record_guard_of_the_request_lists(gpu_ctx, small_seq):
    """x265_hip_ctx_check_records mode 2 for the request lists (cost, dist, refine, refine in runs, bidir): a thread per
    record checks it on the device and the kernel runs a list in which the records that fail are replaced by harmless
    ones; their results come back as -1 (all nine of a ring), every other result is the usual one, nothing is copied back
    and the refusals are counted"""
    from conftest import random_bidir_reqs, random_cost_reqs
    from x265_amd import abi, lib
    from x265_amd.hipmem import DeviceBuffer
    W, H = small_seq[0].w, small_seq[0].h
    rng = np.random.default_rng(4242)

    def spoil(recs, fields):
        bad = recs.copy()
        idx = rng.choice(len(recs), len(fields), replace=False)
        for i, (k, v) in zip(idx, fields):
            bad[k][i] = v
        return bad, np.sort(idx)

    def run(call, recs, per):
        d, o = DeviceBuffer.from_array(recs), DeviceBuffer(4 * per * len(recs))
        call(d.ptr, len(recs), o.ptr)
        n_ref, first = gpu_ctx.guard_status()
        got = o.download(np.int32).reshape(len(recs), per)
        d.free()
        o.free()
        return got, n_ref, first

    cost = random_cost_reqs(rng, 3000, W, H, abi.COST_REQ, abi.PARTS)
    ring = np.zeros(2000, abi.REFINE_REQ)
    for k in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "qmv_x", "qmv_y"):
        ring[k] = cost[k][:2000]
    ring["metric"] = cost["metric"][:2000] & 1
    ring["pattern"] = rng.choice([abi.RING_HPEL, abi.RING_QPEL, abi.RING_HPEL | abi.RING_8, abi.RING_QPEL | abi.RING_8 | abi.RING_CENTRE], 2000)
    bidir = random_bidir_reqs(rng, 2000, W, H, abi.BIDIR_REQ, abi.PARTS)
    far = [("ref_plane", 300), ("fenc_plane", -1), ("pu_w", 20), ("metric", 9), ("qmv_y", 4 * (H + 400)), ("qmv_y", -4 * (H + 400)), ("pu_y", H + 300)]
    gpu_ctx.check_records(2)
    try:
        for name, recs, fields, call, per, host in (
                ("cost", cost, far, gpu_ctx.cost_batch_device, 1, gpu_ctx.cost_batch),
                ("dist", cost, far, gpu_ctx.dist_batch_device, 1, gpu_ctx.dist_batch),
                ("refine", ring, far[:3] + [("pattern", 0x40), ("qmv_y", 4 * (H + 400)), ("pu_y", H + 300)], gpu_ctx.refine_batch_device, 9,
                 gpu_ctx.refine_batch),
                ("bidir", bidir, [("ref1_plane", 300), ("ref0_plane", 511), ("pu_h", 20), ("metric", 0xC1), ("qmv1_y", 4 * (H + 400)),
                                  ("qmv0_y", -4 * (H + 400))], gpu_ctx.bidir_cost_batch_device, 1, gpu_ctx.bidir_cost_batch)):
            if name == "dist":
                recs = recs.copy()
                recs["metric"] = rng.integers(0, 4, len(recs))
            want = np.asarray(host(recs)).reshape(len(recs), per)
            bad, idx = spoil(recs, fields)
            for i in idx:  # the host-array form refuses each of them
                with pytest.raises(lib.X265HipError):
                    host(bad[i:i + 1])
            got, n_ref, first = run(call, bad, per)
            ok = np.ones(len(recs), bool)
            ok[idx] = False
            assert np.array_equal(got[ok], want[ok]), name
            assert (got[~ok] == -1).all() and (n_ref, first) == (len(idx), int(idx[0])), (name, n_ref, first, idx)
            got, n_ref, first = run(call, recs, per)  # a good list
            assert np.array_equal(got, want) and n_ref == 0, name
        # prediction requests measured against the source (x265_hip_mc_satd_batch_device)
        from conftest import random_pred_reqs
        pr = np.concatenate([random_pred_reqs(rng, W, H, abi.PRED_REQ, abi.PARTS, dst_plane=1, refs=(0, 2), weighted_bi=True) for _ in range(20)])
        want = gpu_ctx.mc_satd_batch(pr)
        bi = np.nonzero(pr["ref1_plane"] >= 0)[0]
        bad, idx = pr.copy(), np.sort(np.concatenate([rng.choice(bi, 3, replace=False), rng.choice(np.setdiff1d(np.arange(len(pr)), bi), 4, replace=False)]))
        for i, (k, v) in zip(idx, [("ref0_plane", 300), ("dst_plane", -1), ("pu_w", 20), ("qmv0_y", 4 * (H + 400)), ("ref1_plane", 300), ("pu_y", H + 300),
                                   ("log2_denom", 9)]):
            bad[k][i] = v
            if k == "log2_denom":
                bad["weighted_bi"][i], bad["ref1_plane"][i] = 1, 0
        for i in idx:
            with pytest.raises(lib.X265HipError):
                gpu_ctx.mc_satd_batch(bad[i:i + 1])
        got, n_ref, first = run(gpu_ctx.mc_satd_batch_device, bad, 1)
        ok = np.ones(len(pr), bool)
        ok[idx] = False
        assert np.array_equal(got[ok, 0], want[ok]) and (got[~ok] == -1).all() and (n_ref, first) == (len(idx), int(idx[0]))
        # CTU full searches (x265_hip_fullsearch_device): refused CTUs give 85 x (cost -1, mv 0)
        cj = np.zeros(12, lib.CTU_JOB)
        cj["fenc_plane"], cj["ref_plane"], cj["qp"] = 1, rng.choice([0, 2], 12), 30
        cj["ctu_x"], cj["ctu_y"] = 64 * rng.integers(0, W // 64, 12), 64 * rng.integers(0, H // 64, 12)
        cj["mvmin_x"], cj["mvmin_y"], cj["mvmax_x"], cj["mvmax_y"] = -8, -6, 7, 9
        want = gpu_ctx.fullsearch(cj)
        bad = cj.copy()
        bad["ref_plane"][2], bad["qp"][5], bad["mvmax_y"][6], bad["mvmax_x"][9], bad["ctu_y"][11] = 300, 81, 400, 200, H + 400
        idx = np.array([2, 5, 6, 9, 11])
        for i in idx:
            with pytest.raises(lib.X265HipError):
                gpu_ctx.fullsearch(bad[i:i + 1])
        d, o = DeviceBuffer.from_array(bad), DeviceBuffer(8 * 85 * len(bad))
        gpu_ctx.fullsearch_device(d.ptr, len(bad), o.ptr)
        assert gpu_ctx.guard_status() == (5, 2)
        got = o.download(abi.ME_RESULT).reshape(len(bad), 85)
        ok = np.ones(len(bad), bool)
        ok[idx] = False
        assert np.array_equal(got[ok], want[ok])
        assert (got["cost"][~ok] == -1).all() and (got["mvx"][~ok] == 0).all() and (got["mvy"][~ok] == 0).all()
        d.free()
        o.free()
        # rings in runs of one size (x265_hip_refine_sections_device)
        runs = [ring[ring["pu_w"] == w_] for w_ in (16, 8)]
        runs = [r_[r_["pu_h"] == r_["pu_w"]] for r_ in runs]
        recs = np.concatenate(runs)
        want = gpu_ctx.refine_batch(recs)
        bad, idx = spoil(recs, [("ref_plane", 300), ("qmv_y", 4 * (H + 400))])
        d, o = DeviceBuffer.from_array(bad), DeviceBuffer(36 * len(bad))
        gpu_ctx.refine_sections_device(d.ptr, [(len(r_), int(r_["pu_w"][0]), int(r_["pu_h"][0])) for r_ in runs], o.ptr)
        assert gpu_ctx.guard_status() == (2, int(idx[0]))
        got = o.download(np.int32).reshape(len(bad), 9)
        ok = np.ones(len(bad), bool)
        ok[idx] = False
        assert np.array_equal(got[ok], want[ok]) and (got[~ok] == -1).all()
        # a good record of another size than its run's (mode 1 refuses it) only costs time: the runs size the batches, the
        # kernel takes a request's strips from the request
        odd = recs.copy()
        big = ring[(ring["pu_w"] == 64) & (ring["pu_h"] == 64)][:2]
        assert len(big) == 2
        odd[3], odd[len(runs[0]) + 5] = big[0], big[1]
        want = gpu_ctx.refine_batch(odd)
        d2 = DeviceBuffer.from_array(odd)
        gpu_ctx.refine_sections_device(d2.ptr, [(len(r_), int(r_["pu_w"][0]), int(r_["pu_h"][0])) for r_ in runs], o.ptr)
        assert gpu_ctx.guard_status() == (0, 0xFFFFFFFF)
        assert np.array_equal(o.download(np.int32).reshape(len(odd), 9), want)
        for b_ in (d, d2, o):
            b_.free()
    finally:
        gpu_ctx.check_records(False)


@pytest.mark.parametrize("one_stream", [False, True])
def test_large_block_jobs_and_their_routing(oracle, small_seq, costtabs, one_stream, monkeypatch):
    """blocks of 128 tiles and more are listed on the device and searched by me_big_kernel beside me_kernel (on a second
    stream, or on the one stream with X265_HIP_ME_ONE_STREAM=1): lists of big blocks only (every such size, every
    method but FULL, every sub-pel level), of none, and mixed ones, with growing and shrinking job counts back to back
    on one context (the two counter sets of the list alternate; the list buffer grows), through the device-array and
    the host-array entry point"""
    from x265_amd import abi, lib, synth
    from x265_amd.hipmem import DeviceBuffer
    if one_stream:
        monkeypatch.setenv("X265_HIP_ME_ONE_STREAM", "1")
    ctx = lib.Context(0)
    for i, p in enumerate(small_seq):
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
    for q, t in costtabs.items():
        ctx.costtab_upload(q, t)
    rng = np.random.default_rng(128 + one_stream)
    W, H = small_seq[0].w, small_seq[0].h
    planes = _planes(small_seq)
    big_parts = [i for i in range(1, 25) if abi.PARTS[i][0] % 16 == 0 and abi.PARTS[i][0] * abi.PARTS[i][1] >= 128 * 16]
    assert sorted(abi.PARTS[i] for i in big_parts) == [(32, 64), (48, 64), (64, 32), (64, 48), (64, 64)]
    small_parts = [i for i in range(1, 25) if i not in big_parts]

    def jobs_of(n, which, method, subme):
        # random_me_jobs draws partition indices 1..24: hand it a table that maps every index into the wanted set
        table = [abi.PARTS[0]] + [abi.PARTS[which[int(rng.integers(len(which)))]] for _ in range(24)]
        return random_me_jobs(rng, n, method, subme, W, H, abi.ME_JOB, synth.search_range, table)

    both = big_parts + small_parts
    lists = [jobs_of(40, big_parts, m, s) for m in (0, 1, 2, 3) for s in (0, 2, 3, 5, 7)]
    lists += [jobs_of(n, which, int(rng.integers(0, 4)), int(rng.integers(0, 8)))
              for n, which in ((700, both), (3, big_parts), (5000, both), (900, small_parts), (130, big_parts), (1, big_parts), (64, both), (2000, both))]
    for k, jobs in enumerate(lists):
        want = oracle.me_batch(planes, costtabs, jobs)
        d = DeviceBuffer.from_array(jobs)
        o = DeviceBuffer(len(jobs) * abi.ME_RESULT.itemsize)
        ctx.me_batch_device(d.ptr, len(jobs), o.ptr)
        ctx.sync()
        got = o.download(abi.ME_RESULT, len(jobs))
        assert np.array_equal(got, want), (k, "device", np.nonzero(got != want)[0][:5])
        assert np.array_equal(ctx.me_batch(jobs), want), (k, "host")
    ctx.close()


def test_long_lists_search_32x32_blocks_in_teams(oracle, small_seq, costtabs):
    """lists of 262,144 calls and more send their 32x32, 32x16, 32x24, 24x32, 32x8 blocks to me_team_kernel's class of
    8x8 cells (shorter lists to me_rest_kernel, me_launch_both): a short list repeated to that length, DIA and HEX at
    every sub-pel level, every copy against the oracle"""
    from x265_amd import abi, lib, synth
    from x265_amd.hipmem import DeviceBuffer
    ctx = lib.Context(0)
    for i, p in enumerate(small_seq):
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
    for q, t in costtabs.items():
        ctx.costtab_upload(q, t)
    rng = np.random.default_rng(3232)
    W, H = small_seq[0].w, small_seq[0].h
    t32 = [i for i in range(1, 25) if 24 <= abi.PARTS[i][0] <= 32 and abi.PARTS[i][1] <= 32 and abi.PARTS[i][0] % 8 == 0 and abi.PARTS[i][1] % 8 == 0]
    assert sorted(abi.PARTS[i] for i in t32) == [(24, 32), (32, 8), (32, 16), (32, 24), (32, 32)]
    mixed = t32 * 3 + list(range(1, 25))
    for method in (abi.DIA, abi.HEX):
        for subme in range(8):
            table = [abi.PARTS[0]] + [abi.PARTS[mixed[int(rng.integers(len(mixed)))]] for _ in range(24)]
            jobs = random_me_jobs(rng, 150, method, subme, W, H, abi.ME_JOB, synth.search_range, table)
            want = oracle.me_batch(_planes(small_seq), costtabs, jobs)
            reps = 262144 // len(jobs) + 1
            d = DeviceBuffer.from_array(np.tile(jobs, reps))
            o = DeviceBuffer(reps * len(jobs) * abi.ME_RESULT.itemsize)
            ctx.me_batch_device(d.ptr, reps * len(jobs), o.ptr)
            ctx.sync()
            got = o.download(abi.ME_RESULT, reps * len(jobs)).reshape(reps, len(jobs))
            bad = np.nonzero((got != want[None, :]).any(axis=0))[0]
            assert len(bad) == 0, f"method {method} subme {subme}: job {jobs[bad[0]]} got {got[:, bad[0]][:4]} want {want[bad[0]]}"
            d.free()
            o.free()
    ctx.close()


def test_empty_and_single_record_lists(gpu_ctx, oracle, small_seq, costtabs):
    """every batched entry point with no record at all (host arrays and the device forms with n = 0: nothing is
    launched, nothing is read) and with exactly one record (against the oracle): the ragged ends of the batch
    hand-out, the routing list of the motionEstimate launch and the staging buffers"""
    from conftest import random_bidir_reqs, random_cost_reqs
    from x265_amd import abi, lib, synth
    rng = np.random.default_rng(1)
    W, H = small_seq[0].w, small_seq[0].h
    planes = _planes(small_seq)
    assert len(gpu_ctx.me_batch(np.zeros(0, abi.ME_JOB))) == 0
    assert len(gpu_ctx.cost_batch(np.zeros(0, abi.COST_REQ))) == 0
    assert len(gpu_ctx.dist_batch(np.zeros(0, abi.COST_REQ))) == 0
    assert len(gpu_ctx.bidir_cost_batch(np.zeros(0, abi.BIDIR_REQ))) == 0
    assert len(gpu_ctx.refine_batch(np.zeros(0, abi.REFINE_REQ))) == 0
    assert len(gpu_ctx.inter_search_batch(np.zeros(0, abi.INTER_JOB), W, H)) == 0
    assert len(gpu_ctx.fullsearch(np.zeros(0, lib.CTU_JOB))) == 0
    gpu_ctx.predict_batch(np.zeros(0, abi.PRED_REQ))
    for call in (gpu_ctx.me_batch_device, gpu_ctx.cost_batch_device, gpu_ctx.cost_batch_device_noalloc, gpu_ctx.dist_batch_device,
                 gpu_ctx.bidir_cost_batch_device, gpu_ctx.refine_batch_device, gpu_ctx.fullsearch_device):
        call(0, 0, 0)
    gpu_ctx.predict_batch_device(0, 0)
    gpu_ctx.inter_search_batch_device(0, 0, W, H, 0)
    gpu_ctx.sync()
    # an empty call must not take one of the two alternating counter sets (the launch after it would start on counters
    # nobody cleared): one empty call between two real launches, for the cost kernels and for the motionEstimate list
    for _ in range(3):
        reqs = random_cost_reqs(rng, 700, W, H, abi.COST_REQ, abi.PARTS)
        assert np.array_equal(gpu_ctx.cost_batch(reqs), oracle.cost_batch(planes, reqs))
        gpu_ctx.cost_batch_device(0, 0, 0)
        jobs = random_me_jobs(rng, 300, abi.HEX, 2, W, H, abi.ME_JOB, synth.search_range, abi.PARTS)
        assert np.array_equal(gpu_ctx.me_batch(jobs), oracle.me_batch(planes, costtabs, jobs))
        gpu_ctx.me_batch_device(0, 0, 0)
    for size in ((8, 8), (64, 64)):  # one job for me_kernel, one for me_big_kernel
        table = [abi.PARTS[0]] + [size] * 24
        jobs = random_me_jobs(rng, 1, abi.HEX, 2, W, H, abi.ME_JOB, synth.search_range, table)
        assert np.array_equal(gpu_ctx.me_batch(jobs), oracle.me_batch(planes, costtabs, jobs)), size
    reqs = random_cost_reqs(rng, 1, W, H, abi.COST_REQ, abi.PARTS)
    assert np.array_equal(gpu_ctx.cost_batch(reqs), oracle.cost_batch(planes, reqs))
    reqs = random_cost_reqs(rng, 1, W, H, abi.COST_REQ, abi.PARTS, metrics=(2,))
    assert np.array_equal(gpu_ctx.dist_batch(reqs), oracle.cost_batch(planes, reqs))
    br = random_bidir_reqs(rng, 1, W, H, abi.BIDIR_REQ, abi.PARTS)
    assert np.array_equal(gpu_ctx.bidir_cost_batch(br), oracle.bidir_cost_batch(planes, br))
    rr = _random_refine_reqs(rng, 1, W, H, abi.PARTS)
    _check_refine(gpu_ctx, oracle, small_seq, rr)
