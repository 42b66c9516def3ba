"""Deterministic synthetic input for the ME cost path (host side, numpy only).

Frames follow the recipe of SURVEY.md section 8(d): a smooth random texture panned a few pixels per
frame, a handful of independently moving patches (every fourth frame displaced by half a pixel so
that sub-pel candidates sometimes win) and +-2 noise.  Planes are laid out like the reference's
TComPicYuv luma plane (source/Lib/TLibCommon/TComPicYuv.cpp:78-97): marginX = 96, marginY = 80,
stride = ceil(W/64)*64 + 192, edge-extended the way extendPicBorder does (source/common/pixel.cpp:893-911).

Nothing here is timed; bench.py uploads the result once and keeps it resident in HBM.
"""
import numpy as np

from .abi import COST_REQ, LOWRES_REF, ME_JOB, MAX_MVC, SA8D, SATD

MARGIN_X = 96
MARGIN_Y = 80
CTU = 64

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on a uint64 array (counter-based, so every sample is reproducible)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def hash2d(h, w, seed):
    """uint64 noise field of shape (h, w)."""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64), indexing="ij")
    with np.errstate(over="ignore"):
        key = (yy * np.uint64(0x100000001B3) + xx + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)) & _M64
    return _mix(key)


def _smooth(a, passes):
    a = a.astype(np.float32)
    for _ in range(passes):
        a = (np.roll(a, 1, 0) + 2 * a + np.roll(a, -1, 0)) * 0.25
        a = (np.roll(a, 1, 1) + 2 * a + np.roll(a, -1, 1)) * 0.25
    return a


class Sequence:
    """Synthetic luma sequence of `width` x `height` pictures."""

    def __init__(self, width, height, seed=0x265, n_patches=8):
        self.w, self.h, self.seed = width, height, seed
        th, tw = height + 128, width + 128  # room for the pan
        coarse = (hash2d((th + 7) // 8, (tw + 7) // 8, seed) >> np.uint64(56)).astype(np.float32)
        coarse = np.kron(coarse, np.ones((8, 8), np.float32))[:th, :tw]
        mid = (hash2d((th + 1) // 2, (tw + 1) // 2, seed + 1) >> np.uint64(58)).astype(np.float32)
        mid = np.kron(mid, np.ones((2, 2), np.float32))[:th, :tw]
        tex = _smooth(coarse, 3) * 0.75 + _smooth(mid, 1) * 0.9 + 16.0
        self.tex = np.clip(tex, 0, 255).astype(np.uint8)
        p = _mix(np.arange(n_patches * 8, dtype=np.uint64) + np.uint64(seed * 7919))
        p = p.reshape(n_patches, 8)
        self.patches = []
        for k in range(n_patches):
            x0 = int(p[k, 0] % np.uint64(max(1, width - 256))) + 64
            y0 = int(p[k, 1] % np.uint64(max(1, height - 256))) + 64
            vx = int(p[k, 2] % np.uint64(9)) - 4
            vy = int(p[k, 3] % np.uint64(7)) - 3
            body = (hash2d(128 // 4, 128 // 4, seed + 100 + k) >> np.uint64(56)).astype(np.float32)
            body = np.clip(_smooth(np.kron(body, np.ones((4, 4), np.float32)), 2), 0, 255).astype(np.uint8)
            self.patches.append((x0, y0, vx, vy, body))

    def frame(self, n):
        w, h = self.w, self.h
        px, py = (3 * n) % 61, (2 * n) % 37
        f = self.tex[py:py + h, px:px + w].astype(np.int16)
        for (x0, y0, vx, vy, body) in self.patches:
            x = (x0 + n * vx) % max(1, w - 130)
            y = (y0 + n * vy) % max(1, h - 130)
            if n % 4 == 3:  # half-pel horizontal displacement: average of two integer placements
                a = f[y:y + 128, x:x + 129].copy()
                a[:, :128] = body
                b = f[y:y + 128, x:x + 129].copy()
                b[:, 1:129] = body
                f[y:y + 128, x:x + 129] = (a + b + 1) >> 1
            else:
                f[y:y + 128, x:x + 128] = body
        noise = (hash2d(h, w, self.seed + 1000 + n) >> np.uint64(60)).astype(np.int16) % 5 - 2
        return np.clip(f + noise, 0, 255).astype(np.uint8)

    def global_motion(self, n_cur, n_ref):
        """integer displacement (dx, dy) such that cur(x, y) ~ ref(x + dx, y + dy) for the background."""
        cx, cy = (3 * n_cur) % 61, (2 * n_cur) % 37
        rx, ry = (3 * n_ref) % 61, (2 * n_ref) % 37
        return cx - rx, cy - ry


class Plane:
    """Padded, edge-extended luma plane in the reference's TComPicYuv layout."""

    def __init__(self, pic):
        h, w = pic.shape
        self.w, self.h = w, h
        self.stride = -(-w // CTU) * CTU + 2 * MARGIN_X
        self.rows = -(-h // CTU) * CTU + 2 * MARGIN_Y
        self.origin = MARGIN_Y * self.stride + MARGIN_X
        buf = np.zeros((self.rows, self.stride), np.uint8)
        buf[MARGIN_Y:MARGIN_Y + h, MARGIN_X:MARGIN_X + w] = pic
        # left / right margins of picture rows, then whole rows up and down (extendPicBorder)
        buf[MARGIN_Y:MARGIN_Y + h, :MARGIN_X] = pic[:, :1]
        buf[MARGIN_Y:MARGIN_Y + h, MARGIN_X + w:MARGIN_X + w + MARGIN_X] = pic[:, -1:]
        buf[:MARGIN_Y, :] = buf[MARGIN_Y]
        buf[MARGIN_Y + h:MARGIN_Y + h + MARGIN_Y, :] = buf[MARGIN_Y + h - 1]
        self.buf = buf.reshape(-1)

    @property
    def nbytes(self):
        return self.buf.size

    @staticmethod
    def nbytes_for(w, h):
        """size of the padded plane of a w x h picture"""
        return (-(-w // CTU) * CTU + 2 * MARGIN_X) * (-(-h // CTU) * CTU + 2 * MARGIN_Y)

    def as_tuple(self):
        return (self.buf, self.origin, self.stride)


def ctu_grid(w, h):
    return -(-w // CTU), -(-h // CTU)


def square_pus(w, h, sizes=(64, 32, 16, 8)):
    """All square PUs of the given sizes lying wholly inside the picture: array of (x, y, size)."""
    out = []
    ncx, ncy = ctu_grid(w, h)
    for cy in range(ncy):
        for cx in range(ncx):
            for s in sizes:
                n = CTU // s
                for by in range(n):
                    for bx in range(n):
                        x, y = cx * CTU + bx * s, cy * CTU + by * s
                        if x + s <= w and y + s <= h:
                            out.append((x, y, s))
    return np.array(out, np.int32)


def clip_mv(qx, qy, cu_x, cu_y, w, h):
    """TComDataCU::clipMv (source/Lib/TLibCommon/TComDataCU.cpp:2502-2514), quarter-pel in/out."""
    xmax = (w + 8 - cu_x - 1) << 2
    xmin = (-CTU - 8 - cu_x + 1) << 2
    ymax = (h + 8 - cu_y - 1) << 2
    ymin = (-CTU - 8 - cu_y + 1) << 2
    return np.minimum(xmax, np.maximum(xmin, qx)), np.minimum(ymax, np.maximum(ymin, qy))


def search_range(qmvp_x, qmvp_y, merange, cu_x, cu_y, w, h):
    """TEncSearch::xSetSearchRange (source/Lib/TLibEncoder/TEncSearch.cpp:2600-2617) without the
    frame-parallel lag clamp: returns full-pel (mvmin_x, mvmin_y, mvmax_x, mvmax_y)."""
    px, py = clip_mv(qmvp_x, qmvp_y, cu_x, cu_y, w, h)
    d = merange << 2
    lox, loy = clip_mv(px - d, py - d, cu_x, cu_y, w, h)
    hix, hiy = clip_mv(px + d, py + d, cu_x, cu_y, w, h)
    return lox >> 2, loy >> 2, hix >> 2, hiy >> 2


def me_jobs(w, h, fenc_plane, ref_planes, motions, method=1, subme=2, merange=57, seed=1, sizes=(64, 32, 16, 8),
            qp_base=30, num_mvc=3):
    """Job list of one ME-stage frame: every square PU x every reference.

    `motions[i]` is the (dx, dy) background displacement towards `ref_planes[i]`; predictors are
    that motion in quarter-pel plus a small hashed perturbation, the way AMVP neighbours scatter
    around the true motion.  Search windows follow xSetSearchRange.
    """
    pus = square_pus(w, h, sizes)
    n = len(pus)
    jobs = np.zeros(n * len(ref_planes), ME_JOB)
    for r, (rp, (dx, dy)) in enumerate(zip(ref_planes, motions)):
        j = jobs[r * n:(r + 1) * n]
        hsh = _mix(np.arange(n, dtype=np.uint64) * np.uint64(16) + np.uint64(seed * 1000003 + r * 7919))
        j["fenc_plane"] = fenc_plane
        j["ref_plane"] = rp
        j["pu_x"], j["pu_y"] = pus[:, 0], pus[:, 1]
        j["pu_w"] = j["pu_h"] = pus[:, 2]
        j["method"], j["subme"], j["merange"] = method, subme, merange
        ctu_x, ctu_y = (pus[:, 0] // CTU) * CTU, (pus[:, 1] // CTU) * CTU
        j["qp"] = qp_base + ((ctu_x // CTU + 3 * (ctu_y // CTU)) % 7)
        px = 4 * dx + ((hsh >> np.uint64(8)) % np.uint64(13)).astype(np.int64) - 6
        py = 4 * dy + ((hsh >> np.uint64(16)) % np.uint64(13)).astype(np.int64) - 6
        j["qmvp_x"], j["qmvp_y"] = px, py
        lox, loy, hix, hiy = search_range(px, py, merange, ctu_x, ctu_y, w, h)
        j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"] = lox, loy, hix, hiy
        j["num_mvc"] = num_mvc
        for c in range(num_mvc):
            hx = ((hsh >> np.uint64(24 + 8 * c)) % np.uint64(17)).astype(np.int64) - 8
            hy = ((hsh >> np.uint64(28 + 8 * c)) % np.uint64(17)).astype(np.int64) - 8
            if c == num_mvc - 1:
                hx = hx * 0 - 4 * dx  # one candidate is the zero vector
                hy = hy * 0 - 4 * dy
            j["mvc"][:, 2 * c] = 4 * dx + hx
            j["mvc"][:, 2 * c + 1] = 4 * dy + hy
    return jobs


def inter_units(me, n_refs, height, n_list1=0, lambda_sad=4771):
    """The iterations of me_jobs() (reference-major: n_refs runs of one job per PU) as prediction units for
    x265_hip_inter_decide_batch: (x265_hip_inter_job records unit-major, x265_hip_inter_pu records).  The last n_list1
    references form list 1 (a B picture; the bidirectional candidate is tried), the others list 0.  Every unit is a 2Nx2N
    CU; its AMVP candidates are the job's predictor and a neighbour of it."""
    from .abi import INTER_JOB, INTER_PU
    n = len(me) // n_refs
    mj = me.reshape(n_refs, n).T.reshape(-1)
    ij = np.zeros(len(mj), INTER_JOB)
    for k in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "method", "subme", "merange", "qp", "num_mvc", "mvc"):
        ij[k] = mj[k]
    ij["pred_plane"] = mj["ref_plane"]
    ij["cu_x"], ij["cu_y"] = mj["pu_x"], mj["pu_y"]
    ij["amvp"][:, 0], ij["amvp"][:, 1] = mj["qmvp_x"], mj["qmvp_y"]
    ij["amvp"][:, 2], ij["amvp"][:, 3] = mj["qmvp_x"] + 3, mj["qmvp_y"] - 2
    ij["num_amvp"], ij["ref_lag"] = 2, height
    pu = np.zeros(n, INTER_PU)
    pu["first_job"] = np.arange(n) * n_refs
    pu["num_ref"] = (n_refs - n_list1, n_list1)
    pu["list_sel_bits"] = (3, 3, 5) if n_list1 else (1, 3, 5)  # xGetBlkBits of a 2Nx2N unit (TEncSearch.cpp:2509-2514)
    pu["no_bidir"] = 0 if n_list1 else 1
    pu["lambda_sad"] = lambda_sad
    return ij, pu


def subpel_requests(w, h, cur, refs, motions):
    """Cost requests of the sub-pel stage of one picture: every square PU x every reference x 16 candidates
    (half-pel ring, then quarter-pel ring, around the integer vector `motions[i]`), SATD after interpolation.
    Ordered reference-major, then candidate, then PU, so neighbouring requests share their fraction."""
    pus = square_pus(w, h)
    ring = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1)]
    offs = [(2 * x, 2 * y) for x, y in ring] + ring
    n = len(pus)
    out = np.zeros(n * len(refs) * len(offs), COST_REQ)
    k = 0
    for rp, (dx, dy) in zip(refs, motions):
        for ox, oy in offs:
            r = out[k:k + n]
            k += n
            r["fenc_plane"], r["ref_plane"] = cur, rp
            r["pu_x"], r["pu_y"] = pus[:, 0], pus[:, 1]
            r["pu_w"] = r["pu_h"] = pus[:, 2]
            r["metric"] = SATD
            r["qmv_x"], r["qmv_y"] = 4 * dx + ox, 4 * dy + oy
    return out


def ctu_pu_mask(w, h, sizes=(64, 32, 16, 8)):
    """which of a CTU-major list of all PUs (85 per CTU in the order 64, 32s, 16s, 8s, each raster: the result order of
    x265_hip_fullsearch) lie wholly inside the picture -- the PUs of square_pus(), in its order"""
    out = []
    ncx, ncy = ctu_grid(w, h)
    for cy in range(ncy):
        for cx in range(ncx):
            for s in sizes:
                n = CTU // s
                for by in range(n):
                    for bx in range(n):
                        out.append(cx * CTU + bx * s + s <= w and cy * CTU + by * s + s <= h)
    return np.array(out, bool)


def subme2_hpel_rings(w, h, cur, refs, winners, metric=SATD):
    """The sub-pel stage the medium preset asks for (workload[2], motion.cpp:51: one half-pel iteration of 4 points with
    the centre re-measured by SATD, one quarter-pel iteration of 4), first ring: per square PU and reference the
    4-point half-pel ring + centre around the integer search's winner.  `winners[i]`: (n_pus, 2) quarter-pel vectors
    (multiples of 4) for reference refs[i], PUs in square_pus() order.  Reference-major."""
    from .abi import REFINE_REQ, RING_CENTRE, RING_HPEL
    pus = square_pus(w, h)
    n = len(pus)
    out = np.zeros(n * len(refs), REFINE_REQ)
    for i, (rp, mv) in enumerate(zip(refs, winners)):
        r = out[i * n:(i + 1) * n]
        r["fenc_plane"], r["ref_plane"] = cur, rp
        r["pu_x"], r["pu_y"] = pus[:, 0], pus[:, 1]
        r["pu_w"] = r["pu_h"] = pus[:, 2]
        r["metric"], r["pattern"] = metric, RING_HPEL | RING_CENTRE
        r["qmv_x"], r["qmv_y"] = mv[:, 0], mv[:, 1]
    return out


def subme2_qpel_rings(hpel_reqs, hpel_costs):
    """second ring: the 4-point quarter-pel ring around the first ring's winner (first minimum of centre, then the
    ring in visiting order: the strict-less updates of motion.cpp:1090-1101; distortion only)"""
    from .abi import RING_QPEL, SQUARE1
    c = np.asarray(hpel_costs).reshape(len(hpel_reqs), -1)[:, :5].astype(np.int64)
    k = np.argmin(c, axis=1)  # the first minimum
    sq = np.array(SQUARE1, np.int64)
    out = hpel_reqs.copy()
    out["pattern"] = RING_QPEL
    out["qmv_x"] = hpel_reqs["qmv_x"] + 2 * sq[k, 0]
    out["qmv_y"] = hpel_reqs["qmv_y"] + 2 * sq[k, 1]
    return out


def satd_lane_ops(reqs):
    """Algorithmic 32-bit lane-ops of a request list: the comparison (SATD 111 per 4x4 tile;
    sa8d 575 per 8x8 block = 64 differences, sixteen 8-point transforms of 24 butterflies, 64 absolute values,
    63 additions; sse and sad 3 per pixel) plus the interpolation: hpp / vpp 8 MAC + 3 per pixel; diagonal
    8 (H + 7) / H + 8 MAC + 3 per pixel."""
    w, hgt = reqs["pu_w"].astype(np.float64), reqs["pu_h"].astype(np.float64)
    px = w * hgt
    fx, fy = (reqs["qmv_x"] & 3) != 0, (reqs["qmv_y"] & 3) != 0
    metric = reqs["metric"]
    blocks8 = ((reqs["pu_w"] | reqs["pu_h"]) & 7) == 0  # sa8d_inter falls back to satd elsewhere
    cmp_px = np.where(metric == SATD, 111.0 / 16, np.where(metric == SA8D, np.where(blocks8, 575.0 / 64, 111.0 / 16), 3.0))
    per_px = cmp_px + np.where(fx & fy, 8 * (hgt + 7) / hgt + 8 + 3, np.where(fx | fy, 11.0, 0.0))
    return float((px * per_px).sum())


def lookahead_jobs(lw, ll, fenc_plane, ref_plane0, motion, method=1, subme=1, merange=16, qp=12, seed=1):
    """Job list of one lookahead pass (slicetype.cpp:1512-1584): every 8x8 CU of a `lw` x `ll` lowres picture
    against the Lowres whose planes are ref_plane0 .. +3.  Search bounds as estimateCUCost sets them
    (:1539-1543); predictors scatter around `motion` (half-resolution full-pel dx, dy), standing in for the
    neighbour medians the lookahead takes from already estimated CUs.  The lookahead's own settings are HEX,
    subme 1, merange 16 and a fixed QP of X265_LOOKAHEAD_QP = 12 (slicetype.h:73-77, common.h:151)."""
    wcu, hcu = lw // 8, ll // 8
    cux, cuy = np.meshgrid(np.arange(wcu), np.arange(hcu), indexing="xy")
    cux, cuy = cux.reshape(-1), cuy.reshape(-1)
    n = len(cux)
    j = np.zeros(n, ME_JOB)
    hsh = _mix(np.arange(n, dtype=np.uint64) * np.uint64(16) + np.uint64(seed * 1000003))
    j["fenc_plane"], j["ref_plane"] = fenc_plane, ref_plane0
    j["pu_x"], j["pu_y"], j["pu_w"], j["pu_h"] = 8 * cux, 8 * cuy, 8, 8
    j["method"], j["subme"], j["merange"], j["qp"] = method | LOWRES_REF, subme, merange, qp
    j["mvmin_x"], j["mvmin_y"] = -cux * 8 - 8, -cuy * 8 - 8
    j["mvmax_x"], j["mvmax_y"] = (wcu - cux - 1) * 8 + 8, (hcu - cuy - 1) * 8 + 8
    dx, dy = motion
    j["qmvp_x"] = 4 * dx + ((hsh >> np.uint64(8)) % np.uint64(9)).astype(np.int64) - 4
    j["qmvp_y"] = 4 * dy + ((hsh >> np.uint64(16)) % np.uint64(9)).astype(np.int64) - 4
    j["num_mvc"] = 3
    for c in range(3):
        j["mvc"][:, 2 * c] = 4 * dx + ((hsh >> np.uint64(24 + 8 * c)) % np.uint64(13)).astype(np.int64) - 6
        j["mvc"][:, 2 * c + 1] = 4 * dy + ((hsh >> np.uint64(28 + 8 * c)) % np.uint64(13)).astype(np.int64) - 6
    return j


def refine_requests(w, h, cur, refs, motions, metric=SATD):
    """The sub-pel stage of subpel_requests() as ring requests (x265_hip_refine_batch): per square PU and reference
    the half-pel ring of 8 and the quarter-pel ring of 8 around the integer vector `motions[i]`: the same 16
    candidates per (PU, reference), reference-major, then ring, then PU."""
    from .abi import REFINE_REQ, RING_8, RING_HPEL, RING_QPEL
    pus = square_pus(w, h)
    n = len(pus)
    out = np.zeros(n * len(refs) * 2, REFINE_REQ)
    k = 0
    for rp, (dx, dy) in zip(refs, motions):
        for ring in (RING_HPEL, RING_QPEL):
            r = out[k:k + n]
            k += n
            r["fenc_plane"], r["ref_plane"] = cur, rp
            r["pu_x"], r["pu_y"] = pus[:, 0], pus[:, 1]
            r["pu_w"] = r["pu_h"] = pus[:, 2]
            r["metric"], r["pattern"] = metric, ring | RING_8
            r["qmv_x"], r["qmv_y"] = 4 * dx, 4 * dy
    return out
