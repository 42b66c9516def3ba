"""BASELINE configs[0] at its stated resolution inside the driver-run suite: where oracle/_ref/enc_capture (the whole
reference encoder built from its own sources with MotionEstimate::motionEstimate intercepted, `make -C oracle capture`)
is present, it encodes 8 pictures of the synthetic 1920x1080 clip with --preset medium on this machine, and every call
it made (real AMVP predictors, candidate vectors, search bounds, per-CU QPs, reconstructed references, the lookahead's
Lowres planes) is replayed through x265_hip_me_batch and must give the reference's own (cost, mv)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "enc_capture")

CALL = np.dtype([
    ("fenc_plane", "<i4"), ("ref_plane", "<i4"), ("pu_x", "<i2"), ("pu_y", "<i2"), ("pu_w", "u1"), ("pu_h", "u1"),
    ("method", "u1"), ("subme", "u1"), ("mvmin_x", "<i2"), ("mvmin_y", "<i2"), ("mvmax_x", "<i2"), ("mvmax_y", "<i2"),
    ("qmvp_x", "<i2"), ("qmvp_y", "<i2"), ("merange", "<i2"), ("num_mvc", "u1"), ("qp", "u1"), ("mvc", "<i2", (24,)),
    ("lowres", "u1"), ("pad", "u1", (3,)), ("cost", "<i4"), ("mvx", "<i2"), ("mvy", "<i2")])


AMVP = np.dtype([("pred_plane", "<i4"), ("cu_x", "<i2"), ("cu_y", "<i2"), ("amvp", "<i2", (4,)), ("num_amvp", "u1"), ("pad", "u1", (3,))])


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/enc_capture not built (needs the reference sources)")
def test_reference_encoder_calls_at_1080p_replay_bit_exactly(tmp_path):
    from x265_amd import abi, lib, synth
    W, H, N = 1920, 1080, 8
    seq = synth.Sequence(W, H, seed=0x265)
    clip = tmp_path / "in.yuv"
    with open(clip, "wb") as f:
        for i in range(N):
            f.write(seq.frame(i).tobytes())
            f.write(bytes([128]) * (W * H // 2))
    r = subprocess.run([EXE, str(clip), str(W), str(H), str(N), str(tmp_path), "120000", "40000", "0", "8", "1"],
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    calls = np.fromfile(tmp_path / "calls.bin", dtype=CALL)
    assert len(calls) > 20000 and calls["lowres"].sum() > 5000 and (calls["lowres"] == 0).sum() > 10000, r.stderr[-500:]
    raw = np.memmap(tmp_path / "planes.bin", dtype=np.uint8, mode="r")
    planes, pos = [], 0
    while pos < len(raw):
        pid, stride, origin, nbytes = (int(v) for v in raw[pos:pos + 16].view("<i4"))
        assert pid == len(planes)
        planes.append((np.array(raw[pos + 16:pos + 16 + nbytes]), origin, stride))
        pos += 16 + nbytes
    assert len(planes) + 4 <= abi.MAX_PLANES  # every plane of the capture resident at once
    jobs = np.zeros(len(calls), abi.ME_JOB)
    for k in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "subme", "mvmin_x", "mvmin_y", "mvmax_x", "mvmax_y",
              "qmvp_x", "qmvp_y", "merange", "num_mvc", "qp"):
        jobs[k] = calls[k]
    jobs["method"] = calls["method"] | (calls["lowres"] * abi.LOWRES_REF)
    jobs["mvc"] = calls["mvc"]
    ctx = lib.Context(0)
    for q in np.unique(calls["qp"]):
        ctx.set_qp(int(q))
    for i, (buf, origin, stride) in enumerate(planes):
        ctx.plane_upload(i, buf, stride, origin)
    got = ctx.me_batch(jobs)
    ctx.close()
    bad = np.nonzero((got["cost"] != calls["cost"]) | (got["mvx"] != calls["mvx"]) | (got["mvy"] != calls["mvy"]))[0]
    assert len(bad) == 0, (len(bad), len(calls), calls[bad[0]], got[bad[0]])
    # the same full-resolution calls one step earlier: from the AMVP candidates TComDataCU::fillMvpCand returned
    # (amvp.bin) through x265_hip_inter_search_batch, which must pick the predictor the reference picked, get
    # xSetSearchRange's window and the same (cost, mv)
    amvp = np.fromfile(tmp_path / "amvp.bin", dtype=AMVP)
    assert len(amvp) == len(calls)
    sel = np.nonzero((calls["lowres"] == 0) & (amvp["num_amvp"] > 0))[0]
    ij = np.zeros(len(sel), abi.INTER_JOB)
    for k in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "method", "subme", "merange", "num_mvc", "qp", "mvc"):
        ij[k] = calls[k][sel]
    for k in ("pred_plane", "cu_x", "cu_y", "amvp", "num_amvp"):
        ij[k] = amvp[k][sel]
    ij["ref_lag"] = H  # one frame thread: m_refLagPixels is the picture height (TEncSearch.cpp:110)
    ctx2 = lib.Context(0)
    for q in np.unique(calls["qp"]):
        ctx2.set_qp(int(q))
    for i, (buf, origin, stride) in enumerate(planes):
        ctx2.plane_upload(i, buf, stride, origin)
    out = ctx2.inter_search_batch(ij, W, H)
    ctx2.close()
    c = calls[sel]
    assert len(sel) > 10000 and 500 < (out["mvp_idx"] == 1).sum()
    assert np.array_equal(out["mvp_x"], c["qmvp_x"]) and np.array_equal(out["mvp_y"], c["qmvp_y"])
    for k in ("mvmin_x", "mvmin_y", "mvmax_x", "mvmax_y", "cost", "mvx", "mvy"):
        assert np.array_equal(out[k], c[k]), k
    # and what predInterSearch left in the CU for every prediction unit whose iterations are all among the recorded calls
    # (pu.bin): bits, xCheckBestMVP, the bidirectional candidate and the choice through x265_hip_inter_decide_batch
    from test_oracle_golden import check_pu_decisions
    PU = np.dtype([("call", "<i4", (8,)), ("call_list", "u1", (8,)), ("call_ref", "u1", (8,)), ("n_calls", "u1"), ("is_p", "u1"),
                   ("part_size", "u1"), ("part_idx", "u1"), ("num_ref", "u1", (2,)), ("bipred_restriction", "u1"), ("complete", "u1"),
                   ("list_sel_bits", "u1", (3,)), ("merge_flag", "u1"), ("inter_dir", "u1"), ("pad", "u1", (3,)), ("lambda_sad", "<u4"),
                   ("mv", "<i2", (4,)), ("ref", "i1", (2,)), ("mvp_idx", "i1", (2,))])
    assert PU.itemsize == 80  # PuRec of oracle/enc_capture.cpp
    pus = np.fromfile(tmp_path / "pu.bin", dtype=PU)
    pus = pus[(pus["complete"] == 1) & (pus["n_calls"] == pus["num_ref"].sum(axis=1))]
    order = np.concatenate([p["call"][:p["n_calls"]] for p in pus])
    uj = np.zeros(len(order), abi.INTER_JOB)
    for k in ("fenc_plane", "ref_plane", "pu_x", "pu_y", "pu_w", "pu_h", "method", "subme", "merange", "num_mvc", "qp", "mvc"):
        uj[k] = calls[k][order]
    for k in ("pred_plane", "cu_x", "cu_y", "amvp", "num_amvp"):
        uj[k] = amvp[k][order]
    uj["ref_lag"] = H
    rec = np.zeros(len(pus), abi.INTER_PU)
    rec["first_job"] = np.concatenate([[0], np.cumsum(pus["n_calls"])[:-1]])
    rec["num_ref"], rec["list_sel_bits"], rec["lambda_sad"] = pus["num_ref"], pus["list_sel_bits"], pus["lambda_sad"]
    rec["no_bidir"] = (pus["is_p"] != 0) | (pus["bipred_restriction"] != 0)
    ctx3 = lib.Context(0)
    for q in np.unique(calls["qp"]):
        ctx3.set_qp(int(q))
    for i, (buf, origin, stride) in enumerate(planes):
        ctx3.plane_upload(i, buf, stride, origin)
    res, units = ctx3.inter_decide_batch(uj, rec, W, H)
    ctx3.close()
    for k in ("cost", "mvx", "mvy"):
        assert np.array_equal(res[k], calls[k][order]), k
    n_dir = check_pu_decisions(units, rec, pus)
    assert len(pus) > 10000 and n_dir[3] > 300 and n_dir[2] > 300, (len(pus), n_dir)
    sizes = {int(s): int((calls["pu_w"][calls["lowres"] == 0] == s).sum()) for s in (8, 16, 32, 64)}
    print(f"{len(calls)} calls of a 1080p medium-preset encode replayed ({int(calls['lowres'].sum())} lookahead; full-resolution "
          f"block sizes {sizes}; QPs {np.unique(calls['qp']).tolist()}), 0 mismatches")


ENC_HIP = os.path.join(ROOT, "oracle", "_ref", "enc_hip")


@pytest.mark.skipif(not os.path.exists(ENC_HIP), reason="oracle/_ref/enc_hip not built (needs the reference sources)")
@pytest.mark.parametrize("fade,mode", [(False, 1), (True, 1), (False, 3), (True, 3)])
def test_whole_frame_lookahead_inside_the_reference_encoder(tmp_path, fade, mode):
    """the reference encoder with every lookahead frame cost search done by x265_hip_lookahead_batch
    (X265_USE_HIP_LOOKAHEAD=1: one launch per (picture, reference) pair; 3: the pairs of a picture batched ahead of their
    use; in both every lowres motionEstimate call is answered from the launches' results and also run by the reference
    itself and compared): no call may differ and the bitstream must be the one the unmodified encoder writes; with
    `fade` the clip fades to black so that the lookahead searches weighted references"""
    import re
    from x265_amd import synth
    W, H, N = 640, 360, 12
    seq = synth.Sequence(W, H, seed=11, n_patches=3)
    clip = tmp_path / "in.yuv"
    with open(clip, "wb") as f:
        for i in range(N):
            y = seq.frame(i).astype(np.float64)
            if fade and i >= N // 3:
                y = y * max(0.15, 1.0 - 0.05 * (i - N // 3)) + 0.5
            f.write(np.clip(y, 0, 255).astype(np.uint8).tobytes())
            f.write(bytes([128]) * (W * H // 2))
    args = [ENC_HIP, str(clip), str(W), str(H), str(N), str(tmp_path), "0", "0", "0", "8", "1"]
    env = {k: v for k, v in os.environ.items() if not k.startswith("X265_USE_HIP")}
    plain = subprocess.run(args, capture_output=True, text=True, timeout=100, env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    hip = subprocess.run(args, capture_output=True, text=True, timeout=100, env=dict(env, X265_USE_HIP_LOOKAHEAD=str(mode)))
    assert hip.returncode == 0, hip.stderr[-2000:]
    h0, h1 = re.search(r"hash ([0-9a-f]+)", plain.stderr), re.search(r"hash ([0-9a-f]+)", hip.stderr)
    assert h0 and h1 and h0.group(1) == h1.group(1), (plain.stderr[-500:], hip.stderr[-500:])
    m = re.search(r"(\d+) lookahead frame searches asked for, (\d+) launches.* (\d+) lowres motionEstimate calls answered from them, (\d+) differed", hip.stderr)
    assert m, hip.stderr[-1000:]
    frames, launches, calls, differed = (int(v) for v in m.groups())
    assert frames >= 10 and calls == frames * ((W // 2 + 7) // 8) * ((H // 2 + 7) // 8) and differed == 0, hip.stderr[-1000:]
    assert launches == frames if mode == 1 else launches < frames, hip.stderr[-1000:]
