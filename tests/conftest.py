import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    if not os.path.exists(pyoracle.oracle_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return pyoracle.load_oracle()


@pytest.fixture(scope="session")
def small_seq():
    """three 416x240 padded planes of a synthetic sequence (plane ids 0, 1, 2)"""
    from x265_amd import synth
    seq = synth.Sequence(416, 240, seed=7, n_patches=3)
    return [synth.Plane(seq.frame(i)) for i in range(3)]


@pytest.fixture(scope="session")
def costtabs(oracle):
    return {q: oracle.bitcost_table(q) for q in range(20, 45)}


@pytest.fixture(scope="session")
def gpu_ctx(small_seq, costtabs):
    from x265_amd import lib
    ctx = lib.Context(0)
    for i, p in enumerate(small_seq):
        ctx.plane_upload(i, p.buf, p.stride, p.origin)
    for q, t in costtabs.items():
        ctx.costtab_upload(q, t)
    yield ctx
    ctx.close()


def random_me_jobs(rng, n, method, subme, W, H, dtype, search_range, parts, meranges=(4, 8, 16, 57), planes=(1, (0, 2))):
    jobs = np.zeros(n, dtype)
    for i in range(n):
        part = int(rng.integers(1, 25))
        w, h = parts[part]
        x = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        j = jobs[i]
        j["fenc_plane"] = planes[0]
        j["ref_plane"] = int(rng.choice(planes[1]))
        j["pu_x"], j["pu_y"], j["pu_w"], j["pu_h"] = x, y, w, h
        j["method"], j["subme"] = method, subme
        mer = int(rng.choice(meranges))
        j["merange"] = mer
        j["qp"] = int(rng.integers(20, 45))
        mode = rng.integers(4)
        if mode == 0:
            px, py = 0, 0
        elif mode == 1:
            px, py = int(rng.integers(-40, 40)), int(rng.integers(-40, 40))
        elif mode == 2:
            px, py = int(rng.integers(-400, 400)), int(rng.integers(-300, 300))
        else:
            px, py = 4 * int(rng.integers(-8, 8)), 4 * int(rng.integers(-8, 8))
        j["qmvp_x"], j["qmvp_y"] = px, py
        cx, cy = (x // 64) * 64, (y // 64) * 64
        lox, loy, hix, hiy = search_range(np.int64(px), np.int64(py), mer, cx, cy, W, H)
        j["mvmin_x"], j["mvmin_y"], j["mvmax_x"], j["mvmax_y"] = lox, loy, hix, hiy
        nm = int(rng.integers(0, 6))
        j["num_mvc"] = nm
        for c in range(nm):
            j["mvc"][2 * c] = px + int(rng.integers(-30, 30)) if rng.integers(3) else 0
            j["mvc"][2 * c + 1] = py + int(rng.integers(-30, 30)) if rng.integers(3) else 0
    return jobs
