"""CPU tests of the C-ABI library and host logic: it loads, exports every declared symbol, and the
host-side pieces that need no GPU behave (no compute call is made here)."""
import os
import re
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from x265_amd import lib
    text = open(os.path.join(ROOT, "include", "x265_hip.h")).read()
    declared = set(re.findall(r"\b(x265_hip_[a-z0-9_]+)\s*\(", text))
    declared = {d for d in declared if not d.endswith("_t")}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    handle = lib.load()
    for name in declared:
        assert getattr(handle, name) is not None


def test_struct_layouts_match_header():
    from x265_amd import abi, lib
    assert abi.ME_JOB.itemsize == 80 and abi.ME_RESULT.itemsize == 8 and abi.COST_REQ.itemsize == 20
    assert lib.CTU_JOB.itemsize == 28
    assert abi.ME_JOB.fields["mvc"][1] == 32 and abi.ME_JOB.fields["qp"][1] == 30


def test_partition_from_sizes():
    from x265_amd import abi, lib
    h = lib.load()
    for p, (w, hh) in enumerate(abi.PARTS):
        assert h.x265_hip_partition_from_sizes(w, hh) == p
    for bad in ((12, 12), (0, 0), (64, 8), (128, 128)):
        assert h.x265_hip_partition_from_sizes(*bad) == -1


def test_costtab_build_matches_reference_fixture():
    from x265_amd import lib
    g = np.load(os.path.join(ROOT, "tests", "golden", "bitcost.npz"))
    for q in range(70):
        t = lib.costtab_build(q)
        assert zlib.crc32(t.tobytes()) == int(g["crc"][q]), q
    with pytest.raises(lib.X265HipError):
        lib.costtab_build(70)


def test_no_gpu_means_error_not_fallback():
    """without a gfx950 device the library refuses to create a context (no CPU path behind it)"""
    from x265_amd import lib
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lib.X265HipError):
        lib.Context(0)
    with pytest.raises(lib.X265HipError):
        lib.setup_primitives(0)


def test_product_does_not_import_oracle():
    """nothing under x265_amd/ or include/ may reference the checker"""
    for base in ("x265_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in text.lower(), (dirpath, f)


def test_synth_search_range_matches_reference_rule():
    """xSetSearchRange / clipMv restated in x265_amd.synth (TEncSearch.cpp:2600-2617, TComDataCU.cpp:2502-2514)"""
    from x265_amd import synth
    # interior CTU: window is predictor +- merange
    assert tuple(int(v) for v in synth.search_range(np.int64(12), np.int64(-8), 57, 640, 320, 1920, 1080)) == (-54, -59, 60, 55)
    # top-left CTU: clipped at -(64 + 8) + 1 = -71
    lox, loy, hix, hiy = (int(v) for v in synth.search_range(np.int64(0), np.int64(0), 57, 0, 0, 1920, 1080))
    assert (lox, loy, hix, hiy) == (-57, -57, 57, 57)
    lox, loy, hix, hiy = (int(v) for v in synth.search_range(np.int64(-200), np.int64(0), 64, 0, 0, 1920, 1080))
    assert lox == -71 and hix == 14


def test_hot_kernels_use_no_scratch(tmp_path):
    """the ME and full-search kernels must keep their state in registers: a struct that slips into scratch memory
    costs several KB of HBM writes per motionEstimate call (seen once: 120 bytes per lane, caught by WRITE_SIZE).
    me_kernel is built for seven waves per SIMD (72 VGPRs) and may spill a handful of dwords, nothing more (a loop that
    cleared 2048 counters at its start once cost it 13 spilled registers and 5 % of every small call); me_big_kernel,
    at four waves per SIMD, spills nothing; the cost
    kernels spill nothing (cost_kernel at six waves spilled 14 dwords, 100 MB of scratch writes per 1080p launch,
    and was 8 % slower than at five waves without spills); refine_kernel at four waves (127 registers) spills nothing;
    me_rest_kernel (the same search inside a list loop, six waves per SIMD) may spill a dozen dwords; me_team_kernel
    (three waves per SIMD for the tile classes) may spill eight in the set-up of a chunk of calls (fifteen scratch
    instructions in all, none inside the rounds of a search)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "x265_amd", "csrc")
    allowed = {"me_kernel": 32, "me_rest_kernel": 64, "me_team_kernel": 32}
    seen = set()
    kernels = ("me_kernel", "me_rest_kernel", "me_big_kernel", "me_team_kernel", "cost_kernel", "dist_kernel", "predict_kernel", "fullsearch2_kernel",
               "refine_kernel")
    for name in ("me_kernel.hip", "me_team.hip", "fullsearch.hip", "refine.hip"):
        out = tmp_path / (name + ".s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                               os.path.join(src, name)], stderr=subprocess.DEVNULL)
        text = out.read_text()
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s*(\d+)", text):
            kernel = next((k for k in kernels if k in m.group(1)), None)
            if kernel:
                seen.add(kernel)
                assert int(m.group(2)) <= allowed.get(kernel, 0), (kernel, m.group(2))
    assert seen == set(kernels)


def test_python_struct_layouts_match_the_header(tmp_path):
    """the ctypes / numpy mirrors of the C structs must have the header's sizes: x265_hip_setup_primitives writes the
    whole table, a shorter Python struct would be overrun"""
    import ctypes
    import shutil
    import subprocess
    from x265_amd import abi, lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "x265_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(x265_hip_primitives), sizeof(x265_hip_me_job), sizeof(x265_hip_me_result), sizeof(x265_hip_cost_req), '
                   'sizeof(x265_hip_bidir_req), sizeof(x265_hip_pred_req), sizeof(x265_hip_ctu_job), sizeof(x265_hip_refine_req), sizeof(x265_hip_inter_job), sizeof(x265_hip_inter_result), sizeof(x265_hip_lookahead_job), sizeof(x265_hip_refine_section), sizeof(x265_hip_inter_pu), sizeof(x265_hip_inter_pu_result), sizeof(x265_hip_merge_cand), sizeof(x265_hip_merge_pu), sizeof(x265_hip_merge_result)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call([cc, "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(lib.Primitives), abi.ME_JOB.itemsize, abi.ME_RESULT.itemsize, abi.COST_REQ.itemsize,
            abi.BIDIR_REQ.itemsize, abi.PRED_REQ.itemsize, lib.CTU_JOB.itemsize, abi.REFINE_REQ.itemsize, abi.INTER_JOB.itemsize, abi.INTER_RESULT.itemsize,
            abi.LOOKAHEAD_JOB.itemsize, abi.REFINE_SECTION.itemsize, abi.INTER_PU.itemsize, abi.INTER_PU_RESULT.itemsize, abi.MERGE_CAND.itemsize,
            abi.MERGE_PU.itemsize, abi.MERGE_RESULT.itemsize]
    assert got == want
