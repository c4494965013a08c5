"""CPU-only checks of the product's host pieces (no GPU, no compute calls through the ABI):

* the C-ABI library loads and exports every symbol include/simuscop_amd.h declares;
* sg_cdf_count_le -- the exact integer form of the reference's `r <= cdf[k]` test
  (lib/mydefine/MyDefine.cpp:176-184 with lib/threadpool/ThreadPool.cpp:203-207) -- agrees with the
  fp64 expression for every probed 32-bit draw, including both edges of each threshold;
* the engine refuses to run without a HIP device instead of falling back to anything.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import simuscop_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZERO = 2.2204e-16


def _r(x):
    """threadPool->randomDouble(ZERO_FINAL, 1) for the 32-bit generator output x, in fp64."""
    return ZERO + (1.0 - ZERO) * (np.float64(x) / 4294967296.0)


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "simuscop_amd.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"sg_ctx"}
    assert declared == set(simuscop_amd.ENGINE_SYMBOLS), declared ^ set(simuscop_amd.ENGINE_SYMBOLS)
    lib = simuscop_amd.load_engine()
    for sym in declared:
        assert hasattr(lib, sym), sym
    host = simuscop_amd.load_host()
    for sym in ("simu_run", "simu_open", "simu_close", "simu_engine", "simu_weighted_length", "simu_set_reads",
                "simu_prepare_batch", "simu_default_options"):
        assert hasattr(host, sym), sym


def test_count_le_matches_fp64_predicate():
    lib = simuscop_amd.load_engine()
    rng = np.random.default_rng(5)
    cdfs = list(rng.random(200)) + [0.0, 1e-300, ZERO, ZERO * (1 + 1e-9), 0.5, 1.0 - 1e-10, 1.0 - 2.4e-10, 1.0,
                                    1.0000000000000002, 0.9999999999999998, 2.0, 3.0e-10, 2.3283064365386963e-10]
    for c in cdfs:
        cnt = int(lib.sg_cdf_count_le(ctypes.c_double(c)))
        assert 0 <= cnt <= 2 ** 32
        # the predicate is true exactly for x < cnt
        if cnt > 0:
            assert _r(cnt - 1) <= c
        if cnt < 2 ** 32:
            assert not (_r(cnt) <= c)


def test_count_le_on_real_profile_rows(oracle_lib):
    """Every quality / substitution CDF value of the shipped XTen profile: edges of the threshold."""
    lib = simuscop_amd.load_engine()
    path = os.path.join(ROOT, "tests", "golden", "testData", "Illumina_HiSeqXTen.profile")
    h = oracle_lib.orc_profile_load(path.encode(), 1, 350)
    assert h
    try:
        bins = oracle_lib.orc_profile_info(h, 2)
        nq = oracle_lib.orc_profile_info(h, 5)
        q = np.ctypeslib.as_array(oracle_lib.orc_profile_array(h, 4), shape=(16 * bins * nq,))
        vals = np.unique(q)
        step = max(1, len(vals) // 400)
        for c in vals[::step]:
            cnt = int(lib.sg_cdf_count_le(ctypes.c_double(float(c))))
            if cnt > 0:
                assert _r(cnt - 1) <= c
            if cnt < 2 ** 32:
                assert not (_r(cnt) <= c)
    finally:
        oracle_lib.orc_profile_free(h)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = simuscop_amd.load_engine()
    ctx = ctypes.c_void_p()
    rc = lib.sg_create(ctypes.byref(ctx), 0, 1)
    assert rc != 0 and not ctx.value
    msg = lib.sg_last_error(None).decode()
    assert "no HIP device" in msg or "hip" in msg.lower()
    # ... and so does the whole host driver: no CPU path hides behind it
    import cases
    import tempfile
    with tempfile.TemporaryDirectory() as wd:
        cfg = cases.build_case("tiny_contigs_pe", wd)
        with pytest.raises(simuscop_amd.SimuError):
            simuscop_amd.run_config(cfg, quiet=1)


def test_config_errors_match_reference_messages(tmp_path):
    """Unknown key / missing profile are fatal with the reference's texts (Config.cpp:90-94,102-105)."""
    cfg = tmp_path / "c.txt"
    cfg.write_text("ref = x.fa\nprofile = p\nname = a\noutput = o\ncoverage = 1\nbogus = 1\n")
    with pytest.raises(simuscop_amd.SimuError, match="unrecognized item"):
        simuscop_amd.run_config(str(cfg), quiet=1)
    cfg.write_text("ref = x.fa\nname = a\noutput = o\ncoverage = 1\n")
    with pytest.raises(simuscop_amd.SimuError, match="sequencing profile must be specified"):
        simuscop_amd.run_config(str(cfg), quiet=1)
    cfg.write_text("ref = x.fa\nprofile = p\nname = a, b\noutput = o\ncoverage = 1\n")
    with pytest.raises(simuscop_amd.SimuError, match="abundance file not specified"):
        simuscop_amd.run_config(str(cfg), quiet=1)


def test_header_is_plain_c(tmp_path):
    """include/simuscop_amd.h is the drop-in boundary: it must compile as C99 (no C++ or torch types) and as
    C++11 (the reference's language level, CMakeLists.txt:4)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "simuscop_amd.h"\nint main(void) { return sg_create(0, 0, 0) == SG_OK; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)])
