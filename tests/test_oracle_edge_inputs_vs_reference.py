"""Degenerate configurations (tests/edge_inputs.py): oracle(mt) against the UNMODIFIED reference binary, here and now --
same files byte for byte where the reference runs, a refusal where it refuses (or never returns).  Needs
oracle/_ref/simuReads (this container only); skipped elsewhere."""
import hashlib
import os
import shutil
import subprocess

import pytest

import cases
import edge_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "simuReads")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libfakeclock.so")


def _md5s(d):
    return {f: hashlib.md5(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))} if os.path.isdir(d) else {}


@pytest.mark.skipif(not os.path.exists(REF), reason="reference binary not built (make -C oracle ref; needs /root/reference)")
@pytest.mark.parametrize("name", edge_inputs.NAMES)
def test_oracle_mt_equals_reference_on_degenerate_configuration(name, oracle_lib, tmp_path):
    cfg = edge_inputs.build(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "out")
    env = dict(os.environ, LD_PRELOAD=SHIM, FAKECLOCK_SEC=str(cases.FAKE_SEC), FAKECLOCK_NSEC=str(cases.FAKE_NSEC))
    try:
        ref_rc = subprocess.run([REF, cfg], env=env, capture_output=True, text=True, timeout=40).returncode   # (these cases take the reference seconds)
    except subprocess.TimeoutExpired:
        ref_rc = None   # (a copy-number gain on a haploid genome: Segment.cpp:188-197 spins for ever)
    want = _md5s(out)
    shutil.rmtree(out, ignore_errors=True)
    rc = oracle_lib.orc_simulate(cfg.encode(), 0, cases.FAKE_SEC, cases.FAKE_NSEC, b"", 1)
    if ref_rc != 0:
        assert rc != 0, "the oracle accepted what the reference refused"
        return
    assert rc == 0, oracle_lib.orc_last_error().decode()
    assert _md5s(out) == want
