"""The two RNG modes of the oracle sample the same distributions.

oracle(mt) is byte-identical to the reference (test_oracle_vs_reference.py); oracle(philox) is
byte-identical to the GPU (test_gpu_parity.py).  Both modes run the same algorithm code and differ
only in where each 32-bit draw comes from, so their outputs must agree in distribution.  This test
compares histograms of two independent runs with a two-sample chi-square statistic.

Tolerance: for every histogram family the statistic's z-score (chi2 - dof) / sqrt(2 dof) must stay
below 5 (false-failure probability < 1e-6 per family under H0); the seeds are fixed, so the test is
deterministic.  Also checked against the analytic expectation: the mean sequencing-substitution rate
implied by the profile tables.
"""
import collections
import os

import numpy as np

import cases

from stats_util import Z_MAX, _chi2_z, _parse, _vec  # noqa: E402


def test_mt_and_philox_modes_agree_in_distribution(oracle_lib, tmp_path):
    wd_a, wd_b = tmp_path / "a", tmp_path / "b"
    cfg_a = cases.build_case("wgs_pe_xten", str(wd_a))
    cfg_b = cases.build_case("wgs_pe_xten", str(wd_b))
    # raise the coverage so that every histogram has plenty of counts (250 kbp x 40x = 33 k pairs)
    for cfg in (cfg_a, cfg_b):
        txt = open(cfg).read().replace("coverage = 5", "coverage = 40")
        open(cfg, "w").write(txt)
    assert oracle_lib.orc_simulate(cfg_a.encode(), 0, 1600000000, 7, b"", 1) == 0, oracle_lib.orc_last_error()
    assert oracle_lib.orc_simulate(cfg_b.encode(), 1, 424242, 99, b"", 4) == 0, oracle_lib.orc_last_error()
    fa, fb = cases.output_files(cfg_a), cases.output_files(cfg_b)
    for pa, pb in zip(fa, fb):
        na, qa, la, sa = _parse(pa, 250000)
        nb, qb, lb, sb = _parse(pb, 250000)
        # the plan (GC weights) is drawn from different normal generators in the two modes, the total is not
        assert abs(na - nb) <= 0.002 * na
        for name, (x, y) in {"quality by cycle": _vec(qa, qb), "read length": _vec(la, lb),
                             "start position": _vec(sa, sb)}.items():
            z, dof = _chi2_z(x, y)
            assert z < Z_MAX, f"{os.path.basename(pa)} {name}: z={z:.2f} (dof {dof})"


def test_substitution_rate_matches_profile_expectation(oracle_lib, tmp_path):
    """Analytic pin: expected per-base mismatch rate of mate 1 from the substitution tables vs the
    rate observed in indel-free reads of a philox-mode run (|z| < 5)."""
    cfg = cases.build_case("wgs_pe_xten", str(tmp_path))
    txt = open(cfg).read().replace("coverage = 5", "coverage = 40")
    open(cfg, "w").write(txt)
    assert oracle_lib.orc_simulate(cfg.encode(), 1, 7, 7, b"", 4) == 0
    prof = os.path.join(cases.TESTDATA, cases.PROFILES["xten"])
    h = oracle_lib.orc_profile_load(prof.encode(), 1, 350)
    bins, kc, L = (oracle_lib.orc_profile_info(h, i) for i in (2, 4, 3))
    cdf = np.ctypeslib.as_array(oracle_lib.orc_profile_array(h, 2), shape=(kc, bins, 4)).copy()
    oracle_lib.orc_profile_free(h)
    pdf = np.diff(np.concatenate([np.zeros((kc, bins, 1)), cdf], axis=2), axis=2)
    # contexts with three real bases: ids 20..83, the context's last base is id % 4 (bases order ACTG)
    ids = np.arange(20, 84)
    p_same = pdf[ids, :, ids % 4]                     # [64, bins]
    # genome: read the haplotype to weight contexts by their frequency
    ref = "".join(l.strip() for l in open(os.path.join(str(tmp_path), "ref.fa")) if not l.startswith(">")).upper()
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    arr = np.array([code.get(c, -1) for c in ref[20000:120000]])
    ok = (arr[:-2] >= 0) & (arr[1:-1] >= 0) & (arr[2:] >= 0)
    ctx = (arr[:-2] * 16 + arr[1:-1] * 4 + arr[2:])[ok]
    w = np.bincount(ctx, minlength=64) / ok.sum()
    exp_mismatch = float((w[:, None] * (1 - p_same)).sum(axis=0).mean())
    # observed: mate-1 reads of nominal length, compared with the reference at their start position
    mism = tot = 0
    with open(cases.output_files(cfg)[0], "rb") as f:
        while True:
            hd = f.readline()
            if not hd:
                break
            s = f.readline().strip().decode()
            f.readline(); f.readline()
            if len(s) != L:
                continue
            pos = int(hd.split(b"#")[2])
            r = ref[pos:pos + L]
            if len(r) != L or "N" in r:
                continue
            d = sum(1 for a, b in zip(s[2:], r[2:]) if a != b)
            if d > 20:      # a read carrying compensating indels; not an indel-free alignment
                continue
            mism += d
            tot += L - 2
    obs = mism / tot
    z = (obs - exp_mismatch) / np.sqrt(exp_mismatch * (1 - exp_mismatch) / tot)
    assert abs(z) < 5, (obs, exp_mismatch, z)
