"""Closed-form histogram families on the CPU: the analysis of tests/histo_util.py applied to the oracle's two RNG modes.

* `mt` mode writes the reference binary's bytes (tests/test_oracle_vs_reference.py), so this shows that the REFERENCE
  itself meets the closed-form expectations derived from the `.profile` -- the expectations are the right ones;
* `philox` mode writes the GPU's bytes (tests/test_gpu_parity.py); tests/test_gpu_histograms.py runs the same analysis on
  what the MI355X emitted, at ten times the sample.

Families G1 (substitutions by bin x context, per mate), G2 (qualities by bin x ref x called), G3 (read lengths), G4
(insert sizes), G5 (GC factor per window); bar |z| < 5 / Bonferroni-corrected exact binomial p > 1e-4."""
import json

import pytest

import cases
import histo_util as H


@pytest.mark.parametrize("profile,layout,mode,coverage,insert", [("xten", "PE", 0, 16, 350), ("gaiix", "SE", 1, 8, 200),
                                                                 ("hs2500", "PE", 1, 12, 200)])
def test_oracle_output_meets_the_profile_expectations(profile, layout, mode, coverage, insert, oracle_lib, tmp_path):
    cfg, fa = H.histogram_config(cases, str(tmp_path), profile, layout, coverage, insert)
    rc = oracle_lib.orc_simulate(cfg.encode(), mode, 1600000000 + coverage, 11, b"", 4 if mode else 1)
    assert rc == 0, oracle_lib.orc_last_error().decode()
    rep = H.analyse_run(oracle_lib, cases, profile, layout, insert, fa, cases.output_files(cfg), f"{profile} {layout} mode {mode}")
    print(json.dumps(rep))
    assert rep["mate1"]["reads_used"] > 50_000 and rep["gc_factor"]["n"] >= 1390


def test_the_analysis_detects_a_wrong_mate_table(oracle_lib, tmp_path):
    """Power of G1: mate-1 counts held against the mate-2 table fail (a wrong `mate2` flag in the sampler would pass every
    byte-parity test between two implementations sharing it)."""
    import numpy as np
    cfg, fa = H.histogram_config(cases, str(tmp_path), "xten", "PE", 16, 350)
    assert oracle_lib.orc_simulate(cfg.encode(), 1, 5, 5, b"", 4) == 0
    import os
    T = H.ProfileTables(oracle_lib, os.path.join(cases.TESTDATA, cases.PROFILES["xten"]), True, 350)
    ref = H.read_fasta_one(fa)
    fq = H.Fastq(cases.output_files(cfg)[0])
    rows = np.flatnonzero((fq.len == T.L) & (fq.pos + 1000 <= len(ref)))
    sub, qual, used, _ = H.sub_and_quality_counts(T, fq, H.forward_source(ref, fq, T.L), rows, False)
    H.check_sub_and_quality(T, sub, qual, False, "mate 1 against its own table")
    with pytest.raises(AssertionError):
        H.check_sub_and_quality(T, sub, qual, True, "mate 1 against the mate-2 table")
