"""Profile training, counting half on the CPU: the restatement of Profile::processRead's counters
(oracle/train_oracle.cpp) applied to SAM lines made from reads the oracle itself sampled.  PARITY UNPINNED -- no samtools,
no BAM in this image, so the restatement is not run against the reference binary -- but the loop closes: normalised, the
counts must give back the profile tables the reads were drawn from (the inverse of Profile::predict)."""
import ctypes as C
import os

import numpy as np

import cases
import histo_util as H
import simuscop_amd
import train_util as TU


def _declare(lib):
    lib.orc_train_count.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                    C.POINTER(simuscop_amd.SgTrainCounts)]
    return lib


def make_sam(oracle_lib, wd, profile="xten", coverage=12, insert=350, crafted=True):
    os.makedirs(wd, exist_ok=True)
    cfg, fa = H.histogram_config(cases, wd, profile, "PE", coverage, insert)
    assert oracle_lib.orc_simulate(cfg.encode(), 1, 99, 7, b"", 4) == 0, oracle_lib.orc_last_error()
    T = H.ProfileTables(oracle_lib, os.path.join(cases.TESTDATA, cases.PROFILES[profile]), True, insert)
    ref = H.read_fasta_one(fa)
    f1, f2 = cases.output_files(cfg)
    fq1, fq2 = H.Fastq(f1), H.Fastq(f2)
    lines = TU.sam_from_pairs(ref, fq1, fq2, T.L, T.isize_min + len(T.isize_pmf) - 1, cuts=(H.mismatch_cut(T, False), H.mismatch_cut(T, True)))
    if crafted:
        lines += TU.filter_lines(T.L)
    return b"\n".join(lines) + b"\n", fa, T


def test_counts_give_back_the_profile(oracle_lib, tmp_path):
    _declare(oracle_lib)
    sam, fa, T = make_sam(oracle_lib, str(tmp_path))
    st, a = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
    assert oracle_lib.orc_train_count(sam, len(sam), fa.encode(), T.bases.encode(), 3, T.bins, 2048, 256, C.byref(st)) == 0
    n_lines = sam.count(b"\n")
    assert st.lines == n_lines and 0.8 * n_lines < st.reads_counted < n_lines
    assert st.insert_events > 0 and st.delete_events > 0 and st.isize_overflow == 1 and st.skipped_overhang == 1
    assert a["ins_len"][2] >= 1 and a["del_len"][3] >= 1            # the crafted 20M2I..3D10M line
    # ... and without the crafted lines (their made-up bases are no sample of the profile):
    sam, fa, T = make_sam(oracle_lib, str(tmp_path / "plain"), crafted=False)
    st, a = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
    assert oracle_lib.orc_train_count(sam, len(sam), fa.encode(), T.bases.encode(), 3, T.bins, 2048, 256, C.byref(st)) == 0
    # the counters against the tables the reads were sampled from (G1 / G2 / G4 of tests/histo_util.py, read backwards)
    for mate, key in ((0, "subs1"), (1, "subs2")):
        z, dof, pmin, cells, worst = H.categorical_report(a[key].astype(np.float64), T.sub[mate])
        assert abs(z) < H.Z_MAX and pmin > H.P_MIN, (key, z, pmin, worst)
    z, dof, pmin, cells, worst = H.categorical_report(a["quality"].astype(np.float64), T.qual)
    assert abs(z) < H.Z_MAX, ("quality", z)
    isz = a["isize"].astype(np.float64)
    k = len(T.isize_pmf)
    z, dof, pmin, cells, worst = H.categorical_report(isz[T.isize_min:T.isize_min + k][None, :], T.isize_pmf[None, :])
    assert abs(z) < H.Z_MAX, ("isize", z)
    assert isz[:T.isize_min].sum() == 0 and isz[T.isize_min + k:].sum() == 0
    # kmersDist is the row sum of the two substitution tables
    assert np.array_equal(a["kmers"], (a["subs1"] + a["subs2"]).sum(axis=2).T)
