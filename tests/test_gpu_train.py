"""Profile training, counting half on the MI355X: sg_train_count (train_parse_kernel + train_count_kernel,
simuscop_amd/csrc/sg_train.hip) through the C ABI against the CPU restatement of Profile::processRead's counters
(oracle/train_oracle.cpp), on SAM lines made from reads THE GPU sampled plus lines exercising every filter and the CIGAR
walk.  Bar: every counter bit-exact (integer work).  PARITY UNPINNED against the reference binary (no samtools / BAM here);
tests/test_train_counts.py closes the loop from the other side (the counts give back the profile tables)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cases
import histo_util as H
import simuscop_amd
import train_util as TU
from simuscop_amd import SgContig

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")


def _reference_on_device(eng, ctx, fasta_path, line_len=60):
    image = open(fasta_path, "rb").read()
    assert eng.sg_reference_begin(ctx, len(image)) == 0
    assert eng.sg_reference_chunk(ctx, 0, image, len(image)) == 0
    assert eng.sg_sync(ctx) == 0
    rows, keys, off = [], [], 0
    while off < len(image):                      # headers and bodies of a write_fasta file
        assert image[off:off + 1] == b">"
        e = image.index(b"\n", off)
        name = image[off + 1:e].split()[0]
        nxt = image.find(b">", e)
        body_end = len(image) if nxt < 0 else nxt
        body = image[e + 1:body_end]
        length = len(body) - body.count(b"\n")
        rows.append(SgContig(e + 1, length, line_len, line_len + 1))
        keys.append(name[3:] if name.startswith(b"chr") else name)
        off = body_end
    tab = (SgContig * len(rows))(*rows)
    assert eng.sg_reference_commit(ctx, tab, len(rows)) == 0, eng.sg_last_error(ctx)
    return keys


@pytest.mark.parametrize("profile,insert", [("xten", 350), ("hs2000", 200)])
def test_device_counts_equal_the_restatement(profile, insert, oracle_lib, tmp_path):
    oracle_lib.orc_train_count.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint32,
                                           C.POINTER(simuscop_amd.SgTrainCounts)]
    wd = str(tmp_path)
    cfg, fa = H.histogram_config(cases, wd, profile, "PE", 40, insert)
    out = os.path.join(wd, "gpu")
    r = subprocess.run([SIMU, cfg, "--seed", "77", "--out", out, "--quiet"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    T = H.ProfileTables(oracle_lib, os.path.join(cases.TESTDATA, cases.PROFILES[profile]), True, insert)
    ref = H.read_fasta_one(fa)
    f1, f2 = sorted(os.path.join(out, f) for f in os.listdir(out))
    fq1, fq2 = H.Fastq(f1), H.Fastq(f2)
    lines = TU.sam_from_pairs(ref, fq1, fq2, T.L, T.isize_min + len(T.isize_pmf) - 1, cuts=(H.mismatch_cut(T, False), H.mismatch_cut(T, True)))
    assert len(lines) > 200_000
    lines += TU.filter_lines(T.L)
    sam = b"\n".join(lines) + b"\n\n"           # (an empty line at the end: dropped)
    want, wa = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 1024)
    assert oracle_lib.orc_train_count(sam, len(sam), fa.encode(), T.bases.encode(), 3, T.bins, 1024, C.byref(want)) == 0
    eng = simuscop_amd.load_engine()
    ctx = C.c_void_p()
    assert eng.sg_create(C.byref(ctx), 0, 1) == 0
    try:
        keys = _reference_on_device(eng, ctx, fa)
        got, ga = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 1024)
        karr = (C.c_char_p * len(keys))(*keys)
        rc = eng.sg_train_count(ctx, sam, len(sam), karr, len(keys), T.bases.encode(), 3, T.bins, 1024, C.byref(got))
        assert rc == 0, eng.sg_last_error(ctx)
        for k in ("subs1", "subs2", "kmers", "quality", "isize"):
            assert np.array_equal(ga[k], wa[k]), k
        for k in ("lines", "reads_counted", "cigar_chars", "insert_events", "delete_events", "isize_overflow", "skipped_overhang"):
            assert getattr(got, k) == getattr(want, k), (k, getattr(got, k), getattr(want, k))
        assert list(got.ins_len) == list(want.ins_len) and list(got.del_len) == list(want.del_len)
        assert got.reads_counted > 200_000 and ga["subs2"].sum() > 0 and ga["quality"].sum() > 10_000_000
        # a line with fewer than eleven fields is an error, as in the reference (Profile.cpp:246-251)
        bad = b"r0\t0\tchr1\t100\t60\n"
        assert eng.sg_train_count(ctx, bad, len(bad), karr, len(keys), T.bases.encode(), 3, T.bins, 1024, C.byref(got)) != 0
        # other base orders / context lengths (the kernel's context index against the restatement's trie)
        for bases, kmer in ((b"ACGT", 2), (b"GTCA", 4)):
            kc = sum(4 ** m for m in range(1, kmer + 1))
            w2, wa2 = TU.count_arrays(simuscop_amd.SgTrainCounts, kc, 20, 1024)
            g2, ga2 = TU.count_arrays(simuscop_amd.SgTrainCounts, kc, 20, 1024)
            part = b"\n".join(lines[:40000]) + b"\n"
            assert oracle_lib.orc_train_count(part, len(part), fa.encode(), bases, kmer, 20, 1024, C.byref(w2)) == 0
            assert eng.sg_train_count(ctx, part, len(part), karr, len(keys), bases, kmer, 20, 1024, C.byref(g2)) == 0, eng.sg_last_error(ctx)
            for k in ("subs1", "subs2", "kmers", "quality", "isize"):
                assert np.array_equal(ga2[k], wa2[k]), (bases, kmer, k)
    finally:
        eng.sg_destroy(ctx)
