"""CPU-only: the per-base sampling tables are exact rearrangements of the reference's inverse-CDF scan.

The reference picks `first k with r <= cdf[k]` from one 32-bit draw (lib/mydefine/MyDefine.cpp:176-184 on
lib/threadpool/ThreadPool.cpp:203-207), which gives outcome k exactly count_le(cdf[k]) - count_le(cdf[k-1]) of the
2^32 draws.  The engine (and the oracle's philox mode) sample substitutions from identity-first rows and
qualities from alias columns; both must give every outcome exactly that many draws.  Checked here for every row
of the four shipped profiles, product tables (through the C ABI, host code only) against the oracle's and against
the counts themselves."""
import ctypes as C
import os

import numpy as np
import pytest

import simuscop_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = ["Illumina_HiSeqXTen.profile", "Illumina_HiSeq2500.profile", "Illumina_HiSeq2000.profile",
            "Illumina_GenomeAnalyzerIIx.profile"]


def _counts(lib, cdf):
    """Draws per outcome of one CDF row, straight from the definition."""
    cnt = [int(lib.sg_cdf_count_le(C.c_double(float(c)))) for c in cdf[:-1]]
    for i in range(1, len(cnt)):
        cnt[i] = max(cnt[i], cnt[i - 1])
    edges = [0] + cnt + [1 << 32]
    return [edges[i + 1] - edges[i] for i in range(len(cdf))]


@pytest.mark.parametrize("prof", PROFILES)
def test_tables_keep_every_count(prof, oracle_lib):
    lib = simuscop_amd.load_engine()
    oracle_lib.orc_profile_sub_row.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]
    oracle_lib.orc_profile_alias_row.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8),
                                                 C.POINTER(C.c_uint8)]
    h = oracle_lib.orc_profile_load(os.path.join(ROOT, "tests", "golden", "testData", prof).encode(), 1, 350)
    assert h
    try:
        bins, kc, nq = (oracle_lib.orc_profile_info(h, i) for i in (2, 4, 5))
        lgw = oracle_lib.orc_profile_info(h, 11)
        W, Ccol = 1 << lgw, 1 << (32 - lgw)
        rng = np.random.default_rng(11)
        # ---- substitution rows (both mates): a sample of contexts x every bin ----
        kbuf = C.create_string_buffer(8)
        for which, mate2 in ((2, 0), (3, 1)):
            sub = np.ctypeslib.as_array(oracle_lib.orc_profile_array(h, which), shape=(kc, bins, 4))
            for ki in sorted(set(rng.integers(0, kc, 24).tolist() + [0, 3, 4, 19, 20, kc - 1])):
                oracle_lib.orc_profile_kmer(h, ki, kbuf)
                cd = "ACTG".index(kbuf.value.decode()[-1])
                for b in range(bins):
                    row = np.ascontiguousarray(sub[ki, b])
                    n = _counts(lib, row)
                    cum, order = (C.c_uint64 * 3)(), (C.c_uint8 * 4)()
                    assert lib.sg_sub_row_identity_first(row.ctypes.data_as(C.POINTER(C.c_double)), cd, cum, order) == 0
                    ocum, oorder = (C.c_uint64 * 3)(), (C.c_uint8 * 4)()
                    oracle_lib.orc_profile_sub_row(h, mate2, ki, b, ocum, oorder)
                    assert list(cum) == list(ocum) and list(order) == list(oorder)
                    assert order[0] == cd and sorted(order) == [0, 1, 2, 3] and list(order[1:]) == sorted(order[1:])
                    edges = [0] + list(cum) + [1 << 32]
                    for j in range(4):
                        assert edges[j + 1] - edges[j] == n[order[j]], (prof, ki, b, j)
        # ---- quality rows: every (reference, called) pair x every bin ----
        qual = np.ctypeslib.as_array(oracle_lib.orc_profile_array(h, 4), shape=(16, bins, nq))
        most = 0
        for bp in range(16):
            for b in range(bins):
                row = np.ascontiguousarray(qual[bp, b])
                ptr = row.ctypes.data_as(C.POINTER(C.c_double))
                most = max(most, lib.sg_row_symbols(ptr, nq))
                thr, lo, hi = (C.c_uint32 * W)(), (C.c_uint8 * W)(), (C.c_uint8 * W)()
                assert lib.sg_alias_row(ptr, nq, lgw, thr, lo, hi) == 0
                othr, olo, ohi = (C.c_uint32 * W)(), (C.c_uint8 * W)(), (C.c_uint8 * W)()
                oracle_lib.orc_profile_alias_row(h, bp, b, othr, olo, ohi)
                assert list(thr) == list(othr) and list(lo) == list(olo) and list(hi) == list(ohi)
                got = [0] * nq
                for c in range(W):
                    assert thr[c] < Ccol and (thr[c] != 0 or lo[c] == hi[c])
                    got[lo[c]] += thr[c]
                    got[hi[c]] += Ccol - thr[c]
                if bp % 5 == 0 or b % 7 == 0:   # the exact counts are 94 bisections a row: diagonal rows and every 7th bin
                    assert got == _counts(lib, row), (prof, bp, b)
                else:
                    assert sum(got) == 1 << 32
        assert W >= most and (W == 4 or W // 2 < most)
    finally:
        oracle_lib.orc_profile_free(h)


def test_alias_row_edge_shapes():
    """One symbol, two symbols with a one-draw sliver, more symbols than columns (refused)."""
    lib = simuscop_amd.load_engine()
    def run(cdf, lgw):
        arr = (C.c_double * len(cdf))(*cdf)
        W = 1 << lgw
        thr, lo, hi = (C.c_uint32 * W)(), (C.c_uint8 * W)(), (C.c_uint8 * W)()
        rc = lib.sg_alias_row(arr, len(cdf), lgw, thr, lo, hi)
        return rc, list(thr), list(lo), list(hi)
    rc, thr, lo, hi = run([0.0, 0.0, 1.0, 1.0], 2)
    assert rc == 0 and thr == [0] * 4 and lo == hi == [2] * 4
    rc, thr, lo, hi = run([2.4e-10, 1.0, 1.0], 2)      # outcome 0 owns two draws (x = 0, 1)
    got = [0, 0, 0]
    for c in range(4):
        got[lo[c]] += thr[c]
        got[hi[c]] += (1 << 30) - thr[c]
    assert rc == 0 and got == _counts(lib, [2.4e-10, 1.0, 1.0]) == [2, (1 << 32) - 2, 0]
    rc, *_ = run([0.1, 0.2, 0.3, 0.4, 0.5, 1.0], 2)   # six symbols do not fit four columns
    assert rc != 0


@pytest.mark.parametrize("prof", ["Illumina_HiSeqXTen.profile", "Illumina_GenomeAnalyzerIIx.profile"])
def test_indel_candidates_by_skipping_ahead(prof, oracle_lib):
    """Sequencing indels in philox mode: the distance to a read's next indel candidate is drawn from a table of
    P(no candidate in k positions) instead of testing every position (Profile.cpp:1560-1570 tests them one by one).
    The table is the exact integer recurrence of DESIGN.md section 4, it stays within k * 2^-64 of (1 - p)^k, and
    reads sampled with it carry indels at the profile's rates, uniformly over the template."""
    from fractions import Fraction
    path = os.path.join(ROOT, "tests", "golden", "testData", prof).encode()
    h = oracle_lib.orc_profile_load(path, 1, 350)
    try:
        L = oracle_lib.orc_profile_info(h, 3)
        ab = (C.c_uint64 * 2)()
        gaps = (C.c_uint64 * L)()
        assert oracle_lib.orc_profile_indel_gaps(h, ab, gaps, L) == L
        A, B = int(ab[0]), int(ab[1])
        ins, dele = oracle_lib.orc_profile_rate(h, 0), oracle_lib.orc_profile_rate(h, 1)
        # A / 2^64 = P(insertion test passes), (B - A) / 2^64 = P(it fails and the deletion test passes): the reference's
        # two 32-bit tests `p <= insertRate`, `p < delRate / (1 - insertRate)`
        c_i = A >> 32
        assert A == c_i << 32 and abs(c_i / 2.0 ** 32 - ins) < 2.0 ** -31
        c_d = (B - A) // ((1 << 32) - c_i)
        assert B == A + ((1 << 32) - c_i) * c_d and abs(c_d / 2.0 ** 32 - dele / (1 - ins)) < 2.0 ** -31
        q = (1 << 64) - B
        want = q
        for k in range(1, L + 1):
            assert int(gaps[k - 1]) == want, k                        # the recurrence, exactly
            exact = Fraction(q, 1 << 64) ** k
            assert abs(Fraction(want, 1 << 64) - exact) < Fraction(k, 1 << 64)
            want = (want * q) >> 64
        # sampled reads: template of L 'A's; an insertion lengthens the read, a deletion shortens it
        n_reads, longer, shorter = 60000, 0, 0
        out_b, out_q = C.create_string_buffer(4 * L + 64), C.create_string_buffer(4 * L + 64)
        for slot in range(n_reads):
            np_ = oracle_lib.orc_predict_philox(h, b"A" * L, L, 1, 12345, 3, slot, out_b, out_q)
            longer += np_ > L
            shorter += np_ < L
        # per read: P(at least one insertion with length > 0) etc.; compare with the per-position rates (events of
        # length 0 and reads with both kinds blur it by a few percent of the rate; the bound is 5 sigma + 5 %)
        for seen, rate in ((longer, A / 2.0 ** 64), (shorter, (B - A) / 2.0 ** 64)):
            expect = n_reads * (1 - (1 - rate) ** L)
            assert abs(seen - expect) < 5 * expect ** 0.5 + 0.08 * expect, (seen, expect)
    finally:
        oracle_lib.orc_profile_free(h)
