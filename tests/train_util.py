"""SAM text for the profile-training counters (test infrastructure): lines in the format of `samtools view`, made from
the simulator's OWN reads -- their positions are in their names (one contig of one segment, ploidy 1, no variants:
tests/histo_util.py), the insert size is recovered from the text -- plus lines that exercise every filter of
Profile::processRead (Profile.cpp:244-288, :294-388)."""
import numpy as np

import histo_util as H


def count_arrays(cls, kmer_count, bins, n_isize):
    """A zeroed counts struct (simuscop_amd.SgTrainCounts; the oracle's has the same layout) and its numpy views."""
    import ctypes as C
    a = {"subs1": np.zeros((kmer_count, bins, 4), np.uint64), "subs2": np.zeros((kmer_count, bins, 4), np.uint64),
         "kmers": np.zeros((bins, kmer_count), np.uint64), "quality": np.zeros((16, bins, 94), np.uint64), "isize": np.zeros(n_isize, np.uint64)}
    st = cls()
    for k, v in a.items():
        setattr(st, k, v.ctypes.data_as(C.POINTER(C.c_uint64)))
    return st, a


def sam_from_pairs(ref, fq1, fq2, L, isz_max, chrom=b"chr1", limit=None, cuts=(10 ** 9, 10 ** 9)):
    """Two lines per pair whose mates both have the nominal length and whose insert size could be placed: mate 1 forward
    (flag 99, TLEN +isz), mate 2 as aligned to the forward strand (flag 147, reverse-complemented text, reversed qualities,
    TLEN -isz).  Pairs with a mate of another length become lines with an insertion or deletion in their CIGAR (they feed
    the indel length counters and are then rejected, Profile.cpp:386-388).  `cuts` = most mismatches of mate 1 / mate 2
    an "nM" line may show (a read of nominal length can still hold an insertion and a deletion of one size: an aligner
    would not report that as nM either)."""
    comp = bytes(H._COMP)
    isz = H.recover_insert_sizes(ref, fq1, fq2, L, isz_max)
    rows = np.flatnonzero((isz >= 0) & (fq1.len == L))
    for c0 in range(0, len(rows), 100000):
        r = rows[c0:c0 + 100000]
        d1 = (fq1.matrix(fq1.seq_off, r, L) != H.forward_source(ref, fq1, L)(r)).sum(axis=1)
        d2 = (fq2.matrix(fq2.seq_off, r, L) != H.reverse_source(ref, fq1.pos + isz, L)(r)).sum(axis=1)
        isz[r[(d1 > cuts[0]) | (d2 > cuts[1])]] = -1
    b1, b2 = fq1.buf.tobytes(), fq2.buf.tobytes()
    out = []
    n = fq1.n if limit is None else min(limit, fq1.n)
    for i in range(n):
        l1, l2 = int(fq1.len[i]), int(fq2.len[i])
        s1 = b1[fq1.seq_off[i]:fq1.seq_off[i] + l1]
        q1 = b1[fq1.qual_off[i]:fq1.qual_off[i] + l1]
        s2 = b2[fq2.seq_off[i]:fq2.seq_off[i] + l2]
        q2 = b2[fq2.qual_off[i]:fq2.qual_off[i] + l2]
        pos = int(fq1.pos[i])
        name = b"r%d" % i
        if isz[i] >= 0 and l1 == L:
            z = int(isz[i])
            p2 = pos + z - L
            out.append(b"\t".join([name, b"99", chrom, b"%d" % (pos + 1), b"60", b"%dM" % L, b"=", b"%d" % (p2 + 1), b"%d" % z, s1, q1]))
            out.append(b"\t".join([name, b"147", chrom, b"%d" % (p2 + 1), b"60", b"%dM" % L, b"=", b"%d" % (pos + 1), b"-%d" % z,
                                   s2.translate(comp)[::-1], q2[::-1]]))
        elif l1 != L and l1 >= 60:
            d = l1 - L
            cigar = b"30M%dI%dM" % (d, l1 - 30 - d) if d > 0 else b"30M%dD%dM" % (-d, l1 - 30)
            out.append(b"\t".join([name, b"99", chrom, b"%d" % (pos + 1), b"60", cigar, b"=", b"0", b"0", s1, q1]))
    return out


def filter_lines(L, chrom=b"chr1"):
    """Lines every one of which must be rejected, or counted in a particular way."""
    s, q = b"ACGT" * (L // 4) + b"A" * (L % 4), b"F" * L
    mk = lambda **kw: b"\t".join([kw.get("name", b"x"), b"0", kw.get("chr", chrom), kw.get("pos", b"1000"), kw.get("mapq", b"60"),  # noqa: E731
                                  kw.get("cigar", b"%dM" % L), b"=", b"0", kw.get("tlen", b"0"), kw.get("seq", s), kw.get("qual", q)])
    return [mk(pos=b"0"), mk(mapq=b"14"), mk(chr=b"chrUn_unknown"), mk(seq=b"*", qual=b"*"), mk(cigar=b"10H%dM" % (L - 10)),
            mk(cigar=b"5S%dM" % (L - 5)), mk(cigar=b"%d" % L), mk(cigar=b"*"), mk(cigar=b"20M2I%dM3D10M" % (L - 32)),
            mk(seq=s.lower()), mk(seq=s[:-1] + b"N"), mk(qual=q[:-1]),               # counted: lower case / N never match `bases`; short quality string
            mk(chr=b"scaffold_chrom" + chrom[3:]), mk(chr=chrom[3:]),                # abbrOfChr: what follows "chrom" / no prefix at all
            mk(tlen=b"-350"), mk(tlen=b"999999"), mk(pos=b"1399990"),                # mate 2 without a mate-1 line; insert size past the row; overhang
            mk(name=b"extra", qual=q + b"\tNM:i:0\tMD:Z:%d" % L)]                   # optional fields after the eleventh
