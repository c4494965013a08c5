"""SAM text for the profile-training counters (test infrastructure): lines in the format of `samtools view`, made from
the simulator's OWN reads -- their positions are in their names (one contig of one segment, ploidy 1, no variants:
tests/histo_util.py), the insert size is recovered from the text -- plus lines that exercise every filter of
Profile::processRead (Profile.cpp:244-288, :294-388)."""
import numpy as np

import histo_util as H


def count_arrays(cls, kmer_count, bins, n_isize, n_indel_len=256):
    """A zeroed counts struct (simuscop_amd.SgTrainCounts; the oracle's has the same layout) and its numpy views."""
    import ctypes as C
    a = {"subs1": np.zeros((kmer_count, bins, 4), np.uint64), "subs2": np.zeros((kmer_count, bins, 4), np.uint64),
         "kmers": np.zeros((bins, kmer_count), np.uint64), "quality": np.zeros((16, bins, 94), np.uint64), "isize": np.zeros(n_isize, np.uint64),
         "ins_len": np.zeros(n_indel_len, np.uint64), "del_len": np.zeros(n_indel_len, np.uint64)}
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


# ---- a whole training input: reference with several contigs, reads, known variants, targets (Profile::train) ----
EXTRA_CONTIGS = (("chr2", 5200), ("chrX", 3000), ("chrS", 700), ("chrM", 1600), ("chr3", 2500))


def _crafted(rng, chrom, seq, pos, L, tlen, cigar=None, alt_at=None, mapq=b"60"):
    """One line on `chrom` at 1-based `pos`: the reference bases with two substitutions (alt_at: (offset, base) put in as well)."""
    s = bytearray(seq[pos - 1:pos - 1 + L].upper())
    for _ in range(2):
        if len(s):
            s[rng.randrange(len(s))] = rng.choice(b"ACGT")
    if alt_at is not None and 0 <= alt_at[0] < len(s):
        s[alt_at[0]] = alt_at[1]
    q = bytes(rng.choice(b"#-7<AFJ") for _ in range(len(s)))
    return b"\t".join([b"c%d" % rng.randrange(10 ** 6), b"0", chrom, b"%d" % pos, mapq, cigar or b"%dM" % len(s), b"=", b"0", b"%d" % tlen, bytes(s), q])


def training_inputs(wd, fa_one, sim_lines, L, exome=False, seed=5):
    """Writes wd/train.fa (chr1 of `fa_one` + EXTRA_CONTIGS), wd/known.vcf, wd/targets.bed (exome) and returns
    (fasta, vcf, bed or None, SAM text).  The lines: `sim_lines` (reads sampled on chr1) sorted by position, crafted reads on
    the other contigs (one of them shorter than a window: countGC's window shrinks for good, Profile.cpp:645-648; X and M:
    turned away, :532-535), a second visit to chr2 and to chr1 (new runs), reads that step backwards, reads over known SNVs
    that show the alternative allele, CIGAR insertions / deletions the VCF knows and does not know."""
    import random
    from simuscop_amd import synth
    rng = random.Random(seed)
    os_ = __import__("os")
    chr1 = open(fa_one, "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
    contigs = {b"chr1": chr1}
    for i, (name, n) in enumerate(EXTRA_CONTIGS):
        contigs[name.encode()] = synth.synth_contig(n, 900 + i, 0, n_runs=(name == "chr3")).tobytes()
    fa = os_.path.join(wd, "train.fa")
    with open(fa, "wb") as f:
        for name, s in contigs.items():
            f.write(b">" + name + b"\n" + b"".join(s[i:i + 60] + b"\n" for i in range(0, len(s), 60)))
    # known variants: SNVs every ~40 kbp of chr1 ("0/1" is filed as homozygous, "1/1" as heterozygous: vcfparser.cpp:81-86),
    # rows the depth / quality filters drop, an insertion and a deletion the crafted CIGARs below hit, rows on an absent contig
    vcf = [b"##fileformat=VCFv4.2", b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS"]
    snvs = []
    for k, pos in enumerate(range(20000, len(chr1) - 1000, 40000)):
        ref_b = chr1[pos - 1:pos].upper()
        alt = bytes([rng.choice([c for c in b"ACGT" if c != ref_b[0]])])
        gt = b"1/1" if k % 2 else b"0/1"
        vcf.append(b"\t".join([b"chr1", b"%d" % pos, b".", ref_b, alt, b"50", b"PASS", b"DP=30;AF=0.5", b"GT:AD", gt + b":10,20"]))
        snvs.append((pos, alt[0]))
    vcf.append(b"\t".join([b"chr1", b"777", b".", b"A", b"C", b"50", b"PASS", b"DP=5", b"GT", b"0/1"]))        # depth below 10
    vcf.append(b"\t".join([b"chr1", b"778", b".", b"A", b"C", b"10", b"PASS", b"DP=50", b"GT", b"0/1"]))       # quality below 20
    vcf.append(b"\t".join([b"chr9", b"100", b".", b"A", b"C", b"50", b"PASS", b"DP=50", b"GT", b"0/1"]))       # contig the FASTA lacks
    vcf.append(b"\t".join([b"chr2", b"1029", b".", b"A", b"ACG", b"50", b"PASS", b"DP=50", b"GT", b"0/1"]))    # insertion of 2 behind 1029
    vcf.append(b"\t".join([b"chr2", b"2019", b".", b"ACGT", b"A", b"50", b"PASS", b"DP=50", b"GT", b"1/1"]))   # deletion of 3 at 2020
    vcf.append(b"\t".join([b"chr2", b"3000", b".", b"A", b"T", b"50", b"PASS", b"DP=50", b"GT", b"0/1"]))
    vcf_path = os_.path.join(wd, "known.vcf")
    open(vcf_path, "wb").write(b"\n".join(vcf) + b"\n")
    bed_path = None
    if exome:
        rows = []
        for a in range(5000, len(chr1) - 5000, 9000):
            rows.append(b"chr1\t%d\t%d" % (a, a + rng.choice([120, 400, 1500, 2600])))
        rows.insert(3, b"chr1\t%d\t%d" % (rows and 40000, 40300))      # out of order
        rows.insert(9, b"chr1\t%d\t%d" % (14100, 14180))               # nested in the padding of another
        rows += [b"chr2\t1000\t1800", b"chr2\t1700\t2600", b"chrS\t10\t300", b"chr7\t1\t100"]
        bed_path = os_.path.join(wd, "targets.bed")
        open(bed_path, "wb").write(b"\n".join(rows) + b"\n")
    # ---- lines ----
    def pos_of(line):
        return int(line.split(b"\t", 4)[3])
    lines = sorted(sim_lines, key=pos_of)
    out = list(lines)
    for pos, alt in snvs[:12]:   # reads that show the alternative allele of a known SNV, forward and as mate 2
        out.append(_crafted(rng, b"chr1", chr1, pos - 40, L, 300, alt_at=(40, alt)))
        out.append(_crafted(rng, b"chr1", chr1, pos - 10, L, -300, alt_at=(10, alt)))
    c2, cx, cs, cm, c3 = (contigs[n.encode()] for n, _ in EXTRA_CONTIGS)
    for p in sorted(rng.randrange(1, 5200 - L) for _ in range(300)):
        out.append(_crafted(rng, b"chr2", c2, p, L, rng.choice([-300, 0, 280, 320])))
    out.append(_crafted(rng, b"chr2", c2, 1000, L, 0, cigar=b"30M2I%dM" % (L - 32)))     # known insertion (position 1029)
    out.append(_crafted(rng, b"chr2", c2, 1000, L, 0, cigar=b"30M3I%dM" % (L - 33)))     # same place, another length: counted
    out.append(_crafted(rng, b"chr2", c2, 2000, L, 0, cigar=b"20M3D%dM" % (L - 20)))     # known deletion (position 2020)
    out.append(_crafted(rng, b"chr2", c2, 2001, L, 0, cigar=b"20M3D%dM" % (L - 20)))     # one base on: counted
    for p in sorted(rng.randrange(1, 3000 - L) for _ in range(40)):
        out.append(_crafted(rng, b"chrX", cx, p, L, 300))
    for p in sorted(rng.randrange(1, 700 - L) for _ in range(30)):
        out.append(_crafted(rng, b"chrS", cs, p, L, 250))                                # a contig shorter than the window
    for p in sorted(rng.randrange(1, 1600 - L) for _ in range(20)):
        out.append(_crafted(rng, b"chrM", cm, p, L, 300))
    for p in sorted(rng.randrange(1, 2500 - L) for _ in range(200)):
        out.append(_crafted(rng, b"chr3", c3, p, L, -280))                               # N runs: windows without a GC content
    for p in sorted(rng.randrange(1, 5200 - L) for _ in range(150)):
        out.append(_crafted(rng, b"chr2", c2, p, L, 300))                                # chr2 again: a new run
    back = [rng.randrange(1, 200000) for _ in range(400)]                                # chr1 again, unsorted: reads step backwards
    for p in back:
        out.append(_crafted(rng, b"chr1", chr1, p, L, 300))
    out.append(_crafted(rng, b"chr1", chr1, len(chr1) - L // 2, L, 300))                 # hangs over the contig's end
    out.append(_crafted(rng, b"chr1", chr1, len(chr1) + 5, L, 300))                      # starts behind it
    out += filter_lines(L)
    return fa, vcf_path, bed_path, b"\n".join(out) + b"\n"
