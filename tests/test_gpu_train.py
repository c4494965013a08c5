"""Profile training on the MI355X: sg_train_* (simuscop_amd/csrc/sg_train.hip) through the C ABI and the
`seqToProfile` command line against the CPU restatement of Profile::train (oracle/train_oracle.cpp), on SAM lines made from
reads THE GPU sampled plus crafted lines (every filter, the CIGAR walk, known variants, several contigs, reads that step
backwards, exome targets).  Bar: every counter and every (GC, read count) pair bit-exact, the written profile byte for byte
(integer work, then the same fp64 operations in the same order).  The counting is PARITY UNPINNED against the reference
binary (no samtools / BAM here); what a reference run can pin is pinned: the unmodified binary loads the file `seqToProfile`
wrote and samples from it exactly as oracle(mt) does (below, and tests/test_train_profile_cpu.py on the CPU)."""
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


ARRAYS = ("subs1", "subs2", "kmers", "quality", "isize", "ins_len", "del_len")
SCALARS = ("lines", "reads_counted", "cigar_chars", "insert_events", "delete_events", "isize_overflow", "indel_len_overflow", "skipped_overhang",
           "gc_rejected", "gc_windows", "capped")


def _same_counts(got, ga, want, wa):
    for k in ARRAYS:
        assert np.array_equal(ga[k], wa[k]), (k, int((ga[k] != wa[k]).sum()), ga[k].sum(), wa[k].sum())
    for k in SCALARS:
        assert getattr(got, k) == getattr(want, k), (k, getattr(got, k), getattr(want, k))


@pytest.mark.parametrize("profile,insert", [("xten", 350), ("hs2000", 200)])
def test_device_counts_equal_the_restatement(profile, insert, oracle_lib, tmp_path):
    oracle_lib.orc_train_count.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
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
    assert oracle_lib.orc_train_count(sam, len(sam), fa.encode(), T.bases.encode(), 3, T.bins, 1024, 256, C.byref(want)) == 0
    eng = simuscop_amd.load_engine()
    ctx = C.c_void_p()
    assert eng.sg_create(C.byref(ctx), 0, 1) == 0
    try:
        keys = _reference_on_device(eng, ctx, fa)
        got, ga = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 1024)
        karr = (C.c_char_p * len(keys))(*keys)
        rc = eng.sg_train_count(ctx, sam, len(sam), karr, len(keys), T.bases.encode(), 3, T.bins, 1024, 256, C.byref(got))
        assert rc == 0, eng.sg_last_error(ctx)
        _same_counts(got, ga, want, wa)
        assert got.reads_counted > 200_000 and ga["subs2"].sum() > 0 and ga["quality"].sum() > 10_000_000
        # a line with fewer than eleven fields is an error, as in the reference (Profile.cpp:246-251)
        bad = b"r0\t0\tchr1\t100\t60\n"
        assert eng.sg_train_count(ctx, bad, len(bad), karr, len(keys), T.bases.encode(), 3, T.bins, 1024, 256, C.byref(got)) != 0
        # other base orders / context lengths (the kernel's context index against the restatement's trie)
        # (1 .. 6: the ABI's whole range; from five bases on the count tables leave LDS for global atomics)
        for bases, kmer in ((b"ACTG", 1), (b"ACGT", 2), (b"GTCA", 4), (b"TGCA", 5), (b"CATG", 6)):
            kc = sum(4 ** m for m in range(1, kmer + 1))
            w2, wa2 = TU.count_arrays(simuscop_amd.SgTrainCounts, kc, 20, 1024)
            g2, ga2 = TU.count_arrays(simuscop_amd.SgTrainCounts, kc, 20, 1024)
            part = b"\n".join(lines[:40000]) + b"\n"
            assert oracle_lib.orc_train_count(part, len(part), fa.encode(), bases, kmer, 20, 1024, 256, C.byref(w2)) == 0
            assert eng.sg_train_count(ctx, part, len(part), karr, len(keys), bases, kmer, 20, 1024, 256, C.byref(g2)) == 0, eng.sg_last_error(ctx)
            for k in ("subs1", "subs2", "kmers", "quality", "isize"):
                assert np.array_equal(ga2[k], wa2[k]), (bases, kmer, k)
    finally:
        eng.sg_destroy(ctx)


# ---- the whole of Profile::train ----
def _sampled_lines(oracle_lib, wd, profile="xten", insert=350, coverage=10):
    cfg, fa = H.histogram_config(cases, wd, profile, "PE", coverage, insert)
    out = os.path.join(wd, "gpu")
    r = subprocess.run([SIMU, cfg, "--seed", "78", "--out", out, "--quiet"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    T = H.ProfileTables(oracle_lib, os.path.join(cases.TESTDATA, cases.PROFILES[profile]), True, insert)
    ref = H.read_fasta_one(fa)
    f1, f2 = sorted(os.path.join(out, f) for f in os.listdir(out))
    fq1, fq2 = H.Fastq(f1), H.Fastq(f2)
    lines = TU.sam_from_pairs(ref, fq1, fq2, T.L, T.isize_min + len(T.isize_pmf) - 1, cuts=(H.mismatch_cut(T, False), H.mismatch_cut(T, True)))
    return lines, fa, T


def _setup_from_files(keys, lens, vcf, bed, T, keep):
    """sg_train_setup from the VCF / BED files, parsed here the way vcfparser.cpp:26-106 and Genome.cpp:238-299, 684-739 do (the
    product's own parsers are exercised through `seqToProfile` below)."""
    row = {k: i for i, k in enumerate(keys)}
    abbr = lambda n: n.split(b"chrom", 1)[1] if b"chrom" in n else (n.split(b"chr", 1)[1] if b"chr" in n else n)   # noqa: E731
    snv, ins, dele = [], [], []
    for line in open(vcf, "rb"):
        if line.startswith(b"#"):
            continue
        f = line.split(b"\t")   # (the line break stays with the last field, as after fgets: "1/1\n" is not "1/1")
        if len(f) < 10:
            continue
        info = f[7]
        if b"DP=" in info and int(info.split(b"DP=", 1)[1].split(b";", 1)[0]) < 10:
            continue
        if float(f[5]) < 20:
            continue
        c = row.get(abbr(f[0]))
        if c is None:
            continue
        pos, homo = int(f[1]), f[9].split(b":")[0] != b"1/1"
        if len(f[3]) > 1:
            dele.append((c, pos + 1, len(f[3]) - 1))
        elif len(f[4]) > 1:
            ins.append((c, pos, len(f[4]) - 1))
        else:
            snv.append((c, pos, f[4][:1], 1 if homo else 0))
    st = simuscop_amd.SgTrainSetup()
    karr = (C.c_char_p * len(keys))(*keys)
    keep += [karr]
    st.contig_keys, st.n_contigs, st.bases, st.kmer, st.bins = karr, len(keys), T.bases.encode(), 3, T.bins
    st.n_isize, st.n_indel_len, st.count_gc, st.window = 2048, 256, 1, 1000

    def arr(ctype, vals):
        a = (ctype * max(1, len(vals)))(*vals)
        keep.append(a)
        return a
    st.n_snv = len(snv)
    st.snv_contig, st.snv_pos = arr(C.c_uint32, [x[0] for x in snv]), arr(C.c_int64, [x[1] for x in snv])
    alt = b"".join(x[2] for x in snv)
    keep.append(alt)
    st.snv_alt, st.snv_homo = alt, arr(C.c_uint8, [x[3] for x in snv])
    st.n_ins, st.ins_contig, st.ins_pos, st.ins_len = len(ins), arr(C.c_uint32, [x[0] for x in ins]), arr(C.c_int64, [x[1] for x in ins]), arr(C.c_int32, [x[2] for x in ins])
    st.n_del, st.del_contig, st.del_pos, st.del_len = len(dele), arr(C.c_uint32, [x[0] for x in dele]), arr(C.c_int64, [x[1] for x in dele]), arr(C.c_int32, [x[2] for x in dele])
    if bed:
        per = {}
        for line in open(bed, "rb"):
            f = line.rstrip(b"\n").split(b"\t")
            c = row.get(abbr(f[0]))
            if c is None or lens[c] <= 0:
                continue
            e = int(f[2])
            sp, ep = max(1, int(f[1]) - 50 + 1), min(lens[c], (lens[c] - (-e) % lens[c] if e <= 0 else e) + 50)
            k, s0 = (ep - sp + 1) // 1000, sp
            for i in range(max(k, 0)):
                e0 = ep if i == k - 1 else s0 + 999
                per.setdefault(c, []).append((s0, e0))
                s0 = e0 + 1
            if s0 <= ep:
                per.setdefault(c, []).append((s0, ep))
        first, sp, ep = [0], [], []
        for c in range(len(keys)):
            for a, b in per.get(c, []):
                sp.append(a)
                ep.append(b)
            first.append(len(sp))
        st.target_first, st.target_spos, st.target_epos = arr(C.c_uint64, first), arr(C.c_int64, sp), arr(C.c_int64, ep)
    return st


@pytest.mark.parametrize("exome", [False, True])
def test_device_training_equals_the_restatement(exome, oracle_lib, tmp_path):
    """sg_train_begin / _feed (the text in nine chunks of unequal size) / _finish against orc_train: counters, and the (GC,
    read count) pairs countGC pushed, in its order."""
    import test_train_profile_cpu as TP
    TP.declare(oracle_lib)
    wd = str(tmp_path)
    lines, fa1, T = _sampled_lines(oracle_lib, wd)
    fa, vcf, bed, sam = TU.training_inputs(wd, fa1, lines, T.L, exome=exome)
    want, wa = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
    cap = 200000
    wgc, wrc, wn = (C.c_double * cap)(), (C.c_double * cap)(), C.c_uint64()
    assert oracle_lib.orc_train(sam, len(sam), fa.encode(), vcf.encode(), (bed or "").encode(), T.bases.encode(), 3, T.bins, 2048, 256,
                                C.byref(want), wgc, wrc, cap, C.byref(wn)) == 0
    eng = simuscop_amd.load_engine()
    ctx = C.c_void_p()
    assert eng.sg_create(C.byref(ctx), 0, 1) == 0
    try:
        keys = _reference_on_device(eng, ctx, fa)
        lens = [len(b"".join(part.split(b"\n")[1:])) for part in open(fa, "rb").read().split(b">")[1:]]
        keep = []
        st = _setup_from_files(keys, lens, vcf, bed, T, keep)
        assert eng.sg_train_begin(ctx, C.byref(st)) == 0, eng.sg_last_error(ctx)
        cuts = [0]
        for frac in (0.001, 0.13, 0.131, 0.4, 0.62, 0.95, 0.9999, 0.99995):
            cuts.append(sam.rfind(b"\n", 0, max(1, int(len(sam) * frac))) + 1)
        cuts.append(len(sam))
        for a, b in zip(cuts, cuts[1:]):
            part = sam[a:b]
            assert eng.sg_train_feed(ctx, part, len(part)) == 0, eng.sg_last_error(ctx)
        got, ga = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
        ggc, grc, gn = (C.c_double * cap)(), (C.c_double * cap)(), C.c_uint64()
        assert eng.sg_train_finish(ctx, C.byref(got), ggc, grc, cap, C.byref(gn)) == 0, eng.sg_last_error(ctx)
        _same_counts(got, ga, want, wa)
        assert gn.value == wn.value and gn.value > (50 if exome else 1000)
        assert list(ggc[:gn.value]) == list(wgc[:wn.value]) and list(grc[:gn.value]) == list(wrc[:wn.value])
        assert got.gc_rejected > 100 and got.reads_counted > 50_000 // (20 if exome else 1)
    finally:
        eng.sg_destroy(ctx)


@pytest.mark.parametrize("exome", [False, True])
def test_seqtoprofile_writes_the_profile_of_the_restatement(exome, oracle_lib, tmp_path):
    """The command line (src/seqToProfile.cpp's options + --sam) end to end: its profile file is the restatement's, byte for
    byte behind the time stamp line, and so is the `.gc` side file of estimateGCParas; the unmodified reference binary -- where
    it has been built (this container; it travels to the GPU box with oracle/_ref) -- loads the file the GPU path wrote and
    samples from it as oracle(mt) does."""
    import hashlib
    import test_train_profile_cpu as TP
    TP.declare(oracle_lib)
    wd = str(tmp_path)
    lines, fa1, T = _sampled_lines(oracle_lib, wd, coverage=12)
    fa, vcf, bed, sam = TU.training_inputs(wd, fa1, lines, T.L, exome=exome)
    sam_path = os.path.join(wd, "reads.sam")
    open(sam_path, "wb").write(sam)
    want = os.path.join(wd, "want.profile")
    assert oracle_lib.orc_train_profile(sam, len(sam), fa.encode(), vcf.encode(), (bed or "").encode(), b"ACTG", 3, 50, want.encode(), sam_path.encode(),
                                        b"stamp\n") == 0
    got = os.path.join(wd, "got.profile")
    exe = os.path.join(ROOT, "simuscop_amd", "lib", "seqToProfile")
    cmd = [exe, "--sam", sam_path, "-v", vcf, "-r", fa, "-o", got, "--quiet"] + (["-t", bed] if bed else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = open(want, "rb").read().split(b"\n", 1), open(got, "rb").read().split(b"\n", 1)
    assert a[0] == b"#model created at stamp" and b[0].startswith(b"#model created at ")
    if a[1] != b[1]:   # say where: section and line
        la, lb, sec, diffs = a[1].split(b"\n"), b[1].split(b"\n"), b"", []
        for i, (x, y) in enumerate(zip(la, lb)):
            if x.startswith((b"[", b"kmer:", b"basePair")):
                sec = x
            if x != y:
                diffs.append((i, sec, x[:200], y[:200]))
                if len(diffs) > 5:
                    break
        assert False, ("profile text differs from the restatement's", len(la), len(lb), diffs)
    assert os.path.exists(want + ".gc") == os.path.exists(got + ".gc")
    if os.path.exists(want + ".gc"):
        assert open(want + ".gc", "rb").read() == open(got + ".gc", "rb").read()
    if not exome:
        assert os.path.exists(got + ".gc")   # 12x: the GC model is fitted
    # the count tables straight in memory (the kernel of context lengths whose tables do not fit LDS): the same file
    got3 = os.path.join(wd, "got3.profile")
    r = subprocess.run([exe, "--sam", sam_path, "-v", vcf, "-r", fa, "-o", got3, "--quiet"] + (["-t", bed] if bed else []), capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, SG_TRAIN_GLOBAL_ATOMICS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(got3, "rb").read().split(b"\n", 1)[1] == b[1]
    # the library call (simu_train, host/train.h) with a fixed time-stamp line: the whole file
    got4 = os.path.join(wd, "got4.profile")
    st = simuscop_amd.train_profile(ref=fa, vcf=vcf, output=got4, sam=sam_path, target=bed or "", stamp="stamp\n")
    assert open(got4, "rb").read() == open(want, "rb").read()
    assert st.read_length == T.L and st.bins == 50 and st.reads_counted > 1000 and st.lines == sam.count(b"\n")
    # through standard input, as `samtools view ... | seqToProfile --sam -` would
    got2 = os.path.join(wd, "got2.profile")
    with open(sam_path, "rb") as f:
        r = subprocess.run([exe, "--sam", "-", "-v", vcf, "-r", fa, "--quiet"] + (["-t", bed] if bed else []), stdin=f, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split(b"\n", 1)[1].replace(b"#reads: -", b"#reads: " + sam_path.encode(), 1) == b[1]
    # option checks of src/seqToProfile.cpp:84-121
    for bad, msg in ((["-v", vcf, "-r", fa], "Use --bam"), (["--sam", sam_path, "-r", fa], "Use --vcf"), (["--sam", sam_path, "-v", vcf], "Use --ref"),
                     (["--sam", sam_path, "-v", vcf, "-r", fa, "-k", "6"], "maximum value of 5"), (["--sam", sam_path, "-v", vcf, "-r", fa, "-B", "9"], "minimum value of 10")):
        r = subprocess.run([exe] + bad, capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and msg in r.stderr, (bad, r.stderr[-300:])
    if not exome:
        # the trained profile drives the GPU sampler like a shipped one: simuReads on it = oracle(philox) on it, byte for byte
        cfg2, out2, out3 = os.path.join(wd, "sim2.txt"), os.path.join(wd, "sim2_gpu"), os.path.join(wd, "sim2_orc")
        cases._config(cfg2, ref=fa1, profile=got, name="t", output=os.path.join(wd, "unused"), layout="PE", threads=1, verbose=0, coverage=2,
                      insertSize=350, ploidy=2)
        r = subprocess.run([SIMU, cfg2, "--seed", "4242", "--out", out2, "--quiet"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert oracle_lib.orc_simulate(cfg2.encode(), 1, 4242 >> 32, 4242 & 0xFFFFFFFF, out3.encode(), 4) == 0, oracle_lib.orc_last_error().decode()
        files = sorted(os.listdir(out3))
        assert files == sorted(os.listdir(out2)) and len(files) == 2
        for f in files:
            assert open(os.path.join(out2, f), "rb").read() == open(os.path.join(out3, f), "rb").read(), f
        # five-base contexts, twenty bins (the count tables of ONE bin per workgroup fill the LDS budget)
        want5, got5 = os.path.join(wd, "want5.profile"), os.path.join(wd, "got5.profile")
        assert oracle_lib.orc_train_profile(sam, len(sam), fa.encode(), vcf.encode(), b"", b"ACTG", 5, 20, want5.encode(), sam_path.encode(), b"stamp\n") == 0
        r = subprocess.run([exe, "--sam", sam_path, "-v", vcf, "-r", fa, "-o", got5, "-k", "5", "-B", "20", "--quiet"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(want5, "rb").read().split(b"\n", 1)[1] == open(got5, "rb").read().split(b"\n", 1)[1]
    REF, SHIM = os.path.join(ROOT, "oracle", "_ref", "simuReads"), os.path.join(ROOT, "oracle", "_ref", "libfakeclock.so")
    if os.path.exists(REF) and not exome:
        cfg, out = os.path.join(wd, "sim.txt"), os.path.join(wd, "sim_out")
        cases._config(cfg, ref=fa1, profile=got, name="t", output=out, layout="PE", threads=1, verbose=0, coverage=1, insertSize=350, ploidy=2)
        env = dict(os.environ, LD_PRELOAD=SHIM, FAKECLOCK_SEC=str(cases.FAKE_SEC), FAKECLOCK_NSEC=str(cases.FAKE_NSEC))
        r = subprocess.run([REF, cfg], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1000:]
        md5 = lambda d: {f: hashlib.md5(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))}   # noqa: E731
        ref_md5 = md5(out)
        for f in os.listdir(out):
            os.remove(os.path.join(out, f))
        assert oracle_lib.orc_simulate(cfg.encode(), 0, cases.FAKE_SEC, cases.FAKE_NSEC, b"", 1) == 0
        assert md5(out) == ref_md5 and ref_md5


def _random_training_case(rng, wd):
    """A small random input for countGC's scan: contigs of random sizes (some shorter than a window, X among them), reads at
    random positions -- mostly ascending, with steps backwards, contig changes at random places, reads past contig ends --,
    random targets (unsorted, overlapping, nested) or none, a few known variants."""
    from simuscop_amd import synth
    names = ["chr%d" % (i + 1) for i in range(rng.randrange(1, 6))] + (["chrX"] if rng.random() < 0.5 else [])
    sizes = [rng.choice([300, 950, 1000, 1001, 2500, 7000, 20000]) for _ in names]
    fa = os.path.join(wd, "ref.fa")
    seqs = {}
    with open(fa, "wb") as f:
        for k, (n, L) in enumerate(zip(names, sizes)):
            s = synth.synth_contig(L, rng.randrange(1, 10 ** 6), 0, n_runs=rng.random() < 0.3).tobytes()
            seqs[n] = s
            f.write(b">" + n.encode() + b"\n" + b"".join(s[i:i + 60] + b"\n" for i in range(0, len(s), 60)))
    RL = rng.choice([36, 75, 100])
    lines = []
    cur = rng.choice(names)
    pos = 1
    for _ in range(rng.randrange(200, 3000)):
        r = rng.random()
        if r < 0.02:
            cur, pos = rng.choice(names), 1
        elif r < 0.06:
            pos = max(1, pos - rng.randrange(1, 3000))
        else:
            pos += rng.choice([0, 0, 1, 3, 17, 60, 400, 1500])
        size = len(seqs[cur])
        if pos > size + 50:
            cur, pos = rng.choice(names), 1
            size = len(seqs[cur])
        lines.append(TU._crafted(rng, cur.encode(), seqs[cur], pos, RL, rng.choice([-250, 0, 200, 300]),
                                 cigar=rng.choice([None] * 8 + [b"10M2I%dM" % (RL - 12), b"10M3D%dM" % (RL - 10), b"5S%dM" % (RL - 5)]),
                                 mapq=rng.choice([b"60"] * 9 + [b"3"])))
    vcf = [b"##fileformat=VCFv4.2"]
    for _ in range(rng.randrange(0, 12)):
        n = rng.choice(names)
        p = rng.randrange(1, len(seqs[n]))
        kind = rng.random()
        ref_b = seqs[n][p - 1:p].upper() or b"A"
        if kind < 0.6:
            row = [n.encode(), b"%d" % p, b".", ref_b, bytes([rng.choice(b"ACGT")]), b"50", b"PASS", b"DP=40", b"GT", rng.choice([b"0/1", b"1/1"])]
        elif kind < 0.8:
            row = [n.encode(), b"%d" % p, b".", ref_b, ref_b + b"AC", b"50", b"PASS", b"DP=40", b"GT", b"0/1"]
        else:
            row = [n.encode(), b"%d" % p, b".", ref_b + b"ACG", ref_b, b"50", b"PASS", b"DP=40", b"GT", b"0/1"]
        vcf.append(b"\t".join(row))
    vcf_path = os.path.join(wd, "k.vcf")
    open(vcf_path, "wb").write(b"\n".join(vcf) + b"\n")
    bed = None
    if rng.random() < 0.5:
        rows = []
        for _ in range(rng.randrange(1, 25)):
            n = rng.choice(names)
            a = rng.randrange(0, len(seqs[n]))
            rows.append(b"%s\t%d\t%d" % (n.encode(), a, a + rng.choice([1, 40, 300, 1200, 2600])))
        bed = os.path.join(wd, "t.bed")
        open(bed, "wb").write(b"\n".join(rows) + b"\n")
    return fa, vcf_path, bed, b"\n".join(lines) + b"\n", RL


def test_device_training_on_random_inputs(oracle_lib, tmp_path):
    """Forty small random inputs (one process, one engine context): counters and (GC, read count) pairs against the
    restatement, the text cut into chunks at random line ends.  SG_TRAIN_FUZZ="first:count" widens the run (the logs of
    such runs are under profiles/*_parity/)."""
    import random
    import test_train_profile_cpu as TP
    TP.declare(oracle_lib)
    T = H.ProfileTables(oracle_lib, os.path.join(cases.TESTDATA, cases.PROFILES["xten"]), True, 350)
    eng = simuscop_amd.load_engine()
    first, count = (int(v) for v in os.environ.get("SG_TRAIN_FUZZ", "0:40").split(":"))
    for seed in range(first, first + count):
        rng = random.Random(7000 + seed)
        wd = str(tmp_path / ("c%d" % seed))
        os.makedirs(wd)
        fa, vcf, bed, sam, RL = _random_training_case(rng, wd)
        bins = rng.choice([10, 50])
        want, wa = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, bins, 2048)
        cap = 20000
        wgc, wrc, wn = (C.c_double * cap)(), (C.c_double * cap)(), C.c_uint64()
        assert oracle_lib.orc_train(sam, len(sam), fa.encode(), vcf.encode(), (bed or "").encode(), T.bases.encode(), 3, bins, 2048, 256,
                                    C.byref(want), wgc, wrc, cap, C.byref(wn)) == 0
        ctx = C.c_void_p()
        assert eng.sg_create(C.byref(ctx), 0, 1) == 0
        try:
            keys = _reference_on_device(eng, ctx, fa)
            lens = [len(b"".join(part.split(b"\n")[1:])) for part in open(fa, "rb").read().split(b">")[1:]]
            keep = []

            class TT:   # (the setup helper reads .bases / .bins)
                pass
            tt = TT()
            tt.bases, tt.bins = T.bases, bins
            st = _setup_from_files(keys, lens, vcf, bed, tt, keep)
            assert eng.sg_train_begin(ctx, C.byref(st)) == 0, eng.sg_last_error(ctx)
            ends = [i + 1 for i, ch in enumerate(sam) if ch == 10]
            cuts = sorted(set([0, len(sam)] + [rng.choice(ends) for _ in range(rng.randrange(0, 6))]))
            for a, b in zip(cuts, cuts[1:]):
                assert eng.sg_train_feed(ctx, sam[a:b], b - a) == 0, eng.sg_last_error(ctx)
            got, ga = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, bins, 2048)
            ggc, grc, gn = (C.c_double * cap)(), (C.c_double * cap)(), C.c_uint64()
            assert eng.sg_train_finish(ctx, C.byref(got), ggc, grc, cap, C.byref(gn)) == 0, eng.sg_last_error(ctx)
            try:
                _same_counts(got, ga, want, wa)
                assert gn.value == wn.value
                assert list(ggc[:gn.value]) == list(wgc[:wn.value]) and list(grc[:gn.value]) == list(wrc[:wn.value])
            except AssertionError as e:
                raise AssertionError((seed, bool(bed), cuts, str(e)[:600]))
        finally:
            eng.sg_destroy(ctx)


def test_seqtoprofile_refusals(tmp_path):
    """What the reference's trainer refuses, with its exit codes: unreadable VCF / target / read files (exit(-1)), a line with
    fewer than eleven fields (exit(1), Profile.cpp:246-251); and no read with a single match, where the reference's
    setReadLength would run off its buffer."""
    from simuscop_amd import synth
    wd = str(tmp_path)
    fa = os.path.join(wd, "r.fa")
    seq = synth.synth_contig(5000, 3, 0, n_runs=False).tobytes()
    open(fa, "wb").write(b">chr1\n" + b"".join(seq[i:i + 60] + b"\n" for i in range(0, len(seq), 60)))
    vcf = os.path.join(wd, "k.vcf")
    open(vcf, "w").write("##fileformat=VCFv4.2\n")
    import random
    rng = random.Random(1)
    good = [TU._crafted(rng, b"chr1", seq, p, 50, 200) for p in range(1, 4000, 7)]
    sam = os.path.join(wd, "ok.sam")
    open(sam, "wb").write(b"\n".join(good) + b"\n")
    exe = os.path.join(ROOT, "simuscop_amd", "lib", "seqToProfile")

    def run(*a):
        return subprocess.run([exe, *a, "--quiet"], capture_output=True, text=True, timeout=300)
    r = run("--sam", sam, "-v", vcf, "-r", fa, "-o", os.path.join(wd, "p.profile"))
    assert r.returncode == 0, r.stderr[-1000:]
    r = run("--sam", sam, "-v", os.path.join(wd, "none.vcf"), "-r", fa, "-o", os.path.join(wd, "p2"))
    assert r.returncode == 255 and "cannot open VCF file" in r.stderr
    r = run("--sam", os.path.join(wd, "none.sam"), "-v", vcf, "-r", fa, "-o", os.path.join(wd, "p2"))
    assert r.returncode == 255 and "cannot open SAM file" in r.stderr
    r = run("--sam", sam, "-v", vcf, "-r", fa, "-t", os.path.join(wd, "none.bed"), "-o", os.path.join(wd, "p2"))
    assert r.returncode == 255 and "can not open target file" in r.stderr
    bad = os.path.join(wd, "bad.sam")
    open(bad, "wb").write(b"\n".join(good[:100] + [b"r\t0\tchr1\t10\t60"] + good[100:]) + b"\n")
    r = run("--sam", bad, "-v", vcf, "-r", fa, "-o", os.path.join(wd, "p2"))
    assert r.returncode == 1 and "malformed read" in r.stderr and "11 mandatory fields" in r.stderr
    clipped = os.path.join(wd, "clipped.sam")
    open(clipped, "wb").write(b"\n".join(l.replace(b"\t50M\t", b"\t5S45M\t") for l in good) + b"\n")
    r = run("--sam", clipped, "-v", vcf, "-r", fa, "-o", os.path.join(wd, "p2"))
    assert r.returncode == 1 and "single match" in r.stderr


def test_a_chunks_verdict_arrives_with_the_next_call(tmp_path):
    """sg_train_feed returns with the chunk's kernels running: a line with fewer than eleven fields (Profile.cpp:246-251)
    in chunk k is reported by the feed of chunk k + 1, or by sg_train_finish when it was the last chunk -- with the
    reference's message either way -- and the caller's buffer may be reused as soon as feed has returned."""
    import random
    import ctypes
    from simuscop_amd import synth
    wd = str(tmp_path)
    fa = os.path.join(wd, "r.fa")
    seq = synth.synth_contig(6000, 5, 0, n_runs=False).tobytes()
    open(fa, "wb").write(b">chr1\n" + b"".join(seq[i:i + 60] + b"\n" for i in range(0, len(seq), 60)))
    vcf = os.path.join(wd, "k.vcf")
    open(vcf, "w").write("##fileformat=VCFv4.2\n")
    rng = random.Random(5)
    good = [TU._crafted(rng, b"chr1", seq, p, 50, 200) for p in range(1, 5000, 5)]
    ok = b"\n".join(good[:400]) + b"\n"
    bad = b"\n".join(good[400:500] + [b"r\t0\tchr1\t10\t60"] + good[500:600]) + b"\n"
    tail = b"\n".join(good[600:]) + b"\n"
    eng = simuscop_amd.load_engine()

    class TT:
        bases, bins = "ACTG", 10
    for chunks, failing_call in (((ok, bad, tail), 2), ((ok, bad), "finish")):
        ctx = C.c_void_p()
        assert eng.sg_create(C.byref(ctx), 0, 1) == 0
        try:
            keys = _reference_on_device(eng, ctx, fa)
            keep = []
            st = _setup_from_files(keys, [len(seq)], vcf, None, TT, keep)
            assert eng.sg_train_begin(ctx, C.byref(st)) == 0, eng.sg_last_error(ctx)
            buf = ctypes.create_string_buffer(max(len(c) for c in chunks))   # ONE buffer for every chunk, rewritten right after feed
            for i, c in enumerate(chunks):
                ctypes.memmove(buf, c, len(c))
                rc = eng.sg_train_feed(ctx, buf, len(c))
                ctypes.memset(buf, 0x41, len(buf))
                if i == failing_call:
                    assert rc != 0 and b"malformed read" in eng.sg_last_error(ctx) and b"11 mandatory fields" in eng.sg_last_error(ctx)
                    break
                assert rc == 0, (i, eng.sg_last_error(ctx))
            else:
                got, ga = TU.count_arrays(simuscop_amd.SgTrainCounts, 84, 10, 2048)
                rc = eng.sg_train_finish(ctx, C.byref(got), None, None, 0, C.byref(C.c_uint64()))
                assert failing_call == "finish" and rc != 0 and b"malformed read" in eng.sg_last_error(ctx)
        finally:
            eng.sg_train_end(ctx)
            eng.sg_destroy(ctx)


@pytest.mark.parametrize("exome", [False, True])
def test_device_training_stops_at_the_cap(exome, oracle_lib, tmp_path):
    """Profile::processRead's cap on counted reads (Profile.cpp:236, 497-507; twice it with targets) with small values: the
    device cuts the chunk at the capping line -- counters, windows and pairs equal the restatement's, whether the line falls
    in the first chunk, in a later one, or on a chunk's last line -- and ignores what is fed afterwards; the command line
    stops reading and writes the restatement's file."""
    import test_train_profile_cpu as TP
    TP.declare(oracle_lib)
    oracle_lib.orc_train_set_max_reads.argtypes = [C.c_uint64]
    oracle_lib.orc_train_set_max_reads.restype = None
    wd = str(tmp_path)
    lines, fa1, T = _sampled_lines(oracle_lib, wd, coverage=4)
    fa, vcf, bed, sam = TU.training_inputs(wd, fa1, lines, T.L, exome=exome)
    eng = simuscop_amd.load_engine()
    ends = [i + 1 for i, ch in enumerate(sam) if ch == 10]
    cap_n = 200000
    try:
        for cap in ((100, 1, 400) if exome else (700, 1, 20011)):
            oracle_lib.orc_train_set_max_reads(cap)
            want, wa = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
            wgc, wrc, wn = (C.c_double * cap_n)(), (C.c_double * cap_n)(), C.c_uint64()
            assert oracle_lib.orc_train(sam, len(sam), fa.encode(), vcf.encode(), (bed or "").encode(), T.bases.encode(), 3, T.bins, 2048, 256,
                                        C.byref(want), wgc, wrc, cap_n, C.byref(wn)) == 0
            assert want.capped == 1 and want.reads_counted == (2 * cap if exome else cap)
            # chunk cuts: one exactly behind the capping line, one before it, a few behind it
            at = ends[want.lines - 1]
            for cuts in ([0, len(sam)], [0, at, len(sam)], sorted({0, ends[max(0, want.lines // 2)], at, ends[min(len(ends) - 1, want.lines + 5)], len(sam)})):
                ctx = C.c_void_p()
                assert eng.sg_create(C.byref(ctx), 0, 1) == 0
                try:
                    keys = _reference_on_device(eng, ctx, fa)
                    lens = [len(b"".join(part.split(b"\n")[1:])) for part in open(fa, "rb").read().split(b">")[1:]]
                    keep = []
                    st = _setup_from_files(keys, lens, vcf, bed, T, keep)
                    st.max_reads = cap
                    assert eng.sg_train_begin(ctx, C.byref(st)) == 0, eng.sg_last_error(ctx)
                    for a, b in zip(cuts, cuts[1:]):
                        assert eng.sg_train_feed(ctx, sam[a:b], b - a) == 0, eng.sg_last_error(ctx)
                    # (a chunk's verdict arrives with the next feed: known here unless the capping chunk was the last one)
                    assert eng.sg_train_capped(ctx) == (1 if min(c for c in cuts if c >= at) < len(sam) else 0)
                    got, ga = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
                    ggc, grc, gn = (C.c_double * cap_n)(), (C.c_double * cap_n)(), C.c_uint64()
                    assert eng.sg_train_finish(ctx, C.byref(got), ggc, grc, cap_n, C.byref(gn)) == 0, eng.sg_last_error(ctx)
                    assert got.capped == 1
                    _same_counts(got, ga, want, wa)
                    assert gn.value == wn.value and list(ggc[:gn.value]) == list(wgc[:wn.value]) and list(grc[:gn.value]) == list(wrc[:wn.value])
                finally:
                    eng.sg_destroy(ctx)
        # the command line with --max-reads: the restatement's file
        cli_cap = 300 if exome else 3000
        oracle_lib.orc_train_set_max_reads(cli_cap)
        sam_path, want_p, got_p = os.path.join(wd, "r.sam"), os.path.join(wd, "w.profile"), os.path.join(wd, "g.profile")
        open(sam_path, "wb").write(sam)
        assert oracle_lib.orc_train_profile(sam, len(sam), fa.encode(), vcf.encode(), (bed or "").encode(), b"ACTG", 3, 50, want_p.encode(), sam_path.encode(), b"stamp\n") == 0
        exe = os.path.join(ROOT, "simuscop_amd", "lib", "seqToProfile")
        r = subprocess.run([exe, "--sam", sam_path, "-v", vcf, "-r", fa, "-o", got_p, "--max-reads", str(cli_cap), "--quiet", "--stats"] + (["-t", bed] if bed else []),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and '"capped": 1' in r.stderr, r.stderr[-1500:]
        assert open(want_p, "rb").read().split(b"\n", 1)[1] == open(got_p, "rb").read().split(b"\n", 1)[1]
    finally:
        oracle_lib.orc_train_set_max_reads(0)
