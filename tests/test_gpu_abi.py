"""GPU tests that drive the C ABI directly (no CLI): GC scan against numpy, call-order / argument
errors, and size-independent properties of a full-size pass (BASELINE C2: chr20-sized contig, XTen
PE151, 30x, 6.4 M pairs): record structure, mate synchronisation, counts, determinism under the seed,
and reads mapping back to the haplotype at the position their name carries."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import cases
import simuscop_amd
from simuscop_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return simuscop_amd.load_engine()


def _ctx(eng, seed=1):
    ctx = C.c_void_p()
    rc = eng.sg_create(C.byref(ctx), 0, seed)
    assert rc == 0, eng.sg_last_error(None)
    return ctx


def test_gc_percent_matches_numpy(eng):
    rng = np.random.default_rng(3)
    n = 300000
    chain = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    chain[5000:5400] = ord("N")
    chain[70000] = ord("R")  # neither GC nor N (MyDefine.cpp:286-292)
    chain2 = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1234, p=[0.1, 0.4, 0.4, 0.1])
    ctx = _ctx(eng)
    try:
        bufs = [chain.tobytes(), chain2.tobytes()]
        arr = (C.c_char_p * 2)(*bufs)
        lens = (C.c_uint64 * 2)(len(bufs[0]), len(bufs[1]))
        assert eng.sg_upload_haplotypes(ctx, 2, arr, lens) == 0
        wins = []
        for s in range(0, n - 1000, 997):
            wins.append((s, 0, 1000))
        wins += [(n - 333, 0, 333), (0, 1, 1234), (7, 1, 1), (100, 1, 17), (4990, 0, 11), (69990, 0, 20)]
        w = (simuscop_amd.SgGcWindow * len(wins))(*[simuscop_amd.SgGcWindow(s, c, l) for s, c, l in wins])
        out = (C.c_int32 * len(wins))()
        assert eng.sg_gc_percent(ctx, w, len(wins), out) == 0, eng.sg_last_error(ctx)
        for i, (s, c, l) in enumerate(wins):
            seg = (chain if c == 0 else chain2)[s:s + l]
            exp = -1 if (seg == ord("N")).any() else 100 * int(((seg == ord("G")) | (seg == ord("C"))).sum()) // l
            assert out[i] == exp, (i, s, c, l, out[i], exp)
        # a window past the chain end is refused on the host, not left to fault on the device
        bad = (simuscop_amd.SgGcWindow * 1)(simuscop_amd.SgGcWindow(n - 10, 0, 100))
        assert eng.sg_gc_percent(ctx, bad, 1, out) != 0
    finally:
        eng.sg_destroy(ctx)


def test_call_order_and_argument_errors(eng):
    ctx = _ctx(eng)
    try:
        b = simuscop_amd.SgBatch()
        assert eng.sg_plan(ctx, C.byref(b)) != 0 and b"sg_load_profile" in eng.sg_last_error(ctx)
        assert eng.sg_sample(ctx) != 0 and b"sg_plan" in eng.sg_last_error(ctx)
        n1 = C.c_uint64()
        assert eng.sg_result(ctx, C.byref(n1), None, None) != 0
    finally:
        eng.sg_destroy(ctx)
    bad = C.c_void_p()
    assert eng.sg_create(C.byref(bad), 9999, 1) != 0 and not bad.value


@pytest.fixture(scope="module")
def c2_session(tmp_path_factory):
    wd = str(tmp_path_factory.mktemp("c2"))
    fa = os.path.join(wd, "ref.fa")
    L = 64444167
    synth.write_fasta(fa, [("chr20", L)], seed=20)
    cfg = os.path.join(wd, "config.txt")
    with open(cfg, "w") as f:
        f.write(f"ref = {fa}\nprofile = {os.path.join(cases.TESTDATA, cases.PROFILES['xten'])}\nname = sim\n"
                f"output = {wd}/out\nlayout = PE\nthreads = 1\nverbose = 0\ncoverage = 30\ninsertSize = 350\n")
    sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=12345)
    sess.weighted_length()
    sess.set_reads(sess.planned_reads)
    assert sess.prepare_batch(0)
    ref = synth.synth_contig(L, 20, 0)
    ref = np.where((ref >= 97) & (ref <= 122), ref - 32, ref).astype(np.uint8)
    yield sess, ref
    sess.close()


def _head(sess, nbytes=32 << 20):
    sess.sample()
    b1, b2, nf = sess.result()
    buf1, buf2 = C.create_string_buffer(nbytes), C.create_string_buffer(nbytes)
    assert sess.eng.sg_fetch_range(sess.ctx, 0, 0, nbytes, buf1) == 0
    assert sess.eng.sg_fetch_range(sess.ctx, 1, 0, nbytes, buf2) == 0
    return b1, b2, nf, buf1.raw, buf2.raw


def test_full_size_pass_properties(c2_session):
    sess, ref = c2_session
    b1, b2, nf, r1, r2 = _head(sess)
    planned = sess.planned_reads
    # ceil(n/2) pairs per window (Segment.cpp:848): at least planned/2, at most one extra pair per window
    assert planned // 2 <= nf <= planned // 2 + 200000
    assert abs(b1 - b2) < 0.001 * b1 and 300 * nf < b1 < 340 * nf   # ~330 B per 151-bp record
    qual_alphabet = set(b")-7<AFJ")
    rec1 = r1[:r1.rfind(b"\n@") + 1].split(b"\n")
    rec2 = r2[:r2.rfind(b"\n@") + 1].split(b"\n")
    n = min(len(rec1), len(rec2)) // 4
    assert n > 90000
    mism = tot = 0
    lens = {}
    for i in range(0, n, 7):
        h1, s1, p1, q1 = rec1[4 * i:4 * i + 4]
        h2, s2, p2, q2 = rec2[4 * i:4 * i + 4]
        assert h1.startswith(b"@sim#20#") and h1.endswith(b"/1") and h2 == h1[:-1] + b"2"   # mates stay in sync
        assert p1 == b"+" and p2 == b"+" and len(s1) == len(q1) and len(s2) == len(q2)
        assert set(s1) <= set(b"ACGTN") and set(s2) <= set(b"ACGTN")
        for s_, q_ in ((s1, q1), (s2, q2)):   # called bases carry profile qualities; 'N' carries U{33..52} (Profile.cpp:1582)
            assert all((q in qual_alphabet) if b != 78 else (33 <= q <= 52) for b, q in zip(s_, q_))
        lens[len(s1)] = lens.get(len(s1), 0) + 1
        if len(s1) == 151:
            pos = int(h1.split(b"#")[2])          # single <=1 Mbp segments: pos % segsize, segment-relative
            seg = i  # unknown segment: try all starts congruent mod 1e6 is costly; use the first segment only
            if i < 20000:                          # first records belong to segment 0 (starts at 1)
                r = ref[pos:pos + 151].tobytes()
                if len(r) == 151 and b"N" not in r:
                    d = sum(a != b for a, b in zip(s1, r))
                    if d <= 20:
                        mism += d
                        tot += 151
    assert 50 <= min(lens) and max(lens) < 151 + 8 * 46
    assert lens.get(151, 0) > 0.75 * sum(lens.values())             # ~83 % of reads carry no sequencing indel
    assert tot > 100 * 151 and mism / tot < 0.01                    # reads map back where their name says


def test_pass_is_a_pure_function_of_the_seed(c2_session):
    sess, _ = c2_session
    sess.set_seed(777)
    a = _head(sess, 8 << 20)
    b = _head(sess, 8 << 20)
    assert a[:3] == b[:3] and hashlib.md5(a[3]).digest() == hashlib.md5(b[3]).digest() and a[4] == b[4]
    sess.set_seed(778)
    c = _head(sess, 8 << 20)
    assert c[2] == a[2]                      # the plan (counts) does not depend on the sampling seed ...
    assert c[3] != a[3] and c[4] != a[4]     # ... the reads do


@pytest.mark.parametrize("prof,want", [("xten", 1), ("hs2500", 1), ("hs2000", 1), ("gaiix", 1)])
def test_shipped_profiles_get_the_straight_line_emit_kernel(prof, want, tmp_path):
    """Regression guard on the LDS budget: the table image of all four shipped profiles (context keep counts +
    diagonal alias columns, 8 or 64 columns wide) must fit the straight-line kernel."""
    fa = str(tmp_path / "ref.fa")
    synth.write_fasta(fa, [("chr1", 40000)], seed=3)
    cfg = str(tmp_path / "c.txt")
    with open(cfg, "w") as f:
        f.write(f"ref = {fa}\nprofile = {os.path.join(cases.TESTDATA, cases.PROFILES[prof])}\nname = s\noutput = {tmp_path}/o\n"
                f"layout = PE\nthreads = 1\nverbose = 0\ncoverage = 2\ninsertSize = 300\n")
    sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=1)
    try:
        assert sess.eng.sg_emit_variant(sess.ctx) == want
    finally:
        sess.close()


def test_detached_outputs_survive_the_next_pass(tmp_path):
    """sg_detach_outputs: the text of pass 1 stays fetchable (own buffers, own stream) while pass 2 runs and
    overwrites nothing of it; released sets are reused; the context refuses fetches after a detach."""
    cfg = cases.build_case("wgs_pe_xten", str(tmp_path))
    sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=21)
    eng = sess.eng
    try:
        sess.weighted_length()
        sess.set_reads(sess.planned_reads)
        assert sess.prepare_batch(0)
        sess.sample()
        b1, b2, nf = sess.result()
        t1, t2 = sess.fetch(b1, b2)
        g1, g2 = sess.compress()
        z1 = sess.fetch_compressed(0, g1)
        h = C.c_void_p()
        assert eng.sg_detach_outputs(sess.ctx, C.byref(h)) == 0
        buf = C.create_string_buffer(16)
        assert eng.sg_fetch_range(sess.ctx, 0, 0, 16, buf) != 0       # nothing left in the context
        sess.set_seed(22)
        sess.sample()                                                  # pass 2 into fresh buffers
        c1, c2, _ = sess.result()
        u1, u2 = sess.fetch(c1, c2)
        assert u1 != t1
        tb, gb = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        assert eng.sg_outputs_sizes(h, tb, gb) == 0 and (tb[0], tb[1], gb[0]) == (b1, b2, g1)
        for mate, want in ((0, t1), (1, t2)):
            got = C.create_string_buffer(len(want))
            assert eng.sg_outputs_fetch(h, mate, 0, 0, len(want), got) == 0
            assert got.raw == want
        got = C.create_string_buffer(g1)
        assert eng.sg_outputs_fetch(h, 0, 1, 0, g1, got) == 0 and got.raw == z1
        assert eng.sg_outputs_fetch(h, 0, 0, b1 - 3, 8, got) != 0    # range check
        assert eng.sg_release_outputs(sess.ctx, h) == 0
        sess.set_seed(21)
        sess.sample()                                                  # reuses the released set
        d1, d2, _ = sess.result()
        assert sess.fetch(d1, d2) == (t1, t2)
    finally:
        sess.close()


def test_a_pass_larger_than_the_buffers_it_was_launched_into(tmp_path, monkeypatch):
    """Every pass but a context's first is queued without waiting for the text size (the emit kernels are launched into
    the buffers the context holds and check the capacity themselves).  A small chromosome first, then the largest one:
    the second pass does not fit, is emitted again into grown buffers, and equals what the same sequence of passes gives
    when every pass waits for its size (SG_NO_SPECULATION)."""
    cfg = cases.build_case("c3_grch38_pe_xten_cov3", str(tmp_path))

    def texts_of(chroms):
        sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=31)
        try:
            sess.weighted_length()
            sess.set_reads(sess.planned_reads)
            out = []
            for c in chroms:
                assert sess.prepare_batch(c)
                sess.sample()
                b1, b2, nf = sess.result()
                # (the items left to emit_slow_kernel are counted once, also when the pass was emitted a second time)
                out.append((b1, b2, nf, hashlib.md5(b"".join(sess.fetch(b1, b2))).hexdigest(), tuple(sess.emit_info())))
            return out
        finally:
            sess.close()

    order = [20, 0, 21, 1]                  # chr21 (smallest), chr1, chr22, chr2
    queued = texts_of(order)
    monkeypatch.setenv("SG_NO_SPECULATION", "1")
    waited = texts_of(order)
    assert queued == waited
    assert queued[1][0] > 4 * queued[0][0] and queued[3][0] > 4 * queued[2][0]


def test_runs_in_one_process_reuse_device_blocks(tmp_path):
    """The device blocks a finished run gave up serve the next run of the process (sg_api.cpp BlockCache; what keeps a
    sequence of whole-genome runs off the allocator, see tools/c3_steps.py).  They come back dirty -- holding the other
    run's tables, rows and text -- so: the same configuration before and after a different one, and after the cache was
    emptied, must write the same bytes; so must a run with the cache turned off in a fresh process (the parity tests)."""
    cfg_a = cases.build_case("indel_rich_n_islands_pe", str(tmp_path / "a"))
    cfg_b = cases.build_case("wes_tight_targets_pe", str(tmp_path / "b"))

    def run(cfg, tag):
        out = str(tmp_path / tag)
        simuscop_amd.run_config(cfg, device=0, quiet=1, write_files=1, seed=77, output_dir=out)
        return {f: hashlib.md5(open(os.path.join(out, f), "rb").read()).hexdigest() for f in sorted(os.listdir(out))}

    first_a, first_b = run(cfg_a, "a1"), run(cfg_b, "b1")
    assert first_a and first_b and first_a != first_b
    assert run(cfg_a, "a2") == first_a           # into blocks that held b's data
    assert run(cfg_b, "b2") == first_b
    simuscop_amd.release_cached_memory()
    assert run(cfg_a, "a3") == first_a           # into fresh blocks
    simuscop_amd.release_cached_memory()
