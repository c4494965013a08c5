"""Reference ingest and haplotype assembly on the device against numpy models, through the
C ABI, and the two haplotype routes of the CLI against each other.

The reference side of this is Fasta.cpp:304-334 (one slice per segment through the .fai arithmetic),
Segment.cpp:143 (upper-casing) and the std::string editing of Segment::generateSegSequences
(Segment.cpp:210-447); the byte-level pin is the FASTQ parity of tests/test_gpu_parity.py, whose cases
with variants now run through this path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cases
import simuscop_amd
from simuscop_amd import SgContig, SgHapPatch, SgHapPiece, synth

pytestmark = pytest.mark.gpu

CODE = np.full(256, 5, dtype=np.uint8)
for ch, v in ((b"A", 0), (b"C", 1), (b"T", 2), (b"G", 3), (b"N", 4)):
    CODE[ch[0]] = v
    CODE[ch.lower()[0]] = v
CODE[ord("X")] = CODE[ord("x")] = 6   # a literal X is the k-mer trie's place holder (Profile.cpp:94-101): a code of its own


@pytest.fixture(scope="module")
def eng():
    return simuscop_amd.load_engine()


def _ctx(eng):
    ctx = C.c_void_p()
    assert eng.sg_create(C.byref(ctx), 0, 1) == 0, eng.sg_last_error(None)
    return ctx


def _fasta_image(contigs, width, newline=b"\n", final_newline=True):
    """(file bytes, [(header offset, first base offset, sequence)])"""
    out, rows = bytearray(), []
    for i, (name, seq) in enumerate(contigs):
        h = len(out)
        out += b">" + name + newline
        first = len(out)
        for s in range(0, len(seq), width):
            out += seq[s:s + width]
            last = i == len(contigs) - 1 and s + width >= len(seq)
            if not last or final_newline:
                out += newline
        rows.append((h, first, seq))
    return bytes(out), rows


def _upload(eng, ctx, image, chunk=100003):
    assert eng.sg_reference_begin(ctx, len(image)) == 0
    for off in range(0, len(image), chunk):
        part = image[off:off + chunk]
        assert eng.sg_reference_chunk(ctx, off, part, len(part)) == 0
        assert eng.sg_sync(ctx) == 0


@pytest.mark.parametrize("width,newline,final_newline", [(60, b"\n", True), (70, b"\r\n", True), (61, b"\n", False),
                                                         (5, b"\n", True), (100000, b"\n", True)])
def test_ingest_and_assembly_match_numpy(eng, width, newline, final_newline):
    rng = np.random.default_rng(width)
    alphabet = np.frombuffer(b"ACGTacgtNnRY", dtype=np.uint8)
    seqs = [rng.choice(alphabet, size=n, p=[.2, .2, .2, .2, .04, .04, .04, .04, .015, .015, .005, .005]).tobytes()
            for n in (40001, 7, 123456, 1)]
    contigs = [(b"chr%d some description" % i, s) for i, s in enumerate(seqs)]
    image, rows = _fasta_image(contigs, width, newline, final_newline)
    ctx = _ctx(eng)
    try:
        _upload(eng, ctx, image)
        offs = (C.c_uint64 * 16)()
        n, flags = C.c_uint32(), C.c_uint32()
        assert eng.sg_reference_scan(ctx, offs, 16, C.byref(n), C.byref(flags)) == 0
        assert sorted(offs[:n.value]) == [r[0] for r in rows] and flags.value == 0
        tab = (SgContig * len(rows))()
        for i, (_, first, seq) in enumerate(rows):
            tab[i] = SgContig(first, len(seq), width, width + len(newline))
        assert eng.sg_reference_commit(ctx, tab, len(rows)) == 0, eng.sg_last_error(ctx)

        # chains: 0 = contig 2 with a deletion, an insertion and a duplicated tail; 1 = contig 0 verbatim + contig 3
        lit = b"acgtNNx"
        pieces = [(0, 0, 1000, 0, 2, 0), (1000, 1500, 70000, 0, 2, 0), (71000, 0, len(lit), 0, 0, 1),
                  (71000 + len(lit), 71500, 123456 - 71500, 0, 2, 0), (71000 + len(lit) + 123456 - 71500, 100000, 23456, 0, 2, 0),
                  (0, 0, 40001, 1, 0, 0), (40001, 0, 1, 1, 3, 0)]
        model = [np.concatenate([CODE[np.frombuffer(seqs[2][0:1000], np.uint8)], CODE[np.frombuffer(seqs[2][1500:71500], np.uint8)],
                                 CODE[np.frombuffer(lit, np.uint8)], CODE[np.frombuffer(seqs[2][71500:], np.uint8)],
                                 CODE[np.frombuffer(seqs[2][100000:], np.uint8)]]),
                 np.concatenate([CODE[np.frombuffer(seqs[0], np.uint8)], CODE[np.frombuffer(seqs[3], np.uint8)]])]
        lens = (C.c_uint64 * 2)(len(model[0]), len(model[1]))
        patches = [(5, 0, ord("g")), (70999, 0, ord("T")), (len(model[0]) - 1, 0, ord("N")), (40000, 1, ord("c")), (17, 1, ord("*"))]
        for dst, ch, b in patches:
            model[ch][dst] = CODE[b]
        pa = (SgHapPiece * len(pieces))(*[SgHapPiece(*p) for p in pieces])
        pp = (SgHapPatch * len(patches))(*[SgHapPatch(*p) for p in patches])
        rc = eng.sg_build_haplotypes(ctx, 2, lens, pa, len(pieces), lit, len(lit), pp, len(patches))
        assert rc == 0, eng.sg_last_error(ctx)
        for ch in range(2):
            buf = C.create_string_buffer(len(model[ch]))
            assert eng.sg_haplotype_codes(ctx, ch, 0, len(model[ch]), buf) == 0
            got = np.frombuffer(buf.raw, np.uint8)
            bad = np.nonzero(got != model[ch])[0]
            assert bad.size == 0, (ch, bad[:10], got[bad[:10]], model[ch][bad[:10]])
        # pieces that do not tile a chain, or leave their contig, are refused
        short = (SgHapPiece * 1)(SgHapPiece(0, 0, 10, 0, 2, 0))
        assert eng.sg_build_haplotypes(ctx, 2, lens, short, 1, None, 0, None, 0) != 0
        over = (SgHapPiece * 1)(SgHapPiece(0, 123450, 100, 0, 2, 0))
        one = (C.c_uint64 * 1)(100)
        assert eng.sg_build_haplotypes(ctx, 1, one, over, 1, None, 0, None, 0) != 0
        assert b"contig" in eng.sg_last_error(ctx)
    finally:
        eng.sg_destroy(ctx)


def test_commit_refuses_lines_of_several_widths(eng):
    seq = b"ACGT" * 50
    image = b">c1\n" + seq[:60] + b"\n" + seq[60:110] + b"\n" + seq[110:] + b"\n"  # 60, 50, 90
    ctx = _ctx(eng)
    try:
        _upload(eng, ctx, image)
        tab = (SgContig * 1)(SgContig(4, len(seq), 60, 61))
        assert eng.sg_reference_commit(ctx, tab, 1) == 5  # SG_ERR_FORMAT
        assert b"width" in eng.sg_last_error(ctx)
        # comment lines are reported by the scan
        _upload(eng, ctx, b";note\n>c1\nACGT\n")
        offs = (C.c_uint64 * 4)()
        n, flags = C.c_uint32(), C.c_uint32()
        assert eng.sg_reference_scan(ctx, offs, 4, C.byref(n), C.byref(flags)) == 0
        assert n.value == 1 and offs[0] == 6 and flags.value & 1
    finally:
        eng.sg_destroy(ctx)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")
SEED = (cases.FAKE_SEC << 32) | cases.FAKE_NSEC


def _run(cfg, out, *extra):
    r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", out, "--quiet", *extra], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}


@pytest.mark.parametrize("name", ["wgs_pe_variants", "tumor_se_mixture", "wes_pe_targets", "tiny_contigs_pe"])
def test_device_and_host_haplotypes_give_the_same_fastq(name, tmp_path):
    cfg = cases.build_case(name, str(tmp_path))
    dev = _run(cfg, str(tmp_path / "dev"))
    host = _run(cfg, str(tmp_path / "host"), "--host-haplotypes")
    assert dev.keys() == host.keys() and dev
    for f in dev:
        assert dev[f] == host[f], f


def test_fasta_layouts_do_not_change_the_reads(tmp_path):
    """Same genome, different files: 60-column lines, 80-column CRLF lines, no final newline, one line
    per contig, ragged lines and a comment line (the last two take the host parser): identical FASTQ."""
    cfg = cases.build_case("wgs_pe_xten", str(tmp_path))
    fa = [l.split("=")[1].strip() for l in open(cfg) if l.startswith("ref")][0]
    lines = open(fa, "rb").read().split(b"\n")
    name, seq = lines[0], b"".join(lines[1:])
    base = _run(cfg, str(tmp_path / "o0"))

    def rewrite(body):
        with open(fa, "wb") as f:
            f.write(body)
    wrap = lambda w, nl: nl.join(seq[i:i + w] for i in range(0, len(seq), w))
    variants = {
        "crlf80": name + b"\r\n" + wrap(80, b"\r\n") + b"\r\n",
        "no_final_newline": name + b"\n" + wrap(60, b"\n"),
        "one_line": name + b"\n" + seq + b"\n",
        "ragged": name + b"\n" + seq[:1000] + b"\n" + b"\n".join(seq[i:i + 77] for i in range(1000, len(seq), 77)) + b"\n",
        "comment": b";a comment\n" + name + b"\n" + wrap(60, b"\n") + b"\n",
        "blank_tail": name + b"\n" + wrap(60, b"\n") + b"\n\n\n",
    }
    for tag, body in variants.items():
        rewrite(body)
        # (carriage returns are bases and name bytes by default, as in fastahack; --crlf-as-lf reads the LF twin)
        got = _run(cfg, str(tmp_path / tag), *(["--crlf-as-lf"] if tag == "crlf80" else []))
        assert got == base, tag
