"""Block-gzip sink on the device: the BGZF members sg_compress produces decompress
(zlib, member by member and as a whole) to exactly the FASTQ text sg_fetch returns; BGZF structure."""
import ctypes as C
import gzip
import os
import struct
import subprocess
import zlib

import pytest

import cases
import simuscop_amd

pytestmark = pytest.mark.gpu


def _members(blob):
    """split a BGZF stream by the BSIZE fields"""
    out, p = [], 0
    while p < len(blob):
        assert blob[p:p + 4] == b"\x1f\x8b\x08\x04" and blob[p + 10:p + 12] == b"\x06\x00" and blob[p + 12:p + 16] == b"BC\x02\x00", p
        size = struct.unpack_from("<H", blob, p + 16)[0] + 1
        out.append(blob[p:p + size])
        p += size
    assert p == len(blob)
    return out


@pytest.mark.parametrize("name", ["wgs_pe_xten", "wgs_pe_variants", "short_reads_se", "tiny_contigs_pe"])
def test_compressed_text_round_trips(name, tmp_path):
    cfg = cases.build_case(name, str(tmp_path))
    sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=77)
    try:
        sess.weighted_length()
        sess.set_reads(sess.planned_reads)
        done = 0
        for chrom in range(sess.n_chromosomes):
            if not sess.prepare_batch(chrom):
                continue
            sess.sample()
            b1, b2, nf = sess.result()
            t1, t2 = sess.fetch(b1, b2)
            g1, g2 = sess.compress()
            for mate, text, gz in ((0, t1, g1), (1, t2, g2)):
                if not text:
                    assert gz == 0
                    continue
                blob = sess.fetch_compressed(mate, gz)
                assert gzip.decompress(blob) == text, (name, chrom, mate)
                mem = _members(blob)
                assert len(mem) == (len(text) + 32767) // 32768
                pos = 0
                for m in mem:   # every member on its own: raw DEFLATE body, CRC-32, ISIZE
                    body = zlib.decompressobj(-15).decompress(m[18:-8])
                    assert body == text[pos:pos + 32768]
                    assert struct.unpack("<II", m[-8:]) == (zlib.crc32(body) & 0xFFFFFFFF, len(body))
                    pos += len(body)
                assert pos == len(text)
                if len(text) > 200000:
                    # literal codes alone give 2.5x (XTen) .. 2.0x (40-symbol qualities); the copies add what the coverage
                    # offers (5x here: ~3.2x on the XTen case, 3.8x at 30x)
                    assert len(text) / gz > (2.9 if name == "wgs_pe_xten" else 1.8), len(text) / gz
                    lens = [len(zlib.decompressobj(-15).decompress(m[18:-8])) for m in mem[:4]]
                    assert lens == [32768] * len(lens)
            done += 1
        assert done
    finally:
        sess.close()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")


@pytest.mark.parametrize("name", ["wgs_pe_variants", "tumor_se_mixture"])
def test_cli_gzip_files_equal_plain_files(name, tmp_path):
    cfg = cases.build_case(name, str(tmp_path))
    for tag, extra in (("plain", []), ("gz", ["--gzip"])):
        r = subprocess.run([SIMU, cfg, "--seed", "5", "--out", str(tmp_path / tag), "--quiet", *extra], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    plain = sorted(os.listdir(tmp_path / "plain"))
    assert sorted(os.listdir(tmp_path / "gz")) == [f + ".gz" for f in plain] and plain
    for f in plain:
        text = open(tmp_path / "plain" / f, "rb").read()
        blob = open(tmp_path / "gz" / (f + ".gz"), "rb").read()
        assert gzip.decompress(blob) == text, f
        assert blob.endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
        zc = subprocess.run(["gzip", "-dc", str(tmp_path / "gz" / (f + ".gz"))], capture_output=True, timeout=120)
        assert zc.returncode == 0 and zc.stdout == text
