"""Host side of the block-gzip sink (sg_deflate.cpp), no GPU: the Huffman code is complete and length
limited, and a member assembled in Python from the plan's prefix + codes is accepted by zlib."""
import ctypes as C
import gzip
import zlib

import numpy as np
import pytest

import simuscop_amd


def _plan(counts):
    eng = simuscop_amd.load_engine()
    cnt = (C.c_uint64 * 256)(*[int(x) for x in counts])
    lens = C.create_string_buffer(257)
    codes = (C.c_uint32 * 257)()
    prefix = (C.c_uint32 * 256)()
    bits = eng.sg_deflate_plan(cnt, lens, codes, prefix, 256)
    assert bits > 144
    return np.frombuffer(lens.raw, np.uint8).copy(), np.array(codes[:], dtype=np.uint64), np.array(prefix[:], dtype=np.uint64), bits


def _member(data, lens, codes, prefix, bits):
    acc, n = 0, 0
    for i, wv in enumerate(prefix):
        acc |= int(wv) << (32 * i)
    acc &= (1 << bits) - 1
    n = bits
    for b in list(data) + [256]:
        acc |= int(codes[b]) << n
        n += int(lens[b])
    body = acc.to_bytes((n + 7) // 8, "little")
    out = bytearray(body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little"))
    out[16:18] = (len(out) - 1).to_bytes(2, "little")   # BSIZE
    return bytes(out)


@pytest.mark.parametrize("kind", ["fastq", "uniform", "one_symbol", "fibonacci"])
def test_plan_gives_a_valid_member(kind):
    rng = np.random.default_rng(5)
    if kind == "fastq":
        rec = b"@test#20#12345#77/1\n" + bytes(rng.choice(list(b"ACGT"), 151)) + b"\n+\n" + bytes(rng.choice(list(b"JJJJFA<7-"), 151)) + b"\n"
        data = rec * 40
    elif kind == "uniform":
        data = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))
    elif kind == "one_symbol":
        data = b"A" * 3000
    else:  # frequencies that make an unlimited Huffman tree deeper than 15
        fib = [1, 1]
        while len(fib) < 40:
            fib.append(fib[-1] + fib[-2])
        data = b"".join(bytes([33 + i]) * min(f, 300) for i, f in enumerate(fib[:30]))
    counts = np.bincount(np.frombuffer(data, np.uint8), minlength=256)
    if kind == "fibonacci":
        counts[33:33 + 40] = fib  # the plan sees the extreme histogram, the data is a bounded sample of it
    lens, codes, prefix, bits = _plan(counts)
    assert lens.min() >= 1 and lens.max() <= 15
    assert sum(2.0 ** -int(l) for l in lens) == 1.0          # complete code (zlib rejects anything else)
    m = _member(data, lens, codes, prefix, bits)
    assert m[:4] == b"\x1f\x8b\x08\x04" and m[12:14] == b"BC"
    assert gzip.decompress(m) == data
    assert gzip.decompress(m + m) == data + data            # members concatenate
    eof = C.create_string_buffer(28)
    assert simuscop_amd.load_engine().sg_bgzf_eof(eof) == 0
    assert gzip.decompress(m + eof.raw) == data
