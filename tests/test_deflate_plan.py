"""Host side of the block-gzip sink (sg_deflate.cpp), no GPU: both Huffman codes are complete and length
limited, and a member assembled in Python from the plan's prefix + codes -- literals and matches, tokenised by
the rule the device kernels follow -- is accepted by zlib."""
import ctypes as C
import gzip
import zlib

import numpy as np
import pytest

import simuscop_amd

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
         12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


def _lsym(length):
    return max(i for i in range(29) if LBASE[i] <= length)


def _dsym(dist):
    return max(i for i in range(30) if DBASE[i] <= dist)


def tokens(data, lane=64, min_gram=8, min_run=5, per_lane=6):
    """sg_deflate.hip's walk over one member (<= 32 KB): per lane of 64 bytes (aligned to the END of the chunk), greedy;
    candidates are the first occurrence -- at an even position of the chunk's 32 KB frame -- of the position's 8-byte gram,
    taken where the run at the position is shorter than eight, and the previous byte; at most six matches per lane.
    (The device hashes grams into 8192 slots and loses some candidates to collisions; any parse is a valid member.)"""
    n = len(data)
    lead = (lane - n % lane) % lane
    first_of = {}
    for p in range(n - 7):
        if (p + 32768 - n) % 2 == 0:
            first_of.setdefault(data[p:p + 8], p)
    out, p, cur, cnt = [], 0, -1, 0
    while p < n:
        ln = (p + lead) // lane
        if ln != cur:
            cur, cnt = ln, 0
        room = (ln + 1) * lane - lead - p
        length = dist = 0
        if cnt < per_lane:
            lr = 0
            while p > 0 and lr < room and data[p + lr] == data[p + lr - 1]:
                lr += 1
            lh, c = 0, first_of.get(data[p:p + 8]) if p + 8 <= n else None
            if c is not None and c < p:
                while lh < room and data[c + lh] == data[p + lh]:
                    lh += 1
            if lh >= min_gram and lr < 8:
                length, dist = lh, p - c
            elif lr >= min_run:
                length, dist = lr, 1
        if length:
            out.append((length, dist))
            p += length
            cnt += 1
        else:
            out.append(data[p])
            p += 1
    return out


def histograms(toks):
    lit, dist = np.zeros(286, np.int64), np.zeros(30, np.int64)
    for t in toks:
        if isinstance(t, tuple):
            lit[257 + _lsym(t[0])] += 1
            dist[_dsym(t[1])] += 1
        else:
            lit[t] += 1
    lit[256] = 1
    return lit, dist


def _plan(lit, dist):
    eng = simuscop_amd.load_engine()
    c1 = (C.c_uint64 * 286)(*[int(x) for x in lit])
    c2 = (C.c_uint64 * 30)(*[int(x) for x in dist])
    l1, l2 = C.create_string_buffer(286), C.create_string_buffer(30)
    k1, k2, lt = (C.c_uint32 * 286)(), (C.c_uint32 * 30)(), (C.c_uint32 * 260)()
    prefix = (C.c_uint32 * 256)()
    bits = eng.sg_deflate_plan(c1, c2, l1, k1, l2, k2, lt, prefix, 256)
    assert bits > 144
    return dict(lit_len=np.frombuffer(l1.raw, np.uint8).copy(), lit_code=[int(x) for x in k1], dist_len=np.frombuffer(l2.raw, np.uint8).copy(),
                dist_code=[int(x) for x in k2], len_token=[int(x) for x in lt], prefix=[int(x) for x in prefix], bits=bits)


def _member(data, toks, P):
    acc = 0
    for i, wv in enumerate(P["prefix"]):
        acc |= wv << (32 * i)
    n = P["bits"]
    acc &= (1 << n) - 1
    for t in list(toks) + [256]:
        if isinstance(t, tuple):
            length, dist = t
            lt = P["len_token"][length]
            acc |= (lt & 0xFFFFFF) << n
            n += lt >> 24
            ds = _dsym(dist)
            acc |= P["dist_code"][ds] << n
            n += int(P["dist_len"][ds])
            acc |= (dist - DBASE[ds]) << n
            n += DEXT[ds]
        else:
            acc |= P["lit_code"][t] << n
            n += int(P["lit_len"][t])
    body = acc.to_bytes((n + 7) // 8, "little")
    out = bytearray(body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little"))
    out[16:18] = (len(out) - 1).to_bytes(2, "little")   # BSIZE
    return bytes(out)


def _fastq(rng, n_reads):
    """Reads of one window: overlapping substrings of one template, XTen-like quality runs."""
    def pick(alphabet, n):
        return rng.choice(np.frombuffer(alphabet, np.uint8), n).tobytes()
    template = pick(b"ACGT", 700)
    recs = []
    for i in range(n_reads):
        s = int(rng.integers(0, 700 - 151))
        q = pick(b"FFFFFFFFFFFFF<,#", 151)
        recs.append(b"@test#20#%d#%d/1\n" % (100000 + s, i) + template[s:s + 151] + b"\n+\n" + q + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("kind", ["fastq", "uniform", "one_symbol", "fibonacci", "short"])
def test_plan_gives_a_valid_member(kind):
    rng = np.random.default_rng(5)
    if kind == "fastq":
        data = _fastq(rng, 90)
    elif kind == "uniform":
        data = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))
    elif kind == "one_symbol":
        data = b"A" * 3000
    elif kind == "short":
        data = b"ACGTACGTACGTAC"
    else:  # frequencies that make an unlimited Huffman tree deeper than 15
        fib = [1, 1]
        while len(fib) < 40:
            fib.append(fib[-1] + fib[-2])
        data = b"".join(bytes([33 + i]) * min(f, 300) for i, f in enumerate(fib[:30]))
    assert len(data) <= 32768
    toks = tokens(data)
    lit, dist = histograms(toks)
    if kind == "fibonacci":
        lit[33:33 + 40] = fib  # the plan sees the extreme histogram, the data is a bounded sample of it
    P = _plan(lit, dist)
    for lens in (P["lit_len"], P["dist_len"]):
        assert lens.min() >= 1 and lens.max() <= 15
        assert sum(2.0 ** -int(l) for l in lens) == 1.0          # complete codes (zlib rejects anything else)
    m = _member(data, toks, P)
    assert m[:4] == b"\x1f\x8b\x08\x04" and m[12:14] == b"BC"
    assert gzip.decompress(m) == data
    assert gzip.decompress(m + m) == data + data            # members concatenate
    eof = C.create_string_buffer(28)
    assert simuscop_amd.load_engine().sg_bgzf_eof(eof) == 0
    assert gzip.decompress(m + eof.raw) == data
    if kind == "fastq":
        assert len(data) / len(m) > 3.3                         # literal-only Huffman stops at ~2.5 on this text
        assert sum(1 for t in toks if isinstance(t, tuple) and t[1] == 1) > 50      # quality runs
        assert sum(1 for t in toks if isinstance(t, tuple) and t[1] > 150) > 100    # overlapping reads
    if kind == "one_symbol":     # 3000 = 46 * 64 + 56: the first lane holds 56 bytes (a literal + a run of 55), every other one run of 64
        assert toks[:3] == [65, (55, 1), (64, 1)] and len(toks) == 48


def test_a_plan_from_one_text_encodes_another():
    """The code is built from a SAMPLE of the members: symbols the sample never saw still have codes."""
    rng = np.random.default_rng(9)
    P = _plan(*histograms(tokens(_fastq(rng, 40))))
    other = bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) + b"N" * 500 + _fastq(rng, 20)
    assert len(other) <= 32768
    assert gzip.decompress(_member(other, tokens(other), P)) == other


def test_length_tokens_follow_rfc1951():
    P = _plan(np.ones(286, np.int64), np.ones(30, np.int64))
    for length in range(3, 259):
        s = _lsym(length)
        lt = P["len_token"][length]
        nb = int(P["lit_len"][257 + s])
        ext = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0][s]
        assert lt >> 24 == nb + ext
        assert (lt & 0xFFFFFF) == P["lit_code"][257 + s] | ((length - LBASE[s]) << nb)
