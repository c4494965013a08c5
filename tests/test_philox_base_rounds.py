"""What fewer rounds could break.  The per-base stream (KIND_BASE: 97 % of all Philox calls) runs Philox4x32 at
kBaseRounds = 7 rounds (oracle/philox.h, sg_device.h) -- the published Crush-resistant minimum -- where every other stream
keeps ten.  Too few rounds show first as dependence between NEIGHBOURING counters, and the BASE addressing puts neighbours
next to each other everywhere: slot and slot + 1 (c0: adjacent fragments), call and call + 1 (c1: the next four bases of one
read), heads and tails (c2 = 0 / 1: the two halves of one base's draws).  This test aims at exactly that, at 7 rounds, with
the counter layout the engine uses; 5 rounds FAIL it (checked below, so that the test is known to have teeth).
CPU only; numpy restatement of the round function, tied to oracle/philox.h by known answers."""
import ctypes

import numpy as np
import pytest

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
KIND_BASE = 6
SEED = 0x5EED0000


def philox(rounds, c0, c1, c2, c3, k0=SEED & 0xFFFFFFFF, k1=SEED >> 32):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in np.broadcast_arrays(c0, c1, c2, c3))
    for _ in range(rounds):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0), p1 & MASK, (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1), p0 & MASK
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3]).astype(np.uint32)   # [word][sample]


def c3_of(mate, batch):
    return KIND_BASE | ((((mate & 1) << 23) | (batch & 0xFFFF)) << 8)


def test_numpy_round_function_is_the_oracles(oracle_lib):
    u32p = ctypes.POINTER(ctypes.c_uint32)
    oracle_lib.orc_philox4x32_r.argtypes = [ctypes.c_int, u32p, u32p, u32p]
    rng = np.random.default_rng(1)
    for rounds in (5, 7, 10):
        for _ in range(20):
            c = rng.integers(0, 2**32, 4, dtype=np.uint64)
            k = rng.integers(0, 2**32, 2, dtype=np.uint64)
            out = (ctypes.c_uint32 * 4)()
            oracle_lib.orc_philox4x32_r(rounds, (ctypes.c_uint32 * 4)(*map(int, c)), (ctypes.c_uint32 * 2)(*map(int, k)), out)
            mine = philox(rounds, c[0], c[1], c[2], c[3], int(k[0]), int(k[1]))
            assert [int(x) for x in mine.reshape(4)] == list(out)


def _stats(rounds, n=1 << 21):
    """z-scores (each ~N(0,1) under independence) of neighbour dependence in the BASE stream's own addressing."""
    z = {}
    slots = np.arange(n, dtype=np.uint64) + np.uint64(12345)
    c3 = c3_of(0, 3)
    a = philox(rounds, slots, 7, 0, c3)                      # heads of call 7 of slots s
    nb = {"slot+1": philox(rounds, slots + np.uint64(1), 7, 0, c3), "call+1": philox(rounds, slots, 8, 0, c3),
          "tails": philox(rounds, slots, 7, 1, c3), "mate 2": philox(rounds, slots, 7, 0, c3_of(1, 3))}
    for name, b in nb.items():
        x = a ^ b
        # avalanche: every output bit flips with probability 1/2 when the counter moves to its neighbour
        for w in range(4):
            ones = np.array([int(((x[w] >> np.uint32(bit)) & np.uint32(1)).sum()) for bit in range(32)])
            z[f"flip {name} w{w}"] = float(np.abs((ones - n / 2) / np.sqrt(n / 4)).max())
        # the 16-bit heads the sampler compares (high halves: substitution, low halves: quality) are uncorrelated
        for w in range(4):
            for half, sh in (("hi", 16), ("lo", 0)):
                u = ((a[w] >> np.uint32(sh)) & np.uint32(0xFFFF)).astype(np.float64)
                v = ((b[w] >> np.uint32(sh)) & np.uint32(0xFFFF)).astype(np.float64)
                z[f"corr {name} w{w} {half}"] = float(abs(np.corrcoef(u, v)[0, 1]) * np.sqrt(n))
        # joint law of the top bytes of a word and its neighbour's: 65,536 cells, chi-square against the uniform
        cell = ((a[0] >> np.uint32(24)).astype(np.int64) << 8) | (b[0] >> np.uint32(24)).astype(np.int64)
        cnt = np.bincount(cell, minlength=65536)
        chi = float(((cnt - n / 65536.0) ** 2 / (n / 65536.0)).sum())
        z[f"joint {name}"] = abs(chi - 65535.0) / np.sqrt(2 * 65535.0)
    # one word: its 16-bit head is uniform (65,536 cells), and the four words of a call are pairwise uncorrelated
    for w in range(4):
        cnt = np.bincount((a[w] >> np.uint32(16)).astype(np.int64), minlength=65536)
        chi = float(((cnt - n / 65536.0) ** 2 / (n / 65536.0)).sum())
        z[f"uniform head w{w}"] = abs(chi - 65535.0) / np.sqrt(2 * 65535.0)
    for w in range(4):
        for v in range(w + 1, 4):
            z[f"corr words {w},{v}"] = float(abs(np.corrcoef(a[w].astype(np.float64), a[v].astype(np.float64))[0, 1]) * np.sqrt(n))
    # serial: word w of slot s against the same word of slots s + d (adjacent fragments of one GC window)
    for d in (1, 2, 63, 64):
        u, v = a[0][:-d].astype(np.float64), a[0][d:].astype(np.float64)
        z[f"serial lag {d}"] = float(abs(np.corrcoef(u, v)[0, 1]) * np.sqrt(n - d))
    return z


def test_seven_rounds_show_no_dependence_between_neighbouring_counters():
    z = _stats(7)
    # ~150 statistics, each ~|N(0,1)| (the maxima over 32 bits a little more): the bar of the histogram families, |z| < 5
    worst = max(z, key=z.get)
    assert z[worst] < 5.0, (worst, z[worst])


def test_the_statistics_have_teeth_at_too_few_rounds():
    """The same statistics on a 3-round Philox fail by a wide margin: the test would see a stream that mixes too little."""
    z = _stats(3, n=1 << 18)
    assert max(z.values()) > 20.0


@pytest.mark.parametrize("rounds", [7, 10])
def test_base_draws_as_the_sampler_uses_them(rounds):
    """The decision variable of nearly every base: `head < keep` with keep ~ 0.999 * 2^16.  Substitution candidates (head >=
    keep) of neighbouring bases -- words 0..3 of one call, then the next call -- must be independent events: the number
    of positions where base i AND base i+1 are candidates has its binomial mean."""
    n = 1 << 21
    slots = np.arange(n, dtype=np.uint64)
    words = np.concatenate([philox(rounds, slots, c, 0, c3_of(0, 1)) for c in (0, 1)])   # eight consecutive bases per slot
    keep = 0xF000   # a candidate one time in sixteen: the dependence a 1/1000 event would hide shows 250 times sooner
    cand = (words >> np.uint32(16)) >= np.uint32(keep)
    p = 1.0 / 16
    for i in range(7):
        both = int((cand[i] & cand[i + 1]).sum())
        zz = (both - n * p * p) / np.sqrt(n * p * p * (1 - p * p))
        assert abs(zz) < 5.0, (rounds, i, zz)
