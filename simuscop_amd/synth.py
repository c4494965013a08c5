"""Seeded synthetic reference genomes (no network: the reference's testData/ref.fa.gz is absent,
`/root/reference/testData/.MISSING_LARGE_BLOBS`, and GRCh38 cannot be fetched).

The generator is a pure function of (seed, contig index, position): every base comes from a
splitmix64 hash, so the same genome is rebuilt bit-for-bit on any box / numpy version.  Shape
follows SURVEY.md section 8(d): per-10-kbp block GC ~ clip(N(0.41, 0.06), 0.2, 0.7), telomere /
centromere-like `N` runs and one soft-masked (lower-case) stretch per contig.
"""
from __future__ import annotations

import os
from typing import Iterable, Sequence, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

# GRCh38 primary assembly lengths (chr1..22, X, Y)
GRCH38_LENGTHS = [
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
    138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
    83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415,
]
GRCH38_NAMES = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _unit(h: np.ndarray) -> np.ndarray:
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synth_contig(length: int, seed: int, contig_index: int = 0, n_runs: bool = True,
                 softmask: bool = True, block: int = 10000, n_islands=None) -> np.ndarray:
    """Return the contig as a uint8 array of ASCII bases.  `n_islands=(period, width)` drops a short
    run of N every `period` bases (assembly gaps): fragments then regularly run into non-ACGT bases."""
    with np.errstate(over="ignore"):
        base_key = _splitmix64(np.array([seed * 1000003 + contig_index * 7919 + 1], dtype=np.uint64))[0]
        nblk = (length + block - 1) // block
        bidx = np.arange(nblk, dtype=np.uint64)
        h1 = _splitmix64(bidx ^ base_key)
        h2 = _splitmix64(h1)
        # Box-Muller on two hashed uniforms -> block GC fraction
        u1 = np.maximum(_unit(h1), 1e-12)
        u2 = _unit(h2)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        gc = np.clip(0.41 + 0.06 * z, 0.2, 0.7)
        out = np.empty(length, dtype=np.uint8)
        chunk = 1 << 22
        pos_key = _splitmix64(np.array([base_key ^ np.uint64(0xA5A5A5A5)], dtype=np.uint64))[0]
        for s in range(0, length, chunk):
            e = min(length, s + chunk)
            p = np.arange(s, e, dtype=np.uint64)
            u = _unit(_splitmix64(p ^ pos_key))
            g = gc[(p // np.uint64(block)).astype(np.int64)]
            # A:(1-g)/2  C:g/2  G:g/2  T:(1-g)/2
            t1 = (1.0 - g) * 0.5
            t2 = t1 + g * 0.5
            t3 = t2 + g * 0.5
            code = (u >= t1).astype(np.uint8) + (u >= t2).astype(np.uint8) + (u >= t3).astype(np.uint8)
            out[s:e] = np.frombuffer(b"ACGT", dtype=np.uint8)[code]
    if n_runs and length >= 20000:
        tel = min(10000, length // 100)
        out[:tel] = ord("N")
        out[length - tel:] = ord("N")
        cen = length // 3
        out[cen:cen + min(50000, length // 50)] = ord("N")
    if n_islands:
        period, width = n_islands
        for s0 in range(period // 2, length - width, period):
            out[s0:s0 + width] = ord("N")
    if softmask and length >= 5000:
        s = length // 2
        e = s + min(2000, length // 20)
        out[s:e] |= 0x20  # lower-case
    return out


def write_fasta(path: str, contigs: Sequence[Tuple[str, int]], seed: int, line_len: int = 60,
                **kw) -> None:
    """Write a multi-contig FASTA (fixed line length, as .fai requires)."""
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        for ci, (name, length) in enumerate(contigs):
            f.write(b">" + name.encode() + b"\n")
            seq = synth_contig(length, seed, ci, **kw)
            full = (length // line_len) * line_len
            if full:
                body = seq[:full].reshape(-1, line_len)
                lines = np.empty((body.shape[0], line_len + 1), dtype=np.uint8)
                lines[:, :line_len] = body
                lines[:, line_len] = 10
                f.write(lines.tobytes())
            if full < length:
                f.write(seq[full:].tobytes() + b"\n")
    os.replace(tmp, path)
    # a stale index from another genome must not survive (the reference trusts any .fai it finds)
    if os.path.exists(path + ".fai"):
        os.remove(path + ".fai")


def grch38_contigs(scale: float = 1.0) -> list:
    return [(n, max(1000, int(l * scale))) for n, l in zip(GRCH38_NAMES, GRCH38_LENGTHS)]


def variation_rows(popu: str, chr_name: str, scale: float) -> list:
    """Rows of a variation file for one (population, chromosome): the pattern of the reference's
    testData/variations.txt (insertions, deletions, SNVs, CNVs on a 63 Mbp chromosome) with the positions x scale."""
    def P(x):
        return max(1, int(x * scale))
    rows = []
    ins = [(4500100, "tcgagtcg", "homo"), (11000100, "tcgagtc", "homo"), (12000100, "tcgagt", "het"),
           (44000100, "tcgagtcg", "het"), (57000100, "tcgagtc", "homo"), (61000100, "tcgagt", "het")]
    dels = [(3000100, 10, "homo"), (5000100, 9, "het"), (9500100, 8, "homo"), (48000100, 8, "het"),
            (58000100, 7, "homo"), (62000100, 6, "het")]
    snvs = [(2000100, "a", "T", "homo"), (4000100, "T", "G", "homo"), (8500100, "A", "G", "het"),
            (9000100, "G", "C", "homo"), (11500100, "G", "C", "homo"), (46000100, "c", "T", "het"),
            (51000100, "g", "A", "het"), (53000100, "C", "G", "het"), (55000100, "C", "T", "homo"),
            (56000100, "A", "T", "homo"), (59000100, "A", "T", "homo")]
    cnvs = [(5000000, 6000000, 1, 1), (10000000, 14500000, 3, 2), (37500000, 40000000, 4, 3),
            (40000000, 43500000, 1, 1), (44000000, 47000000, 2, 2), (49000000, 53000000, 1, 1)]
    for p, s, t in ins:
        rows.append(f"i\t{popu}\t{chr_name}\t{P(p)}\t{s}\t{t}")
    for p, l, t in dels:
        rows.append(f"d\t{popu}\t{chr_name}\t{P(p)}\t{l}\t{t}")
    for p, r, a, t in snvs:
        rows.append(f"s\t{popu}\t{chr_name}\t{P(p)}\t{r}\t{a}\t{t}")
    for s, e, cn, m in cnvs:
        rows.append(f"c\t{popu}\t{chr_name}\t{P(s)}\t{P(e)}\t{cn}\t{m}")
    return rows


def snp_rows(chr_name: str, length: int, every: int, seed: int) -> list:
    """6-column SNP rows (the reference's testData/snp.txt format), both strands, one every `every`..2*`every` bases."""
    rows = []
    x = seed
    pos = 137
    i = 0
    while pos < length - 200:
        x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        a, b = [("A", "C"), ("C", "T"), ("A", "G"), ("G", "T"), ("C", "G"), ("A", "T")][(x >> 33) % 6]
        strand = "+" if (x >> 40) & 1 else "-"
        ref = a if (x >> 41) & 1 else b
        rows.append(f"rs{i}\t{chr_name}\t{pos}\t{a}/{b}\t{strand}\t{ref}")
        pos += every + int((x >> 45) % every)
        i += 1
    return rows
