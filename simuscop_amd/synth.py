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


def _contig_keys(length: int, seed: int, contig_index: int, block: int):
    """Per-contig hash keys and the GC fraction of every `block`-base block."""
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
        pos_key = _splitmix64(np.array([base_key ^ np.uint64(0xA5A5A5A5)], dtype=np.uint64))[0]
    return gc, pos_key


def synth_range(length: int, seed: int, contig_index: int, lo: int, hi: int, n_runs: bool = True, softmask: bool = True,
                block: int = 10000, n_islands=None, _keys=None) -> np.ndarray:
    """Bases [lo, hi) of the contig `synth_contig(length, seed, contig_index, ...)` as ASCII: every base is a function of
    (seed, contig index, position), so any range can be made on its own (write_fasta makes them on all cores)."""
    gc, pos_key = _keys if _keys is not None else _contig_keys(length, seed, contig_index, block)
    out = np.empty(hi - lo, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    chunk = 1 << 22
    with np.errstate(over="ignore"):
        for s in range(lo, hi, chunk):
            e = min(hi, s + chunk)
            p = np.arange(s, e, dtype=np.uint64)
            u = _unit(_splitmix64(p ^ pos_key))
            g = gc[(p // np.uint64(block)).astype(np.int64)]
            # A:(1-g)/2  C:g/2  G:g/2  T:(1-g)/2
            t1 = (1.0 - g) * 0.5
            t2 = t1 + g * 0.5
            t3 = t2 + g * 0.5
            code = (u >= t1).astype(np.uint8) + (u >= t2).astype(np.uint8) + (u >= t3).astype(np.uint8)
            out[s - lo:e - lo] = lut[code]

    def paint(a, b, fn):   # [a, b) of the contig, clipped to the range
        a, b = max(a, lo), min(b, hi)
        if a < b:
            fn(out[a - lo:b - lo])

    def to_n(v):
        v[:] = ord("N")

    def lower(v):
        v |= 0x20

    if n_runs and length >= 20000:
        tel = min(10000, length // 100)
        paint(0, tel, to_n)
        paint(length - tel, length, to_n)
        cen = length // 3
        paint(cen, cen + min(50000, length // 50), to_n)
    if n_islands:
        period, width = n_islands
        first = period // 2
        k0 = max(0, (lo - width - first) // period)
        for s0 in range(first + k0 * period, min(length - width, hi), period):
            paint(s0, s0 + width, to_n)
    if softmask and length >= 5000:
        s = length // 2
        paint(s, s + min(2000, length // 20), lower)
    return out


def synth_contig(length: int, seed: int, contig_index: int = 0, n_runs: bool = True,
                 softmask: bool = True, block: int = 10000, n_islands=None) -> np.ndarray:
    """Return the contig as a uint8 array of ASCII bases.  `n_islands=(period, width)` drops a short
    run of N every `period` bases (assembly gaps): fragments then regularly run into non-ACGT bases."""
    return synth_range(length, seed, contig_index, 0, length, n_runs=n_runs, softmask=softmask, block=block, n_islands=n_islands)


def _fasta_piece(job):
    """One piece of a contig's body -- whole lines -- written at its place in the file (worker of write_fasta)."""
    path, offset, length, seed, ci, lo, hi, line_len, kw = job
    seq = synth_range(length, seed, ci, lo, hi, **kw)
    n = hi - lo
    full = (n // line_len) * line_len
    parts = []
    if full:
        lines = np.empty((full // line_len, line_len + 1), dtype=np.uint8)
        lines[:, :line_len] = seq[:full].reshape(-1, line_len)
        lines[:, line_len] = 10
        parts.append(lines.tobytes())
    if full < n:   # the contig's last, shorter line
        parts.append(seq[full:].tobytes() + b"\n")
    fd = os.open(path, os.O_WRONLY)
    try:
        os.pwrite(fd, b"".join(parts), offset)
    finally:
        os.close(fd)
    return n


def write_fasta(path: str, contigs: Sequence[Tuple[str, int]], seed: int, line_len: int = 60, workers: int = 0,
                **kw) -> None:
    """Write a multi-contig FASTA (fixed line length, as .fai requires).  Genomes of more than 64 Mbp are made by
    `workers` processes (default: the host's cores, at most 32), each writing whole pieces of 8 M bases at their offsets."""
    tmp = path + ".tmp"
    piece = (8 << 20) // line_len * line_len
    jobs, headers, off = [], [], 0
    for ci, (name, length) in enumerate(contigs):
        hdr = b">" + name.encode() + b"\n"
        headers.append((off, hdr))
        off += len(hdr)
        for lo in range(0, length, piece):
            hi = min(length, lo + piece)
            jobs.append((tmp, off + lo + lo // line_len, length, seed, ci, lo, hi, line_len, kw))
        off += length + (length + line_len - 1) // line_len
    with open(tmp, "wb") as f:
        f.truncate(off)
        for o, hdr in headers:
            f.seek(o)
            f.write(hdr)
    total = sum(l for _, l in contigs)
    if workers <= 0:
        workers = min(32, os.cpu_count() or 1)
    if total <= (64 << 20) or workers == 1 or len(jobs) == 1:
        for j in jobs:
            _fasta_piece(j)
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:   # (before any GPU work of the caller)
            for _ in pool.imap_unordered(_fasta_piece, jobs, chunksize=1):
                pass
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


def write_fai(path: str, contigs: Sequence[Tuple[str, int]], line_len: int = 60) -> None:
    """The samtools-style index of a FASTA written by write_fasta (names without the `chr` prefix, as the reference's
    fastahack keys them)."""
    off, rows = 0, []
    for name, length in contigs:
        off += len(name) + 2
        rows.append(f"{name[3:] if name.startswith('chr') else name}\t{length}\t{off}\t{line_len}\t{line_len + 1}\n")
        off += length + (length + line_len - 1) // line_len
    with open(path + ".fai.tmp", "w") as f:
        f.writelines(rows)
    os.replace(path + ".fai.tmp", path + ".fai")


def main(argv=None) -> None:
    """python -m simuscop_amd.synth OUT.fa --scale S [--seed N] [--workers W]: the 24-contig genome in GRCh38 proportions
    (bench.py's C3 / C4 workloads) and its .fai -- a program of its own so that a caller that has initialised the GPU can
    have it made by a child process."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=67)
    ap.add_argument("--workers", type=int, default=0)
    a = ap.parse_args(argv)
    contigs = grch38_contigs(a.scale)
    write_fasta(a.out, contigs, seed=a.seed, workers=a.workers)
    write_fai(a.out, contigs)


if __name__ == "__main__":
    main()
