"""Histogram helpers shared by the distributional parity tests."""
import collections

import numpy as np

Z_MAX = 5.0


def _chi2_z(h1, h2):
    """Two-sample chi-square on count vectors with pooled expected frequencies; returns (z, dof)."""
    h1 = np.asarray(h1, dtype=np.float64).ravel()
    h2 = np.asarray(h2, dtype=np.float64).ravel()
    keep = (h1 + h2) >= 20  # pool sparse cells away
    rest1, rest2 = h1[~keep].sum(), h2[~keep].sum()
    a = np.append(h1[keep], rest1)
    b = np.append(h2[keep], rest2)
    if a[-1] + b[-1] == 0:
        a, b = a[:-1], b[:-1]
    n1, n2 = a.sum(), b.sum()
    k1, k2 = np.sqrt(n2 / n1), np.sqrt(n1 / n2)
    chi2 = (((k1 * a - k2 * b) ** 2) / (a + b)).sum()
    dof = len(a) - 1
    return (chi2 - dof) / np.sqrt(2 * dof), dof


def _parse(fq_path, ref_len):
    """Histograms from one FASTQ file."""
    qual_by_cycle = collections.Counter()
    length = collections.Counter()
    starts = collections.Counter()
    n = 0
    with open(fq_path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\n")
            f.readline()
            q = f.readline().rstrip(b"\n")
            assert h.startswith(b"@") and len(s) == len(q)
            n += 1
            length[len(s)] += 1
            pos = int(h.split(b"#")[2])
            starts[pos * 20 // ref_len] += 1
            qa = np.frombuffer(q, dtype=np.uint8)
            L = len(qa)
            cyc = (np.arange(L) * 10 // L)  # 10 cycle bins
            for c, v in zip(cyc[::7], qa[::7]):  # thin: every 7th base keeps cells independent enough
                qual_by_cycle[(int(c), int(v))] += 1
    return n, qual_by_cycle, length, starts


def _vec(counter_a, counter_b):
    keys = sorted(set(counter_a) | set(counter_b))
    return [counter_a.get(k, 0) for k in keys], [counter_b.get(k, 0) for k in keys]


