"""Multi-GPU read-count balancing: the one exchange step of the sharded path.

Each rank owns whole chromosomes (or runs of segments).  The reference apportions reads to
chromosomes by GC-weighted length (Genome::setReadCounts, lib/genome/Genome.cpp:783-825):
`chrReads = reads * (chrWL / WL)` truncated, the last chromosome taking the remainder.  WL is a sum
over ALL chromosomes, so ranks all_gather their per-chromosome weighted lengths (a few fp64 per rank;
RCCL over xGMI when the backend is "nccl", gloo in the CPU tests) and then every rank evaluates the
same fp64 expression in the same order -- an all_reduce(sum) would change the summation order and
with it, occasionally, a truncated read count.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def apportion(reads: int, chr_wl: Sequence[float]) -> List[int]:
    """Genome.cpp:787-811 for one population: reads per chromosome, in chromosome order."""
    WL = 0.0
    for w in chr_wl:
        WL += w
    out, cur = [], 0
    for i, w in enumerate(chr_wl):
        if i < len(chr_wl) - 1:
            n = int(reads * (w / WL))
        else:
            n = reads - cur
        out.append(n)
        cur += n
    return out


def balance_reads(my_wl: float, my_target_len: int, coverage: int, read_length: int, device=None) -> Tuple[int, int]:
    """Returns (my_reads, total_reads) for a rank that owns ONE chromosome (rank order = chromosome
    order).  Uses torch.distributed when initialised, else behaves as a single rank."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        rank, world = dist.get_rank(), dist.get_world_size()
        t = torch.tensor([my_wl, float(my_target_len)], dtype=torch.float64, device=device)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        wls = [float(p[0]) for p in parts]
        total_len = int(sum(int(p[1]) for p in parts))
    else:
        rank, wls, total_len = 0, [my_wl], my_target_len
    reads = total_len * coverage // read_length  # Genome.cpp:831
    per_chr = apportion(reads, wls)
    return per_chr[rank], reads


def make_exchange(device=None):
    """simu_options.exchange for the chromosome-sharded mode: an in-place all-reduce(sum) of the per-chromosome
    weighted lengths over torch.distributed (RCCL when the backend is "nccl", gloo in the CPU tests).  Every entry has
    exactly one non-zero contributor -- the rank that owns the chromosome -- so the sum is exact in any order.
    Returns the ctypes callback (keep a reference to it while the run lasts)."""
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    from . import EXCHANGE_FN

    def _exchange(_user, values, n):
        try:
            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
                return 0
            arr = np.ctypeslib.as_array(values, shape=(n,))
            t = torch.from_numpy(arr.copy()).to(device) if device is not None else torch.from_numpy(arr.copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            arr[:] = t.cpu().numpy()
            return 0
        except Exception:   # a Python exception must not unwind through the C++ caller
            import traceback
            traceback.print_exc()
            return 1

    return EXCHANGE_FN(_exchange)
