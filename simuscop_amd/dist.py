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

import os
from typing import List, Sequence, Tuple


def pg_timeout_s() -> int:
    """Seconds a collective may wait for its peers before the group gives up (SIMUSCOP_PG_TIMEOUT_S, default 120): a rank
    that died must take the others out, not leave them in a collective until the launcher's own limit."""
    return int(os.environ.get("SIMUSCOP_PG_TIMEOUT_S", "120"))


def init_process_group(backend: str, local_rank: int, world: int, force: bool = False) -> bool:
    """The one place a process group is made (bench.py, bench_c3.py, simuscop_amd.run).  `backend` "nccl" is RCCL on ROCm
    (device tensors, communicator bound to cuda:`local_rank`), anything else (gloo) is for CPU rehearsals.  With one rank
    no group is needed; `force` (or SIMUSCOP_FORCE_PG=1) makes one anyway so that the collectives of the sharded path run
    through the real transport on a single leased GPU (tests/test_gpu_nccl_world1.py).  Every group has a timeout.
    Returns whether a group exists."""
    import datetime

    import torch
    import torch.distributed as dist

    force = force or os.environ.get("SIMUSCOP_FORCE_PG") == "1"
    if world <= 1 and not force:
        return False
    if dist.is_initialized():
        return True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = dict(timeout=datetime.timedelta(seconds=pg_timeout_s()))
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ or "MASTER_PORT" not in os.environ:
        # a single process without a launcher: a one-rank group on a free local port
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        kw.update(init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
    else:
        dist.init_process_group(backend, **kw)
    return True


# collectives this process ran through torch.distributed, by name (bench.py reports them: evidence that the transport ran)
COLLECTIVES = {"all_gather": 0, "all_reduce": 0}


def group_active() -> bool:
    """True when collectives should run: a process group exists (also a one-rank group made with `force`)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


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

    if group_active():   # (a one-rank group too: the all_gather then runs through the transport alone)
        rank, world = dist.get_rank(), dist.get_world_size()
        t = torch.tensor([my_wl, float(my_target_len)], dtype=torch.float64, device=device)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        COLLECTIVES["all_gather"] += 1
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
            if not group_active():
                return 0
            arr = np.ctypeslib.as_array(values, shape=(n,))
            t = torch.from_numpy(arr.copy()).to(device) if device is not None else torch.from_numpy(arr.copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            COLLECTIVES["all_reduce"] += 1
            arr[:] = t.cpu().numpy()
            return 0
        except Exception:   # a Python exception must not unwind through the C++ caller
            import traceback
            traceback.print_exc()
            return 1

    return EXCHANGE_FN(_exchange)
