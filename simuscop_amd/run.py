"""Multi-GPU front end of simuReads: one process per GPU (torchrun), RCCL only for the small exchanges.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m simuscop_amd.run config.txt [--seed N] [--no-write] [--merge]

Two ways to shard (host/simulate.cpp): by default every (population, chromosome) batch is split over the ranks
by runs of segments balanced on planned fragments (`shard_rank/shard_world`; every rank holds the whole genome);
with --shard-contigs the ranks OWN whole chromosomes (longest-first by length) and ingest, scan and sample only
those -- the per-chromosome GC-weighted lengths of Genome::setReadCounts are all-reduced once per population.  Because every draw is addressed inside the
whole batch, the ranks' part files `<name>_1.fq.part<r>` hold exactly the reads of the 1-GPU run
(tests/test_gpu_parity.py::test_sharded_run_equals_unsharded).  No bulk data moves between GPUs:
the collectives here are a barrier and an all_gather of per-rank statistics.  Each rank evaluates the
read apportioning (Genome::setReadCounts) on the full genome itself, so the weighted-length exchange
of `simuscop_amd.dist` is only needed when ranks own different chromosomes (bench.py does that).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m simuscop_amd.run")
    ap.add_argument("config")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--no-write", action="store_true", help="keep results on the devices (throughput runs)")
    ap.add_argument("--merge", action="store_true", help="rank 0 concatenates the part files per output file")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--force-process-group", action="store_true",
                    help="make a process group even for one rank (rehearsal of the RCCL path on a single GPU)")
    ap.add_argument("--shard-contigs", action="store_true",
                    help="ranks own whole chromosomes: each ingests, scans and samples only its own; the per-chromosome "
                         "weighted lengths are exchanged with one all-reduce per population")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    import simuscop_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("simuscop_amd.run needs MI355X devices: the engine has no CPU path")
    if os.environ.get("SIMUSCOP_SAME_DEVICE"):   # rehearsal on a 1-GPU box (gloo backend only)
        local_rank = 0
    from simuscop_amd import dist as sdist
    torch.cuda.set_device(local_rank)
    pg = sdist.init_process_group(args.backend, local_rank, world, force=args.force_process_group)

    opts = dict(device=local_rank, quiet=0 if rank == 0 else 1, shard_rank=rank, shard_world=world,
                write_files=0 if args.no_write else 1)
    if args.seed is not None:
        opts["seed"] = args.seed
    exchange = None
    if args.shard_contigs and pg:
        exchange = sdist.make_exchange("cuda" if args.backend == "nccl" else None)
        opts["shard_contigs"] = 1
        opts["exchange"] = exchange
    t0 = time.time()
    st = simuscop_amd.run_config(args.config, **opts)
    dt = time.time() - t0
    mine = {"rank": rank, "reads": int(st.reads), "fragments": int(st.fragments), "bytes": int(st.fastq_bytes),
            "seconds": dt, "t_sample": st.t_sample}
    if pg:
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        dist.barrier()
    else:
        allv = [mine]
    if rank == 0:
        total = sum(v["reads"] for v in allv)
        wall = max(v["seconds"] for v in allv)
        print(json.dumps({"ranks": world, "reads": total, "fragments": sum(v["fragments"] for v in allv),
                          "fastq_bytes": sum(v["bytes"] for v in allv), "wall_s": wall,
                          "reads_per_s": total / wall, "per_rank": allv}), flush=True)
        if args.merge and not args.no_write and world > 1:
            out_dir = None
            for line in open(args.config):
                s = line.strip()
                if s.startswith("output") and "=" in s:
                    out_dir = s.split("=", 1)[1].strip()
            names = sorted({f.rsplit(".part", 1)[0] for f in os.listdir(out_dir) if ".part" in f})
            for base in names:
                # NB: parts interleave per chromosome; the merged file holds the same records, chromosome
                # blocks grouped by rank (the reference's record order depends on thread timing anyway)
                with open(os.path.join(out_dir, base), "wb") as dst:
                    for r in range(world):
                        part = os.path.join(out_dir, f"{base}.part{r}")
                        with open(part, "rb") as src:
                            shutil.copyfileobj(src, dst, 1 << 24)
                        os.remove(part)
    if pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
