"""bench.py --workload c3: strong scaling of ONE multi-contig genome, end to end.

Workload (BASELINE.json configs[3], SURVEY.md 8(d) "C3"): 24 contigs with the GRCh38 primary-assembly lengths x
`--scale` (1.0 = 3.1 Gbp, ~308 M pairs; default 0.1), HiSeqXTen profile (151 bp), PE, 30x, insertSize 350.  A step is
the WHOLE run of `simuReads` on that genome -- reference ingest, haplotype assembly, GC scan and read apportioning,
sampling of every chromosome -- with the FASTQ text left in HBM.  With N ranks the chromosomes are owned by ranks
(longest first by length, host/variants.cpp Genome::assign_contigs): a rank ingests, scans and samples only its own;
the one exchange is the all-reduce of the per-chromosome GC-weighted lengths (RCCL; 24 doubles per population).
value = pairs of the whole genome / slowest rank's wall time per step.

`--workload c4` (BASELINE.json configs[4], SURVEY.md 8(d) "C4"): the same genome, four populations clone1, clone2, clone3,
normal with the reference's testData/variations.txt pattern on every contig of the three clones, a SNP every ~2 kb,
abundance 0.3 / 0.25 / 0.35 / 0.1, 60x: 96 (population, chromosome) batches, haplotypes with insertions, deletions, SNVs
and copy-number changes assembled on the device.
"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _genome(scale, rank, barrier):
    """The synthetic genome, made once per box (rank 0) and kept under /tmp; a .fai is written next to it."""
    from simuscop_amd import synth
    contigs = synth.grch38_contigs(scale)
    path = f"/tmp/simuscop_c3_scale{scale:g}_seed67.fa"
    if rank == 0 and not (os.path.exists(path) and os.path.exists(path + ".fai")):
        synth.write_fasta(path, contigs, seed=67)
        off, rows = 0, []
        for name, length in contigs:   # 60 bases per line (synth.write_fasta)
            off += len(name) + 2
            rows.append(f"{name[3:] if name.startswith('chr') else name}\t{length}\t{off}\t60\t61\n")
            off += length + (length + 59) // 60
        with open(path + ".fai.tmp", "w") as f:
            f.writelines(rows)
        os.replace(path + ".fai.tmp", path + ".fai")
    barrier()
    return path, contigs


def _tumour_files(scale, contigs, rank, barrier):
    """Variation, SNP and abundance files of the C4 mixture (made once per box by rank 0)."""
    from simuscop_amd import synth
    base = f"/tmp/simuscop_c4_scale{scale:g}"
    paths = {k: f"{base}_{k}.txt" for k in ("variation", "snp", "abundance")}
    if rank == 0 and not all(os.path.exists(p) for p in paths.values()):
        rows, snps = [], []
        for i, (name, length) in enumerate(contigs):
            key = name[3:] if name.startswith("chr") else name
            for popu in ("clone1", "clone2", "clone3"):
                rows += synth.variation_rows(popu, key, length / 63025520.0)
            snps += synth.snp_rows(key, length, 1500, 100 + i)
        for k, data in (("variation", rows), ("snp", snps), ("abundance", ["0.3\t0.25\t0.35\t0.1"])):
            with open(paths[k] + ".tmp", "w") as f:
                f.write("\n".join(data) + "\n")
            os.replace(paths[k] + ".tmp", paths[k])
    barrier()
    return paths


def main(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if os.environ.get("BENCH_SAME_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import bench
    import simuscop_amd
    from simuscop_amd import dist as sdist

    def barrier():
        if world > 1:
            dist.barrier()

    fasta, contigs = _genome(args.scale, rank, barrier)
    prof_file, L = bench.PROFILES[args.profile]
    tumour = args.workload == "c4"
    coverage = args.coverage if args.coverage is not None else (60 if tumour else 30)
    cfg = f"/tmp/simuscop_{args.workload}_config_r{rank}.txt"
    bench.write_config(cfg, fasta, f"/tmp/simuscop_{args.workload}_out_r{rank}", coverage=coverage, threads=min(64, os.cpu_count() or 1),
                       profile=prof_file)
    if tumour:
        extra = _tumour_files(args.scale, contigs, rank, barrier)
        with open(cfg) as f:
            text = f.read().replace("name = sim\n", "name = clone1, clone2, clone3, normal\n")
        with open(cfg, "w") as f:
            f.write(text + f"variation = {extra['variation']}\nsnp = {extra['snp']}\nabundance = {extra['abundance']}\n")
    opts = dict(device=local_rank, quiet=1, write_files=0, seed=0x5EED0C3, shard_rank=rank, shard_world=world)
    exchange = None
    if world > 1:
        exchange = sdist.make_exchange("cuda" if args.backend == "nccl" else None)
        opts.update(shard_contigs=1, exchange=exchange)

    def step():
        return simuscop_amd.run_config(cfg, **opts)

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frags = 0
    last = None
    for _ in range(args.steps):
        last = step()
        frags += int(last.fragments)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    phases = {k: getattr(last, k) for k in ("t_total", "t_load", "t_engine", "t_reference", "t_haplotypes", "t_plan", "t_sample", "t_hap_device")}
    kernel_ms = {n: float(last.kernel_ms[i]) for i, n in enumerate(simuscop_amd.SG_K_NAMES)}
    mine = {"rank": rank, "dt": dt, "pairs": frags, "bytes": int(last.fastq_bytes), "phases": phases, "kernel_ms": kernel_ms}
    if world > 1:
        allv = [None] * world
        dist.all_gather_object(allv, mine)
    else:
        allv = [mine]
    if rank == 0:
        dt_max = max(v["dt"] for v in allv)
        total_pairs = sum(v["pairs"] for v in allv)
        pairs_per_step = total_pairs / args.steps
        # dominant kernel: emit_fast_kernel, summed over the chromosomes of the slowest rank's last run
        slow = max(allv, key=lambda v: v["kernel_ms"]["emit"])
        bpp = 2 * L + slow["bytes"] / max(slow["pairs"] / args.steps, 1)
        achieved = (slow["pairs"] / args.steps) * bpp / (slow["kernel_ms"]["emit"] * 1e-3) / 1e9
        out = {
            "metric": "simulated paired reads/sec (whole node) at 30x WGS PE150",
            "value": total_pairs / dt_max, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("C4: four populations (variations + SNPs, abundance 0.3/0.25/0.35/0.1) of " if tumour else "C3: ") +
                                   f"24 contigs, GRCh38 primary lengths x {args.scale:g} ({sum(l for _, l in contigs)} bp), "
                                   f"{prof_file[:-8]} profile ({L} bp), PE, {coverage}x, insertSize 350; whole simuReads run per step "
                                   f"(ingest + haplotypes + GC scan + apportioning + sampling), text left in HBM",
                       "pairs_per_step": pairs_per_step,
                       "parallelism": f"{world} rank(s), whole chromosomes per rank (longest first); all-reduce of 24 weighted lengths"
                                      + (" per population" if tumour else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                         "kernel": "emit_fast_kernel", "kernel_ms": slow["kernel_ms"]["emit"],
                         "note": "kernel time summed over the chromosomes of the rank with the largest share (its last run)",
                         "algorithmic_bytes_per_pair": bpp, "bytes_note": bench.BYTES_PER_PAIR_FMT},
            "per_rank": [{"rank": v["rank"], "pairs_per_step": v["pairs"] / args.steps, "s_per_step": v["dt"] / args.steps,
                          "phases_last_run_s": v["phases"], "kernel_ms_last_run": v["kernel_ms"]} for v in allv],
        }
        if world == 1 and not args.no_cpu_baseline:
            import tempfile
            try:
                out["cpu_baseline"] = bench.cpu_baseline(tempfile.mkdtemp(prefix="simuscop_c3_cpu_"), prof_file)
            except Exception as e:
                out["cpu_baseline"] = None
                out["cpu_baseline_error"] = repr(e)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
