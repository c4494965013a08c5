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


def genome_path(scale):
    return f"/tmp/simuscop_c3_scale{scale:g}_seed67.fa"


def ensure_genome(scale, rank, timeout=1500):
    """The synthetic genome, made once per box and kept under /tmp (its .fai is written last: the sign that it is whole).
    Rank 0 has it made by a child process (python -m simuscop_amd.synth: all host cores; never a fork of a process that
    holds the GPU), the other ranks wait for the index.  No collective: this runs before the process group exists."""
    import subprocess
    from simuscop_amd import synth
    path = genome_path(scale)
    if not os.path.exists(path + ".fai"):
        if rank == 0:
            t0 = time.time()
            subprocess.check_call([sys.executable, "-m", "simuscop_amd.synth", path, "--scale", repr(scale), "--seed", "67"], cwd=ROOT)
            sys.stderr.write(f"bench: genome x{scale:g} ({os.path.getsize(path) >> 20} MiB) written in {time.time() - t0:.1f} s\n")
        else:
            t0 = time.time()
            while not os.path.exists(path + ".fai"):
                if time.time() - t0 > timeout:
                    raise SystemExit("bench: timed out waiting for rank 0 to write " + path)
                time.sleep(0.2)
    return path, synth.grch38_contigs(scale)


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
    """bench.py --workload c3 / c4: the whole line is this measurement."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ensure_genome(args.scale, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if os.environ.get("BENCH_SAME_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from simuscop_amd import dist as sdist
    pg = sdist.init_process_group(args.backend, local_rank, world, force=getattr(args, "force_process_group", False))
    out = measure(args.workload, args.scale, args.steps, args.warmup, args.profile, args.coverage, args.backend, rank, local_rank, world)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            import tempfile
            import bench
            try:
                out["cpu_baseline"] = bench.cpu_baseline(tempfile.mkdtemp(prefix="simuscop_c3_cpu_"), bench.PROFILES[args.profile][0])
            except Exception as e:
                out["cpu_baseline"] = None
                out["cpu_baseline_error"] = repr(e)
        print(json.dumps(out), flush=True)
    if pg:
        dist.barrier()
        dist.destroy_process_group()


def measure(workload, scale, steps, warmup, profile, coverage, backend, rank, local_rank, world):
    """`steps` whole runs of the genome on `world` ranks (the process group exists, the genome is on disk); every rank
    calls it, rank 0 gets the bench object, the others None."""
    import torch
    import torch.distributed as dist

    import bench
    import simuscop_amd
    from simuscop_amd import dist as sdist
    from simuscop_amd import synth

    pg = sdist.group_active()   # (also a one-rank group made with --force-process-group: every collective below then runs)

    def barrier():
        if pg:
            dist.barrier()

    fasta, contigs = genome_path(scale), synth.grch38_contigs(scale)
    prof_file, L = bench.PROFILES[profile]
    tumour = workload == "c4"
    coverage = coverage if coverage is not None else (60 if tumour else 30)
    cfg = f"/tmp/simuscop_{workload}_config_r{rank}.txt"
    bench.write_config(cfg, fasta, f"/tmp/simuscop_{workload}_out_r{rank}", coverage=coverage, threads=min(64, os.cpu_count() or 1),
                       profile=prof_file)
    if tumour:
        extra = _tumour_files(scale, contigs, rank, barrier)
        with open(cfg) as f:
            text = f.read().replace("name = sim\n", "name = clone1, clone2, clone3, normal\n")
        with open(cfg, "w") as f:
            f.write(text + f"variation = {extra['variation']}\nsnp = {extra['snp']}\nabundance = {extra['abundance']}\n")
    opts = dict(device=local_rank, quiet=1, write_files=0, seed=0x5EED0C3, shard_rank=rank, shard_world=world)
    exchange = None
    if pg:
        exchange = sdist.make_exchange("cuda" if backend == "nccl" else None)
        opts.update(shard_contigs=1, exchange=exchange)

    def step():
        return simuscop_amd.run_config(cfg, **opts)

    for _ in range(warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frags = 0
    last = None
    for _ in range(steps):
        last = step()
        frags += int(last.fragments)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    phases = {k: getattr(last, k) for k in ("t_total", "t_load", "t_engine", "t_reference", "t_haplotypes", "t_plan", "t_sample", "t_hap_device")}
    kernel_ms = {n: float(last.kernel_ms[i]) for i, n in enumerate(simuscop_amd.SG_K_NAMES)}
    mine = {"rank": rank, "dt": dt, "pairs": frags, "bytes": int(last.fastq_bytes), "phases": phases, "kernel_ms": kernel_ms,
            "work": {"windows": int(last.windows), "segments": int(last.segments), "batches": int(last.batches)}}
    if pg:
        allv = [None] * world
        dist.all_gather_object(allv, mine)
    else:
        allv = [mine]
    if rank != 0:
        return None
    dt_max = max(v["dt"] for v in allv)
    total_pairs = sum(v["pairs"] for v in allv)
    pairs_per_step = total_pairs / steps
    # dominant kernel: emit_fast_kernel, summed over the chromosomes of the slowest rank's last run
    slow = max(allv, key=lambda v: v["kernel_ms"]["emit"])
    bpp = 2 * L + slow["bytes"] / max(slow["pairs"] / steps, 1)
    achieved = (slow["pairs"] / steps) * bpp / (slow["kernel_ms"]["emit"] * 1e-3) / 1e9
    return {
        "metric": "simulated paired reads/sec (whole node) at 30x WGS PE150",
        "value": total_pairs / dt_max, "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": ("C4: four populations (variations + SNPs, abundance 0.3/0.25/0.35/0.1) of " if tumour else "C3: ") +
                               f"24 contigs, GRCh38 primary lengths x {scale:g} ({sum(l for _, l in contigs)} bp), "
                               f"{prof_file[:-8]} profile ({L} bp), PE, {coverage}x, insertSize 350; whole simuReads run per step "
                               f"(ingest + haplotypes + GC scan + apportioning + sampling), text left in HBM",
                   "pairs_per_step": pairs_per_step,
                   "parallelism": f"{world} rank(s), whole chromosomes per rank (longest first); all-reduce of 24 weighted lengths"
                                  + (" per population" if tumour else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                     "kernel": "emit_fast_kernel", "kernel_ms": slow["kernel_ms"]["emit"],
                     "note": "kernel time summed over the chromosomes of the rank with the largest share (its last run)",
                     "algorithmic_bytes_per_pair": bpp, "bytes_note": bench.BYTES_PER_PAIR_FMT},
        "per_rank": [{"rank": v["rank"], "pairs_per_step": v["pairs"] / steps, "s_per_step": v["dt"] / steps,
                      "phases_last_run_s": v["phases"], "kernel_ms_last_run": v["kernel_ms"], "work_last_run": v["work"]} for v in allv],
    }
