#!/usr/bin/env python3
"""bench.py -- simulated paired reads/sec of the MI355X read-sampling pass.

Workload (BASELINE.json configs[2], SURVEY.md 8(d) "C2"): one synthetic contig of GRCh38 chr20 length
(64,444,167 bp), HiSeqXTen profile (151 bp), PE, 30x, insertSize 350 -> ~6.4 M pairs per step.
A step = one full sampling pass over that chromosome (plan draws, indel pass, offset scan, per-base
substitution/quality sampling, FASTQ formatting) with haplotypes, tables and plan already resident in
HBM.  `value` counts a pair when its FASTQ text is complete in HBM (inputs resident, no PCIe in the
timed region).  SURVEY 8(d)'s own metric -- pairs whose text is complete in a PINNED HOST buffer -- is
measured in the same run right after (`host_pinned`: every pass drained through detached output sets while
the next pass is sampled, plain and block-gzipped on the device first), with the PCIe rate it achieved.

Multi-GPU (`--gpus N`: one rank per GPU over torchrun; started by the driver or by this script itself):
weak scaling -- every rank owns one chr20-sized chromosome of an N-chromosome genome.  The only exchange
is the read-count balancing step of Genome::setReadCounts (Genome.cpp:783-825): an all_gather of the
per-chromosome GC-weighted lengths over RCCL, after which every rank derives its read count exactly as
the reference apportions `reads*chrWL/WL`.  No data-path collective.
`--workload c3` is strong scaling of ONE 24-contig genome in GRCh38 proportions (`--scale`), whole run
including ingest: ranks own whole contigs balanced by length, ingest only those (simuReads --rank/--world).
`--workload c4` is the same for the tumour mixture of BASELINE configs[4]: four populations with variants and SNPs, 60x
(bench_c3.py).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR20_LEN = 64444167
TESTDATA = os.path.join(ROOT, "tests", "golden", "testData")
BYTES_PER_PAIR_FMT = "2*L haplotype bytes read + 2 FASTQ records written"
# issue cost of one VALU wave-instruction in this kernel's instruction mix, cycles per SIMD (tools/valu_microbench.hip:
# 2.3 for add/sub/logic/right shift alone, 4.2 for everything else, ~4.2 for any realistic mixture of the two)
VALU_CYCLES_PER_INSTRUCTION = 4.2

PROFILES = {  # name -> (file, read length)
    "xten": ("Illumina_HiSeqXTen.profile", 151), "hs2500": ("Illumina_HiSeq2500.profile", 125),
    "hs2000": ("Illumina_HiSeq2000.profile", 75), "gaiix": ("Illumina_GenomeAnalyzerIIx.profile", 74),
}


def write_config(path, fasta, out_dir, coverage=30, threads=1, profile="Illumina_HiSeqXTen.profile"):
    with open(path, "w") as f:
        f.write(f"ref = {fasta}\nprofile = {os.path.join(TESTDATA, profile)}\nname = sim\noutput = {out_dir}\n"
                f"layout = PE\nthreads = {threads}\nverbose = 0\ncoverage = {coverage}\ninsertSize = 350\n")


def cpu_baseline(workdir, profile="Illumina_HiSeqXTen.profile"):
    """Time the reference's CPU thread-pool path on a bounded sample of the same workload, beside the GPU command line on
    the same inputs.

    Preferred: the UNMODIFIED reference binary built by oracle/Makefile (kind "reference").
    Fallback (binary absent): the oracle restatement (kind "port").  Sample: the C2 chromosome itself (64.4 Mbp at
    30x) on boxes with many cores, a 16 Mbp contig of the same genome on small ones.  The reference's unit of
    parallelism is the <=1 Mbp segment with a barrier per chromosome (Genome.cpp:876-883): 65 work items for this
    chromosome, whatever `threads` says; its load and haplotype passes are serial and its writer is one mutex
    (lib/seqwriter/SeqWriter.cpp:49-54), so more threads than ~16 make it SLOWER.  It is therefore run at `threads` = 16
    and = all host cores; `value` is the better of the two.  `gpu_cli_same_inputs` is the like-for-like number: the whole
    `simuReads` run of this repo on the same config -- HIP context, FASTA load, FASTQ files written -- on one GPU."""
    from simuscop_amd import synth
    cores = os.cpu_count() or 1
    sample_len = CHR20_LEN if cores >= 32 else 16_000_000
    fa = os.path.join(workdir, "cpu_sample.fa")
    synth.write_fasta(fa, [("chr20", sample_len)], seed=20, workers=1)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "simuReads")
    kind = "reference" if os.path.exists(ref_bin) else "port"
    out = os.path.join(workdir, "cpu_out")

    def count_pairs():
        lines = 0
        with open(os.path.join(out, "sim_1.fq"), "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                lines += blk.count(b"\n")
        for fn in os.listdir(out):
            os.remove(os.path.join(out, fn))
        return lines // 4

    runs = []
    for threads in sorted({min(16, cores), cores}):
        cfg = os.path.join(workdir, f"cpu_config_t{threads}.txt")
        write_config(cfg, fa, out, threads=threads, profile=profile)
        cmd = [ref_bin, cfg] if kind == "reference" else [os.path.join(ROOT, "oracle", "oracle_cli"), cfg, "--rng", "philox", "--threads", str(threads)]
        t0 = time.time()
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        if r.returncode != 0:
            continue
        pairs = count_pairs()
        runs.append({"threads": threads, "value": pairs / dt, "pairs": pairs, "seconds": round(dt, 2)})
    if not runs:
        return None
    best = max(runs, key=lambda x: x["value"])
    res = {"value": best["value"], "unit": "pairs/s", "cores": best["threads"], "host_cores": f"{best['threads']} of {cores}", "kind": kind,
           "runs": runs,
           "sample": f"{sample_len} bp contig, {profile[9:-8]} PE 30x insertSize 350, {best['pairs']} pairs in {best['seconds']} s wall "
                     f"(whole run incl. input load and FASTQ files; the better of threads = {[x['threads'] for x in runs]})"}
    # the same config through this repo's command line, files written: what a user of the reference would run instead
    try:
        cfg = os.path.join(workdir, "cpu_config_t%d.txt" % best["threads"])
        t0 = time.time()
        r = subprocess.run([os.path.join(ROOT, "simuscop_amd", "lib", "simuReads"), cfg, "--out", out, "--quiet", "--seed", "1"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        if r.returncode == 0:
            pairs = count_pairs()
            res["gpu_cli_same_inputs"] = {"value": pairs / dt, "unit": "pairs/s", "pairs": pairs, "seconds": round(dt, 2),
                                          "includes": "process start, HIP context, FASTA load, sampling, FASTQ files written (page cache)",
                                          "ratio_to_cpu_best": (pairs / dt) / best["value"]}
    except Exception as e:  # noqa: BLE001
        res["gpu_cli_same_inputs_error"] = repr(e)
    return res


def pmc_evidence():
    """Per-launch counters of the dominant kernel from the newest committed rocprofv3 PMC passes
    (profiles/*/pmc_summary.json; FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same
    bench).  gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request for wide
    coalesced reads -> doubled; both counters are in KiB."""
    import glob
    from simuscop_amd.build import engine_source_digest
    now = engine_source_digest()
    best = None
    # the passes taken on THIS run's engine sources if there are any (several directories of one round do not sort by age);
    # otherwise the last by name, which the caller then refuses as stale
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        same = d.get("_meta", {}).get("engine_source_digest") == now
        for k, v in d.items():
            if k != "_meta" and "emit_fast_kernel" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                if same or best is None or not best[2]:
                    best = (path, v, same)
    if not best:
        return None
    path, v = best[0], best[1]
    meta = json.load(open(path)).get("_meta", {})
    out = {"source": os.path.relpath(path, ROOT), "kernel_ms_profiled": meta.get("emit_fast_kernel_avg_ms"), "digest": meta.get("engine_source_digest"),
           "traffic": (2.0 * v["FETCH_SIZE"]["mean_per_dispatch"] + v["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0}
    if "SQ_INSTS_VALU" in v:
        out["valu"] = v["SQ_INSTS_VALU"]["mean_per_dispatch"]
    if "GRBM_GUI_ACTIVE" in v:   # summed over the 8 XCDs: / 8 = the kernel's duration in shader clocks
        out["kernel_cycles"] = v["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8.0
    return out


def host_pinned_rate(sess, steps, gzip):
    """SURVEY 8(d) metric: pairs whose FASTQ text is complete in pinned host memory.  Each pass's text leaves the
    context as a detached output set and drains over PCIe (64 MB pinned buffers, two per mate) on a worker thread
    while the next pass is sampled (and compressed) on the main stream."""
    CH = int(os.environ.get("BENCH_DRAIN_MB", "64")) << 20
    bufs = [sess.host_alloc(CH) for _ in range(4)]
    moved = [0]

    def drain(h):
        text, gz = sess.outputs_sizes(h)
        sizes = gz if gzip else text
        for mate in (0, 1):
            k = 0
            for off in range(0, sizes[mate], CH):
                n = min(CH, sizes[mate] - off)
                sess.outputs_fetch_into(h, mate, gzip, off, n, bufs[2 * mate + (k & 1)])
                moved[0] += n
                k += 1

    pending = None
    pairs = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        sess.sample()
        _, _, nf = sess.result()
        if gzip:
            sess.compress()
        h = sess.detach_outputs()
        if pending:
            pending[0].join()
            sess.release_outputs(pending[1])
        th = threading.Thread(target=drain, args=(h,))
        th.start()
        pending = (th, h)
        pairs += nf
    pending[0].join()
    sess.release_outputs(pending[1])
    dt = time.perf_counter() - t0
    for b in bufs:
        sess.host_free(b)
    return {"value": pairs / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "pcie_GBps": moved[0] / dt / 1e9, "bytes_per_step": moved[0] / steps}


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as a child torchrun and relay its
    output.  Nothing here has touched the GPU yet (no torch import, no engine): the child processes do."""
    import socket
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        # the profiler's preloaded library has initialised the GPU in this process: starting the ranks from here would be the
        # exec-after-GPU-init the pool forbids (tools/README.md: profile one rank of a multi-GPU run instead)
        raise SystemExit("bench.py --gpus N cannot start its ranks under rocprofv3: profile one rank (see tools/README.md)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4"])
    ap.add_argument("--scale", type=float, default=1.0, help="c3: contig lengths = GRCh38 primary lengths x scale")
    ap.add_argument("--contig-len", type=int, default=CHR20_LEN)
    ap.add_argument("--coverage", type=int, default=None, help="default: 30 (c2, c3), 60 (c4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-pinned", action="store_true", help="skip the pinned-host legs (plain + gzip) after the timed region")
    ap.add_argument("--no-md5", action="store_true")
    ap.add_argument("--profile", default="xten", choices=sorted(PROFILES),
                    help="sequencing profile of the workload (default: HiSeqXTen, the configuration the metric is quoted on)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="make a process group even for ONE rank (also SIMUSCOP_FORCE_PG=1): the all_gather of the read balancing, "
                         "the max/sum all-reduces and the strong leg's exchange then run through RCCL on the one GPU at hand")
    ap.add_argument("--strong-scale", type=float, default=None,
                    help="c2: size of the genome of the strong-scaling leg (24 contigs, GRCh38 lengths x this; default 1.0, 0.25 on hosts "
                         "with fewer than 16 cores; 0 = no leg)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.workload in ("c3", "c4"):
        import bench_c3
        return bench_c3.main(args)
    if args.coverage is None:
        args.coverage = 30
    strong_scale = args.strong_scale if args.strong_scale is not None else (1.0 if (os.cpu_count() or 1) >= 16 else 0.25)
    if strong_scale:   # the leg's genome: made (by rank 0, on all host cores) before anything here touches the GPU
        import bench_c3
        bench_c3.ensure_genome(strong_scale, int(os.environ.get("RANK", "0")))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if os.environ.get("BENCH_SAME_DEVICE"):   # rehearsal on a 1-GPU box: every rank on device 0 (gloo only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    import simuscop_amd
    from simuscop_amd import dist as sdist
    from simuscop_amd import synth
    # one group for the whole run, with a timeout: a rank that dies takes the others out of their collective after
    # SIMUSCOP_PG_TIMEOUT_S (120 s) instead of leaving them there until the launcher's own limit
    pg = sdist.init_process_group(args.backend, local_rank, world, force=args.force_process_group)

    workdir = tempfile.mkdtemp(prefix=f"simuscop_bench_r{rank}_")
    fasta = os.path.join(workdir, "ref.fa")
    # rank r owns chromosome r of an N-chromosome synthetic genome (contig index = rank)
    seq = synth.synth_contig(args.contig_len, seed=20, contig_index=rank)
    with open(fasta, "wb") as f:
        f.write(b">chr20\n")
        full = (args.contig_len // 60) * 60
        import numpy as np
        body = np.empty((full // 60, 61), dtype=np.uint8)
        body[:, :60] = seq[:full].reshape(-1, 60)
        body[:, 60] = 10
        f.write(body.tobytes())
        if full < args.contig_len:
            f.write(seq[full:].tobytes() + b"\n")
    del seq
    cfg = os.path.join(workdir, "config.txt")
    write_config(cfg, fasta, os.path.join(workdir, "out"), coverage=args.coverage, profile=PROFILES[args.profile][0])

    sess = simuscop_amd.Session(cfg, device=local_rank, write_files=0, quiet=1, seed=0x5EED0000 + rank)
    L = PROFILES[args.profile][1]
    # ---- read-count balancing (Genome::setReadCounts) ----
    my_wl = sess.weighted_length()
    # all_gather of one fp64 pair per rank (RCCL over xGMI), then the reference's apportioning formula
    my_reads, _ = sdist.balance_reads(my_wl, args.contig_len, args.coverage, L, device=coll_dev)
    sess.set_reads(my_reads)
    assert sess.prepare_batch(0)
    stream = torch.cuda.current_stream()
    sess.set_stream(stream.cuda_stream)

    def step():
        sess.sample()
        return sess.result()   # waits for the pass (stream sync) and returns sizes

    # A context's first passes are not steady state: the first one allocates the output buffers (and is the only one that
    # waits for its size in mid-pass), code and tables page in, the clocks ramp.  Five untimed passes before the W
    # warm-up steps of the contract, so that a run with few steps measures what a long one does.
    for _ in range(5):
        step()
    for _ in range(args.warmup):
        step()
    if pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pairs = 0
    kms = {k: 0.0 for k in simuscop_amd.SG_K_NAMES}
    fq_bytes = 0
    for _ in range(args.steps):
        b1, b2, nf = step()
        pairs += nf
        fq_bytes = b1 + b2
        for k, v in sess.kernel_times().items():   # HIP events recorded on this stream inside the engine
            kms[k] += v
    queued_items = sess.emit_info()[0]   # of the last pass
    torch.cuda.synchronize()
    if pg:
        dist.barrier()
    dt = time.perf_counter() - t0

    if pg:
        t = torch.tensor([dt, float(pairs)], dtype=torch.float64, device=coll_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, total_pairs = float(tmax[0]), float(tsum[1])
    else:
        dt_max, total_pairs = dt, float(pairs)

    if rank == 0:
        default_workload = args.contig_len == CHR20_LEN and args.coverage == 30 and args.profile == "xten"
        pairs_per_step = pairs / args.steps
        emit_ms = kms["emit"] / args.steps            # emit_fast_kernel alone (HIP events around that launch)
        slow_ms = kms["emit_slow"] / args.steps       # emit_slow_kernel: the items it queued
        bytes_per_pair = 2 * L + fq_bytes / max(pairs_per_step, 1)   # measured FASTQ bytes/pair + 2L reference bytes
        # the main kernel's share of the algorithmic bytes = the share of the 8-base items it did itself
        total_items = 2.0 * pairs_per_step * ((L + 7) // 8)
        main_share = 1.0 - queued_items / max(total_items, 1.0)
        achieved = pairs_per_step * bytes_per_pair * main_share / (emit_ms * 1e-3) / 1e9
        pmc = pmc_evidence() if default_workload else None
        pmc_stale = None
        if pmc:
            # Counters of another build say nothing about this one: they are quoted only when they were taken on exactly the
            # engine sources this run uses (tools/profile_round.sh stores their digest with the counters).  The profiled
            # launch's time is reported next to the live one; boxes differ by up to ~10 % on this kernel.
            from simuscop_amd.build import engine_source_digest
            now = engine_source_digest()
            if pmc.get("digest") != now:
                pmc_stale = {"source": pmc["source"], "counters_taken_on_sources": pmc.get("digest"), "this_run": now}
                pmc = None
        compute_side = None
        if pmc and "valu" in pmc:
            # VALU issue occupancy of the LIVE launch: wave-instructions of the committed counter pass (the count does not
            # depend on the run) x the microbenchmarked cycles per instruction, over 1024 SIMDs x this run's kernel time
            compute_side = {"bound": "integer VALU issue", "valu_wave_instructions_per_launch": pmc["valu"],
                            "cycles_per_instruction": VALU_CYCLES_PER_INSTRUCTION, "cost_model": "tools/valu_microbench.hip",
                            "valu_issue_occupancy_live": pmc["valu"] * VALU_CYCLES_PER_INSTRUCTION / (1024 * emit_ms * 1e-3 * 2.35e9),
                            "clock_GHz_assumed": 2.35, "source": pmc["source"]}
            if "kernel_cycles" in pmc:  # both numbers from the same counter passes: no clock assumed
                compute_side["kernel_cycles_counter_pass"] = pmc["kernel_cycles"]
                compute_side["valu_issue_occupancy"] = pmc["valu"] * VALU_CYCLES_PER_INSTRUCTION / (1024 * pmc["kernel_cycles"])
        out = {
            "metric": "simulated paired reads/sec (whole node) at 30x WGS PE150",
            "value": total_pairs / dt_max,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"C2: one {args.contig_len} bp contig per GPU" + (" (GRCh38 chr20 length)" if args.contig_len == CHR20_LEN else "") + f", {PROFILES[args.profile][0][:-8]} profile "
                                   f"({L} bp), PE, {args.coverage}x, insertSize 350",
                       "pairs_per_step_per_gpu": pairs_per_step, "parallelism": f"{world} x 1 chromosome shard"},
            "value_counts": "pairs whose FASTQ text is complete in HBM (inputs resident); the pinned-host rates are in host_pinned",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0,
                         # PMC traffic was collected on the default workload only
                         "traffic": pmc["traffic"] if pmc else None,
                         "traffic_unit": "bytes per launch",
                         "traffic_source": pmc["source"] if pmc else None,
                         "traffic_refused_as_stale": pmc_stale,
                         "kernel_ms_of_the_profiled_run": pmc.get("kernel_ms_profiled") if pmc else None,
                         "algorithmic_bytes_per_launch": pairs_per_step * bytes_per_pair * main_share,
                         "kernel": "emit_fast_kernel", "kernel_ms": emit_ms,
                         "items_left_to_emit_slow_kernel": 1.0 - main_share, "emit_slow_kernel_ms": slow_ms,
                         "compute_side": compute_side,
                         "algorithmic_bytes_per_pair": bytes_per_pair, "bytes_note": BYTES_PER_PAIR_FMT},
            "kernel_ms_per_step": {k: v / args.steps for k, v in kms.items()},
        }
        if not args.no_md5:
            out["output_md5"] = sess.output_md5()   # of the last timed pass (tests/test_gpu_full_size.py checks it against the oracle)
            out["output_seed"] = 0x5EED0000 + rank
        if world == 1 and not args.no_host_pinned:
            try:
                hs = max(2, min(args.steps, 10))   # the first pass of a leg is not overlapped with a drain: more passes, closer to the steady rate
                out["host_pinned"] = {"plain": host_pinned_rate(sess, hs, False), "gzip": host_pinned_rate(sess, hs, True),
                                      "counts": "pairs whose FASTQ text (plain, or BGZF made on the device) is complete in pinned host "
                                                "memory, passes drained while the next one is sampled",
                                      "bound": "PCIe Gen5 x16, 63 GB/s (spec)"}
                # SURVEY 8(d) words the metric at the pinned host buffer: that figure, by name, next to the device-resident `value`
                hp = out["host_pinned"]
                out["value_contract"] = {"value": hp["plain"]["value"], "unit": "pairs/s",
                                         "counts": "pairs whose plain FASTQ text is complete in a pinned host buffer (SURVEY 8(d))",
                                         "bound": "pcie", "pcie_GBps": hp["plain"]["pcie_GBps"], "pcie_frac_of_63_GBps": hp["plain"]["pcie_GBps"] / 63.0,
                                         "as_bgzf": hp["gzip"]["value"]}
            except Exception as e:
                out["host_pinned"] = None
                out["host_pinned_error"] = repr(e)
        out["value_device_resident"] = out["value"]
        # what a step spends outside its kernels (launch gaps, event records, the host's wait for the pass): the headline is a
        # step rate, the roofline a kernel rate -- the line says how far apart the two are
        ksum = sum(out["kernel_ms_per_step"].values())
        out["gap_ms_per_step"] = out["ms_per_step"] - ksum
        out["step_over_kernels"] = out["ms_per_step"] / ksum if ksum else None
        out["roofline"]["frac_at_step_level"] = pairs_per_step * bytes_per_pair / (out["ms_per_step"] * 1e-3) / 1e9 / 8000.0
        if world == 1 and not args.no_cpu_baseline:
            # (before the strong leg: with one rank no collective can hang it, and the line below stays the only one)
            try:
                out["cpu_baseline"] = cpu_baseline(workdir, PROFILES[args.profile][0])
            except Exception as e:  # the baseline is a report, never a reason to lose the bench line
                out["cpu_baseline"] = None
                out["cpu_baseline_error"] = repr(e)
        if world > 1 and strong_scale:
            # N > 1: the weak-scaling line leaves NOW, flushed.  The strong leg below is a second whole measurement with
            # collectives of its own; whatever happens to it, the driver has this line.  The full line (this one plus
            # `strong_c3`) follows as the last line of the run.
            print(json.dumps(dict(out, line="weak-scaling result; the full line (with strong_c3) follows the strong leg")), flush=True)
    sess.close()
    del sess
    # ---- strong scaling of ONE genome (BASELINE configs[3]) in the same line: every N, the flag-less run included ----
    rc = 0
    if strong_scale:
        import bench_c3
        # Every rank says whether it can enter the leg (its genome is on disk, its engine is alive); one "no" keeps ALL
        # ranks out of it -- alike, so nobody is left in a collective.
        ready = 1.0
        try:
            if not os.path.exists(bench_c3.genome_path(strong_scale) + ".fai"):
                ready = 0.0
        except Exception:  # noqa: BLE001
            ready = 0.0
        if pg:
            flag = torch.tensor([ready], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ready = float(flag[0])
        leg = {"error": "a rank was not ready for the strong leg (genome missing)"}
        if ready:
            try:
                leg = bench_c3.measure("c3", strong_scale, max(1, min(args.steps, 3)), 1, "xten", None, args.backend, rank, local_rank, world)
            except Exception as e:  # noqa: BLE001
                traceback.print_exc()
                sys.stderr.flush()
                if pg and world > 1:
                    # The other ranks are inside (or about to enter) a collective of the leg that this rank will never
                    # join.  The weak line is out already; leave with a non-zero code so that the launcher ends the others
                    # (and their group timeout does if it does not) -- never re-enter, never re-exec.
                    os._exit(3)
                leg = {"error": repr(e)}   # one rank: nothing waits for us, the line survives with the error in it
        if rank == 0:
            if isinstance(leg, dict) and "value" in leg:
                leg = {"value": leg["value"], "unit": leg["unit"], "scaling": "strong", "n_gpus": world, "s_per_run": leg["ms_per_step"] / 1e3,
                       "runs_timed": leg["steps"], "workload": leg["config"]["workload"], "pairs_per_run": leg["config"]["pairs_per_step"],
                       "per_rank": leg["per_rank"], "emit_roofline_frac": leg["roofline"]["frac"]}
            out["strong_c3"] = leg
    if rank == 0:
        if pg:
            out["process_group"] = {"backend": "rccl (torch.distributed nccl)" if args.backend == "nccl" else args.backend, "world": world,
                                    "timeout_s": sdist.pg_timeout_s(), "exchange_collectives_rank0": dict(sdist.COLLECTIVES)}
        print(json.dumps(out), flush=True)
    if pg:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 -- a peer that left early: the lines are out, say so with the exit code
            traceback.print_exc()
            rc = 4
    import shutil
    shutil.rmtree(workdir, ignore_errors=True)
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
