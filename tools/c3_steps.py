#!/usr/bin/env python3
"""Whole C3 runs (GRCh38-sized genome, one GPU) one after the other: wall time and phases of every run.  usage: python tools/c3_steps.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, bench_c3, simuscop_amd
from simuscop_amd import synth
bench_c3.ensure_genome(1.0, 0)
fasta = bench_c3.genome_path(1.0)
cfg = "/tmp/c3_steps.cfg"
bench.write_config(cfg, fasta, "/tmp/c3_steps_out", coverage=30, threads=64)
for i in range(int(os.environ.get("C3_RUNS", "8"))):
    t0 = time.perf_counter()
    r = simuscop_amd.run_config(cfg, device=0, quiet=1, write_files=0, seed=0x5EED0C3, shard_rank=0, shard_world=1)
    torch.cuda.synchronize()
    print(i, "wall %.3f" % (time.perf_counter() - t0), "t_total %.3f t_load %.3f t_plan %.3f t_sample %.3f" % (r.t_total, r.t_load, r.t_plan, r.t_sample),
          "emit ms %.1f" % r.kernel_ms[4], "free GB %.1f" % (torch.cuda.mem_get_info()[0] / 1e9), flush=True)
