#!/usr/bin/env python3
"""How much of bench.py's step is the host in the loop?  K passes queued back to back (one sg_result at the end)
against K passes each waited for.  usage: python tools/pipelined_steps.py [K]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, simuscop_amd
from simuscop_amd import synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wd = tempfile.mkdtemp(prefix="pipe_")
fa = os.path.join(wd, "ref.fa")
synth.write_fasta(fa, [("chr20", bench.CHR20_LEN)], seed=20)
cfg = os.path.join(wd, "config.txt")
bench.write_config(cfg, fa, os.path.join(wd, "out"))
sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=0x5EED0000)
sess.weighted_length(); sess.set_reads(sess.planned_reads); assert sess.prepare_batch(0)
for _ in range(3):
    sess.sample(); sess.result()
for mode in ("waited", "queued", "waited", "queued"):
    t0 = time.perf_counter()
    for _ in range(K):
        sess.sample()
        if mode == "waited":
            nf = sess.result()[2]
    nf = sess.result()[2]
    dt = time.perf_counter() - t0
    print(mode, "%.3f ms/pass  %.1f M pairs/s" % (dt / K * 1e3, nf * K / dt / 1e6))
sess.close()
