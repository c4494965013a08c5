"""GPU box: how fast the training path takes SAM text (SURVEY 8(f)-4 measurement).  Reads sampled by the engine on a 1.4 Mbp
contig (40x, XTen PE) become `samtools view` lines sorted by position; K contigs chr1..chrK of that one sequence and K copies
of the lines (contig field rewritten) make a text of K x 250 MB.  Prints the --stats line of `seqToProfile` and GB/s of SAM
text through the kernels; with `rocprofv3 --kernel-trace --stats -- python3 tools/train_bench.py` the per-kernel times.
usage: python tools/train_bench.py [K=8] [workdir]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import histo_util as H  # noqa: E402
import train_util as TU  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wd = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="trainbench_")
os.makedirs(wd, exist_ok=True)
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")
EXE = os.path.join(ROOT, "simuscop_amd", "lib", "seqToProfile")
sam_path, fa_path, vcf_path = os.path.join(wd, "reads.sam"), os.path.join(wd, "ref.fa"), os.path.join(wd, "none.vcf")
if not os.path.exists(sam_path):
    t0 = time.time()
    cfg, fa1 = H.histogram_config(cases, wd, "xten", "PE", 40, 350)
    out = os.path.join(wd, "gpu")
    subprocess.run([SIMU, cfg, "--seed", "78", "--out", out, "--quiet"], check=True)
    L, isz_max = 151, 551   # HiSeqXTen profile, insertSize 350: the support of its insert-size law (Profile.cpp:912-930)
    ref = H.read_fasta_one(fa1)
    f1, f2 = sorted(os.path.join(out, f) for f in os.listdir(out))
    lines = TU.sam_from_pairs(ref, H.Fastq(f1), H.Fastq(f2), L, isz_max, cuts=(10 ** 9, 10 ** 9))
    lines.sort(key=lambda l: int(l.split(b"\t", 4)[3]))
    text = b"\n".join(lines) + b"\n"
    seq = open(fa1, "rb").read().split(b"\n", 1)[1]
    with open(fa_path, "wb") as f, open(sam_path, "wb") as s:
        for k in range(1, K + 1):
            f.write(b">chr%d\n" % k + seq)
            s.write(text.replace(b"\tchr1\t", b"\tchr%d\t" % k))
    open(vcf_path, "w").write("##fileformat=VCFv4.2\n")
    print("inputs made in %.1f s: %d lines x %d" % (time.time() - t0, len(lines), K), file=sys.stderr)
best = None
for rep in range(3):
    r = subprocess.run([EXE, "--sam", sam_path, "-v", vcf_path, "-r", fa_path, "-o", os.path.join(wd, "out.profile"), "--quiet", "--stats"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    st = json.loads(r.stderr.strip().split("\n")[-1])
    st["sam_GBps_through_the_kernels"] = st["sam_bytes"] / st["t_reads"] / 1e9
    st["lines_per_s"] = st["lines"] / st["t_reads"]
    if best is None or st["t_reads"] < best["t_reads"]:
        best = st
print(json.dumps(best))
