"""`seqToProfile` option handling (src/seqToProfile.cpp:46-121) without a device: the checks that run before anything touches
the GPU -- usage, the three required inputs, the ranges of -k and -B -- with the reference's messages and exit codes.  (The
reference's own trainer cannot be run past them here: it needs samtools.)"""
import os
import subprocess

import simuscop_amd.build as build

EXE = os.path.join(build.LIBDIR, "seqToProfile")


def _run(*args):
    if not os.path.exists(EXE):
        build.build_all()
    return subprocess.run([EXE, *args], capture_output=True, text=True, timeout=60)


def test_help_and_unknown_options():
    r = _run("-h")
    assert r.returncode == 0 and "--bam" in r.stderr and "--vcf" in r.stderr and "--ref" in r.stderr and "--sam" in r.stderr
    r = _run("--no-such-option")
    assert r.returncode == 1 and "Usage" in r.stderr


def test_required_inputs_in_the_reference_order():
    # seqToProfile.cpp:84-107: bam first, then vcf, then ref
    r = _run()
    assert r.returncode == 1 and "Use --bam to specify a normal BAM file." in r.stderr
    r = _run("-b", "x.bam")
    assert r.returncode == 1 and "Use --vcf to specify the VCF file generated from the normal BAM." in r.stderr
    r = _run("-b", "x.bam", "-v", "x.vcf")
    assert r.returncode == 1 and "Use --ref to specify the reference file(.fasta) to which the reads are aligned." in r.stderr
    r = _run("--sam", "x.sam", "-v", "x.vcf")          # --sam stands in for --bam
    assert r.returncode == 1 and "Use --ref" in r.stderr


def test_ranges_of_kmer_and_bins():
    base = ["-b", "x.bam", "-v", "x.vcf", "-r", "x.fa"]
    for k in ("0", "6", "-1"):
        r = _run(*base, "-k", k)
        assert r.returncode == 1 and 'parameter "kmer" should be a positive integer with maximum value of 5' in r.stderr, k
    r = _run(*base, "-B", "9")
    assert r.returncode == 1 and 'parameter "bins" should be a positive integer with minimum value of 10' in r.stderr
    # without -s the reference warns that samtools is taken from the PATH (seqToProfile.cpp:109-112); --sam needs none
    r = _run(*base, "-k", "6")
    assert "the path of samtools not specified" in r.stderr
    r = _run("--sam", "x.sam", "-v", "x.vcf", "-r", "x.fa", "-k", "6")
    assert "the path of samtools not specified" not in r.stderr
