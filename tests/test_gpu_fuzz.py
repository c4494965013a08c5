"""Seeded random configurations, GPU (C ABI through the CLI) against the oracle in Philox mode, byte for
byte: contig count and sizes, profile, layout, coverage, insert size, ploidy, variants, SNPs, seed."""
import os
import random
import subprocess

import pytest

import cases
import gpu_run
from simuscop_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")


def _make(rng, wd, extended=False, tight_targets=False):
    fa = os.path.join(wd, "ref.fa")
    n_contigs = rng.choice([1, 1, 2, 3, 5])
    contigs = [("chr%d" % (i + 1), rng.choice([900, 4000, 30000, 90000, 250000, 1200000])) for i in range(n_contigs)]
    synth.write_fasta(fa, contigs, seed=rng.randrange(1, 1000), line_len=rng.choice([50, 60, 70, 80]),
                      n_islands=rng.choice([None, None, (5000, 7), (20000, 300)]))
    prof = rng.choice(list(cases.PROFILES))
    layout = rng.choice(["PE", "PE", "SE"])
    kv = dict(ref=fa, profile=os.path.join(cases.TESTDATA, cases.PROFILES[prof]), name="fz", output=os.path.join(wd, "out"),
              layout=layout, threads=1, verbose=0, coverage=rng.choice([1, 2, 3, 5, 8]),
              insertSize=rng.choice([200, 250, 350, 500]), ploidy=rng.choice([2, 2, 2, 1, 3]))
    big = max(contigs, key=lambda c: c[1])
    if rng.random() < 0.5 and big[1] >= 90000:
        rows = cases._variations("fz", big[0][3:], big[1] / 63025520.0)
        cases._write(os.path.join(wd, "variations.txt"), rows)
        kv["variation"] = os.path.join(wd, "variations.txt")
    if rng.random() < 0.5:
        rows = []
        for name, length in contigs:
            if length >= 4000:
                rows += cases._snps(name[3:], length, rng.choice([300, 1500, 4000]), rng.randrange(1, 99))
        if rows:
            cases._write(os.path.join(wd, "snp.txt"), rows)
            kv["snp"] = os.path.join(wd, "snp.txt")
    if extended and rng.random() < 0.35:   # another read length / indel rate on the same tables
        src = {v: k for k, v in cases.PROFILES.items()}[os.path.basename(kv["profile"])]
        base_len = {"xten": 151, "hs2500": 125, "hs2000": 75, "gaiix": 74}[src]
        prof2 = os.path.join(wd, "derived.profile")
        cases.derive_profile(src, prof2, read_length=rng.choice([52, 60, base_len + 9, 2 * base_len + 1, 400, 700]),
                             indel_scale=rng.choice([None, None, 3.0, 12.0]))
        kv["profile"] = prof2
        kv["insertSize"] = max(kv["insertSize"], 900)
    if tight_targets:   # exome whose targets overlap after the +-50 bp padding, nest, abut, come unsorted (cases._bed_tight)
        rows = []
        for name, length in contigs:
            if length >= 90000 and (name == big[0] or rng.random() < 0.5):
                rows += cases._bed_tight(name[3:] if rng.random() < 0.5 else name, length, rng.randrange(1, 999), rng.choice([8, 40, 150]),
                                         shuffle=rng.random() < 0.7)
        if rows:
            if rng.random() < 0.5:
                rng.shuffle(rows)   # contigs interleaved in the file
            cases._write(os.path.join(wd, "targets.bed"), rows)
            kv["target"] = os.path.join(wd, "targets.bed")
    elif extended:
        if rng.random() < 0.4 and big[1] >= 90000:   # exome: BED targets on the largest contig
            cases._write(os.path.join(wd, "targets.bed"), cases._bed(big[0][3:] if rng.random() < 0.5 else big[0], big[1], rng.randrange(1, 99), rng.choice([3, 20, 60])))
            kv["target"] = os.path.join(wd, "targets.bed")
        if rng.random() < 0.35:                        # mixture of populations
            names = ["fz", "p2", "p3"][:rng.choice([2, 3])]
            kv["name"] = ", ".join(names)
            rows = []
            for _ in range(rng.choice([1, 2])):
                w = [rng.random() + 0.05 for _ in names]
                rows.append("\t".join("%.3f" % (x / sum(w)) for x in w[:-1]) + "\t%.3f" % (1 - sum(float("%.3f" % (x / sum(w))) for x in w[:-1])))
            cases._write(os.path.join(wd, "abundance.txt"), rows)
            kv["abundance"] = os.path.join(wd, "abundance.txt")
            kv["ploidy"] = 2
    cfg = os.path.join(wd, "config.txt")
    cases._config(cfg, **kv)
    return cfg


def _seeds():
    """default: the committed seeds; SIMU_FUZZ_SEEDS=a-b widens the hunt (seeds >= 200 add targets / mixtures, seeds >= 1000
    are exome runs with overlapping / nested / unsorted targets)"""
    env = os.environ.get("SIMU_FUZZ_SEEDS")
    if env:
        a, b = env.split("-")
        return list(range(int(a), int(b) + 1))
    return list(range(101, 149)) + list(range(201, 261)) + list(range(1001, 1031))


@pytest.mark.timeout(180)
@pytest.mark.parametrize("case_seed", _seeds())
def test_random_configuration(case_seed, oracle_lib, tmp_path):
    rng = random.Random(case_seed)
    cfg = _make(rng, str(tmp_path), extended=case_seed >= 200, tight_targets=case_seed >= 1000)
    seed = rng.getrandbits(63)
    odir, gdir = str(tmp_path / "o"), str(tmp_path / "g")
    rc = oracle_lib.orc_simulate(cfg.encode(), 1, seed >> 32, seed & 0xFFFFFFFF, odir.encode(), 4)
    extra, env = [], {}
    if case_seed >= 200:   # engine / host variants that must not change a byte
        if rng.random() < 0.3:
            extra.append("--host-haplotypes")
        if rng.random() < 0.3:
            env["SIMU_PIECE_SLOTS"] = str(rng.choice([1, 700, 5000]))
        if rng.random() < 0.2:
            env["SIMU_HOST_PLAN"] = "1"   # GC windows, weights and read counts on the host instead of the device planner
        if rng.random() < 0.2:
            env["SG_SLOWQ_CAP"] = str(rng.choice([1, 64, 4096]))
    # in this process (warm HIP context and device blocks); every seventh seed through the command line in a child process
    ok, msg = gpu_run.run_gpu(cfg, seed, gdir, host_haplotypes=bool(extra), env=env, via_cli=case_seed % 7 == 0, timeout=120)
    if rc != 0:   # whatever the oracle refuses (e.g. a copy-number gain on a haploid genome) the GPU path must refuse too
        assert not ok, (oracle_lib.orc_last_error().decode(), msg[-500:])
        return
    assert ok, msg
    files = sorted(os.listdir(odir))
    assert files == sorted(os.listdir(gdir)) and files
    for f in files:
        a, b = open(os.path.join(odir, f), "rb").read(), open(os.path.join(gdir, f), "rb").read()
        assert a == b, (case_seed, f, len(a), len(b), extra, env, open(cfg).read())
