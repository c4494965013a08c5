"""Pin the CPU oracle against the UNMODIFIED reference binary.

tests/golden/golden.json holds md5 sums of FASTQ written by oracle/_ref/simuReads (the reference's
own sources compiled by oracle/Makefile) under the frozen clock of oracle/fakeclock.c.  The oracle in
`mt` RNG mode (libstdc++ mt19937 / default_random_engine / glibc rand(), consumed in the reference's
order) must reproduce every file byte for byte.  Bar: bit-exact.
"""
import hashlib
import json
import os

import pytest

import cases

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def _md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_oracle_mt_reproduces_reference(name, oracle_lib, tmp_path):
    """(The full-coverage C3 / C4 cases and the shipped-fixture cases take minutes on the sequential mt mode: they run in
    the background of the same session, tests/test_zz_long_oracle_cases.py.)"""
    assert name in GOLDEN, "run tests/golden/make_golden.py"
    cfg = cases.build_case(name, str(tmp_path))
    g = GOLDEN[name]
    rc = oracle_lib.orc_simulate(cfg.encode(), 0, g["fake_sec"], g["fake_nsec"], b"", 1)
    assert rc == 0, oracle_lib.orc_last_error().decode()
    files = cases.output_files(cfg)
    assert sorted(os.path.basename(f) for f in files) == sorted(g["files"])
    total = 0
    for fq in files:
        exp = g["files"][os.path.basename(fq)]
        assert os.path.getsize(fq) == exp["bytes"]
        assert _md5(fq) == exp["md5"], f"{name}/{os.path.basename(fq)} differs from the reference"
        total += exp["reads"]
    assert oracle_lib.orc_last_read_count() == total


def test_golden_pins_the_named_baseline_configs():
    """BASELINE configs[3] / [4] (C3: 24 contigs in GRCh38 proportions, XTen PE; C4: four populations, tumour
    variation pattern, abundance row, XTen PE) are pinned on the reference binary at reduced AND at their own
    coverage (30x / 60x)."""
    for name in ("c3_grch38_pe_xten_cov3", "c3_grch38_pe_xten_cov30", "c4_tumor_pe_xten_cov6", "c4_tumor_pe_xten_cov60"):
        assert name in GOLDEN, name
    assert sum(f["reads"] for f in GOLDEN["c3_grch38_pe_xten_cov30"]["files"].values()) > 4_000_000
    c4 = GOLDEN["c4_tumor_pe_xten_cov60"]["files"]
    assert sorted(c4) == ["clone1_0.300+clone2_0.250+clone3_0.350+normal_0.100_1.fq", "clone1_0.300+clone2_0.250+clone3_0.350+normal_0.100_2.fq"]


def test_golden_covers_edge_cases():
    # tiny_contigs_pe must exercise chromosome-end clipping / window abandonment: a 100 bp contig
    # can never yield a 125 bp read, so fewer reads than planned come out.
    f = GOLDEN["tiny_contigs_pe"]["files"]
    assert f["tiny_1.fq"]["reads"] == f["tiny_2.fq"]["reads"] > 0
    # mixtures: one file per abundance row, named popu_%.3f joined by '+' (Genome.cpp:908-928)
    assert any(k.startswith("clone1_0.300+clone2_0.250") for k in GOLDEN["tumor_se_mixture"]["files"])
