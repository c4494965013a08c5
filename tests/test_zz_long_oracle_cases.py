"""The long pins: oracle(mt) against the reference binary's md5 sums on

* the reference's OWN fixtures used verbatim (cases.SHIPPED_CASES: testData/exon_regions.bed, snp.txt, variations.txt,
  variations_tumor.txt, abundance_tumor.txt through the shapes of configFiles/config_test_{wes,wgs,tumor}.txt, on a
  seeded 63,025,520 bp chr20) -- BASELINE configs[0] and configs[1] as they are worded;
* C3 / C4 at their own coverage (cases.SLOW_CASES; SIMU_SLOW_TESTS=1 adds config_test_wgs.txt at its own 10x).

Each takes 20-200 s on the sequential mt mode, so they run in the background from session start (tests/bg_oracle.py,
started by tests/conftest.py) and are collected last.  Bar: bit-exact."""
import json
import os
import time

import pytest

import cases

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))

DEFAULT = sorted(cases.SHIPPED_CASES) + ["c3_grch38_pe_xten_cov30", "c4_tumor_pe_xten_cov60"]
EXTRA = [n for n in sorted(cases.SLOW_CASES) if n not in DEFAULT] if os.environ.get("SIMU_SLOW_TESTS") else []


@pytest.mark.long_oracle
@pytest.mark.parametrize("name", DEFAULT + EXTRA)
def test_oracle_mt_reproduces_reference_on_long_cases(name, long_oracle_runs):
    assert name in GOLDEN, "run tests/golden/make_golden.py --slow"
    path = os.path.join(long_oracle_runs["dir"], name + ".json")
    t0 = time.time()
    while not os.path.exists(path):
        assert long_oracle_runs["proc"].poll() is None or os.path.exists(path), "background oracle runner ended without " + name
        assert time.time() - t0 < 1800, "background oracle runner: timeout on " + name
        time.sleep(0.5)
    got = json.load(open(path))
    assert "error" not in got, got.get("error")
    want = GOLDEN[name]["files"]
    assert sorted(got["files"]) == sorted(want)
    for f, exp in want.items():
        assert got["files"][f]["bytes"] == exp["bytes"], (name, f)
        assert got["files"][f]["md5"] == exp["md5"], f"{name}/{f} differs from the reference"
    assert got["reads"] == sum(e["reads"] for e in want.values())


def test_golden_pins_the_shipped_fixture_configs():
    """BASELINE configs[1] as worded (shipped exon_regions.bed, HiSeq2000, PE, 50x) and the three shipped test configs."""
    for name in cases.SHIPPED_CASES:
        assert name in GOLDEN, name
    c1 = GOLDEN["c1_wes_shipped_hs2000_cov50"]["files"]
    assert sorted(c1) == ["test_1.fq", "test_2.fq"] and c1["test_1.fq"]["reads"] == c1["test_2.fq"]["reads"] > 500_000
    t = GOLDEN["tumor_shipped"]["files"]
    assert sorted(t) == ["clone1_0.300+clone2_0.250+clone3_0.350+normal_0.100.fq", "clone1_1.000+clone2_0.000+clone3_0.000+normal_0.000.fq"]
