"""The command line refuses what the reference's Config refuses (lib/config/Config.cpp:12-175), with the same last line on
stderr and the same exit code -- checked before any device is touched, so this runs without a GPU.  The expected lines
below were written by the reference binary (oracle/_ref/simuReads) on these very configurations; where that binary exists
(this container) it is also run here and now."""
import os
import re
import subprocess

import pytest

import cases
import simuscop_amd.build as build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "simuReads")
SIMU = os.path.join(build.LIBDIR, "simuReads")


def _sub(key, value):
    return lambda t: re.sub(rf"^{key} = .*$", f"{key} = {value}", t, flags=re.M)


def _drop(key):
    return lambda t: re.sub(rf"^{key} = .*\n", "", t, flags=re.M)


VARIANTS = {   # name -> (edit of the wgs_pe_xten configuration, last stderr line of the reference)
    "unknown_key": (lambda t: t + "bogus = 1\n", "bogus = 1"),
    "profile_missing": (_drop("profile"), "Error: sequencing profile must be specified!"),
    "reference_missing": (_drop("ref"), "Error: reference file not specified!"),
    "comment_behind_a_value": (lambda t: t.replace("layout = PE", "layout = PE   # paired"), "should be SE or PE"),
    "layout_in_lower_case": (_sub("layout", "pe"), "should be SE or PE"),
    "layout_unknown": (_sub("layout", "XX"), "should be SE or PE"),
    "coverage_as_text": (_sub("coverage", "ten"), "Error: sequence coverage should be a positive integer!"),
    "coverage_0": (_sub("coverage", "0"), "Error: sequence coverage should be a positive integer!"),
    "threads_0": (_sub("threads", "0"), "Error: number of threads should be a positive integer!"),
    "population_names_empty": (_sub("name", ""), "Error: population names not specified!"),
}


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(SIMU):   # (the driver runs __graft_entry__.build() first; a bare checkout gets the build here)
        build.build_all()


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_cli_refuses_like_the_reference(name, tmp_path):
    base = cases.build_case("wgs_pe_xten", str(tmp_path / "base"))
    edit, want = VARIANTS[name]
    cfg = str(tmp_path / "config.txt")
    with open(cfg, "w") as f:
        f.write(edit(open(base).read()))
    r = subprocess.run([SIMU, cfg, "--quiet"], capture_output=True, text=True, timeout=120)
    last = (r.stderr.strip().splitlines() or [""])[-1]
    assert r.returncode == 1 and last == want, (r.returncode, r.stderr[-300:])
    if os.path.exists(REF):
        ref = subprocess.run([REF, cfg], capture_output=True, text=True, timeout=120)
        assert ref.returncode == 1 and (ref.stderr.strip().splitlines() or [""])[-1] == want, ref.stderr[-300:]
