"""Degenerate configurations (tests/edge_inputs.py) through the GPU command line: the same files as oracle(philox), byte
for byte, or a refusal where the oracle refuses (test_oracle_edge_inputs_vs_reference.py ties the oracle to the reference
binary on the same inputs)."""
import os
import subprocess

import pytest

import edge_inputs
import gpu_run
import simuscop_amd.build as build

pytestmark = pytest.mark.gpu
SIMU = os.path.join(build.LIBDIR, "simuReads")
SEED = 0x5EED0E


@pytest.mark.parametrize("name", edge_inputs.GPU_NAMES)
def test_gpu_equals_oracle_on_degenerate_configuration(name, oracle_lib, tmp_path):
    cfg = edge_inputs.build(name, str(tmp_path))
    odir, gdir = str(tmp_path / "o"), str(tmp_path / "g")
    rc = oracle_lib.orc_simulate(cfg.encode(), 1, SEED >> 32, SEED & 0xFFFFFFFF, odir.encode(), 4)
    ok, msg = gpu_run.run_gpu(cfg, SEED, gdir, via_cli=sum(map(ord, name)) % 5 == 0)   # (a fifth of the names through the command line)
    if rc != 0:
        assert not ok, (oracle_lib.orc_last_error().decode(), msg[-500:])
        return
    assert ok, msg
    files = sorted(os.listdir(odir))
    assert files == sorted(os.listdir(gdir)) and files
    for f in files:
        assert open(os.path.join(odir, f), "rb").read() == open(os.path.join(gdir, f), "rb").read(), (name, f)


@pytest.mark.parametrize("via_cli", [False, True])
def test_crlf_as_lf_flag_reads_a_crlf_fasta_like_its_lf_twin(via_cli, tmp_path):
    """--crlf-as-lf (additive): carriage returns are line ends, not bases and not name bytes.  (Default: fastahack's reading,
    the byte-parity case "fasta_crlf" above.)"""
    outs = []
    for name in ("fasta_lf", "fasta_crlf"):
        cfg = edge_inputs.build(name, str(tmp_path / name))
        gdir = str(tmp_path / (name + "_out"))
        ok, msg = gpu_run.run_gpu(cfg, SEED, gdir, via_cli=via_cli, flags=("crlf_as_lf",) if name == "fasta_crlf" else ())
        assert ok, msg
        outs.append({f: open(os.path.join(gdir, f), "rb").read() for f in sorted(os.listdir(gdir))})
    assert outs[0] == outs[1] and outs[0]


def test_strict_bases_flag_makes_a_literal_x_an_unknown_base(oracle_lib, tmp_path):
    """--strict-bases (additive): the run on a genome with X's equals the oracle's run on the same genome with R's in their
    places -- an X is then an unknown base like any other."""
    cfg = edge_inputs.build("fasta_literal_x", str(tmp_path / "x"))
    gdir = str(tmp_path / "g")
    ok, msg = gpu_run.run_gpu(cfg, SEED, gdir, flags=("strict_bases",))
    assert ok, msg
    # the twin genome: R / r where the X / x stand
    cfg_r = edge_inputs.build("fasta_literal_x", str(tmp_path / "r"))
    fa = [l.split("=", 1)[1].strip() for l in open(cfg_r) if l.startswith("ref")][0]
    parts = edge_inputs._fasta_variants()["fasta_literal_x"]
    from simuscop_amd import synth
    a = synth.synth_contig(30000, 77, 0, n_runs=False).tobytes()
    body = edge_inputs.literal_x(a, other=True)
    with open(fa, "wb") as f:
        f.write(parts[0] + b"".join(body[i:i + 60] + b"\n" for i in range(0, len(body), 60)) + parts[2] + parts[3])
    odir = str(tmp_path / "o")
    assert oracle_lib.orc_simulate(cfg_r.encode(), 1, SEED >> 32, SEED & 0xFFFFFFFF, odir.encode(), 4) == 0
    files = sorted(os.listdir(odir))
    assert files == sorted(os.listdir(gdir)) and files
    for f in files:
        assert open(os.path.join(odir, f), "rb").read() == open(os.path.join(gdir, f), "rb").read(), f
    # (Whether the default run differs from the strict one is a matter of the profile: the place-holder rows "Xbb" / "XXb" of
    # the shipped profiles hold counts in the first bin only and are identity rows elsewhere (normParas, Profile.cpp:845-862),
    # so an X in the middle of a read changes nothing a draw can show -- the oracle's own runs on the X genome and on its R
    # twin are byte-identical.  The default reading is pinned by the byte-parity case "fasta_literal_x" above.)


@pytest.mark.parametrize("via_cli", [False, True])
def test_unique_contigs_flag_refuses_a_repeated_name(via_cli, tmp_path):
    cfg = edge_inputs.build("fasta_repeated_name", str(tmp_path / "c"))
    ok, msg = gpu_run.run_gpu(cfg, SEED, str(tmp_path / "g"), via_cli=via_cli, flags=("unique_contigs",))
    assert not ok and "more than once" in msg, msg
