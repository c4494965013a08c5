"""Degenerate configurations (tests/edge_inputs.py) through the GPU command line: the same files as oracle(philox), byte
for byte, or a refusal where the oracle refuses (test_oracle_edge_inputs_vs_reference.py ties the oracle to the reference
binary on the same inputs)."""
import os
import subprocess

import pytest

import edge_inputs
import simuscop_amd.build as build

pytestmark = pytest.mark.gpu
SIMU = os.path.join(build.LIBDIR, "simuReads")
SEED = 0x5EED0E


@pytest.mark.parametrize("name", edge_inputs.GPU_NAMES)
def test_gpu_equals_oracle_on_degenerate_configuration(name, oracle_lib, tmp_path):
    cfg = edge_inputs.build(name, str(tmp_path))
    odir, gdir = str(tmp_path / "o"), str(tmp_path / "g")
    rc = oracle_lib.orc_simulate(cfg.encode(), 1, SEED >> 32, SEED & 0xFFFFFFFF, odir.encode(), 4)
    r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", gdir, "--quiet"], capture_output=True, text=True, timeout=300)
    if rc != 0:
        assert r.returncode != 0, (oracle_lib.orc_last_error().decode(), r.stderr[-500:])
        return
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(os.listdir(odir))
    assert files == sorted(os.listdir(gdir)) and files
    for f in files:
        assert open(os.path.join(odir, f), "rb").read() == open(os.path.join(gdir, f), "rb").read(), (name, f)


def test_crlf_fasta_reads_like_its_lf_twin(tmp_path):
    """The documented divergence (edge_inputs.GPU_NAMES): carriage returns are line ends, not bases and not name bytes."""
    outs = []
    for name in ("fasta_lf", "fasta_crlf"):
        cfg = edge_inputs.build(name, str(tmp_path / name))
        gdir = str(tmp_path / (name + "_out"))
        r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", gdir, "--quiet"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append({f: open(os.path.join(gdir, f), "rb").read() for f in sorted(os.listdir(gdir))})
    assert outs[0] == outs[1] and outs[0]
