"""GPU parity on the reference's OWN fixtures, used verbatim at their real coordinates (cases.SHIPPED_CASES):

* BASELINE configs[1] as worded: config_test_wes.txt with testData/exon_regions.bed (4,677 targets, 76 padded-overlap
  pairs, 111 targets > 1 kbp), variations.txt, snp.txt, HiSeq2000, PE, 50x on a 63,025,520 bp chr20;
* config_test_wes.txt exactly as shipped (HiSeq2500, 100x), config_test_wgs.txt (BASELINE configs[0], GAIIx PE),
  config_test_tumor.txt (four populations, both rows of abundance_tumor.txt, SE).

The reference binary pins the same cases on the oracle's mt mode (tests/test_zz_long_oracle_cases.py, golden.json);
here the HIP path (CLI -> C ABI) must equal the oracle's Philox mode byte for byte -- also as BGZF, as three
chromosome-owning ranks, and cut into memory-bounded pieces.  Bar: bit-exact."""
import hashlib
import os
import subprocess

import pytest

import cases
from simuscop_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")
SEED = (cases.FAKE_SEC << 32) | cases.FAKE_NSEC

pytestmark = pytest.mark.gpu


def _md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


@pytest.fixture(scope="module")
def chr20_fasta(tmp_path_factory):
    """The 63,025,520 bp contig every shipped case runs on, written once."""
    fa = str(tmp_path_factory.mktemp("shipped_ref") / "ref.fa")
    synth.write_fasta(fa, [("chr20", cases.SHIPPED_CHR20_LEN)], seed=20)
    return fa


def _build(name, wd, fasta):
    os.makedirs(wd, exist_ok=True)
    os.symlink(fasta, os.path.join(wd, "ref.fa"))   # cases._shipped keeps a ref.fa it finds
    return cases.build_case(name, wd)


def _oracle(oracle_lib, cfg, out_dir):
    rc = oracle_lib.orc_simulate(cfg.encode(), 1, SEED >> 32, SEED & 0xFFFFFFFF, out_dir.encode(), os.cpu_count() or 1)
    assert rc == 0, oracle_lib.orc_last_error().decode()
    return {f: (_md5_file(os.path.join(out_dir, f)), os.path.getsize(os.path.join(out_dir, f))) for f in sorted(os.listdir(out_dir))}


def _gpu(cfg, out, extra=(), env=None):
    r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", out, "--quiet", *extra], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return {f: (_md5_file(os.path.join(out, f)), os.path.getsize(os.path.join(out, f))) for f in sorted(os.listdir(out))}


@pytest.mark.parametrize("name", sorted(cases.SHIPPED_CASES) + ["c0_wgs_shipped_cov10"])
def test_shipped_fixture_config_identical_to_oracle(name, oracle_lib, chr20_fasta, tmp_path):
    cfg = _build(name, str(tmp_path / "case"), chr20_fasta)
    want = _oracle(oracle_lib, cfg, str(tmp_path / "oracle_out"))
    got = _gpu(cfg, str(tmp_path / "gpu_out"))
    assert got == want and want
    assert sum(sz for _, sz in want.values()) > 50_000_000


def test_baseline_c1_variants_of_the_engine(oracle_lib, chr20_fasta, tmp_path):
    """BASELINE configs[1] through the other routes that must not change a byte: BGZF sink, pieces of a few thousand
    fragments, the host planner, host haplotypes, and three ranks owning whole chromosomes (one contig: two idle ranks)."""
    cfg = _build("c1_wes_shipped_hs2000_cov50", str(tmp_path / "case"), chr20_fasta)
    want = _oracle(oracle_lib, cfg, str(tmp_path / "oracle_out"))
    for tag, extra, env in (("pieces", (), {"SIMU_PIECE_SLOTS": "20000"}), ("host_haps", ("--host-haplotypes",), {}),
                            ("ranks", ("--gpus", "3", "--shard-contigs"), {"SIMUSCOP_SAME_DEVICE": "1"})):
        got = _gpu(cfg, str(tmp_path / tag), extra, dict(os.environ, **env))
        assert got == want, tag
    gz = str(tmp_path / "gz")
    _gpu(cfg, gz, ("--gzip",))
    for f, (md5, size) in want.items():
        p = subprocess.run(["gzip", "-dc", os.path.join(gz, f + ".gz")], capture_output=True, timeout=600)
        assert p.returncode == 0 and len(p.stdout) == size and hashlib.md5(p.stdout).hexdigest() == md5, f


def _records(path):
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    return sorted(b"\n".join(lines[i:i + 4]) for i in range(0, len(lines) - 1, 4))


def test_exome_ranks_owning_whole_chromosomes(oracle_lib, tmp_path):
    """--shard-contigs on an exome run whose contigs' file order (chr9, chr10, chr2) differs from the target map's name
    order ("10", "2", "9"): ownership is by contig name, every rank plans exactly the contigs it ingested; the merged
    records are those of the one-GPU run, which is the oracle's text."""
    cfg = cases.build_case("wes_tight_targets_pe", str(tmp_path / "case"))
    want = _oracle(oracle_lib, cfg, str(tmp_path / "oracle_out"))
    one = str(tmp_path / "one")
    assert _gpu(cfg, one) == want
    d = str(tmp_path / "ranks")
    _gpu(cfg, d, ("--gpus", "3", "--shard-contigs"), dict(os.environ, SIMUSCOP_SAME_DEVICE="1"))
    assert sorted(os.listdir(d)) == sorted(os.listdir(one))
    for f in os.listdir(one):
        assert _records(os.path.join(d, f)) == _records(os.path.join(one, f)), f
