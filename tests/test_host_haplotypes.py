"""CPU-only: the haplotype copy lists the device assembles from (Genome::segment_pieces -> sg_build_haplotypes)
against the reference's std::string editing restated on the host (Genome::segment_haplotypes,
lib/segment/Segment.cpp:210-447), byte for byte, for every (population, chromosome) of a configuration."""
import ctypes as C
import random

import pytest

import cases
import simuscop_amd


def _selftest(cfg, seed):
    host = simuscop_amd.load_host()
    err = C.create_string_buffer(2048)
    rc = host.simu_selftest_haplotypes(cfg.encode(), seed, err, len(err))
    return rc, err.value.decode(errors="replace")


@pytest.mark.parametrize("name", ["wgs_pe_variants", "tumor_se_mixture", "wes_pe_targets", "tiny_contigs_pe",
                                  "degenerate_inputs_pe", "wgs_se_hs2000"])
def test_copy_lists_reproduce_the_edited_strings(name, tmp_path):
    cfg = cases.build_case(name, str(tmp_path))
    for seed in (1, 0x5EED5EED1234):
        rc, msg = _selftest(cfg, seed)
        assert rc == 0, msg


@pytest.mark.parametrize("case_seed", range(700, 740))
def test_copy_lists_on_random_variant_sets(case_seed, tmp_path):
    """Dense random variants on a small contig: overlapping and adjacent indels, events at segment ends,
    het/homo mixes, copy-number gains and losses with 1-3 copies of a haplotype."""
    from simuscop_amd import synth
    rng = random.Random(case_seed)
    L = rng.choice([3000, 20000, 120000])
    fa = str(tmp_path / "ref.fa")
    synth.write_fasta(fa, [("chr1", L), ("chr2", 2500)], seed=case_seed, n_runs=False)
    rows = []
    for _ in range(rng.choice([3, 20, 120])):
        pos = rng.randrange(1, L)
        kind = rng.choice("sid")
        zyg = rng.choice(["het", "homo"])
        if kind == "s":
            ref = rng.choice("ACGT")
            alt = rng.choice([b for b in "acgtACGT" if b.upper() != ref])
            rows.append(f"s\tpop\t1\t{pos}\t{ref}\t{alt}\t{zyg}")
        elif kind == "i":
            rows.append(f"i\tpop\t1\t{pos}\t{''.join(rng.choice('acgtACGTN') for _ in range(rng.choice([1, 2, 9, 40])))}\t{zyg}")
        else:
            rows.append(f"d\tpop\t1\t{pos}\t{rng.choice([1, 2, 7, 55])}\t{zyg}")
    pos = 1
    while pos < L - 500 and rng.random() < 0.8:   # non-overlapping copy-number segments
        s = pos + rng.randrange(1, max(2, L // 6))
        e = min(L, s + rng.randrange(50, max(51, L // 5)))
        cn = rng.choice([0, 1, 1, 3, 4, 5])
        mcn = rng.randrange(max(0, cn - 2) if cn >= 2 else 0, cn + 1) if cn else 0
        if cn >= 2:
            mcn = max(mcn, (cn + 1) // 2)
        rows.append(f"c\tpop\t1\t{s}\t{e}\t{cn}\t{mcn}")
        pos = e + 1
    rng.shuffle(rows)
    # the parsers expect each class sorted by position as in the reference's test data
    rows.sort(key=lambda r: (r.split("\t")[0], int(r.split("\t")[3])))
    cases._write(str(tmp_path / "variations.txt"), rows)
    snps = cases._snps("1", L, rng.choice([97, 800]), case_seed) if rng.random() < 0.6 else []
    kv = dict(ref=fa, profile=cases.os.path.join(cases.TESTDATA, cases.PROFILES["xten"]), name="pop",
              output=str(tmp_path / "out"), layout="PE", threads=1, verbose=0, coverage=1, insertSize=350,
              variation=str(tmp_path / "variations.txt"))
    if snps:
        cases._write(str(tmp_path / "snp.txt"), snps)
        kv["snp"] = str(tmp_path / "snp.txt")
    cfg = str(tmp_path / "config.txt")
    cases._config(cfg, **kv)
    rc, msg = _selftest(cfg, case_seed)
    # inputs both routes refuse (an edit outside its haplotype, a segment lost to CN 0 ...) must be refused alike:
    # the self-test then reports the string route's own error, never a difference
    assert rc == 0 or ("differs" not in msg and "lengths" not in msg and "tile" not in msg and "offsets" not in msg), msg
