"""What the MI355X EMITTED against the closed-form expectation from the `.profile`, and against the reference binary.

The byte-parity chain is reference == oracle(mt) and oracle(philox) == GPU; the two oracle modes differ in four sampling
primitives.  These tests close the loop from the GPU's own text, without the oracle's sampler:

  primitive                                   | pinned here by
  --------------------------------------------+-------------------------------------------------------------------------
  1 addressed Philox draws (all of them)      | every family below
  2 identity-first rows / alias columns       | G1 substitution counts per (mate, bin, 3-mer context, called base) and G2
                                              | quality counts per (bin, ref, called, symbol), all four shipped profiles (W = 8
                                              | and W = 64 alias columns), both mates of PE runs (mate-2 table, Profile.cpp:
                                              | 1420-1430) and SE runs with both strands -- against subsCdf1/2 and qualityCdf
  3 indel candidates by geometric skip-ahead  | G3 read-length histogram against its exact law (first loop of predict,
                                              | Profile.cpp:1607-1634, as a forward recursion)
  4 normal variate from the quantile table    | G5 per-window pair counts -> the GC factor each window drew, against
                                              | N(gcMeans[gc], gcStd) truncated at 0 (getGCFactor, :1507-1517): moments and
                                              | Kolmogorov distance; the same statistic on the REFERENCE BINARY's output, and
                                              | a two-sample test between the two; mean and variance by GC%
  -- insert-size rows (threshold compare)     | G4 recovered fragment lengths against normParas' pmf (:912-930)
  -- population split of mixtures             | G8 reads per population against reads * prop * ACN / w_acn (Genome.cpp:935)

Bar: |z| < 5, Bonferroni-corrected exact binomial p > 1e-4; seeds are fixed, the tests are deterministic."""
import json
import os
import subprocess

import numpy as np
import pytest
from scipy import stats as sps

import cases
import histo_util as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")
REF = os.path.join(ROOT, "oracle", "_ref", "simuReads")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libfakeclock.so")

pytestmark = pytest.mark.gpu

INSERT = {"xten": 350, "hs2500": 200, "hs2000": 200, "gaiix": 250}


def _gpu(cfg, seed, out):
    r = subprocess.run([SIMU, cfg, "--seed", str(seed), "--out", out, "--quiet"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return sorted(os.path.join(out, f) for f in os.listdir(out))


@pytest.mark.parametrize("layout", ["PE", "SE"])
@pytest.mark.parametrize("profile", sorted(cases.PROFILES))
def test_emitted_histograms_meet_the_profile(profile, layout, oracle_lib, tmp_path):
    """~100 M sampled bases per configuration (G1 needs counts in 50 bins x 64 contexts x 3 substituted outcomes)."""
    L = {"xten": 151, "hs2500": 125, "hs2000": 75, "gaiix": 74}[profile]
    coverage = 70 if layout == "PE" else 36
    cfg, fa = H.histogram_config(cases, str(tmp_path), profile, layout, coverage, INSERT[profile])
    files = _gpu(cfg, 20261004 + L, str(tmp_path / "gpu"))
    rep = H.analyse_run(oracle_lib, cases, profile, layout, INSERT[profile], fa, files, f"GPU {profile} {layout}")
    print(json.dumps(rep))
    assert rep["mate1"]["reads_used"] > 250_000 and rep["mate1"]["g1_offdiag_dof"] > 1000 and rep["gc_factor"]["n"] >= 1390
    if layout == "PE":
        assert rep["mate2"]["reads_used"] > 250_000 and rep["insert_size"]["pairs"] > 250_000


@pytest.mark.skipif(not os.path.exists(REF), reason="reference binary not built (make -C oracle ref)")
@pytest.mark.parametrize("profile", ["xten", "gaiix"])
def test_gc_factor_law_gpu_and_reference_binary(profile, oracle_lib, tmp_path):
    """G5 on both sides: three runs each (other seeds / other frozen clocks) over the same 1.4 Mbp contig; the factor every
    window drew, mapped through the model's law, is standard normal on the GPU AND on the reference binary (which draws
    from std::normal_distribution), the two samples agree (two-sample Kolmogorov-Smirnov), and so do the mean and the
    variance of the pair counts by GC% -- what a bent quantile table would move."""
    insert = INSERT[profile]
    T = H.ProfileTables(oracle_lib, os.path.join(cases.TESTDATA, cases.PROFILES[profile]), True, insert)
    zs = {"gpu": [], "ref": []}
    fs = {"gpu": [], "ref": []}
    gcs = {"gpu": [], "ref": []}
    # the reference binary's three runs start first and work beside the GPU's (CPU time: two thirds of this test)
    runs = {}
    for k in range(3):
        for side in ("ref", "gpu"):
            wd = str(tmp_path / f"{side}{k}")
            os.makedirs(wd)
            cfg, fa = H.histogram_config(cases, wd, profile, "PE", 40, insert)
            if side == "gpu":
                runs[side, k] = (cfg, fa, _gpu(cfg, 777 + k, os.path.join(wd, "out")))
            else:
                txt = open(cfg).read().replace("threads = 1", "threads = 5")
                open(cfg, "w").write(txt)
                # a clock that advances between calls, as in the field: with a frozen one the reference's 101 GC generators
                # (Profile.cpp:1409-1415) share one seed and every GC% draws the same sequence of factors
                env = dict(os.environ, LD_PRELOAD=SHIM, FAKECLOCK_SEC=str(1700000000 + 977 * k), FAKECLOCK_NSEC="5", FAKECLOCK_STEP_NSEC="1373", FAKECLOCK_NO_PIN="1")
                runs[side, k] = (cfg, fa, subprocess.Popen([REF, cfg], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    for k in range(3):
        for side in ("gpu", "ref"):
            cfg, fa, files = runs[side, k]
            if side == "ref":
                _, err = files.communicate(timeout=900)
                assert files.returncode == 0, err[-2000:]
                files = cases.output_files(cfg)
            ref = H.read_fasta_one(fa)
            fq = H.Fastq(files[0])
            z, g, c = H.gc_factor_z(T, ref, fq.pos, True)
            zs[side].append(z)
            gcs[side].append(g)
            cnt = np.bincount(fq.pos // 1000, minlength=len(ref) // 1000 + 1)[1:len(ref) // 1000]
            fs[side].append(cnt / c)
    rep = {}
    for side in ("gpu", "ref"):
        z = np.concatenate(zs[side])
        r = H.normal_report(z)
        rep[side] = r
        assert abs(r["z_var"]) < H.Z_MAX and abs(r["z_skew"]) < H.Z_MAX and abs(r["z_kurt"]) < H.Z_MAX and r["ks_p"] > 1e-6, (side, r)
    ks = sps.ks_2samp(np.concatenate(zs["gpu"]), np.concatenate(zs["ref"]))
    rep["two_sample_ks"] = {"D": round(float(ks.statistic), 5), "p": float(ks.pvalue)}
    assert ks.pvalue > 1e-6, rep
    # mean and variance of the windows' factors by GC%
    fg, fr = np.concatenate(fs["gpu"]), np.concatenate(fs["ref"])
    gg, gr = np.concatenate(gcs["gpu"]), np.concatenate(gcs["ref"])
    worst_m = worst_v = 0.0
    for g in np.unique(gg):
        a, b = fg[gg == g], fr[gr == g]
        if len(a) < 40 or len(b) < 40:
            continue
        zm = (a.mean() - b.mean()) / np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        zv = (np.log(a.var(ddof=1)) - np.log(b.var(ddof=1))) / np.sqrt(2.0 / (len(a) - 1) + 2.0 / (len(b) - 1))
        worst_m, worst_v = max(worst_m, abs(zm)), max(worst_v, abs(zv))
    rep["by_gc"] = {"max_abs_z_mean": round(float(worst_m), 2), "max_abs_z_log_variance": round(float(worst_v), 2)}
    print(json.dumps(rep))
    assert worst_m < H.Z_MAX and worst_v < H.Z_MAX, rep


def _acn(rows, popu, chr_len, ploidy=2):
    """Genome::calculateACNs (Genome.cpp:765-781) from the CNV rows of one population on one contig, cut the way
    generateSegments walks them (:634-682: file order, overlaps kept)."""
    nxt, total = 1, 0
    for r in rows:
        f = r.split("\t")
        if f[0] != "c" or f[1] != popu:
            continue
        s, e, cn = int(f[3]), min(int(f[4]), chr_len), int(float(f[5]))
        if nxt > chr_len:
            break
        if nxt < s:
            total += ploidy * (s - nxt)
        total += cn * (e - s + 1)
        nxt = e + 1
    if nxt <= chr_len:
        total += ploidy * (chr_len - nxt + 1)
    return total / chr_len


def test_population_split_of_a_mixture(tmp_path):
    """G8: reads per population of an SE tumour mixture = (long)(reads * prop * ACN / w_acn) (Genome.cpp:935; SE emits
    exactly the planned reads of every window)."""
    cfg = cases.build_case("tumor_se_mixture", str(tmp_path))
    conf = dict(l.strip().split(" = ", 1) for l in open(cfg) if " = " in l)
    files = _gpu(cfg, 4242, str(tmp_path / "gpu"))
    rows = open(conf["variation"]).read().split("\n")
    popus = [p.strip() for p in conf["name"].split(",")]
    chr_len = 2600000
    acn = {p: _acn(rows, p, chr_len) for p in popus}
    assert len(set(acn.values())) > 1                      # the clones really differ in copy number
    reads = chr_len * int(conf["coverage"]) // 74
    mixes = [[np.float32(x) for x in l.split()] for l in open(conf["abundance"]).read().strip().split("\n")]
    assert len(files) == len(mixes) == 2
    for props in mixes:
        stem = "+".join("%s_%.3f" % (p, x) for p, x in zip(popus, props))
        path = [f for f in files if os.path.basename(f) == stem + ".fq"]
        assert path, stem
        got = H.Fastq(path[0]).populations()
        w_acn = sum(float(x) * acn[p] for p, x in zip(popus, props))
        for p, x in zip(popus, props):
            want = int(float(np.float32(reads) * x) * acn[p] / w_acn)
            assert abs(got.get(p.encode(), 0) - want) <= 2, (stem, p, got.get(p.encode(), 0), want)
