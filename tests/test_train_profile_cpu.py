"""Profile training end to end on the CPU: the restatement of Profile::train (oracle/train_oracle.cpp:
processRead with countGC, known variants, targets, estimateGCParas, normParas(false), saveResults) on reads the oracle
sampled, and the ONE pin a reference run can give it here (no samtools, no BAM: the counting stays "parity unpinned"):
the unmodified reference binary LOADS the profile file written from those counts (Profile::load, Profile.cpp:934-1238) and
simulates from it exactly as oracle(mt) does.  tests/test_gpu_train.py holds the GPU path (the product: `seqToProfile`,
sg_train_*) against this restatement, byte for byte."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

import cases
import histo_util as H
import simuscop_amd
import test_train_counts as TC
import train_util as TU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "simuReads")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libfakeclock.so")


def declare(lib):
    cp, u64, u32, dp = C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)
    lib.orc_train.argtypes = [cp, u64, cp, cp, cp, cp, C.c_int, C.c_int, u32, u32, C.POINTER(simuscop_amd.SgTrainCounts), dp, dp, u64, C.POINTER(u64)]
    lib.orc_train_profile.argtypes = [cp, u64, cp, cp, cp, cp, C.c_int, C.c_int, cp, cp, cp]
    return lib


def parse_profile(path):
    """The sections of a .profile file as numbers (the format of Profile::saveResults / Profile::load)."""
    out, sec = {"head": {}}, None
    for line in open(path):
        line = line.rstrip("\n")
        if not line or line.startswith("#"):
            continue
        if line.startswith("["):
            sec = line.strip("[]")
            out[sec] = []
        elif sec is None:
            k, v = line.split(":")
            out["head"][k.strip()] = v.strip()
        elif line.startswith("kmer:") or line.startswith("basePairIndx:"):
            out[sec].append(line)
        else:
            out[sec].append([float(x) for x in line.split("\t")])
    return out


@pytest.fixture(scope="module")
def trained(oracle_lib, tmp_path_factory):
    wd = str(tmp_path_factory.mktemp("train"))
    declare(TC._declare(oracle_lib))
    sam1, fa1, T = TC.make_sam(oracle_lib, os.path.join(wd, "sim"), coverage=12, crafted=False)
    fa, vcf, bed, sam = TU.training_inputs(wd, fa1, sam1.rstrip(b"\n").split(b"\n"), T.L)
    prof = os.path.join(wd, "trained.profile")
    rc = oracle_lib.orc_train_profile(sam, len(sam), fa.encode(), vcf.encode(), b"", b"ACTG", 3, 50, prof.encode(), b"reads.bam", b"Thu Jan  1 00:00:00 1970\n")
    assert rc == 0
    return dict(wd=wd, fa=fa, fa1=fa1, vcf=vcf, sam=sam, profile=prof, T=T)


def test_trained_profile_has_the_shape_of_a_shipped_one(trained):
    P = parse_profile(trained["profile"])
    S = parse_profile(os.path.join(cases.TESTDATA, cases.PROFILES["xten"]))
    assert P["head"] == S["head"]
    for sec in ("Substitution Probs", "Base Quality Distribution", "Log Ratio Mean Value"):
        assert len(P[sec]) == len(S[sec]), sec
    assert os.path.exists(trained["profile"] + ".gc")          # the GC model was fitted (median window count >= 5)
    means = np.array([r[1] for r in P["Log Ratio Mean Value"]])
    assert means.shape == (101,) and 0.8 < means[35:55].mean() < 1.2 and P["Log Ratio Standard Deviation"][0][0] > 0
    # rates and the insert-size spread come back close to what the reads were sampled with
    # (the sampler cuts its normal at three deviations and at the read length, Profile.cpp:912-930: narrower than the parameter)
    assert 0.8 < P["Insert Size Standard Deviation"][0][0] / S["Insert Size Standard Deviation"][0][0] < 1.0
    # (rates: events per CIGAR character, Profile.cpp:294,897 -- not per base; only their order of magnitude is the sampler's)
    assert 0 < P["Insert Rate"][0][0] < 1 and 0 < P["Deletion Rate"][0][0] < 1


@pytest.mark.skipif(not os.path.exists(REF), reason="reference binary not built (make -C oracle ref; needs /root/reference)")
@pytest.mark.parametrize("layout", ["PE", "SE"])
def test_reference_binary_loads_the_trained_profile_and_samples_like_the_oracle(layout, trained, oracle_lib, tmp_path):
    """Profile::load of the unmodified binary reads the written file; a run from it is, byte for byte, oracle(mt)'s."""
    cfg = os.path.join(str(tmp_path), "config.txt")
    out = os.path.join(str(tmp_path), "out")
    cases._config(cfg, ref=trained["fa1"], profile=trained["profile"], name="t", output=out, layout=layout, threads=1, verbose=0,
                  coverage=2, insertSize=350, ploidy=2)
    env = dict(os.environ, LD_PRELOAD=SHIM, FAKECLOCK_SEC=str(cases.FAKE_SEC), FAKECLOCK_NSEC=str(cases.FAKE_NSEC))
    r = subprocess.run([REF, cfg], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    md5 = lambda d: {f: hashlib.md5(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))}   # noqa: E731
    want = md5(out)
    assert want and all(os.path.getsize(os.path.join(out, f)) > 1_000_000 for f in want)
    for f in os.listdir(out):
        os.remove(os.path.join(out, f))
    assert oracle_lib.orc_simulate(cfg.encode(), 0, cases.FAKE_SEC, cases.FAKE_NSEC, b"", 1) == 0, oracle_lib.orc_last_error().decode()
    assert md5(out) == want


def test_count_gc_turns_reads_away_and_opens_windows(trained, oracle_lib):
    """countGC on the whole-genome input: X / M reads, reads behind the window, the shrunken window after the short contig."""
    T = trained["T"]
    sam, fa, vcf = trained["sam"], trained["fa"], trained["vcf"]
    st, a = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
    gc, rc, n = (C.c_double * 100000)(), (C.c_double * 100000)(), C.c_uint64()
    assert oracle_lib.orc_train(sam, len(sam), fa.encode(), vcf.encode(), b"", T.bases.encode(), 3, T.bins, 2048, 256, C.byref(st), gc, rc, 100000, C.byref(n)) == 0
    plain, pa = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
    assert oracle_lib.orc_train_count(sam, len(sam), fa.encode(), T.bases.encode(), 3, T.bins, 2048, 256, C.byref(plain)) == 0
    assert st.gc_rejected >= 60 + 100                      # the X and M reads, most of the reads that step backwards
    assert st.reads_counted < plain.reads_counted and st.reads_counted > 0.9 * plain.reads_counted
    assert 1300 < n.value <= st.gc_windows                  # ~1,400 windows of chr1, a few of the small contigs
    g, r = np.array(gc[:n.value]), np.array(rc[:n.value])
    assert (g > 0).all() and (g <= 1).all() and (r >= 1).all() and np.median(r) > 50
    # known events are not counted: one of the two crafted insertions / deletions each (the other length / place is)
    assert a["ins_len"][3] >= 1 and a["del_len"][3] >= 1
    # (without the VCF both of each count; reads countGC turned away are not walked at all)
    assert pa["ins_len"][2] >= a["ins_len"][2] + 1 and pa["del_len"][3] >= a["del_len"][3] + 1
    assert plain.insert_events > st.insert_events and plain.delete_events > st.delete_events


def test_exome_targets_are_the_windows(trained, oracle_lib, tmp_path):
    T = trained["T"]
    fa, vcf, bed, sam = TU.training_inputs(str(tmp_path), trained["fa1"], trained["sam"].split(b"\n")[:60000], T.L, exome=True)
    st, a = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
    gc, rc, n = (C.c_double * 100000)(), (C.c_double * 100000)(), C.c_uint64()
    assert oracle_lib.orc_train(sam, len(sam), fa.encode(), vcf.encode(), bed.encode(), T.bases.encode(), 3, T.bins, 2048, 256, C.byref(st), gc, rc, 100000, C.byref(n)) == 0
    assert 50 < n.value < 1000 and st.gc_rejected > st.reads_counted      # most reads lie between the targets


def test_the_run_ends_at_the_cap_of_counted_reads(trained, oracle_lib):
    """Profile::processRead returns 2 once `readCount` reaches `maxCount` (300,000,000; Profile.cpp:236, 497-507) and
    Profile::train leaves its loop (:1461-1464): lines behind that read are never read.  With a small cap: exactly so many
    reads counted, fewer lines seen, fewer windows; the text cut behind the capping line gives the same counters."""
    T, sam, fa, vcf = trained["T"], trained["sam"], trained["fa"], trained["vcf"]
    oracle_lib.orc_train_set_max_reads.argtypes = [C.c_uint64]
    oracle_lib.orc_train_set_max_reads.restype = None
    full, _ = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
    n = C.c_uint64()
    assert oracle_lib.orc_train(sam, len(sam), fa.encode(), vcf.encode(), b"", T.bases.encode(), 3, T.bins, 2048, 256, C.byref(full), None, None, 0, C.byref(n)) == 0
    assert full.capped == 0
    try:
        oracle_lib.orc_train_set_max_reads(5000)
        cut, ca = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
        assert oracle_lib.orc_train(sam, len(sam), fa.encode(), vcf.encode(), b"", T.bases.encode(), 3, T.bins, 2048, 256, C.byref(cut), None, None, 0, C.byref(n)) == 0
        assert cut.capped == 1 and cut.reads_counted == 5000 and cut.lines < full.lines and cut.gc_windows < full.gc_windows
        # the same text without what lies behind the capping line: the same counters
        lines = sam.split(b"\n")
        part = b"\n".join(lines[:cut.lines]) + b"\n"
        oracle_lib.orc_train_set_max_reads(0)
        same, sa = TU.count_arrays(simuscop_amd.SgTrainCounts, T.kc, T.bins, 2048)
        assert oracle_lib.orc_train(part, len(part), fa.encode(), vcf.encode(), b"", T.bases.encode(), 3, T.bins, 2048, 256, C.byref(same), None, None, 0, C.byref(n)) == 0
        assert same.capped == 0 and same.reads_counted == 5000 and same.lines == cut.lines
        for k in ca:
            assert np.array_equal(ca[k], sa[k]), k
    finally:
        oracle_lib.orc_train_set_max_reads(0)
