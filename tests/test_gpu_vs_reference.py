"""North-star acceptance: reads produced on the MI355X have the same per-cycle quality, read-length and
start-position histograms as the UNMODIFIED reference CPU binary (oracle/_ref/simuReads, built by
oracle/Makefile from the reference's own sources; it travels to the GPU box as a prebuilt file).

The reference cannot be seeded, so this is a two-sample test: z = (chi2 - dof)/sqrt(2 dof) < 5 per
histogram family (false-failure probability < 1e-6 under H0).  The reference runs with a frozen clock
(oracle/fakeclock.c) so the test is deterministic."""
import os
import subprocess

import numpy as np
import pytest

import cases
from stats_util import Z_MAX, _chi2_z, _parse, _vec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "simuReads")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libfakeclock.so")
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(REF), reason="reference binary not built (make -C oracle ref)")
def test_histograms_match_reference_binary(tmp_path):
    wd_r, wd_g = tmp_path / "ref", tmp_path / "gpu"
    cfgs = {}
    for wd in (wd_r, wd_g):
        cfg = cases.build_case("wgs_pe_xten", str(wd))
        txt = open(cfg).read().replace("coverage = 5", "coverage = 60").replace("threads = 1", "threads = 4")
        open(cfg, "w").write(txt)
        cfgs[wd] = cfg
    env = dict(os.environ, LD_PRELOAD=SHIM, FAKECLOCK_SEC="1700000000", FAKECLOCK_NSEC="5")
    r = subprocess.run([REF, cfgs[wd_r]], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    g = subprocess.run([SIMU, cfgs[wd_g], "--seed", "20261003", "--quiet"], capture_output=True, text=True, timeout=600)
    assert g.returncode == 0, g.stderr[-2000:]
    fr, fg = cases.output_files(cfgs[wd_r]), cases.output_files(cfgs[wd_g])
    assert [os.path.basename(x) for x in fr] == [os.path.basename(x) for x in fg]
    for pr, pg in zip(fr, fg):
        nr, qr, lr, sr = _parse(pr, 250000)
        ng, qg, lg, sg = _parse(pg, 250000)
        assert abs(nr - ng) <= 0.002 * nr, (nr, ng)
        for name, (x, y) in {"quality by cycle": _vec(qr, qg), "read length": _vec(lr, lg),
                             "start position": _vec(sr, sg)}.items():
            z, dof = _chi2_z(x, y)
            assert z < Z_MAX, f"{os.path.basename(pr)} {name}: z={z:.2f} (dof {dof})"
    # per-cycle mismatch rate against the haplotype (mate 1, reads of nominal length)
    ref = "".join(l.strip() for l in open(os.path.join(str(wd_g), "ref.fa")) if not l.startswith(">")).upper()

    def mismatch_by_cycle(path):
        mm = np.zeros(10)
        tot = np.zeros(10)
        with open(path, "rb") as f:
            while True:
                hd = f.readline()
                if not hd:
                    break
                s = f.readline().strip().decode()
                f.readline(); f.readline()
                if len(s) != 151:
                    continue
                pos = int(hd.split(b"#")[2])
                rr = ref[pos:pos + 151]
                if len(rr) != 151 or "N" in rr:
                    continue
                d = np.frombuffer(s.encode(), dtype=np.uint8) != np.frombuffer(rr.encode(), dtype=np.uint8)
                if d.sum() > 20:
                    continue
                cyc = np.arange(151) * 10 // 151
                mm += np.bincount(cyc, weights=d, minlength=10)
                tot += np.bincount(cyc, minlength=10)
        return mm, tot

    mr, tr = mismatch_by_cycle(fr[0])
    mg, tg = mismatch_by_cycle(fg[0])
    p = (mr + mg) / (tr + tg)
    z = (mr / tr - mg / tg) / np.sqrt(p * (1 - p) * (1 / tr + 1 / tg))
    assert np.all(np.abs(z) < 5), z
