#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the UNMODIFIED reference binary.

Needs /root/reference (this container only).  Builds oracle/_ref/simuReads (make -C oracle ref:
g++ on the reference's own sources where they lie) and runs it on every case of tests/cases.py
under oracle/fakeclock.c (frozen wall clock => frozen RNG seeds) with threads = 1.  Only md5 sums,
sizes and read counts are kept (data, not source).

    python tests/golden/make_golden.py [case ...]     named cases (default: every case of cases.CASES and of
                                                      cases.SHIPPED_CASES, the reference's own testData fixtures)
    python tests/golden/make_golden.py --slow          also the full-coverage C3 / C4 / C0 cases (cases.SLOW_CASES: minutes each)
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-j8"], stdout=subprocess.DEVNULL)
    ref = os.path.join(ROOT, "oracle", "_ref", "simuReads")
    shim = os.path.join(ROOT, "oracle", "_ref", "libfakeclock.so")
    only = [a for a in sys.argv[1:] if a != "--slow"] or list(cases.CASES) + list(cases.SHIPPED_CASES)
    if "--slow" in sys.argv:
        only += [n for n in cases.SLOW_CASES if n not in only]
    path = os.path.join(HERE, "golden.json")
    golden = json.load(open(path)) if os.path.exists(path) else {}
    for name in only:
        wd = tempfile.mkdtemp(prefix="golden_" + name + "_")
        try:
            cfg = cases.build_case(name, wd)
            env = dict(os.environ, LD_PRELOAD=shim, FAKECLOCK_SEC=str(cases.FAKE_SEC), FAKECLOCK_NSEC=str(cases.FAKE_NSEC))
            subprocess.check_call([ref, cfg], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            files = {}
            for fq in cases.output_files(cfg):
                with open(fq, "rb") as f:
                    nl = sum(blk.count(b"\n") for blk in iter(lambda: f.read(1 << 20), b""))
                files[os.path.basename(fq)] = {"md5": md5(fq), "bytes": os.path.getsize(fq), "reads": nl // 4}
            golden[name] = {"fake_sec": cases.FAKE_SEC, "fake_nsec": cases.FAKE_NSEC, "files": files}
            print(name, json.dumps(files))
        finally:
            shutil.rmtree(wd, ignore_errors=True)
    with open(path, "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
