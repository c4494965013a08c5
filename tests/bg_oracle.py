"""Background runner of the long oracle(mt) cases (TEST INFRASTRUCTURE).

The shipped-fixture cases (cases.SHIPPED_CASES: the reference's own testData on a 63 Mbp contig) and the
full-coverage C3 / C4 cases (cases.SLOW_CASES) take 20-200 s each on the sequential mt-mode oracle.  The CPU suite
starts this script once at session start (tests/conftest.py); it runs the cases on a few worker processes while the
other tests go on, and tests/test_zz_long_oracle_cases.py -- collected last -- compares the md5 sums it leaves with the
reference binary's in tests/golden/golden.json.

    python tests/bg_oracle.py <result dir> <workers> <case> [<case> ...]
writes <result dir>/<case>.json = {"files": {name: {"md5", "bytes"}}, "reads": n, "seconds": t} or {"error": text}.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import cases  # noqa: E402

# longest first (measured on this container's cores, seconds): the pool then finishes as early as it can
COST = {"c4_tumor_pe_xten_cov60": 200, "c0_wgs_shipped_cov10": 190, "c3_grch38_pe_xten_cov30": 160, "c1_wes_shipped_hs2500": 50,
        "tumor_shipped": 45, "c1_wes_shipped_hs2000_cov50": 35, "c0_wgs_shipped": 25}


def _md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def run_case(name, result_dir):
    wd = tempfile.mkdtemp(prefix="bg_oracle_" + name + "_")
    out = {}
    try:
        t0 = time.time()
        cfg = cases.build_case(name, wd)
        r = subprocess.run([os.path.join(ROOT, "oracle", "oracle_cli"), cfg, "--rng", "mt", "--sec", str(cases.FAKE_SEC),
                            "--nsec", str(cases.FAKE_NSEC)], capture_output=True, text=True)
        if r.returncode != 0:
            out = {"error": r.stderr[-2000:]}
        else:
            files = {os.path.basename(f): {"md5": _md5(f), "bytes": os.path.getsize(f)} for f in cases.output_files(cfg)}
            reads = int(r.stderr.split("oracle:")[1].split()[0])
            out = {"files": files, "reads": reads, "seconds": round(time.time() - t0, 1)}
    except Exception as e:  # noqa: BLE001 -- whatever went wrong is the test's message
        out = {"error": repr(e)}
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    tmp = os.path.join(result_dir, name + ".json.tmp")
    with open(tmp, "w") as f:
        json.dump(out, f)
    os.rename(tmp, os.path.join(result_dir, name + ".json"))


def main():
    result_dir, workers, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    os.makedirs(result_dir, exist_ok=True)
    names.sort(key=lambda n: -COST.get(n, 30))
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:   # threads that wait for child processes
        list(ex.map(lambda n: run_case(n, result_dir), names))


if __name__ == "__main__":
    main()
