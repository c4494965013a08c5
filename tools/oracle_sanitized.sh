#!/bin/bash
# usage: tools/oracle_sanitized.sh   (CPU only) the oracle under AddressSanitizer + UndefinedBehaviorSanitizer: the CPU restatement is
# the checker of every parity test, so it should at least be free of out-of-bounds reads and undefined arithmetic.  Builds
# oracle_cli into /tmp/oracle_san and runs five parity cases in both draw modes; prints one line per run.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/oracle_san; mkdir -p $OUT
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -I$ROOT/oracle \
    $ROOT/oracle/oracle.cpp $ROOT/oracle/train_oracle.cpp $ROOT/oracle/oracle_cli.cpp -o $OUT/oracle_cli_san
python3 - "$ROOT" "$OUT" <<'PY'
import os, subprocess, sys
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import cases
bad = 0
for name in ("wgs_pe_variants", "wes_tight_targets_pe", "indel_rich_n_islands_pe", "indel_storm_se", "tiny_contigs_pe"):
    wd = os.path.join(out, name)
    cfg = cases.build_case(name, wd)
    for mode, threads in (("mt", "1"), ("philox", "2")):
        r = subprocess.run([os.path.join(out, "oracle_cli_san"), cfg, "--rng", mode, "--sec", str(cases.FAKE_SEC), "--nsec", str(cases.FAKE_NSEC),
                            "--out", os.path.join(wd, "out_" + mode), "--threads", threads], capture_output=True, text=True)
        hits = [l for l in r.stderr.splitlines() if "runtime error" in l or "ERROR: " in l]
        bad += r.returncode != 0 or bool(hits)
        print(name, mode, "rc", r.returncode, hits[:2] or "clean")
sys.exit(1 if bad else 0)
PY
