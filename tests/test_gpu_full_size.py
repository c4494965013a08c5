"""GPU parity at production size: the bench workload itself and the named multi-contig / tumour configurations.

* C2 (BASELINE configs[2], the workload bench.py times): BOTH whole mates of the full-size pass, md5 against the CPU
  oracle in Philox mode on all host cores -- and the md5 bench.py prints for the pass it timed is that same value.
* The same chromosome at 64x: each mate is > 4 GiB of text (64-bit record offsets, > 2^24 slots).
* C3 (24 contigs in GRCh38 proportions, XTen PE 30x) through the CLI, unsharded and as eight shards (--world 8), and
  C4 (four populations, tumour variation pattern, abundance row, XTen PE 60x): byte-identical to the oracle.
  The reference binary pins the same cases in tests/golden/golden.json (mt mode).
Bar: bit-exact (integer / byte work)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

import cases
import simuscop_amd
from simuscop_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

BENCH_SEED = 0x5EED0000   # bench.py rank 0


def _md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def _oracle_md5(oracle_lib, cfg, out_dir, seed):
    """FASTQ md5 sums of the oracle in Philox mode, threads = all host cores (results do not depend on the count)."""
    rc = oracle_lib.orc_simulate(cfg.encode(), 1, seed >> 32, seed & 0xFFFFFFFF, out_dir.encode(), os.cpu_count() or 1)
    assert rc == 0, oracle_lib.orc_last_error().decode()
    from concurrent.futures import ThreadPoolExecutor
    names = sorted(os.listdir(out_dir))
    with ThreadPoolExecutor(max_workers=2) as pool:   # (hashlib releases the GIL: the mates' digests side by side)
        sums = list(pool.map(lambda f: _md5_file(os.path.join(out_dir, f)), names))
    out = {}
    for f, m in zip(names, sums):
        out[f] = (m, os.path.getsize(os.path.join(out_dir, f)))
        os.remove(os.path.join(out_dir, f))   # gigabytes each
    return out


def _c2_config(wd, coverage):
    import bench
    fa = os.path.join(wd, "ref.fa")
    synth.write_fasta(fa, [("chr20", bench.CHR20_LEN)], seed=20)   # the bytes bench.py writes for rank 0
    cfg = os.path.join(wd, "config.txt")
    bench.write_config(cfg, fa, os.path.join(wd, "out"), coverage=coverage)
    return cfg


def _gpu_md5(cfg, seed):
    sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=seed)
    try:
        sess.weighted_length()
        sess.set_reads(sess.planned_reads)
        assert sess.prepare_batch(0)
        sess.sample()
        b1, b2, nf = sess.result()
        return sess.output_md5(), (b1, b2), nf
    finally:
        sess.close()


def test_c2_whole_output_md5_equals_oracle_and_bench(oracle_lib, tmp_path):
    cfg = _c2_config(str(tmp_path), 30)
    want = _oracle_md5(oracle_lib, cfg, str(tmp_path / "oracle_out"), BENCH_SEED)
    md5s, sizes, nf = _gpu_md5(cfg, BENCH_SEED)
    assert nf > 6_000_000
    assert (md5s[0], sizes[0]) == want["sim_1.fq"] and (md5s[1], sizes[1]) == want["sim_2.fq"]
    # ... and bench.py prints the digest of what it timed
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-host-pinned", "--strong-scale", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["output_md5"] == md5s and line["output_seed"] == BENCH_SEED


def test_mates_larger_than_4_GiB(oracle_lib, tmp_path):
    cfg = _c2_config(str(tmp_path), 64)
    want = _oracle_md5(oracle_lib, cfg, str(tmp_path / "oracle_out"), 99)
    md5s, sizes, nf = _gpu_md5(cfg, 99)
    assert sizes[0] > (1 << 32) and sizes[1] > (1 << 32) and nf > (1 << 23)
    assert (md5s[0], sizes[0]) == want["sim_1.fq"] and (md5s[1], sizes[1]) == want["sim_2.fq"]


def _run_cli(cfg, out, extra=()):
    r = subprocess.run([SIMU, cfg, "--seed", str(BENCH_SEED), "--out", out, "--quiet", *extra], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]


def _record_digest(oracle_lib, paths):
    """Order-independent digest of the FASTQ records of the files: (count, sum of 64-bit record hashes); the loop over
    the bytes is oracle/fastq_digest.cpp (four million records in Python took a third of this test)."""
    import ctypes as C
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    n, total = C.c_uint64(), C.c_uint64()
    oracle_lib.orc_fastq_record_digest.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    oracle_lib.orc_fastq_record_digest.restype = C.c_int
    assert oracle_lib.orc_fastq_record_digest(arr, len(paths), C.byref(n), C.byref(total)) == 0, paths
    return n.value, total.value


@pytest.mark.parametrize("name", ["c3_grch38_pe_xten_cov30", "c4_tumor_pe_xten_cov60"])
def test_named_configs_at_their_own_coverage(name, oracle_lib, tmp_path):
    cfg = cases.build_case(name, str(tmp_path))
    want = _oracle_md5(oracle_lib, cfg, str(tmp_path / "oracle_out"), BENCH_SEED)
    one = str(tmp_path / "one")
    _run_cli(cfg, one)
    got = {f: (_md5_file(os.path.join(one, f)), os.path.getsize(os.path.join(one, f))) for f in sorted(os.listdir(one))}
    assert got == want and len(want) == 2
    if name.startswith("c3"):
        # eight shards (runs of segments of every chromosome): together exactly the records of the one-GPU run
        parts = str(tmp_path / "parts")
        for r in range(8):
            _run_cli(cfg, parts, ("--rank", str(r), "--world", "8"))
        for f in sorted(os.listdir(one)):
            whole = _record_digest(oracle_lib, [os.path.join(one, f)])
            sharded = _record_digest(oracle_lib, [os.path.join(parts, f"{f}.part{r}") for r in range(8)])
            assert whole == sharded and whole[0] > 2_000_000, f


def test_bgzf_of_mates_larger_than_4_GiB(tmp_path):
    """The block-gzip sink at production size: C2 at 64x, each mate > 4 GiB of text and > 140 k members; the whole
    compressed stream of both mates, piped through `gzip -dc`, is exactly the text the context holds."""
    import threading
    cfg = _c2_config(str(tmp_path), 64)
    sess = simuscop_amd.Session(cfg, device=0, write_files=0, quiet=1, seed=99)
    try:
        sess.weighted_length()
        sess.set_reads(sess.planned_reads)
        assert sess.prepare_batch(0)
        sess.sample()
        b1, b2, nf = sess.result()
        assert b1 > (1 << 32) and b2 > (1 << 32)
        want = sess.output_md5()
        g = sess.compress()
        # both mates at once: `gzip -dc` on one core is the slow part (the context is used under a lock: one call at a time)
        lock = threading.Lock()
        results, errors = {}, []

        def one_mate(mate, total, gz):
            try:
                p = subprocess.Popen(["gzip", "-dc"], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
                got = {}

                def digest():
                    h, n = hashlib.md5(), 0
                    for blk in iter(lambda: p.stdout.read(1 << 24), b""):
                        h.update(blk)
                        n += len(blk)
                    got["md5"], got["bytes"] = h.hexdigest(), n

                t = threading.Thread(target=digest)
                t.start()
                for off in range(0, gz, 1 << 27):
                    with lock:
                        piece = sess.fetch_compressed(mate, min(1 << 27, gz - off), off)
                    p.stdin.write(piece)
                    print(f"mate {mate}: {off >> 20} MiB of {gz >> 20} piped", flush=True)   # (a long quiet run is taken for a hang)
                p.stdin.close()
                t.join()
                got["exit"] = p.wait()
                results[mate] = got
            except Exception as e:  # noqa: BLE001
                errors.append((mate, repr(e)))

        mates = ((0, b1, g[0]), (1, b2, g[1]))
        for mate, total, gz in mates:
            assert total / gz > 3.2
        workers = [threading.Thread(target=one_mate, args=m) for m in mates]
        for w in workers:
            w.start()
        for w in workers:
            w.join()
        assert not errors, errors
        for mate, total, gz in mates:
            assert results[mate] == {"md5": want[mate], "bytes": total, "exit": 0}, mate
    finally:
        sess.close()
