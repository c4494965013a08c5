"""GPU parity: the HIP path (through the C ABI, driven by the C++ host `simuReads`) against the CPU
oracle in Philox mode, same seed, same inputs.  Bar: FASTQ files byte-identical (integer/byte work)."""
import os
import re
import subprocess

import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMU = os.path.join(ROOT, "simuscop_amd", "lib", "simuReads")
SEED = (cases.FAKE_SEC << 32) | cases.FAKE_NSEC

pytestmark = pytest.mark.gpu


def _free_port():
    """A port nobody listens on right now (a fixed one can still be held by the previous run's rendezvous)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _run_gpu(cfg, out, extra=(), env=None):
    r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", out, "--quiet", "--stats", *extra],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def _files(d):
    return sorted(x for x in os.listdir(d) if ".fq" in x)


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_fastq_identical_to_oracle(name, oracle_lib, tmp_path):
    cfg = cases.build_case(name, str(tmp_path))
    odir, gdir = str(tmp_path / "oracle_out"), str(tmp_path / "gpu_out")
    rc = oracle_lib.orc_simulate(cfg.encode(), 1, cases.FAKE_SEC, cases.FAKE_NSEC, odir.encode(), 4)
    assert rc == 0, oracle_lib.orc_last_error().decode()
    _run_gpu(cfg, gdir)
    assert _files(odir) == _files(gdir) and _files(odir)
    for f in _files(odir):
        a = open(os.path.join(odir, f), "rb").read()
        b = open(os.path.join(gdir, f), "rb").read()
        if a != b:
            n = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
            lo = a.rfind(b"\n@", 0, n) + 1
            pytest.fail(f"{name}/{f}: first difference at byte {n} (sizes {len(a)} vs {len(b)})\n"
                        f"oracle: {a[lo:lo + 400]!r}\ngpu:    {b[lo:lo + 400]!r}")


def _stat(stderr, key):
    return int(re.search(key + r"=(\d+)", stderr).group(1))


@pytest.mark.parametrize("name", ["indel_rich_n_islands_pe", "wgs_pe_variants"])
def test_generic_item_queue_and_its_overflow_path(name, tmp_path):
    """The straight-line emit kernel queues the items it cannot do (non-ACGT windows, >= 2 indels) for
    emit_slow_kernel; a queue too small makes the engine emit the batch again with the generic kernel.
    Both routes and the generic kernel alone (SG_DIAG=0) must write the same bytes, and so must a run whose
    batches are cut into many memory-bounded pieces (SIMU_PIECE_SLOTS), and a run that plans its windows on the host
    (SIMU_HOST_PLAN; the default makes the window table on the device: sg_windows_build / sg_plan_windows), and a run
    whose one-indel reads stay whole in the general steps (SG_NO_CLEAN_STEPS: what profiles with a table image too large
    for the clean-item list get) or only some of them fit the list (SG_CLEAN_CAP=64: the first in lane order are split,
    the others stay whole)."""
    cfg = cases.build_case(name, str(tmp_path))
    outs = {}
    for tag, extra_env in (("queue", {}), ("overflow", {"SG_SLOWQ_CAP": "3"}), ("generic", {"SG_DIAG": "0"}),
                           ("pieces", {"SIMU_PIECE_SLOTS": "1500"}), ("host_plan", {"SIMU_HOST_PLAN": "1"}),
                           ("host_plan_pieces", {"SIMU_HOST_PLAN": "1", "SIMU_PIECE_SLOTS": "1500"}),
                           ("no_clean_steps", {"SG_NO_CLEAN_STEPS": "1"}), ("small_clean_list", {"SG_CLEAN_CAP": "64"})):
        d = str(tmp_path / tag)
        err = _run_gpu(cfg, d, env=dict(os.environ, **extra_env))
        outs[tag] = {f: open(os.path.join(d, f), "rb").read() for f in _files(d)}
        if tag == "queue":
            assert _stat(err, "queued_items") > 0 and _stat(err, "requeued_batches") == 0, err
        if tag == "overflow":
            assert _stat(err, "requeued_batches") > 0, err
        if tag == "generic":
            assert _stat(err, "queued_items") == 0, err
    assert outs["queue"] == outs["overflow"] == outs["generic"] == outs["pieces"] == outs["host_plan"] == outs["host_plan_pieces"] == outs["no_clean_steps"] == outs["small_clean_list"] and outs["queue"]


def _records(blob):
    lines = blob.split(b"\n")
    assert lines[-1] == b"" and (len(lines) - 1) % 4 == 0
    return [b"\n".join(lines[i:i + 4]) for i in range(0, len(lines) - 1, 4)]


def test_sharded_run_equals_unsharded(tmp_path):
    """Multi-GPU sharding by runs of segments keeps every draw's address: per batch the shards' FASTQ
    concatenate to the single-GPU text, so over the whole run the part files hold exactly the same
    records (parts interleave per chromosome, hence the multiset comparison)."""
    cfg = cases.build_case("wgs_pe_variants", str(tmp_path))
    one = str(tmp_path / "one")
    _run_gpu(cfg, one)
    parts = str(tmp_path / "parts")
    world = 3
    for r in range(world):
        _run_gpu(cfg, parts, ("--rank", str(r), "--world", str(world)))
    for f in _files(one):
        whole = _records(open(os.path.join(one, f), "rb").read())
        got, nonempty = [], 0
        for r in range(world):
            recs = _records(open(os.path.join(parts, f + f".part{r}"), "rb").read())
            nonempty += bool(recs)
            got += recs
        assert nonempty >= 2
        assert sorted(whole) == sorted(got), f


def test_launcher_two_ranks_merge(tmp_path):
    """simuscop_amd.run with two ranks (gloo rehearsal, both on device 0): merged output holds exactly the
    records of the single-process run."""
    import sys
    cfg = cases.build_case("wes_pe_targets", str(tmp_path))
    one = str(tmp_path / "one")
    _run_gpu(cfg, one)
    env = dict(os.environ, SIMUSCOP_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), "-m", "simuscop_amd.run", cfg,
                        "--seed", str(SEED), "--merge", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out_dir = [l.split("=", 1)[1].strip() for l in open(cfg) if l.startswith("output")][0]
    for f in _files(one):
        whole = _records(open(os.path.join(one, f), "rb").read())
        merged = _records(open(os.path.join(out_dir, f), "rb").read())
        assert sorted(whole) == sorted(merged), f


def test_cli_gpus_option_equals_one_gpu(tmp_path):
    """simuReads --gpus 3 (child process per GPU, here all on device 0): the merged files hold exactly the
    records of the one-GPU run, also as BGZF."""
    cfg = cases.build_case("wgs_pe_variants", str(tmp_path))
    one = str(tmp_path / "one")
    _run_gpu(cfg, one)
    env = dict(os.environ, SIMUSCOP_SAME_DEVICE="1")
    for tag, extra in (("three", []), ("three_gz", ["--gzip"])):
        d = str(tmp_path / tag)
        r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", d, "--quiet", "--gpus", "3", *extra],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        for f in _files(one):
            name = f + (".gz" if extra else "")
            assert sorted(os.listdir(d)) == sorted(x + (".gz" if extra else "") for x in _files(one))
            blob = open(os.path.join(d, name), "rb").read()
            if extra:
                import gzip
                blob = gzip.decompress(blob)
            assert sorted(_records(blob)) == sorted(_records(open(os.path.join(one, f), "rb").read())), (tag, f)


@pytest.mark.parametrize("name", ["wgs_pe_variants", "c3_grch38_pe_xten_cov3", "c4_tumor_pe_xten_cov6"])
def test_ranks_owning_whole_chromosomes(name, tmp_path):
    """--shard-contigs: every rank ingests, scans and samples only the chromosomes it owns (the weighted lengths of the
    others come through the exchange); the merged files hold exactly the records of the one-GPU run.  Through the CLI's
    own process-per-GPU launcher (pipes to the parent) and through the torchrun front end (gloo all-reduce)."""
    import sys
    cfg = cases.build_case(name, str(tmp_path))
    one = str(tmp_path / "one")
    _run_gpu(cfg, one)
    env = dict(os.environ, SIMUSCOP_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    d = str(tmp_path / "cli")
    r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", d, "--quiet", "--gpus", "3", "--shard-contigs"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _files(d) == _files(one)
    for f in _files(one):
        assert sorted(_records(open(os.path.join(d, f), "rb").read())) == sorted(_records(open(os.path.join(one, f), "rb").read())), f
    if name != "wgs_pe_variants":
        return
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", _free_port(), "-m", "simuscop_amd.run", cfg, "--seed", str(SEED), "--merge", "--backend", "gloo",
                        "--shard-contigs"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out_dir = [l.split("=", 1)[1].strip() for l in open(cfg) if l.startswith("output")][0]
    for f in _files(one):
        assert sorted(_records(open(os.path.join(out_dir, f), "rb").read())) == sorted(_records(open(os.path.join(one, f), "rb").read())), f


@pytest.mark.parametrize("workload", ["c3", "c4"])
def test_bench_strong_scaling_modes(workload):
    """bench.py --workload c3 / c4 (the whole run of a 24-contig genome per step; c4: four populations with variants):
    one rank and two ranks sharing the device (gloo rehearsal of the RCCL exchange) sample the same number of pairs, the
    number the configuration asks for."""
    import json
    import sys
    from simuscop_amd import synth
    scale = 0.02
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--scale", str(scale), "--steps", "1", "--warmup", "0",
            "--no-cpu-baseline"]
    lines = []
    for extra, env in (([], {}), (["--gpus", "2", "--backend", "gloo"], {"BENCH_SAME_DEVICE": "1"})):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        lines.append(json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]))
    one, two = lines
    assert (one["n_gpus"], two["n_gpus"]) == (1, 2) and one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["workload"].startswith(workload.upper())
    pairs = one["config"]["pairs_per_step"]
    assert pairs == two["config"]["pairs_per_step"] == sum(r["pairs_per_step"] for r in two["per_rank"])
    coverage = 60 if workload == "c4" else 30
    want = sum(length for _, length in synth.grch38_contigs(scale)) * coverage / 151 / 2
    assert abs(pairs - want) < 0.02 * want      # ceil(n/2) pairs per window: a little above reads / 2
    assert all(r["pairs_per_step"] > 0 for r in two["per_rank"])


def test_sharded_ranks_do_only_their_own_share():
    """Strong scaling is by construction: with N ranks owning whole chromosomes, what a rank ingests, scans, plans and samples
    is its chromosomes' share of the genome and nothing else.  bench.py --workload c3 on four ranks sharing the device (gloo
    rehearsal of the RCCL exchange) against one rank: the ranks' windows, segments and pairs add up to the one-rank run's,
    each rank's part follows its share of the bases."""
    import json
    import sys
    from simuscop_amd import synth
    scale = 0.25
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--scale", str(scale), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    lines = []
    for extra, env in (([], {}), (["--gpus", "4", "--backend", "gloo"], {"BENCH_SAME_DEVICE": "1"})):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        lines.append(json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]))
    one, four = lines
    one_r = one["per_rank"][0]
    lens = [length for _, length in synth.grch38_contigs(scale)]
    import ctypes
    import simuscop_amd
    host = simuscop_amd.load_host()
    owner = (ctypes.c_int32 * len(lens))()
    host.simu_assign_contigs((ctypes.c_uint64 * len(lens))(*lens), len(lens), 4, owner)
    share = [sum(l for l, o in zip(lens, owner) if o == r) / sum(lens) for r in range(4)]
    assert max(share) < 0.27 and min(share) > 0.23
    for key in ("windows", "segments", "batches"):
        assert sum(v["work_last_run"][key] for v in four["per_rank"]) == one_r["work_last_run"][key], key
    assert abs(sum(v["pairs_per_step"] for v in four["per_rank"]) - one_r["pairs_per_step"]) < 1e-6 * one_r["pairs_per_step"]
    for v in four["per_rank"]:
        s_r = share[v["rank"]]
        assert abs(v["work_last_run"]["windows"] / one_r["work_last_run"]["windows"] - s_r) < 0.01, v
        assert abs(v["pairs_per_step"] / one_r["pairs_per_step"] - s_r) < 0.02, v
        # (phase TIMES are not held to a bound here: the four processes share one device, one PCIe link and one page cache;
        # what a rank does -- windows, segments, batches, pairs -- is what is checked)


def test_a_stale_fasta_index_is_not_trusted(tmp_path):
    """--shard-contigs reads a rank's contigs through the .fai when one is there.  An index that is newer than the file but
    belongs to another genome (same size, other line breaks / another contig order) must not be believed: the header scan
    takes over and the records are those of the one-GPU run."""
    from simuscop_amd import synth
    cfg = cases.build_case("wgs_pe_variants", str(tmp_path))
    one = str(tmp_path / "one")
    _run_gpu(cfg, one)
    fa = os.path.join(str(tmp_path), "ref.fa")
    contigs = [("chr20", 3200000), ("chr21", 1234567), ("chrM", 777)]
    synth.write_fai(fa, contigs)                       # the true index ...
    rows = open(fa + ".fai").read().split("\n")
    f0, f1 = rows[0].split("\t"), rows[1].split("\t")
    # ... bent: the first two contigs trade 10 kbp (offsets stay inside the file, every range ends inside it)
    f0[1] = str(int(f0[1]) - 10020)
    f1[2] = str(int(f1[2]) - 10020 - 10020 // 60)
    f1[1] = str(int(f1[1]) + 10020)
    open(fa + ".fai", "w").write("\n".join(["\t".join(f0), "\t".join(f1)] + rows[2:]))
    env = dict(os.environ, SIMUSCOP_SAME_DEVICE="1")
    d = str(tmp_path / "ranks")
    r = subprocess.run([SIMU, cfg, "--seed", str(SEED), "--out", d, "--quiet", "--gpus", "3", "--shard-contigs"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in _files(one):
        assert sorted(_records(open(os.path.join(d, f), "rb").read())) == sorted(_records(open(os.path.join(one, f), "rb").read())), f


def test_default_bench_line_on_two_ranks():
    """The command the driver's scaling run uses -- `bench.py --gpus N`, no workload flag -- on two ranks sharing the device
    (gloo rehearsal of the RCCL collectives): one line with the weak C2 figure of both ranks AND the strong-scaling leg of one
    genome over both (`strong_c3`), whose ranks' pairs add up to the genome's."""
    import json
    import sys
    from simuscop_amd import synth
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--contig-len", "8000000", "--strong-scale", "0.02", "--no-md5"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, BENCH_SAME_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    leg = line["strong_c3"]
    assert leg["n_gpus"] == 2 and leg["scaling"] == "strong" and len(leg["per_rank"]) == 2
    want = sum(length for _, length in synth.grch38_contigs(0.02)) * 30 / 151 / 2
    assert abs(leg["pairs_per_run"] - want) < 0.02 * want
    assert abs(sum(v["pairs_per_step"] for v in leg["per_rank"]) - leg["pairs_per_run"]) < 1e-6 * want
    assert all(v["pairs_per_step"] > 0.4 * leg["pairs_per_run"] for v in leg["per_rank"])
