"""N>1 path on CPU: two gloo processes run the read-count balancing exchange of simuscop_amd.dist and
must agree with the reference's apportioning formula evaluated centrally."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port nobody listens on right now (a fixed one can still be held by the previous run's rendezvous)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from simuscop_amd import dist as sd
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    wl = [31234.56789, 30111.2223][rank]
    length = [64444167, 58617616][rank]
    mine, total = sd.balance_reads(wl, length, 30, 151)
    print(json.dumps({"rank": rank, "mine": mine, "total": total}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_balance_two_ranks(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    import re
    rows = sorted((json.loads(m) for m in re.findall(r'\{"rank".*?\}', r.stdout)), key=lambda d: d["rank"])
    assert len(rows) == 2
    from simuscop_amd import dist as sd
    total = (64444167 + 58617616) * 30 // 151
    exp = sd.apportion(total, [31234.56789, 30111.2223])
    assert [d["mine"] for d in rows] == exp
    assert sum(exp) == total and all(d["total"] == total for d in rows)


def test_apportion_matches_reference_rounding():
    from simuscop_amd import dist as sd
    # last chromosome takes the remainder, others truncate reads*(w/WL)
    out = sd.apportion(1000, [1.0, 1.0, 1.0])
    assert out == [333, 333, 334]
    assert sd.apportion(7, [5.0]) == [7]


# ---- ranks owning different chromosomes (simuReads --shard-contigs / simuscop_amd.run --shard-contigs) ----
OWNER_WORKER = textwrap.dedent("""
    import os, sys, json, ctypes as C
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import simuscop_amd
    from simuscop_amd import dist as sd, synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    host = simuscop_amd.load_host()
    lens = synth.GRCH38_LENGTHS
    owner = (C.c_int32 * len(lens))()
    host.simu_assign_contigs((C.c_uint64 * len(lens))(*lens), len(lens), world, owner)
    # a rank only knows the weighted lengths of its own chromosomes; the exchange is the callback the C++ driver gets
    wl = [l * (0.9 + 0.01 * (i %% 7)) / 1000.0 for i, l in enumerate(lens)]
    mine = (C.c_double * len(lens))(*[w if owner[i] == rank else 0.0 for i, w in enumerate(wl)])
    ex = sd.make_exchange()
    rc = ex(None, mine, len(lens))
    reads = sum(lens) * 30 // 151
    print(json.dumps({"rank": rank, "rc": rc, "owner": list(owner), "wl": list(mine), "per_chr": sd.apportion(reads, list(mine))}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_ranks_owning_different_chromosomes_apportion_like_one_rank(tmp_path):
    """Each rank owns the chromosomes simu_assign_contigs gives it and contributes only their weighted lengths; after the
    exchange every rank holds the full table bit for bit and derives the same per-chromosome read counts as one rank."""
    script = tmp_path / "w.py"
    script.write_text(OWNER_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    import re
    # (the two ranks share the pipe: their lines can arrive glued together, so objects are cut out, not lines)
    rows = sorted((json.loads(m) for m in re.findall(r'\{"rank".*?\}', r.stdout)), key=lambda d: d["rank"])
    assert len(rows) == 2 and all(d["rc"] == 0 for d in rows)
    from simuscop_amd import dist as sd, synth
    lens = synth.GRCH38_LENGTHS
    wl = [l * (0.9 + 0.01 * (i % 7)) / 1000.0 for i, l in enumerate(lens)]
    assert rows[0]["wl"] == wl and rows[1]["wl"] == wl              # exact: one contributor per entry
    reads = sum(lens) * 30 // 151
    assert rows[0]["per_chr"] == rows[1]["per_chr"] == sd.apportion(reads, wl)
    owner = rows[0]["owner"]
    assert owner == rows[1]["owner"] and set(owner) == {0, 1}
    load = [sum(l for l, o in zip(lens, owner) if o == k) for k in (0, 1)]
    assert abs(load[0] - load[1]) < 0.02 * sum(lens)               # longest-first keeps the two shares within 2 %


def test_contig_assignment_is_longest_first():
    import ctypes as C
    import simuscop_amd
    host = simuscop_amd.load_host()
    lens = [10, 50, 20, 50, 5]
    owner = (C.c_int32 * 5)()
    host.simu_assign_contigs((C.c_uint64 * 5)(*lens), 5, 3, owner)
    # 50 (index 1) -> rank 0, 50 (index 3) -> rank 1, 20 -> rank 2, 10 -> rank 2 (load 20 is the smallest), 5 -> rank 2
    assert list(owner) == [2, 0, 2, 1, 2]
    host.simu_assign_contigs((C.c_uint64 * 5)(*lens), 5, 1, owner)
    assert list(owner) == [0] * 5
