"""N>1 path on CPU: two gloo processes run the read-count balancing exchange of simuscop_amd.dist and
must agree with the reference's apportioning formula evaluated centrally."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
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
