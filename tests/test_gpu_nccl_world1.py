"""The RCCL transport of the sharded path on the ONE GPU a test box has: a process group of world size 1 over the `nccl`
backend (= RCCL on ROCm), so that the all_gather of the read balancing (simuscop_amd.dist.balance_reads), the max / sum
all-reduces of bench.py, the ok-flag all-reduce before its strong leg and the weighted-length exchange of the chromosome-owning
mode (make_exchange("cuda"), called back from the C++ driver) have all executed on device tensors through the real
backend before the first 8-GPU run.  (N > 1 is covered by the gloo tests -- tests/test_dist_gloo.py on the CPU,
tests/test_gpu_parity.py with ranks sharing the device; RCCL with N > 1 is the driver's SCALE run.)"""
import json
import os
import subprocess
import sys

import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


@pytest.mark.timeout(600)
def test_bench_line_through_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-process-group", "--steps", "2", "--warmup", "1",
                        "--contig-len", "3000000", "--strong-scale", "0.01", "--no-cpu-baseline", "--no-host-pinned", "--no-md5"],
                       capture_output=True, text=True, timeout=580, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0
    pg = d["process_group"]
    assert pg["backend"].startswith("rccl") and pg["world"] == 1
    # the balancing all_gather ran once for the weak leg; the strong leg's runs each exchanged their weighted lengths
    assert pg["exchange_collectives_rank0"]["all_gather"] >= 1
    assert pg["exchange_collectives_rank0"]["all_reduce"] >= 2, pg
    leg = d["strong_c3"]
    assert "value" in leg and leg["value"] > 0, leg
    assert "gap_ms_per_step" in d and "frac_at_step_level" in d["roofline"]


@pytest.mark.timeout(600)
def test_chromosome_owning_front_end_through_rccl_with_one_rank(tmp_path):
    """python -m simuscop_amd.run --shard-contigs with a one-rank nccl group: same reads as the plain run, byte for byte."""
    cfg = cases.build_case("wgs_pe_variants", str(tmp_path / "c"))   # three contigs, variants, copy-number changes
    outs = []
    for tag, extra in (("plain", []), ("pg", ["--force-process-group", "--shard-contigs"])):
        cfg_t = str(tmp_path / (tag + ".txt"))
        out_dir = str(tmp_path / tag)
        text = "".join(("output = " + out_dir + "\n") if l.strip().startswith("output") else l for l in open(cfg))
        open(cfg_t, "w").write(text)
        r = subprocess.run([sys.executable, "-m", "simuscop_amd.run", cfg_t, "--seed", "77", *extra], capture_output=True, text=True,
                           timeout=580, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append({f: open(os.path.join(out_dir, f), "rb").read() for f in sorted(os.listdir(out_dir))})
    assert outs[0] and outs[0] == outs[1]
