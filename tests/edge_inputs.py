"""Degenerate configurations around two parity cases (shared by test_oracle_edge_inputs_vs_reference.py and
test_gpu_edge_inputs.py): what the reference does with them -- mostly quietly -- the oracle and the GPU path must do too."""
import os
import re

import cases


def _edit(text, **kv):
    for k, v in kv.items():
        if re.search(rf"^{k} = .*$", text, re.M):
            text = re.sub(rf"^{k} = .*$", f"{k} = {v}", text, flags=re.M)
        else:
            text += f"{k} = {v}\n"
    return text


def _bed_variants(bed):
    five = bed[:5]
    return {
        "bed_empty": [],
        "bed_absent_chromosome": five + ["chrZ\t100\t900"],
        "bed_beyond_contig_end": five + ["chr21\t899900\t900500"],
        "bed_start_after_end": five + ["chr21\t5000\t4000"],
        "bed_zero_length": five + ["chr21\t5000\t5000"],
        "bed_at_position_0": ["chr20\t0\t400"] + five,
        "bed_rows_twice": five + five,
        "bed_one_ten_base_target": ["chr20\t40000\t40010"],
        "bed_names_without_chr": [r.replace("chr", "") for r in bed[:6]],
    }


WGS_VARIANTS = {
    "coverage_0": dict(coverage=0), "coverage_fraction_below_1": dict(coverage=0.01), "coverage_3_7": dict(coverage=3.7),
    "insert_below_read_length": dict(insertSize=60), "insert_equal_read_length": dict(insertSize=74), "insert_5000": dict(insertSize=5000),
    "ploidy_4": dict(ploidy=4), "ploidy_1_with_cn_gain": dict(ploidy=1), "layout_se": dict(layout="SE"), "threads_0": dict(threads=0),
}
NAMES = sorted(WGS_VARIANTS) + sorted(_bed_variants(["chr20\t1\t2"] * 6))


def build(name, workdir):
    """Write the inputs of variant `name` under workdir; returns the config path (its output directory: workdir/out)."""
    os.makedirs(workdir, exist_ok=True)
    if name in WGS_VARIANTS:
        base = cases.build_case("wgs_pe_variants", os.path.join(workdir, "base"))
        text = _edit(open(base).read(), **WGS_VARIANTS[name])
    else:
        base = cases.build_case("wes_pe_targets", os.path.join(workdir, "base"))
        bed = open(os.path.join(workdir, "base", "targets.bed")).read().splitlines()
        rows = _bed_variants(bed)[name]
        bp = os.path.join(workdir, "targets.bed")
        with open(bp, "w") as f:
            f.write("\n".join(rows) + ("\n" if rows else ""))
        text = _edit(open(base).read(), target=bp, coverage=10)
    text = _edit(text, output=os.path.join(workdir, "out"))
    cfg = os.path.join(workdir, "config.txt")
    with open(cfg, "w") as f:
        f.write(text)
    return cfg
