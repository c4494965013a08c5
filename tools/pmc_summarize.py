#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/pmc_passes.sh to per-kernel means.

    python tools/pmc_summarize.py <tag> <out.json>

Reads gpurun_out/pmc_<tag>_*/**/*counter_collection.csv; every (kernel, counter) gets the number of
dispatches seen and the mean counter value per dispatch (FETCH_SIZE / WRITE_SIZE stay in the KiB
rocprofv3 reports; bench.py applies the gfx950 corrections when it quotes `roofline.traffic`)."""
import csv
import glob
import json
import sys


def main():
    tag, out = sys.argv[1], sys.argv[2]
    acc = {}
    for path in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    res = {k: {c: {"dispatches": n, "mean_per_dispatch": s / n} for c, (n, s) in sorted(v.items())}
           for k, v in sorted(acc.items())}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        if "emit" in k or "indel" in k:
            print(k, {c: round(d["mean_per_dispatch"], 1) for c, d in v.items()})


if __name__ == "__main__":
    main()
