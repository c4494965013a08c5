#!/bin/bash
# usage: tools/gz_pmc.sh <tag> -- SQ counter passes over a --gzip CLI run of the C2 workload (makes /tmp/e2e_c2 if missing)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
[ -f /tmp/e2e_c2/config.txt ] || bash tools/e2e_c2.sh > /dev/null 2>&1
R="./simuscop_amd/lib/simuReads /tmp/e2e_c2/config.txt --no-write --fetch --gzip --quiet"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_$1_a -- $R > /dev/null 2> gpurun_out/pmc_$1_a.err || echo "pmc a failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_$1_b -- $R > /dev/null 2> gpurun_out/pmc_$1_b.err || echo "pmc b failed"
python3 - <<PY
import csv, glob, json
acc = {}
for path in glob.glob("gpurun_out/pmc_$1_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "gz_" not in k: continue
        a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
res = {k: {c: round(s / n, 1) for c, (n, s) in sorted(v.items())} for k, v in sorted(acc.items())}
json.dump(res, open("gpurun_out/gz_pmc_$1.json", "w"), indent=1)
for k, v in res.items(): print(k, v)
PY
rm -rf gpurun_out/pmc_$1_a gpurun_out/pmc_$1_b
