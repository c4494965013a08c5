#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-name-substring> <tag> -- <command ...>   (GPU box)
# SQ / byte counters of ONE kernel of any command (separate --pmc passes), per-dispatch means printed as JSON.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
k=$1; tag=$2; shift 3
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmck_${tag}_${name} -- "$@" > /dev/null 2> gpurun_out/pmck_${tag}_${name}.err || echo "pmc $name failed"
done
python3 - "$k" "$tag" <<'PY'
import csv, glob, json, sys
k, tag = sys.argv[1], sys.argv[2]
acc = {}
for path in glob.glob(f"gpurun_out/pmck_{tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if k not in row["Kernel_Name"]:
            continue
        a = acc.setdefault(row["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
print(json.dumps({c: round(s / n, 1) for c, (n, s) in sorted(acc.items())}))
PY
rm -rf gpurun_out/pmck_${tag}_*
