#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-name-substring> <tag> -- <command ...>   (GPU box)
# SQ / byte counters of ONE kernel (or, with ALL, of every kernel) of any command (separate --pmc passes), per-dispatch means
# printed as JSON lines, longest kernels first.
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
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("sg::", "").replace("(anonymous namespace)::", "")
        if k != "ALL" and k not in row["Kernel_Name"]:
            continue
        a = acc.setdefault(name if k == "ALL" else k, {}).setdefault(row["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
for name, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [1, 0])[1]):
    m = {c: s / n for c, (n, s) in cs.items()}
    calls = max(n for n, _ in cs.values())
    print(json.dumps({"kernel": name[:60], "dispatches_seen": calls, **{c: round(v, 1) for c, v in sorted(m.items())}}))
PY
rm -rf gpurun_out/pmck_${tag}_*
