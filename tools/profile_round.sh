#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box, from the repo root)
# The evidence of one kernel generation, written to gpurun_out/<tag>/ (copy it to profiles/<tag>/ afterwards):
#   kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5` (C2 workload)
#   pmc_summary.json    per-kernel means of separate --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ groups + GRBM_GUI_ACTIVE)
#   bench.json          the bench line of an unprofiled run of the same build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/$tag
mkdir -p $out
B="python3 bench.py --steps 20 --warmup 5 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5"   # (as many launches as the plain bench line times: the first few run at lower clocks)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > /dev/null 2> $out/trace.err || echo "trace failed"
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py --steps 2 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5 > /dev/null 2> $out/pmc_${name}.err || echo "pmc $name failed"
done
python3 tools/pmc_summarize.py $tag $out/pmc_summary.json > $out/pmc_summarize.txt
# the kernel time of the traced run travels with the counters: bench.py refuses counters whose kernel is not the live one
python3 - $out <<'PY'
import csv, json, sys
out = sys.argv[1]
ms = None
for row in csv.DictReader(open(out + "/kernel_stats.csv", newline="")):
    if "emit_fast_kernel" in row["Name"]:
        ms = float(row["AverageNs"]) / 1e6
d = json.load(open(out + "/pmc_summary.json"))
sys.path.insert(0, ".")
from simuscop_amd.build import engine_source_digest
d["_meta"] = {"emit_fast_kernel_avg_ms": ms, "from": "kernel_stats.csv of the same profile_round.sh run", "engine_source_digest": engine_source_digest()}
json.dump(d, open(out + "/pmc_summary.json", "w"), indent=1)
PY
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
rm -rf $out/trace gpurun_out/pmc_${tag}_*
ls -la $out
