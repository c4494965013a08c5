#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box, from the repo root)
# The evidence of one kernel generation, written to gpurun_out/<tag>/ (copy it to profiles/<tag>/ afterwards):
#   kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 1` (C2 workload)
#   pmc_summary.json    per-kernel means of separate --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ groups + GRBM_GUI_ACTIVE)
#   bench.json          the bench line of an unprofiled run of the same build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/$tag
mkdir -p $out
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-pinned --no-md5"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > /dev/null 2> $out/trace.err || echo "trace failed"
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-pinned --no-md5 > /dev/null 2> $out/pmc_${name}.err || echo "pmc $name failed"
done
python3 tools/pmc_summarize.py $tag $out/pmc_summary.json > $out/pmc_summarize.txt
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
rm -rf $out/trace gpurun_out/pmc_${tag}_*
ls -la $out
