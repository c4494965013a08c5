#!/bin/bash
# usage: pmc_quick.sh <tag> [bench.py args, e.g. --profile hs2000] -- SQ counter passes over bench.py (instruction mix, wave
# states, clock of the kernels); summary printed and written to gpurun_out/pmc_<tag>.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift
B="python3 bench.py --steps 2 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_${tag}_SQ_WAVE_CYCLES -- $B > /dev/null 2> gpurun_out/pmc_$tag.err || echo "pmc failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_SQ_INSTS_LDS -- $B > /dev/null 2> gpurun_out/pmc_${tag}b.err || echo "pmc b failed"
python3 tools/pmc_summarize.py $tag gpurun_out/pmc_$tag.json
rm -rf gpurun_out/pmc_${tag}_SQ_WAVE_CYCLES gpurun_out/pmc_${tag}_SQ_INSTS_LDS
