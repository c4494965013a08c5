#!/bin/bash
# usage: tools_pmc.sh <tag>  -- PMC passes over bench.py (each counter group in its own run)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py --steps 2 --warmup 1 --strong-scale 0 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_${tag}_${name}.err || echo "pmc $name failed"
done
