#!/bin/bash
# usage: tools/pmc_stalls.sh <tag> [repo dir]   (on the GPU box, from the repo root)
# Where the emit kernel's cycles go beyond instruction counts: issue waits, the vector-memory write path (TA fifos, TCP -> TCC
# write latency, TCC -> fabric write stalls).  Separate --pmc passes (the counters do not fit one), summary in
# gpurun_out/<tag>/pmc_stalls.json.  [repo dir]: a built copy of the repo to profile instead of this one (A/B: _ab).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
dir=${2:-.}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
           "TA_BUSY_avr TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  (cd $dir && rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_g$i -- python3 bench.py --steps 2 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5 > /dev/null 2> $out/pmc_g$i.err) || echo "pmc pass $i failed"
done
python3 tools/pmc_summarize.py $tag $out/pmc_stalls.json > $out/pmc_stalls.txt
rm -rf gpurun_out/pmc_${tag}_g*
cat $out/pmc_stalls.txt
