#!/bin/bash
# usage: tools/fdiag_p16.sh [profile]   the 16-base plain steps against the 8-base ones inside ONE build (SG_FDIAG 512 = no 16-base
# steps; 1024 = the ablation build with nothing ablated), alone and with stores (1) / Philox (8) / haplotype fetch (4) compiled out
P=${1:-xten}
for d in 1024 1536 1025 1537 1032 1544 1028 1540 1152 1280; do
  SG_FDIAG=$d python bench.py --steps 10 --warmup 2 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5 --profile $P 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$P fdiag', $d, 'emit_ms %.3f' % d['kernel_ms_per_step']['emit'], 'indel %.3f' % d['kernel_ms_per_step']['indel'])"
done
