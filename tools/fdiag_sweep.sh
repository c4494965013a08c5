#!/bin/bash
# usage: tools/fdiag_sweep.sh [profile]   straight-line emit kernel ablation sweep (timing only; outputs are wrong when SG_FDIAG != 0)
#   1 no item stores, 2 no per-read pass, 4 no haplotype fetch, 8 no Philox, 16 no fix-up loop
P=${1:-xten}
for d in 0 1 2 3 4 8 16 24 28 31; do
  SG_FDIAG=$d python bench.py --steps 5 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5 --profile $P 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$P fdiag', $d, 'emit_ms %.3f' % d['kernel_ms_per_step']['emit'], 'indel %.3f' % d['kernel_ms_per_step']['indel'])"
done
