#!/bin/bash
# emit-kernel ablation sweep (timing only; outputs are wrong when SG_DIAG != 0)
for d in 0 1 2 4 8 16 32 64 128 192 255; do
  SG_DIAG=$d python bench.py --steps 3 --warmup 1 --strong-scale 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('diag', $d, 'emit_ms %.2f' % d['kernel_ms_per_step']['emit'], 'indel %.2f' % d['kernel_ms_per_step']['indel'])"
done
