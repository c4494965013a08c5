for v in default 1 0; do
  if [ $v = default ]; then unset HSA_ENABLE_SDMA; else export HSA_ENABLE_SDMA=$v; fi
  python3 bench.py --steps 5 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-md5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_pinned']; print('HSA_ENABLE_SDMA=$v', 'value %.0fM' % (d['value']/1e6), 'emit %.3f' % d['kernel_ms_per_step']['emit'], 'plain %.1fM %.1f GB/s' % (h['plain']['value']/1e6, h['plain']['pcie_GBps']), 'gzip %.1fM %.1f ms' % (h['gzip']['value']/1e6, h['gzip']['ms_per_step']))"
done
