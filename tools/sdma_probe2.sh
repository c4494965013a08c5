# usage: tools/sdma_probe2.sh   the pinned-host legs with the box's own setting and with HSA_ENABLE_SDMA=0 (copies made by a
# shader, as on some boxes of the pool), the runtime's copy against drain_copy_kernel (SG_D2H_KERNEL=<workgroups>)
echo "box: HSA_ENABLE_SDMA=${HSA_ENABLE_SDMA-unset} GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES-unset}"
run() {
  python3 bench.py --steps 5 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-md5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_pinned']; print('$1', 'plain %.1fM %.1f GB/s' % (h['plain']['value']/1e6, h['plain']['pcie_GBps']), 'gzip %.1fM %.1f ms' % (h['gzip']['value']/1e6, h['gzip']['ms_per_step']))"
}
run "box default                 "
for w in 16 64 256; do SG_D2H_KERNEL=$w run "box, kernel copy $w wgs     "; done
export HSA_ENABLE_SDMA=0
run "SDMA=0                      "
for w in 16 64 256; do SG_D2H_KERNEL=$w run "SDMA=0, kernel copy $w wgs  "; done
SG_D2H_KERNEL=64 SG_OUT_STREAM_PRIORITY=-1 run "SDMA=0, kernel 64, high prio"
