# usage: tools/ab_write_size.sh [dirs...]   WRITE_SIZE (KiB per launch of emit_fast_kernel) and emit time of engine builds on one box
# ('.' = this tree, others = built copies such as _ab)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in ${@:-_ab .}; do
  (cd $d && rm -rf /tmp/ws_$$ && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ws_$$ -- python3 bench.py --steps 3 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5 > /dev/null 2>&1
   python3 - <<PY
import csv, glob
v=[]
for f in glob.glob('/tmp/ws_$$/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'emit_fast_kernel' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE': v.append(float(r['Counter_Value']))
print('$d', 'WRITE_SIZE per launch: %.0f KiB over %d launches' % (sum(v)/max(1,len(v)), len(v)))
PY
  )
done
AB_ROUNDS=2 bash tools/ab_compare.sh $@
