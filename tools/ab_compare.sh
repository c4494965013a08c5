# A/B timing of engine builds on one box: every directory given holds a built copy of the repo ('.' = this one)
dirs="${@:-_ab .}"
for r in $(seq 1 ${AB_ROUNDS:-2}); do
for d in $dirs; do
  for p in ${AB_PROFILES:-xten}; do
    (cd $d && python3 bench.py --steps 20 --warmup 3 --strong-scale 0 --profile $p --no-cpu-baseline --no-host-pinned --no-md5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$d $p', round(d['value']/1e6,1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})")
  done
done
done
