#!/bin/bash
# usage: tools/final_round.sh <tag>   the evidence a round commits under profiles/<tag>/: kernel trace + counters + bench line
# (profile_round.sh), bench lines of the other shipped profiles, kernel trace of a --gzip run, BGZF bit budget, C3 / C4 bench lines
tag=$1; out=gpurun_out/$tag
bash tools/profile_round.sh $tag > $out.round.log 2>&1
for p in hs2500 hs2000 gaiix; do
  python3 bench.py --steps 10 --warmup 2 --strong-scale 0 --no-cpu-baseline --no-md5 --profile $p > $out/bench_$p.json 2>/dev/null
done
bash tools/e2e_c2.sh > $out/e2e_c2.log 2>&1
bash tools/gzip_prof.sh > /dev/null 2>&1 && cp $(ls gpurun_out/prof_gzip/*/*kernel_stats.csv | head -1) $out/gzip_kernel_stats.csv; rm -rf gpurun_out/prof_gzip
SG_GZ_TRACE=1 python3 bench.py --steps 1 --warmup 0 --strong-scale 0 --no-cpu-baseline --no-md5 2> $out/gz_trace.err > /dev/null; grep "^\[gz\]" $out/gz_trace.err | tail -4 > $out/gzip_bit_budget.txt; rm -f $out/gz_trace.err
python3 bench.py --workload c3 --scale 1.0 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_full.json 2> /dev/null
python3 bench.py --workload c4 --scale 1.0 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c4_full.json 2> /dev/null
ls -la $out
