#!/bin/bash
# usage: tools/e2e_all.sh <tag>   every end-to-end CLI measurement DESIGN.md quotes, logs under gpurun_out/<tag>/
# (C0, C1, C2 through the CLI; C3 and C4 at 10 %; C3 at full size device-resident, as BGZF and as text in pinned host
# memory; kernel trace of the scaled C3 run)
out=gpurun_out/$1; mkdir -p $out
bash tools/e2e_c0_c1.sh > $out/e2e_c0_c1.log 2>&1
bash tools/e2e_c2.sh > $out/e2e_c2.log 2>&1
MODES="--no-write;--no-write;--no-write --host-haplotypes;--no-write --fetch;--no-write --fetch --gzip;;--gzip" bash tools/e2e_c3_scaled.sh > $out/e2e_c3_scaled.log 2>&1
bash tools/e2e_c3_prof.sh && cp $(ls gpurun_out/prof_e2e_c3/*/*kernel_stats.csv | head -1) $out/e2e_c3_scaled_kernel_stats.csv && rm -rf gpurun_out/prof_e2e_c3
rm -rf /tmp/e2e_c3/out
bash tools/e2e_c4_scaled.sh > $out/e2e_c4_scaled.log 2>&1
SCALE=1.0 MODES="--no-write;--no-write;--no-write --fetch --gzip;--no-write --fetch" bash tools/e2e_c3_scaled.sh > $out/e2e_c3_full.log 2>&1
SIMU_HOST_PLAN=1 ./simuscop_amd/lib/simuReads /tmp/e2e_c3/config.txt --quiet --stats --no-write 2>&1 | tail -1 | sed 's/.*reads=/host planner: reads=/' >> $out/e2e_c3_full.log
ls -la $out
