#!/bin/bash
# usage: tools/r04_extras.sh <tag>   (GPU box) -- round-4 evidence that is not the emit kernel's: the VALU microbench with the
# packed-16 / DPP / SDWA rows, counters of the BGZF kernels, the training path's throughput and kernel trace, and reference
# ingest by several ranks at once (six of the eight ranks of a --world 8 --shard-contigs run sharing this device: the box
# admits six processes on the card).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_microbench tools/valu_microbench.hip 2> $out/valu_build.err && /tmp/valu_microbench > $out/valu_microbench.txt 2>&1
echo "microbench done"
bash tools/gz_pmc.sh $tag > $out/gz_pmc.txt 2>&1; cp gpurun_out/gz_pmc_$tag.json $out/gz_pmc.json
bash tools/gzip_prof.sh > /dev/null 2>&1 && cp $(ls gpurun_out/prof_gzip/*/*kernel_stats.csv | head -1) $out/gzip_kernel_stats.csv; rm -rf gpurun_out/prof_gzip
echo "gzip done"
python3 tools/train_bench.py 8 /tmp/trainbench > $out/train_bench.json 2> $out/train_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_trace -- ./simuscop_amd/lib/seqToProfile --sam /tmp/trainbench/reads.sam -v /tmp/trainbench/none.vcf -r /tmp/trainbench/ref.fa -o /tmp/trainbench/p.profile --quiet --stats > /dev/null 2> $out/train_trace.err
cp $(ls $out/train_trace/*/*kernel_stats.csv | head -1) $out/train_kernel_stats.csv; rm -rf $out/train_trace
echo "train done"
# ---- ingest rehearsal: the GRCh38-sized genome, six ranks that own whole chromosomes each, all on this one device ----
python3 -c "
import sys; sys.path.insert(0, '.')
import bench_c3
print(bench_c3.ensure_genome(1.0, 0)[0])" > $out/ingest_genome.txt 2>&1
G=$(tail -1 $out/ingest_genome.txt)
printf "ref = $G\nprofile = tests/golden/testData/Illumina_HiSeqXTen.profile\nname = sim\noutput = /tmp/ingest_out\nlayout = PE\nthreads = 16\nverbose = 0\ncoverage = 30\ninsertSize = 350\n" > /tmp/ingest_cfg.txt
for n in 1 6; do
  for rep in 1 2; do
    echo "== --gpus $n (same device) run $rep" >> $out/ingest_rehearsal.txt
    SIMUSCOP_SAME_DEVICE=1 SIMU_TRACE_LOAD=1 timeout -k 10 300 ./simuscop_amd/lib/simuReads /tmp/ingest_cfg.txt --gpus $n --shard-contigs --no-write --quiet --stats >> $out/ingest_rehearsal.txt 2>&1
  done
done
echo "ingest done"
ls -la $out
