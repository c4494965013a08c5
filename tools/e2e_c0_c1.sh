#!/bin/bash
# SURVEY 8d C0 (config_test_wgs.txt shape: GAIIx 74 bp PE, coverage 10, insertSize 250, variations + SNPs,
# one 63,025,520 bp contig) and C1 (config_test_wes.txt shape: HiSeq2500 125 bp PE, coverage 100, insertSize
# 200, 4,677 BED targets, variations + SNPs) at full size on one GPU, with files
set -e
W=/tmp/e2e_c01; rm -rf $W; mkdir -p $W
python - <<PY
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from simuscop_amd import synth
import cases
L = 63025520
synth.write_fasta('$W/ref.fa', [('chr20', L)], seed=20)
cases._write('$W/variations.txt', cases._variations('test', '20', 1.0))
cases._write('$W/snp.txt', cases._snps('20', L, 1500, 3))
cases._write('$W/targets.bed', cases._bed('20', L, 7, 4677))
T = 'tests/golden/testData/'
cases._config('$W/c0.txt', ref='$W/ref.fa', profile=T + cases.PROFILES['gaiix'], variation='$W/variations.txt', snp='$W/snp.txt',
              name='test', output='$W/out0', layout='PE', threads=16, verbose=0, coverage=10, insertSize=250)
cases._config('$W/c1.txt', ref='$W/ref.fa', profile=T + cases.PROFILES['hs2500'], variation='$W/variations.txt', snp='$W/snp.txt',
              target='$W/targets.bed', name='test', output='$W/out1', layout='PE', threads=16, verbose=0, coverage=100, insertSize=200)
PY
for c in c0 c1; do
  for mode in "--no-write" ""; do
    echo "== $c $mode"; ./simuscop_amd/lib/simuReads $W/$c.txt --quiet --stats $mode 2>&1 | tail -1 | sed 's/.*reads=/reads=/'
  done
done
