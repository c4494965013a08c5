#!/bin/bash
# 24 contigs with GRCh38 proportions at 10 % scale (309 Mbp), XTen PE 30x, one GPU: device-only with both
# haplotype routes, then +fetch
set -e
W=/tmp/e2e_c3; rm -rf $W; mkdir -p $W
python - <<PY
import sys, time; sys.path.insert(0,'.')
from simuscop_amd import synth
t=time.time(); synth.write_fasta('$W/ref.fa', synth.grch38_contigs(${SCALE:-0.1}), seed=38); print('fasta %.1fs' % (time.time()-t))
open('$W/config.txt','w').write("ref = $W/ref.fa\nprofile = tests/golden/testData/Illumina_HiSeqXTen.profile\nname = sim\noutput = $W/out\nlayout = PE\nthreads = ${THREADS:-16}\nverbose = 0\ncoverage = ${COVERAGE:-30}\ninsertSize = 350\n")
PY
IFS=";" read -ra MODES_ARR <<< "${MODES:---no-write;--no-write;--no-write --host-haplotypes;--no-write --fetch}"
for mode in "${MODES_ARR[@]}"; do
  echo "== simuReads $mode"; ./simuscop_amd/lib/simuReads $W/config.txt --quiet --stats $mode 2>&1 | tail -1 | sed 's/.*reads=/reads=/'
done
