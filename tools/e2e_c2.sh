#!/bin/bash
# end-to-end timings of the C2 workload through the simuReads CLI (device-only / +PCIe fetch / +files)
set -e
W=/tmp/e2e_c2; rm -rf $W; mkdir -p $W
python - <<PY
import sys; sys.path.insert(0,'.')
from simuscop_amd import synth
synth.write_fasta('$W/ref.fa', [('chr20', 64444167)], seed=20)
open('$W/config.txt','w').write("ref = $W/ref.fa\nprofile = tests/golden/testData/Illumina_HiSeqXTen.profile\nname = sim\noutput = $W/out\nlayout = PE\nthreads = 1\nverbose = 0\ncoverage = 30\ninsertSize = 350\n")
PY
for mode in "--no-write" "--no-write --fetch" "--no-write --fetch --gzip" "" "--gzip"; do
  echo "== simuReads $mode"
  ./simuscop_amd/lib/simuReads $W/config.txt --quiet --stats $mode 2>&1 | tail -2
done
ls -la $W/out | head -8
