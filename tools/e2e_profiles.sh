#!/bin/bash
# kernel timings of the C2-shaped workload with each shipped profile (device-resident, no fetch)
set -e
W=/tmp/e2e_prof; rm -rf $W; mkdir -p $W
python - <<PY
import sys; sys.path.insert(0,'.')
from simuscop_amd import synth
synth.write_fasta('$W/ref.fa', [('chr20', 64444167)], seed=20)
for name, ins in (('Illumina_HiSeqXTen',350), ('Illumina_HiSeq2500',300), ('Illumina_HiSeq2000',250), ('Illumina_GenomeAnalyzerIIx',250)):
    open('$W/%s.txt' % name,'w').write("ref = $W/ref.fa\nprofile = tests/golden/testData/%s.profile\nname = sim\noutput = $W/out\nlayout = PE\nthreads = 1\nverbose = 0\ncoverage = 30\ninsertSize = %d\n" % (name, ins))
PY
for p in Illumina_HiSeqXTen Illumina_HiSeq2500 Illumina_HiSeq2000 Illumina_GenomeAnalyzerIIx; do
  echo "== $p"; ./simuscop_amd/lib/simuReads $W/$p.txt --quiet --stats --no-write 2>&1 | tail -1 | sed 's/.*reads=/reads=/'
done
