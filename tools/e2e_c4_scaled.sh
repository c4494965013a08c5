#!/bin/bash
# SURVEY 8d C4 shape, scaled: 24 contigs (GRCh38 proportions x SCALE), four populations clone1..3 + normal,
# the variations.txt pattern on every contig, SNPs every ~1.5 kb, abundance row 0.3/0.25/0.35/0.1, XTen PE 60x
set -e
W=/tmp/e2e_c4; rm -rf $W; mkdir -p $W
python - <<PY
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from simuscop_amd import synth
import cases
contigs = synth.grch38_contigs(${SCALE:-0.1})
t=time.time(); synth.write_fasta('$W/ref.fa', contigs, seed=44); print('fasta %.1fs' % (time.time()-t))
rows, snps = [], []
for i, (name, L) in enumerate(contigs):
    key = name[3:] if name.startswith('chr') else name
    for popu in ('clone1', 'clone2', 'clone3'):
        rows += cases._variations(popu, key, L / 63025520.0)
    snps += cases._snps(key, L, 1500, 100 + i)
cases._write('$W/variations.txt', rows); cases._write('$W/snp.txt', snps)
cases._write('$W/abundance.txt', ['0.3\t0.25\t0.35\t0.1'])
open('$W/config.txt','w').write("ref = $W/ref.fa\nprofile = tests/golden/testData/Illumina_HiSeqXTen.profile\nvariation = $W/variations.txt\nsnp = $W/snp.txt\nabundance = $W/abundance.txt\nname = clone1, clone2, clone3, normal\noutput = $W/out\nlayout = PE\nthreads = 16\nverbose = 0\ncoverage = ${COVERAGE:-60}\ninsertSize = 350\n")
print('variant rows', len(rows), 'snps', len(snps))
PY
IFS=";" read -ra MODES_ARR <<< "${MODES:---no-write;--no-write;--no-write --host-haplotypes}"
for mode in "${MODES_ARR[@]}"; do
  echo "== simuReads $mode"; ./simuscop_amd/lib/simuReads $W/config.txt --quiet --stats $mode 2>&1 | tail -1 | sed 's/.*reads=/reads=/'
done
