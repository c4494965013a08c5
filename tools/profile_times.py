"""Steady-state kernel times of the C2-shaped pass for each shipped profile (run after tools/e2e_profiles.sh
has created /tmp/e2e_prof)."""
import sys
sys.path.insert(0, ".")
import simuscop_amd

for prof in ("Illumina_HiSeqXTen", "Illumina_HiSeq2500", "Illumina_HiSeq2000", "Illumina_GenomeAnalyzerIIx"):
    s = simuscop_amd.Session("/tmp/e2e_prof/%s.txt" % prof, device=0, write_files=0, quiet=1)
    s.weighted_length()
    s.set_reads(s.planned_reads)
    s.prepare_batch(0)
    out = []
    for i in range(4):
        s.sample()
        b1, b2, nf = s.result()
        out.append(s.kernel_times())
    print(prof, "pairs", nf, " first emit %.2f ms" % out[0]["emit"], " steady:", {k: round(v, 2) for k, v in out[-1].items()})
    s.close()
