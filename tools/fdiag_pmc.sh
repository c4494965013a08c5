#!/bin/bash
# usage: tools/fdiag_pmc.sh [profile]   VALU / SALU / LDS instruction counts of the straight-line emit kernel under the SG_FDIAG
# ablations (0 none, 1 item stores, 2 names + per-read pass, 8 Philox, 16 fix-up, 31 all of these; 128 no plain steps, 256 no general
# steps, 384 neither: phase 0, composition and the per-read pass alone): what each part costs in instructions.  FDIAGS="0 128" picks.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=${1:-xten}
for d in ${FDIAGS:-0 1 2 8 16 31 128 256 384}; do
  rm -rf gpurun_out/fdpmc
  SG_FDIAG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/fdpmc -- python3 bench.py --steps 2 --warmup 1 --strong-scale 0 --no-cpu-baseline --no-host-pinned --no-md5 --profile $P > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob("gpurun_out/fdpmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "emit_fast" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
print("fdiag", sys.argv[1], {k: round(v[1] / v[0] / 1e6, 2) for k, v in sorted(acc.items())})
PY
done
rm -rf gpurun_out/fdpmc
