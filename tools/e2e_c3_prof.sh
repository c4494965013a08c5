#!/bin/bash
# kernel trace of the whole CLI run on the scaled C3 genome (after tools/e2e_c3_scaled.sh has made /tmp/e2e_c3)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e2e_c3 -- ./simuscop_amd/lib/simuReads /tmp/e2e_c3/config.txt --no-write --quiet --stats 2> gpurun_out/prof_e2e_c3.err
