#!/bin/bash
# kernel trace of a --gzip run of the C2 workload (after tools/e2e_c2.sh has made /tmp/e2e_c2)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gzip -- ./simuscop_amd/lib/simuReads /tmp/e2e_c2/config.txt --no-write --fetch --gzip --quiet --stats 2> gpurun_out/prof_gzip.err
