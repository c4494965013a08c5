#!/bin/bash
# usage: pmc_write.sh <tag> -- WRITE_SIZE pass over bench.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_$1_WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --strong-scale 0 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_$1.err || echo "pmc failed"
