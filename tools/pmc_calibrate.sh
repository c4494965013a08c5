#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_calib_$c -- ./tools/pmc_calibrate > gpurun_out/pmc_calib_$c.out 2> gpurun_out/pmc_calib_$c.err || echo "pmc $c failed"
done
cat gpurun_out/pmc_calib_WRITE_SIZE.out
