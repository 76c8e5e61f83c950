#!/bin/bash
# Development helper: L2 request, hit and miss counts of the frame's kernels (one pass, four TCC counters).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum --output-format csv -d $R/gpurun_out/pmc_l2 -- python3 $R/tools/frames.py 3 3 > $R/gpurun_out/pmc_l2.log 2>&1 || { tail -5 $R/gpurun_out/pmc_l2.log; exit 1; }
for k in k_setup k_bin k_scan k_raster; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_l2 $k; done
