#!/bin/bash
# Development helper: LDS and issue counters of the two large kernels (one pass, eight SQ counters).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_lds -- python3 $R/tools/frames.py 3 3 > $R/gpurun_out/pmc_lds.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_lds k_setup
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_lds k_raster
