#!/bin/bash
# Development helper: vector instruction mix of the two large kernels (one pass, eight SQ counters).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d $R/gpurun_out/pmc_mix -- python3 $R/tools/frames.py 3 3 > $R/gpurun_out/pmc_mix.log 2>&1 || { tail -5 $R/gpurun_out/pmc_mix.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_mix k_setup
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_mix k_raster
