#!/bin/bash
# Development helper: SQ counters of k_setup (one pass; eight SQ counters fit).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_setup -- python3 $R/tools/frames.py 3 3 ${DEBUG_FLAGS:-0} > $R/gpurun_out/pmc_setup.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_setup k_setup
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_setup k_bin
