#!/bin/bash
# Development helper: SQ counters of k_raster_shade with parts of the kernel switched off.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 0x80000000 0x40000000 0x10000000 0x20000000}; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmcv_$v -- python3 $R/tools/frames.py 3 3 $v > $R/gpurun_out/pmcv_$v.log 2>&1
  echo "== debug $v"; tail -1 $R/gpurun_out/pmcv_$v.log | head -c 400; echo
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcv_$v k_raster
done
