#!/bin/bash
# VALU / LDS / wait counters of the frame kernels on config 3: three rocprofv3 --pmc passes (at most
# eight SQ counters and two GRBM counters each), summarised into profiles/pmc_valu_lds.json.
# Run on the GPU box from the repository root:  bash tools/pmc_valu_lds.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_valu_lds
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/tools/frames.py 3 6 > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }
}
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
cd $R && python3 tools/pmc_valu_lds.py $O profiles/pmc_valu_lds.json
