#!/bin/bash
# Vector-memory pipeline counters (address unit TA, data return TD, L1 TCP) of the frame kernels on
# config 3: rocprofv3 --pmc passes of at most 2 TA, 2 TD and 4 TCP counters each, summarised into
# profiles/pmc_mem.json.  Run on the GPU box from the repository root:  bash tools/pmc_mem.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_mem
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/tools/frames.py 3 6 > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }
}
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
pass tb TA_FLAT_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE
pass tc TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE
cd $R && python3 tools/pmc_mem.py $O profiles/pmc_mem.json && cp profiles/pmc_mem.json $O/pmc_mem.json
