#!/bin/bash
# Development: L1-miss and timing counters of the frame kernels with each experimental build exp/libsxe_<k>.so.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in $R/exp/libsxe_*.so; do
  b=$(basename $f .so)
  export SXE_LIB=$f
  timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d $O/$b -- python3 $R/tools/frames.py 3 6 > $O/$b.log 2>&1 || { tail -5 $O/$b.log; exit 1; }
done
cd $R && python3 - <<'PY'
import os, sys
sys.path.insert(0, "tools")
from pmc_summary import summarise
root = "gpurun_out/exp_pmc"
for name in sorted(os.listdir(root)):
    d = os.path.join(root, name)
    if os.path.isdir(d):
        for k, cs in summarise(d).items():
            if "raster_shade" in k or "k_setup" in k:
                print(name, k[:30], {c: round(v / 1e6, 2) for c, v in cs.items() if isinstance(v, float)})
PY
