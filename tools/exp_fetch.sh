#!/bin/bash
# Development: FETCH_SIZE (raw KiB x 1024, uncorrected) and L1->L2 read requests of the tile kernel with
# each experimental build exp/libsxe_<k>.so.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp_fetch; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in $R/exp/libsxe_*.so; do
  b=$(basename $f .so)
  export SXE_LIB=$f
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$b.f -- python3 $R/tools/frames.py 3 6 > $O/$b.f.log 2>&1 || { tail -5 $O/$b.f.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/$b.r -- python3 $R/tools/frames.py 3 6 > $O/$b.r.log 2>&1 || { tail -5 $O/$b.r.log; exit 1; }
done
cd $R && python3 - <<'PY'
import os, sys
sys.path.insert(0, "tools")
from pmc_summary import summarise
root = "gpurun_out/exp_fetch"
for name in sorted(os.listdir(root)):
    d = os.path.join(root, name)
    if os.path.isdir(d):
        for k, cs in summarise(d).items():
            if "raster_shade" in k or "k_setup" in k:
                print(name, k[10:24], {c: round(v * (1024 if c == "FETCH_SIZE" else 1) / 1e6, 2) for c, v in cs.items() if isinstance(v, float)})
PY
