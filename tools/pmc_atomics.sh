R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_atomics; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_ATOMIC_WITH_RET_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum TCP_TOTAL_WRITE_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 $R/tools/frames.py 3 6 > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
cd $R && python3 - <<'PY'
import sys
sys.path.insert(0, "tools")
from pmc_summary import summarise
for k, cs in summarise("gpurun_out/pmc_atomics/a").items():
    print(k[:40], {c: round(v / 1e6, 3) for c, v in cs.items() if isinstance(v, float)})
PY
