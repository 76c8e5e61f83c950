#!/bin/bash
# Development: instruction and wait counters of the ordered (blending) tile kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_blend; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 $R/tools/blend_frames.py 6 > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -- python3 $R/tools/blend_frames.py 6 > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/blend_frames.py 20 > $O/t.log 2>&1 || { tail -5 $O/t.log; exit 1; }
cd $R && python3 - <<'PY'
import sys, glob, csv
sys.path.insert(0, "tools")
from pmc_summary import summarise
for d in ("a", "b"):
    for k, cs in summarise("gpurun_out/pmc_blend/" + d).items():
        if "blend" in k:
            print(k[:40], {c: round(v / 1e6, 3) for c, v in cs.items() if isinstance(v, float)})
for f in glob.glob("gpurun_out/pmc_blend/t/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print(row["Name"][:50], row["Calls"], row["AverageNs"])
PY
