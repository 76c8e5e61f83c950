#!/bin/bash
# Calibrates FETCH_SIZE for gathered reads (run on the GPU box from the repository root, after
# building tools/calib/fetch_calib here with hipcc):  bash tools/calib/fetch_calib.sh
# Leaves profiles/fetch_calib.json.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fetch_calib
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/calib/fetch_calib > $O/plain.json 2> $O/plain.err || { cat $O/plain.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $R/tools/calib/fetch_calib > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/rdreq -- $R/tools/calib/fetch_calib > $O/rdreq.log 2>&1 || { tail -5 $O/rdreq.log; echo "(request counters not collected)"; }
cd $R && python3 tools/calib/fetch_calib.py $O profiles/fetch_calib.json
