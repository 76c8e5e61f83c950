#!/bin/bash
# Round artefacts on the GPU box: bench line, kernel-trace statistics of the same command, the two
# HBM-traffic counter passes and the VALU/LDS counter passes.  Everything lands under
# gpurun_out/round/; copy what is to be kept into profiles/ afterwards (tools/keep_round.sh <tag>).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/frames.py 3 6 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/frames.py 3 6 > $O/pmc_write.log 2>&1 || exit 1
cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > /dev/null || exit 1
bash tools/pmc_valu_lds.sh > $O/pmc_valu_lds.out 2>&1 || exit 1
cp profiles/pmc_valu_lds.json $O/pmc_valu_lds.json
bash tools/pmc_mem.sh > $O/pmc_mem.out 2>&1 || exit 1
cp profiles/pmc_mem.json $O/pmc_mem.json
# (the counter summaries first: the bench line quotes them)
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cd $R && timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-depth-variant > $O/trace.json 2> $O/trace.err || exit 1
cd $R
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cat $O/bench.json
