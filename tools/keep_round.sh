#!/bin/bash
# Copies the artefacts of the last tools/profile_round.sh run from gpurun_out/round into profiles/.
# usage: tools/keep_round.sh <tag>      e.g. r2_v1
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/round
cp $O/bench.json profiles/$1_bench.json && cp $O/kernel_stats.csv profiles/$1_kernel_stats.csv &&
cp $O/pmc_traffic.json profiles/pmc_traffic.json && cp $O/pmc_valu_lds.json profiles/pmc_valu_lds.json &&
cp $O/pmc_mem.json profiles/pmc_mem.json && cp $O/pmc_mem.json profiles/$1_pmc_mem.json &&
cp $O/pmc_traffic.json profiles/$1_pmc_traffic.json && cp $O/pmc_valu_lds.json profiles/$1_pmc_valu_lds.json && ls -la profiles | tail -8
