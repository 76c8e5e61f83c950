#!/bin/bash
# Development: sustained frame time of config 3 under two settings of an environment variable,
# interleaved.  usage (on the GPU box): bash tools/ab.sh VAR valueA valueB [rounds]
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for x in $A $B; do
    env $V=$x timeout -k 10 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V=$x', 'steps200 %.4f' % d['ms_per_step'], 'sustained %.4f' % d['sustained']['ms_per_step'])" || exit 1
  done
done
