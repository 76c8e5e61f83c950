#!/bin/bash
# Development: sustained frame time of config 3 (bench.py) and the kernel times of another
# configuration (frames.py) under each experimental build exp/libsxe_<k>.so, interleaved.
# usage (on the GPU box): bash tools/ab_lib.sh [rounds] [other config]
N=${1:-3}; C=${2:-4}
for i in $(seq $N); do
  for f in exp/libsxe_*.so; do
    SXE_LIB=$PWD/$f timeout -k 10 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', 'steps200 %.4f' % d['ms_per_step'], 'sustained %.4f' % d['sustained']['ms_per_step'], 'raster %.4f setup %.4f' % (d['kernels_ms']['raster_shade_ms'], d['kernels_ms']['setup_ms']))" || exit 1
    echo "$f cfg$C $(SXE_LIB=$PWD/$f timeout -k 10 120 python tools/frames.py $C 60 | tail -1)" || exit 1
  done
done
