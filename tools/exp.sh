#!/bin/bash
# Development: time the frame kernels with each experimental build exp/libsxe_<k>.so (tools/build_exp.sh),
# in turn, several rounds; the builds are loaded through SXE_LIB, the tree's own library stays as it is.
# usage (through gpurun): bash tools/exp.sh [rounds] [debug_flags]
O=gpurun_out/exp; mkdir -p $O
for round in $(seq ${1:-2}); do
for f in exp/libsxe_*.so; do
  SXE_LIB=$PWD/$f timeout -k 10 120 python tools/frames.py 3 60 ${2:-0} > $O/$(basename $f).log 2>&1 || { tail -5 $O/$(basename $f).log; exit 1; }
  echo "$f $(tail -1 $O/$(basename $f).log | cut -c40-)"
done
done
