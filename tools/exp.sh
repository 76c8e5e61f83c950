#!/bin/bash
# Development: time the frame kernels with each experimental build exp/libsxe_<k>.so, twice in turn.
# usage (through gpurun): bash tools/exp.sh [debug_flags]
O=gpurun_out/exp; mkdir -p $O
for round in 1 2; do
for f in exp/libsxe_*.so; do
  cp $f renderer_amd/libsxe_hip.so
  timeout -k 10 120 python tools/frames.py 3 60 ${1:-0} > $O/$(basename $f).log 2>&1 || { tail -5 $O/$(basename $f).log; exit 1; }
  echo "$f $(tail -1 $O/$(basename $f).log | cut -c40-)"
done
done
