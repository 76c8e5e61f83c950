#!/bin/bash
# Development loop on the GPU box: the full-size parity test, then mean kernel times of config 3.
# usage (through gpurun): bash tools/quick.sh <tag>
O=gpurun_out/quick_$1; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "full_size_config3 or soups or micro or shading or golden or clipping or 8192" > $O/parity.log 2>&1 || { tail -15 $O/parity.log; exit 1; }
tail -1 $O/parity.log
timeout -k 10 120 python tools/frames.py 3 60 > $O/frames.log 2>&1 || { tail -5 $O/frames.log; exit 1; }
tail -1 $O/frames.log
