#!/bin/bash
# Development helper: k_bin time with its later stages switched off (the tile kernel skips the bins).
for v in 0x40000000 0x50000000 0x60000000; do
  echo "== debug $v"; timeout -k 10 100 python3 tools/frames.py 3 5 $v | tail -1 || exit 1
done
