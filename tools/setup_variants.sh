#!/bin/bash
# Development helper: k_setup time with parts of the kernel switched off (outputs are wrong).
for v in 0 0x40000000 0x80000000 0x08000000 0x88000000; do
  echo "== debug $v"; timeout -k 10 100 python3 tools/frames.py 3 4 $v | tail -1 || exit 1
done
