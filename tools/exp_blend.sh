#!/bin/bash
# Development: the blending probe with each experimental build exp/libsxe_b*.so (loaded through SXE_LIB).
for f in exp/libsxe_b*.so; do echo $f; SXE_LIB=$PWD/$f timeout -k 10 200 python tools/blend_probe.py 2>&1 | grep translucent || exit 1; done
