#!/bin/bash
# Register, LDS and spill figures of every kernel, as the compiler reports them.
# usage: tools/resources.sh [source.hip ...]   (default: all of renderer_amd/csrc/*.hip)
cd "$(dirname "$0")/.." || exit 1
srcs=("$@"); [ ${#srcs[@]} -eq 0 ] && srcs=(renderer_amd/csrc/*.hip)
for f in "${srcs[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -c "$f" -o /dev/null \
      -Rpass-analysis=kernel-resource-usage 2>&1 | sed 's/ \[-Rpass-analysis.*//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /TotalSGPRs:/ {s=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF} /SGPRs Spill/ {ss=$NF} /LDS Size/ {printf "%-58s vgpr %3s sgpr %3s (spilled %s) scratch %3s waves/simd %s lds %s\n", name, v, s, ss, sc, o, $NF}'
done
