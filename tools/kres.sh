#!/bin/bash
# compact per-kernel resource usage of one HIP source:  tools/kres.sh csrc/attention.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {fn=$0; sub(/.*Function Name: /,"",fn); sub(/ \[.*/,"",fn)}
       / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /Occupancy/ {o=$(NF-1)} /VGPRs Spill/ {s=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /LDS Size/ {printf "%-60s vgpr %4s agpr %4s occ %s spill %s scratch %s lds %s\n", fn, v, a, o, s, sc, $(NF-1)}'
