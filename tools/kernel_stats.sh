#!/bin/bash
# Per-kernel register/occupancy/spill report for the HIP library (no GPU needed).
cd "$(dirname "$0")/../raytracer_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../include -Ihost -Icsrc "$@" \
  -c csrc/rt_api.hip -o /tmp/rt_stats.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name/ {name=$NF; sub(/.*rt_trace_kernel/,"",name); sub(/EEv8DevScene.*/,"",name)}
       /TotalSGPRs/ {sg=$NF} / VGPRs:/ {vg=$NF} /SGPRs Spill/ {ss=$NF} /VGPRs Spill/ {vs=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF}
       /LDS Size/ {print name, "sgpr="sg, "vgpr="vg, "sgpr_spill="ss, "vgpr_spill="vs, "scratch="sc, "occ="oc}'
