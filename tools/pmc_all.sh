#!/bin/bash
# Every PMC summary the design notes quote, at the current build: tools/pmc_all.sh <tag>  ->  gpurun_out/pmc_<tag>_<scene>/summary.json
cd "$(dirname "$0")/.."
tag=${1:-r}
run() { name=$1; shift; echo "== $name"; bash tools/pmc.sh pmc_${tag}_$name "$@" > gpurun_out/pmc_${tag}_$name.log 2>&1 || { echo "$name failed"; tail -5 gpurun_out/pmc_${tag}_$name.log; return 1; }; grep -E "derived_valu_issue_busy|derived_active_lanes|Scratch_Size|VGPR_Count" gpurun_out/pmc_${tag}_$name/summary.json; }
run cube_1080p cube auto 20 || exit 1
run spheres_1080p spheres auto 20 || exit 1
run lighthouse_bvh_1080p lighthouse bvh 20 || exit 1
run lighthouse_refl_bvh_1080p lighthouse_refl bvh 20 || exit 1
run lighthouse_exhaustive_1080p lighthouse brute 10 || exit 1
run c4_lighthouse_refl_bvh_4k lighthouse_refl bvh 10 3840 2160 || exit 1
run c5_cube_spheres_refl_8k_16spp cube_spheres_refl auto 3 7680 4320 4 || exit 1
