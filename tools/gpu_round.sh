#!/bin/bash
# One GPU visit: smoke, C host CLI, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 120 python -u __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log || exit 1
( cd gpurun_out && timeout -k 10 60 ../raytracer_amd/bin/raytracer_hip lighthouse --tris-f32 ../raytracer_amd/data/lighthouse_tris_f32.bin --out canvas.ppm && sha256sum canvas.ppm ) 2>&1 | tee gpurun_out/cli.log || exit 1
timeout -k 10 300 python -u bench.py 2>&1 | tail -2 | tee gpurun_out/bench.log || exit 1
rm -rf gpurun_out/prof && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --no-cpu-baseline --no-side > gpurun_out/prof.log 2>&1 || { tail -20 gpurun_out/prof.log; exit 1; }
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f" | head -20
