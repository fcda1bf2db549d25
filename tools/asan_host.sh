#!/bin/bash
# The command-line host built by `make -C raytracer_amd asan` over a set of scenes (GPU box): host code and the host side of
# the library under the address sanitizer; leak detection off (the ROCm runtime's own allocations drown everything else).
set -o pipefail
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0
printf 'leak:libhsa-runtime64\nleak:libamdhip64\nleak:libamd_comgr\nleak:librocprofiler\nleak:libhiprtc\n' > /tmp/lsan.supp
export LSAN_OPTIONS=suppressions=/tmp/lsan.supp:print_suppressions=0
B=raytracer_amd/bin/raytracer_hip_asan
D=raytracer_amd/data/lighthouse_tris_f32.bin
cd gpurun_out
for args in "lighthouse --tris-f32 ../$D --size 1920 1080 --bvh --tri-reflective 0.3 --frames 3 --out b1.ppm" "lighthouse --tris-f32 ../$D --size 3840 2160 --frames 2 --out b2.ppm" "lighthouse --tris-f32 ../$D --size 1920 1080 --brute --frames 2 --out b3.ppm" "cube --size 1920 1080 --frames 2 --out b4.ppm" "spheres --size 1920 1080 --frames 2 --out b5.ppm" "world_with_plane --size 640 480 --out b6.ppm" "lighthouse --tris-f32 ../$D --out a1.ppm" "lighthouse --tris-f32 ../$D --host-ppm --out a2.ppm --size 640 360 --bvh --tri-reflective 0.3" "cube_spheres --size 333 217 --spp-grid 2 --sph-reflective 0.3 --out a3.ppm" "refraction --out a4.ppm" "lighthouse --tris-f32 ../$D --device-bvh --frames 3 --out a5.ppm"; do
  echo "== $args"
  timeout -k 10 90 ../$B $args 2>&1 | tail -12 || exit 1
done
sha256sum a1.ppm; rm -f b?.ppm a?.ppm
