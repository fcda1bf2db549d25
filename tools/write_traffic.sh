#!/bin/bash
# WRITE_SIZE of the cube frame under a few environment settings (one rocprofv3 --pmc pass each).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export RT_HIP_LIB=dev  # the RT_HIP_* switches exist in the development build only
for cfg in "default" "RT_HIP_COVER_TILES=0" "RT_HIP_TILE_MASKS=0" "RT_HIP_LIGHT_PER_BLOCK=4" "RT_HIP_TILE_PLAN=0"; do
  out=gpurun_out/wt_$(echo $cfg | tr '=' '_')
  rm -rf $out
  if [ "$cfg" = default ]; then
    timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_COUNT --output-format csv -d $out -- python3 tools/one_scene.py cube auto 20 > $out.log 2>&1 || exit 1
  else
    export $cfg
    timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_COUNT --output-format csv -d $out -- python3 tools/one_scene.py cube auto 20 > $out.log 2>&1 || exit 1
    unset ${cfg%%=*}
  fi
  python3 - "$out" "$cfg" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_trace_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            rows.append(float(r["Counter_Value"]))
print(sys.argv[2], "dispatches", len(rows), "WRITE_SIZE KiB: first", rows[:3], "last", rows[-3:])
PY
done
