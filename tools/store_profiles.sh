#!/bin/bash
# After tools/pmc_all.sh <tag> and tools/gpu_round.sh have run on the GPU box: copy what they left under gpurun_out/
# into profiles/ (usage: tools/store_profiles.sh <tag> <round prefix, e.g. r03>).
cd "$(dirname "$0")/.."
tag=${1:-r03}; pre=${2:-r03}
for d in gpurun_out/pmc_${tag}_*/; do n=$(basename $d); cp $d/summary.json profiles/${pre}_pmc_${n#pmc_${tag}_}.json; done
f=$(ls -t gpurun_out/prof/vm/*_kernel_stats.csv | head -1); t=${f/_kernel_stats.csv/_kernel_trace.csv}
grep '^{' gpurun_out/bench.log | tail -1 > profiles/${pre}_bench_c3.json
cp $f profiles/${pre}_bench_c3_kernel_stats.csv
grep '^{' gpurun_out/prof.log | tail -1 > profiles/${pre}_bench_c3_under_rocprof.json
T=$t PRE=$pre python3 - <<'P'
import csv, statistics as s, json, os
pre = os.environ['PRE']
rows = list(csv.DictReader(open(os.environ['T'])))
r = [x for x in rows if 'rt_trace_kernel<false' in x['Kernel_Name']]
d = [int(x['End_Timestamp']) - int(x['Start_Timestamp']) for x in r]
u = json.loads(open(f'profiles/{pre}_bench_c3_under_rocprof.json').read())
j = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-side",
     "commit": open('raytracer_amd/lib/BUILD_COMMIT').read().strip(), "kernel": r[100]['Kernel_Name'].split('(')[0], "launches": len(d),
     "vgpr_count_column": int(r[100]['VGPR_Count']), "lds_block_size": int(r[100]['LDS_Block_Size']), "scratch_size": int(r[100]['Scratch_Size']),
     "windows_ns": {k: {"mean": round(s.mean(d[a:b]), 1), "min": min(d[a:b])} for k, (a, b) in
                    {"setup_1500": (0, 1500), "warmup_300": (1500, 1800), "timed_2000_value": (1800, 3800),
                     "warmup_300_cold_leg": (3800, 4100), "timed_2000_cold_leg": (4100, 6100)}.items()},
     "note": "with the profiler attached the launches are spaced %.1f us apart (ms_per_step of the same run, %s_bench_c3_under_rocprof.json), "
             "so every kernel starts on an idle chip; back to back without the profiler (%s_bench_c3.json) the HIP events give the headline's "
             "per-launch time, the tail of one launch overlapping the head of the next" % (u['ms_per_step'] * 1e3, pre, pre)}
json.dump(j, open(f'profiles/{pre}_bench_c3_trace_windows.json', 'w'), indent=1)
b = json.loads(open(f'profiles/{pre}_bench_c3.json').read())
print("trace timed window", j['windows_ns']['timed_2000_value'], "| bench value", b['value'], "cold", b['value_cold'], "us/step", b['ms_per_step'] * 1e3, "frac", b['roofline']['frac'], "commit", b['config']['commit'])
for k, v in list(b['other_scenes_1080p'].items()) + list(b['full_size_configs_one_gpu'].items()) + list(b['general_kernel_scenes'].items()):
    print(" ", k, v['kernel_ms'])
P
