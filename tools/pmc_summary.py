"""Average the rocprofv3 --pmc CSV rows of the trace kernel's steady-state dispatches into one JSON summary.

    python tools/pmc_summary.py <dir> [n]

n: only the last n dispatches of every counter count (tools/one_scene.py renders n frames after the tile plan has
been tuned; the tuning's probe launches come before them and stay out).  Every counter is therefore averaged over the
same number of launches.
"""
import csv, glob, json, os, subprocess, sys
from collections import defaultdict
out = sys.argv[1]
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    rows = defaultdict(list)
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "rt_trace_kernel" not in name and "rt_general_kernel" not in name:
            continue
        rows[row["Counter_Name"]].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
            if k in row:
                meta[k] = row[k]
        meta["Kernel_Name"] = name.split("(")[0]
    for k, v in rows.items():
        v.sort()
        acc[k] += [x for _, x in (v[-last_n:] if last_n else v)]
summ = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
summ["_dispatches_per_counter"] = {k: len(v) for k, v in acc.items()}
summ["_kernel"] = meta
summ["_kernel"]["note"] = ("VGPR_Count as rocprofv3 reports it: the architected vector registers of a wave in units of "
                           "two (gfx950 allocates from a unified file of 512 per SIMD lane in blocks of 8); "
                           "tools/kernel_info.py reads the kernel descriptor itself")
d = summ
if "SQ_WAVE_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
    d["derived_valu_active_share_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_INSTS_VALU"):
    # thread-cycles per issued VALU instruction: 64 = every lane active in every instruction
    d["derived_active_lanes_per_valu_inst_of_64"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_INSTS_VALU"]
if "FETCH_SIZE" in d:
    d["derived_hbm_read_bytes_corrected_x2"] = d["FETCH_SIZE"] * 1024 * 2
if "WRITE_SIZE" in d:
    d["derived_hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
if "GRBM_GUI_ACTIVE" in d and d.get("SQ_INSTS_VALU"):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs.  A SIMD of this part executes 32 lanes per cycle: a wave-instruction
    # holds its vector issue port for 2 cycles, a double-precision add/mul/fma for 4 (16 lanes per cycle: the FP64
    # vector peak is half the FP32 one), a single-precision transcendental for 8 (a double-precision one is priced
    # at 16).  One wave alone issues a vector instruction every 4 cycles at best, so it takes two or more waves
    # per SIMD to reach a share of 1.
    cycles = d["GRBM_GUI_ACTIVE"] / 8.0
    d["derived_kernel_cycles_per_xcd"] = cycles
    f64 = sum(d.get("SQ_INSTS_VALU_" + k, 0.0) for k in ("ADD_F64", "MUL_F64", "FMA_F64"))
    t32, t64 = d.get("SQ_INSTS_VALU_TRANS_F32", 0.0), d.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
    rest = max(0.0, d["SQ_INSTS_VALU"] - f64 - t32 - t64)
    port_cycles = 4.0 * f64 + 8.0 * t32 + 16.0 * t64 + 2.0 * rest
    d["derived_valu_issue_busy"] = port_cycles / (1024.0 * cycles)
    d["derived_valu_issue_port_cycles_per_instruction"] = port_cycles / d["SQ_INSTS_VALU"]
    d["derived_valu_instructions_x4_over_simd_cycles"] = d["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles)  # the figure quoted before
classes = {k[len("SQ_INSTS_VALU_"):]: d[k] for k in d if k.startswith("SQ_INSTS_VALU_")}
if classes and d.get("SQ_INSTS_VALU"):
    total = d["SQ_INSTS_VALU"]
    named = sum(classes.values())
    mix = {k.lower(): round(v / total, 4) for k, v in sorted(classes.items())}
    mix["other (moves, selects, compares, lane and bit operations)"] = round(max(0.0, 1.0 - named / total), 4)
    d["derived_valu_mix_share_of_wave_instructions"] = mix
try:
    d["_commit"] = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_amd", "lib", "BUILD_COMMIT")).read().strip()
except OSError:
    pass
print(json.dumps(d, indent=1))
json.dump(d, open(os.path.join(out, "summary.json"), "w"), indent=1)
