"""Average the rocprofv3 --pmc CSV rows of rt_trace_kernel dispatches into one JSON summary."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "rt_trace_kernel" not in name and "rt_general_kernel" not in name:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
            if k in row:
                meta[k] = row[k]
summ = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
summ["_dispatches_per_counter"] = {k: len(v) for k, v in acc.items()}
summ["_kernel"] = meta
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
print(json.dumps(d, indent=1))
json.dump(d, open(os.path.join(out, "summary.json"), "w"), indent=1)
