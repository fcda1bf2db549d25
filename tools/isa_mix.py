#!/usr/bin/env python3
"""Static instruction mix of the kernels in a device assembly file.

  hipcc <the library's flags> --cuda-device-only -S -o rt_api.s raytracer_amd/csrc/rt_api.hip
  python tools/isa_mix.py rt_api.s [substring of a mangled kernel name ...]

For every kernel whose name holds one of the substrings (all kernels without any): registers, scratch and LDS from
its descriptor, and the number of instructions per class.  Static counts say what the compiler emitted, not what
a frame executes; the executed mix is what the SQ_INSTS_VALU_* counters in profiles/*pmc*.json hold.
"""
import collections
import json
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(fma|fmac|mul|add|sub|min|max|fmamk|fmaak|mad)\w*_f64", op):
            return "valu_f64_arith"
        if re.match(r"v_(fma|fmac|mul|add|sub|subrev|min|max|min3|max3|fmamk|fmaak|mad|pk_\w+)\w*_f32", op):
            return "valu_f32_arith"
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)\w*", op):
            return "valu_transcendental"
        if re.match(r"v_(div_scale|div_fmas|div_fixup|ldexp|frexp|trig|fract|floor|ceil|rndne|trunc)\w*", op):
            return "valu_f64_f32_special"
        if op.startswith("v_cvt"):
            return "valu_convert"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "valu_compare"
        if op.startswith("v_cndmask"):
            return "valu_select"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "valu_move"
        if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
            return "valu_lane_spill_or_broadcast"
        return "valu_integer_bit_other"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "scalar_memory"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait_nop_barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch_memory"
    if re.match(r"(global|flat|buffer)_", op):
        return "vector_memory"
    return "other"


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    kernels, cur = {}, None
    meta = {}
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            cur = m.group(1)
            kernels[cur] = collections.Counter()
            continue
        m = re.match(r"^\s+\.amdhsa_kernel\s+(\S+)", line)
        if m:
            cur = None
            meta_name = m.group(1)
            meta[meta_name] = {}
            continue
        m = re.match(r"^\s+\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\S+)", line)
        if m and meta:
            meta[list(meta)[-1]][m.group(1)] = m.group(2)
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^\s+([a-z]\w+)", line)
        if m and not line.lstrip().startswith("."):
            kernels[cur][classify(m.group(1))] += 1
    out = {}
    for name, c in kernels.items():
        if wanted and not any(w in name for w in wanted):
            continue
        if not c:
            continue
        total = sum(c.values())
        out[name[:60]] = {"descriptor": meta.get(name, {}), "instructions": total,
                          "classes": dict(sorted(c.items(), key=lambda kv: -kv[1]))}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
