"""Registers, scratch and LDS of every kernel in lib/librt_hip.so (read from the code object's metadata note)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "raytracer_amd", "lib", "librt_hip.so")
d = open(so, "rb").read()
pos = [m.start() for m in re.finditer(b"\x7fELF", d)]
co = "/tmp/rt_dev_%d.co" % os.getpid()
open(co, "wb").write(d[pos[1]:])
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows, cur = [], {}
keys = ("name", "vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size")
for line in out.splitlines():
    m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k in keys:
        cur[k] = v
    if k == "wavefront_size":  # last key of a kernel's entry
        rows.append(cur)
        cur = {}
for r in rows:
    n = r.get("name", "?")
    if "rocprim" in n:
        continue
    short = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.split("(")[0].replace("void ", "")
    print(f"{short:55s} vgpr {r.get('vgpr_count','?'):>4s} agpr {r.get('agpr_count','0'):>3s} sgpr {r.get('sgpr_count','?'):>4s} "
          f"sgpr_spill {r.get('sgpr_spill_count','0'):>4s} vgpr_spill {r.get('vgpr_spill_count','0'):>3s} "
          f"scratch {r.get('private_segment_fixed_size','?'):>5s} lds {r.get('group_segment_fixed_size','?')}")
