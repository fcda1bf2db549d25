#!/bin/bash
# Disassembly of one kernel of lib/librt_hip.so: tools/disasm.sh '<mangled-name substring>' > out.s
so=${2:-$(dirname "$0")/../raytracer_amd/lib/librt_hip.so}
python3 - "$so" <<'PY'
import re,sys
d=open(sys.argv[1],'rb').read()
pos=[m.start() for m in re.finditer(b'\x7fELF',d)]
open('/tmp/rt_disasm.co','wb').write(d[pos[1]:])
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/rt_disasm.co 2>/dev/null | awk -v pat="$1" '/^[0-9a-f]+ </{on = index($0, pat) > 0} on{print}'
