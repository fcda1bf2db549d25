"""Static instruction mix of one kernel's disassembly (tools/disasm.sh output): python tools/isa_mix.py file.s"""
import collections, re, sys

c = collections.Counter()
for l in open(sys.argv[1]):
    m = re.match(r'\s+(\w+)', l)
    if m:
        c[m.group(1)] += 1
g = collections.Counter()
for k, v in c.items():
    if k.startswith(('v_readlane', 'v_writelane')): g['lane (sgpr spill)'] += v
    elif k.startswith('v_') and 'f64' in k: g['valu f64'] += v
    elif k.startswith('v_') and 'f32' in k: g['valu f32'] += v
    elif k.startswith('v_'): g['valu other'] += v
    elif k.startswith(('s_load', 's_buffer')): g['smem'] += v
    elif k.startswith('s_'): g['salu'] += v
    elif k.startswith('ds_'): g['lds'] += v
    elif k.startswith(('scratch', 'buffer', 'global', 'flat')): g['vmem ' + k.split('_')[0]] += v
    else: g['other'] += v
print(sum(c.values()), dict(g))
print(c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 30))
