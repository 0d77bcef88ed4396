"""Instruction mix of the kernels in a gfx950 .s file (hipcc -save-temps): python tools/isa_mix.py file.s [name-substring]"""
import collections, re, sys
s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    if want not in m.group(1):
        continue
    ops = collections.Counter()
    for line in m.group(2).split('\n'):
        line = line.strip()
        if not line or line[0] in ';.' or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    groups = collections.Counter()
    for k, v in ops.items():
        key = ('v_pk' if k.startswith('v_pk') else 'mfma' if 'mfma' in k else 'valu' if k.startswith('v_') else 'lds' if k.startswith('ds_') else
               'vmem' if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'salu' if k.startswith('s_') else k)
        groups[key] += v
    print(m.group(1)[:90], "total", sum(ops.values()), dict(groups))
    print("   ", ops.most_common(30))
