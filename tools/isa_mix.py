#!/usr/bin/env python3
"""Static instruction mix per kernel of a gfx950 assembly file (hipcc --save-temps=obj): VALU / SALU / memory / LDS instructions and dot products.
usage: tools/isa_mix.py file.s [kernel-name-substring]"""
import re, sys
s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\w+):\s*; @', s, re.M)]
for k, (pos, name) in enumerate(starts):
    if want not in name: continue
    end = s.find('s_endpgm', pos)
    nxt = starts[k + 1][0] if k + 1 < len(starts) else len(s)
    body = s[pos:nxt]
    ins = []
    for l in body.split('\n'):
        t = l.strip()
        if not t or t[0] in ';._' or t.endswith(':') or t.startswith('BB'): continue
        ins.append(t.split()[0])
    c = lambda *p: sum(1 for i in ins if i.startswith(p))
    print(f"{name[:48]:48s} total {len(ins):5d} valu {c('v_'):5d} salu {c('s_'):5d} vmem {c('global_', 'flat_', 'buffer_', 'scratch_'):4d} lds {c('ds_'):4d} dot {sum(1 for i in ins if 'dot' in i):4d} waitcnt {c('s_waitcnt'):4d} branch {c('s_cbranch', 's_branch'):4d}")
