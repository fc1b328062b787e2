#!/usr/bin/env python3
"""Instruction mix per kernel of a hipcc -save-temps assembly file (*.s): totals, vector instructions, LDS and
address-arithmetic opcodes.  Development tool: python tools/isa_mix.py file.s [substring ...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2:]
for m in re.finditer(r'^(_ZN4pdsp\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want and not all(w in name for w in want):
        continue
    ins = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    keep = {k: n for k, n in sorted(c.items()) if k.startswith(('ds_', 'scratch_', 'buffer_', 'global_')) or
            k in ('v_xad_u32', 'v_xor_b32', 'v_add_u32', 'v_lshl_add_u32', 's_barrier', 'v_or_b32', 'v_lshl_or_b32', 'v_add3_u32', 's_nop')}
    print(name, '\n   total', len(ins), 'vector', v, keep)
