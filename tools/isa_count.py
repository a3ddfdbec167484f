"""Static instruction counts per kernel from a hipcc -save-temps assembly file (the .s of the device side):
python tools/isa_count.py file.s [name filter ...]   — VALU / v_mad_u64_u32 / v_mov / LDS / global / s_nop / waits per kernel."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
filters = sys.argv[2:]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if filters and not any(f in name for f in filters):
        continue
    c = Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith((';', '.')):
            continue
        op = l.split()[0]
        for pre, key in (('v_', 'valu'), ('v_mad_u64', 'mad64'), ('v_mov', 'vmov'), ('v_cndmask', 'cnd'), ('ds_', 'ds'), ('global_load', 'gld'),
                         ('global_store', 'gst'), ('s_waitcnt', 'wait'), ('s_nop', 'nop'), ('s_barrier', 'bar'), ('s_', 'salu'), ('scratch', 'scratch'),
                         ('v_add_u32', 'add32'), ('v_sub_u32', 'sub32'), ('v_and', 'and'), ('v_lshl', 'lshl'), ('v_ashr', 'ashr'), ('v_add_co', 'addco'),
                         ('v_addc', 'addc'), ('v_sub_co', 'subco'), ('v_subb', 'subb'), ('v_lshl_add_u64', 'lshladd64'), ('v_alignbit', 'align'),
                         ('v_bfe', 'bfe'), ('v_perm', 'perm'), ('v_or', 'or'), ('v_mul', 'mul'), ('v_cmp', 'cmp')):
            if op.startswith(pre):
                c[key] += 1
    print(name[:90])
    print('   ', dict(sorted(c.items(), key=lambda kv: -kv[1])))
