#!/usr/bin/env python3
"""usage: isa_blocks.py <asm.s> <kernel-substring> [min_valu] -- VALU / LDS / SALU counts per basic block"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = re.search(r'^(_Z\w*' + re.escape(name) + r'\w*):(.*?)\.end_amdhsa_kernel', s, re.S | re.M)
print(m.group(1))
blocks = []; cur = ['entry', 0, 0, 0, 0]; blocks.append(cur)
for l in m.group(2).splitlines():
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        cur = [t, 0, 0, 0, 0]; blocks.append(cur); continue
    if t.startswith('v_'): cur[1] += 1
    elif t.startswith('ds_'): cur[2] += 1
    elif t.startswith('s_'): cur[3] += 1
    elif t.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur[4] += 1
print("block valu lds salu vmem")
for b in blocks:
    if b[1] >= thr: print(*b)
print("total valu", sum(b[1] for b in blocks))
for k in ("next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size", "group_segment_fixed_size"):
    mm = re.search(r'amdhsa_kernel ' + re.escape(m.group(1)) + r'.*?\.amdhsa_' + k + r' (\d+)', s, re.S)
    print(k, mm.group(1) if mm else None)
