#!/usr/bin/env python3
"""Instruction timeline of one kernel in a hipcc -S dump: runs of MFMA / spill / LDS / global ops in program order.
usage: asm_timeline.py file.s mangled_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end]
cls = [('v_mfma', 'M'), ('scratch_store', 'S'), ('scratch_load', 'L'), ('ds_read_b64_tr', 't'), ('ds_read', 'r'), ('ds_write', 'w'),
       ('global_load', 'g'), ('global_store', 's'), ('global_atomic', 'a'), ('s_cbranch', '|'), ('s_waitcnt', '.'), ('v_', 'v')]
s = []
for l in body:
    t = l.strip().split(' ')[0]
    for pre, c in cls:
        if t.startswith(pre):
            s.append(c)
            break
out, prev, cnt = [], None, 0
for c in s:
    if c == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{prev}{cnt if cnt > 1 else ''}")
        prev, cnt = c, 1
out.append(f"{prev}{cnt}")
print(len(body), 'lines;', {c: s.count(c) for _, c in cls})
print(' '.join(out))
