"""Instruction histogram of the hot loop (the innermost backward-branch span containing MFMAs) of a kernel.
usage: python tools/isa_hist.py file.s kernel_substring [--dump]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [m.group(1) for m in re.finditer(r'^(_Z\w+):', s, re.M) if all(p in m.group(1) for p in pat.split(','))]
for name in names[:1]:
    i = s.index(name + ':')
    j = s.index('.Lfunc_end', i)
    lines = s[i:j].split('\n')
    labels = {}
    for n, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = n
    spans = []
    for n, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            a, b = labels[m.group(1)], n
            nm = sum('v_mfma' in x for x in lines[a:b])
            if nm:
                spans.append((b - a, a, b, nm))
    spans.sort()
    print(name)
    meta = re.search(r'\.vgpr_count:\s+(\d+)', s[s.index(name, j):]) if name in s[j:] else None
    for length, a, b, nm in spans[:1]:
        c = Counter()
        for l in lines[a:b + 1]:
            t = l.strip().split()
            if not t or t[0].startswith(('.', ';', '//')) or t[0].endswith(':'):
                continue
            c[t[0]] += 1
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        print(f'loop lines {a}-{b}: {tot} instrs, mfma {nm}, valu {valu}, salu {sum(v for k, v in c.items() if k.startswith("s_"))}, '
              f'ds {sum(v for k, v in c.items() if k.startswith("ds_"))}, vmem {sum(v for k, v in c.items() if k.startswith(("global_", "buffer_")))}')
        for k, v in c.most_common(70):
            print(f'{v:5d} {k}')
        if '--dump' in sys.argv:
            print('\n'.join(lines[a:b + 1]))
