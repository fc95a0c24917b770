#!/usr/bin/env python3
"""Static instruction mix of one kernel in hipcc's -S output:  tools/isa_stats.py file.s <kernel name substring>"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2]
    labels = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+):', s, re.M)]
    for pos, name in labels:
        if want in name:
            end = s.index('s_endpgm', pos)
            body = s[pos:end]
            break
    else:
        raise SystemExit('kernel not found')
    c = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith(('.', ';', '/')) or line.endswith(':'):
            continue
        c[line.split()[0]] += 1
    groups = collections.Counter()
    for k, v in c.items():
        g = ('valu' if k.startswith('v_') else 'lds' if k.startswith('ds_') else 'salu' if k.startswith('s_')
             else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_')) else 'other')
        groups[g] += v
    print(name)
    print('total', sum(c.values()), dict(groups))
    print(c.most_common(45))


main()
