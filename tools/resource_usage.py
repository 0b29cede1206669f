#!/usr/bin/env python3
"""Summarises `hipcc -Rpass-analysis=kernel-resource-usage` listings: one line per kernel with VGPRs, spills,
occupancy and LDS.  usage: resource_usage.py listing.res [filter-substring ...]"""
import re
import subprocess
import sys


def parse(path):
    txt = open(path, errors='replace').read()
    out = []
    for blk in txt.split('Function Name: ')[1:]:
        name = blk.split()[0]
        get = lambda key: (re.search(key + r': (\d+)', blk) or [None, '?'])[1]   # noqa: E731
        out.append((name, get('VGPRs'), get('AGPRs'), get('ScratchSize \[bytes/lane\]'), get('Occupancy \[waves/SIMD\]'),
                    get('SGPRs'), get('LDS Size \[bytes/block\]')))
    names = subprocess.run(['c++filt'], input='\n'.join(n for n, *_ in out), stdout=subprocess.PIPE,
                           universal_newlines=True).stdout.split('\n')
    return [(dn,) + o[1:] for dn, o in zip(names, out)]


if __name__ == '__main__':
    rows = parse(sys.argv[1])
    for r in rows:
        name = re.sub(r'slf::|\(slf::[^)]*\)|void ', '', r[0])
        if all(f in name for f in sys.argv[2:]):
            print('%-100s vgpr %3s agpr %2s scratch %3s occ %s sgpr %3s lds %s' % ((name[:100],) + r[1:]))
