#!/usr/bin/env python3
"""Instruction mix of the kernels in a `hipcc --cuda-device-only -S` listing: VALU / packed / SALU / memory / LDS counts
per kernel (static counts of the straight-line body; loops are not weighted).
usage: asm_mix.py listing.s [filter-substring ...]"""
import re
import subprocess
import sys
from collections import Counter


def kernels(path):
    txt = open(path, errors='replace').read()
    parts = re.split(r'\n(_Z\w+):[^\n]*\n', txt)
    names, bodies = parts[1::2], parts[2::2]
    dn = subprocess.run(['c++filt'], input='\n'.join(names), stdout=subprocess.PIPE,
                        universal_newlines=True).stdout.split('\n')
    for n, b in zip(dn, bodies):
        b = b.split('.Lfunc_end')[0]
        ins = [ln.split()[0] for ln in b.split('\n') if ln.startswith('\t') and not ln.strip().startswith(('.', ';'))]
        yield re.sub(r'slf::|\(slf::[^)]*\)|void ', '', n), Counter(ins)


if __name__ == '__main__':
    for name, c in kernels(sys.argv[1]):
        if not all(f in name for f in sys.argv[2:]):
            continue
        grp = lambda *p: sum(v for k, v in c.items() if k.startswith(p))   # noqa: E731
        print('%-84s total %5d valu %5d (pk %4d, fp %4d) salu %4d vmem %3d lds %3d' % (
            name[:84], sum(c.values()), grp('v_'), grp('v_pk'),
            grp('v_add_f', 'v_sub_f', 'v_mul_f', 'v_fma', 'v_pk_add_f', 'v_pk_mul_f', 'v_pk_fma', 'v_mac', 'v_rcp', 'v_div'),
            grp('s_'), grp('global_', 'buffer_', 'flat_', 'scratch_'), grp('ds_')))
