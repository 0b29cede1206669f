#!/usr/bin/env python3
"""Compares two .npz result files field by field, NaN-aware and exact (tool of the reference:
utils/compare_results.py; used by its AA-vs-AB and checkpoint scripts, tests/gpu/*.sh).

    utils/compare_results.py a.npz b.npz      exit status = number of differing fields
"""
import sys

import numpy as np


def compare(path_a, path_b, out=sys.stderr):
    a, b = np.load(path_a), np.load(path_b)
    if sorted(a.files) != sorted(b.files):
        print('Different fields: %s vs %s' % (a.files, b.files), file=out)
        return 1
    bad = 0
    for name in a.files:
        x, y = a[name], b[name]
        if x.shape != y.shape:
            print('Field "%s": shapes differ, %s vs %s.' % (name, x.shape, y.shape), file=out)
            bad += 1
        elif not np.array_equal(np.nan_to_num(x), np.nan_to_num(y)) or \
                not np.array_equal(np.isnan(x), np.isnan(y)):
            with np.errstate(invalid='ignore'):
                dev = np.nanmax(np.abs(x - y))
            print('Difference in field "%s", max deviation is: %e.' % (name, dev), file=out)
            bad += 1
    return bad


if __name__ == '__main__':
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    sys.exit(compare(sys.argv[1], sys.argv[2]))
