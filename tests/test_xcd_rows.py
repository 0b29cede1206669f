"""The row -> XCD regrouping of the whole-row launches (xcd_row() / xcd_shift_for(), sailfish_amd/csrc/slf_sweep.h), restated:
workgroup b of a launch (dealt to XCD b mod 8 by the hardware) sweeps row xcd_row(b, s).  The properties the kernels rely
on: a permutation inside every block of 8 << s rows, every XCD's rows of a block consecutive, identity for s = 0, and the
shift the launchers pick divides the row count.  (That the regrouped launches compute the same as the plain ones is what
the GPU parity tests with 64-row planes show: tests/test_gpu_runner.py, tests/test_gpu_sc.py.)"""
import re
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def xcd_row(by, s):
    if s == 0:
        return by
    r = by & ((8 << s) - 1)
    return by - r + ((r & 7) << s) + (r >> 3)


def xcd_shift_for(rows, grid_x, max_shift=5):
    if grid_x != 1:
        return 0
    s = max_shift
    while s > 0 and rows % (8 << s):
        s -= 1
    return s


@pytest.mark.parametrize('s', range(0, 7))
def test_rows_of_a_block_are_permuted_and_every_xcd_gets_consecutive_ones(s):
    n = (8 << s) * 3
    m = [xcd_row(b, s) for b in range(n)]
    assert sorted(m) == list(range(n))
    for blk in range(3):
        base = blk * (8 << s)
        assert sorted(m[base:base + (8 << s)]) == list(range(base, base + (8 << s)))       # rows stay inside their block
        for c in range(8):
            rows = [m[b] for b in range(base, base + (8 << s)) if b % 8 == c]
            assert rows == list(range(rows[0], rows[0] + (1 << s)))
    if s == 0:
        assert m == list(range(n))


@pytest.mark.parametrize('rows,grid_x,expect', [(512, 1, 5), (256, 1, 5), (384, 1, 4), (64, 1, 3), (9, 1, 0), (512, 2, 0),
                                                (40, 1, 0), (48, 1, 1), (8, 1, 0)])
def test_shift_divides_the_row_count(rows, grid_x, expect):
    s = xcd_shift_for(rows, grid_x)
    assert s == expect
    assert s == 0 or rows % (8 << s) == 0


def test_restatement_matches_the_source():
    src = open(os.path.join(ROOT, 'sailfish_amd', 'csrc', 'slf_sweep.h')).read()
    body = src[src.index('__device__ __forceinline__ int xcd_row('):src.index('static inline int xcd_shift_for(')]
    assert re.search(r'by & \(\(8 << s\) - 1\)', body) and re.search(r'by - r \+ \(\(r & 7\) << s\) \+ \(r >> 3\)', body)
    assert 'atoi(e) : 5' in src and '(rows % (8u << s)) != 0' in src
