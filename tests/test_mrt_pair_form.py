"""The D3Q19 moment transform through the pairs of opposite directions (sailfish_amd/csrc/slf_node.h
mrt_forward_d3q19 / mrt_inverse_d3q19, oracle/lbm_oracle.c) is the SAME linear map as the integer matrix of the reference
(sym.py:331-378, M^-1 = M^T diag(1 / |row|^2), sym.py:716-735): the sequence of operations the kernels and the oracle
execute, restated here over exact rationals and compared entry by entry with M f and M^T (m / |row|^2)."""
import random
from fractions import Fraction as Fr

from sailfish_amd import sym


def fma(a, b, c):
    return a * b + c


def forward(f):
    f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11, f12, f13, f14, f15, f16, f17, f18 = f
    sx, dx, sy, dy, sz, dz = f1 + f2, f1 - f2, f3 + f4, f3 - f4, f5 + f6, f5 - f6
    sA, dA, sB, dB = f7 + f10, f7 - f10, f8 + f9, f9 - f8
    sC, dC, sD, dD = f11 + f14, f11 - f14, f12 + f13, f13 - f12
    sE, dE, sF, dF = f15 + f18, f15 - f18, f16 + f17, f17 - f16
    xab, yab, ycd, zcd, xef, zef = dA + dB, dA - dB, dC + dD, dC - dD, dE + dF, dE - dF
    X, Y, Z = xab + xef, yab + ycd, zcd + zef
    m = [None] * 19
    m[3], m[4], m[16] = dx + X, fma(-4, dx, X), xab - xef
    m[5], m[6], m[17] = dy + Y, fma(-4, dy, Y), ycd - yab
    m[7], m[8], m[18] = dz + Z, fma(-4, dz, Z), zef - zcd
    Sxy, Syz, Szx = sA + sB, sC + sD, sE + sF
    A1, B1 = (sx + sy) + sz, (Sxy + Syz) + Szx
    m[0] = (f0 + A1) + B1
    m[1] = fma(8, B1, fma(-11, A1, -30 * f0))
    m[2] = fma(-4, A1, fma(12, f0, B1))
    P, Qd = fma(2, sx, 0 - (sy + sz)), fma(-2, Syz, Sxy + Szx)
    m[9], m[10] = P + Qd, fma(-2, P, Qd)
    W, V = sy - sz, Sxy - Szx
    m[11], m[12] = W + V, fma(-2, W, V)
    m[13], m[14], m[15] = sA - sB, sC - sD, sE - sF
    return m


def inverse(m):
    f = [None] * 19
    K0 = fma(-4, m[2], fma(-11, m[1], m[0]))
    K1 = fma(8, m[1], m[0]) + m[2]
    f[0] = fma(12, m[2], fma(-30, m[1], m[0]))
    G, H = fma(2, m[10], 0 - m[9]), fma(-2, m[12], m[11])
    Ex, KG = fma(-2, G, K0), K0 + G
    Ey, Ez = KG + H, KG - H
    Ox, Oy, Oz = fma(-4, m[4], m[3]), fma(-4, m[6], m[5]), fma(-4, m[8], m[7])
    f[1], f[2], f[3], f[4], f[5], f[6] = Ex + Ox, Ex - Ox, Ey + Oy, Ey - Oy, Ez + Oz, Ez - Oz
    S9, S11 = m[9] + m[10], m[11] + m[12]
    K19 = K1 + S9
    Txy, Tzx, Tyz = K19 + S11, K19 - S11, fma(-2, S9, K1)
    ax, ay, az = m[3] + m[4], m[5] + m[6], m[7] + m[8]
    ep, em = Txy + m[13], Txy - m[13]
    oA, oB = (ax + ay) + (m[16] - m[17]), (ax - ay) + (m[16] + m[17])
    f[7], f[10], f[9], f[8] = ep + oA, ep - oA, em + oB, em - oB
    ep, em = Tyz + m[14], Tyz - m[14]
    oC, oD = (ay + az) + (m[17] - m[18]), (ay - az) + (m[17] + m[18])
    f[11], f[14], f[13], f[12] = ep + oC, ep - oC, em + oD, em - oD
    ep, em = Tzx + m[15], Tzx - m[15]
    oE, oF = (ax + az) + (m[18] - m[16]), (ax - az) - (m[16] + m[18])
    f[15], f[18], f[17], f[16] = ep + oE, ep - oE, em + oF, em - oF
    return f


def test_pair_form_is_the_reference_matrix():
    g = sym.D3Q19
    M = [[int(c) for c in row] for row in g.mrt_matrix]
    norm = [int(n) for n in g.mrt_norms]
    assert [tuple(e) for e in g.basis[1:7]] == [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    assert list(g.idx_opposite) == [0, 2, 1, 4, 3, 6, 5, 10, 9, 8, 7, 14, 13, 12, 11, 18, 17, 16, 15]
    rng = random.Random(5)
    for _ in range(5):
        f = [Fr(rng.randint(-1000, 1000), 997) for _ in range(19)]
        assert forward(f) == [sum(M[k][i] * f[i] for i in range(19)) for k in range(19)]
        m = [Fr(rng.randint(-1000, 1000), 991) / norm[k] for k in range(19)]
        assert inverse(m) == [sum(M[k][i] * m[k] for k in range(19)) for i in range(19)]
    # and the round trip is the identity: M^T diag(1 / |row|^2) M = 1
    f = [Fr(rng.randint(1, 1000), 1009) for _ in range(19)]
    m = forward(f)
    assert inverse([m[k] / norm[k] for k in range(19)]) == f


def test_source_order_matches_this_restatement():
    """The C and HIP sources spell the same operations (a guard against editing one side only): every assignment of the
    restatement above appears in both files, up to the language's spelling."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c_src = open(os.path.join(root, 'oracle', 'lbm_oracle.c')).read()
    h_src = open(os.path.join(root, 'sailfish_amd', 'csrc', 'slf_node.h')).read()

    def body(src, name):
        i = src.index('void ' + name)
        text = re.sub(r'//[^\n]*', '', src[i:src.index('\n}', i)])
        return re.sub(r'\s+', '', text)
    for name in ('mrt_forward_d3q19', 'mrt_inverse_d3q19'):
        c = body(c_src, name)
        h = body(h_src, name)
        # normalise the two spellings: FMA((real)-4, dx, X) <-> fma_<R>((R)-4, dx, X); real <-> R
        c = c.replace('FMA(', 'fma(').replace('(real)', '(R)').replace('constreal', 'constR').replace('realep', 'Rep')
        h = h.replace('fma_<R>(', 'fma(')
        assert c[c.index('{'):] == h[h.index('{'):].replace('(&f)[19]', '*f'), name
