"""Lattice algebra for the host side (plain numpy / fractions, no sympy at run time).

Mirrors the *interface* of the reference's sailfish/sym.py for the parts the
hot path needs (grid classes with basis / weights / idx_name / idx_opposite /
dir2vecidx / mrt_matrix, and the helper functions get_prop_dists,
get_interblock_dists, get_missing_dists, bb_swap_pairs, relaxation_time).
The reference builds these with sympy (sym.py:61-149, 312-406, 944-1045); here
they are computed directly.  Pinned against tests/golden/lattices.json.
"""
from fractions import Fraction

import numpy as np


class DxQy(object):
    """Container class, never instantiated (reference sym.py:24-58)."""
    cssq = Fraction(1, 3)

    @classmethod
    def vec_idx(cls, vec):
        return cls.basis.index(tuple(int(x) for x in vec))

    @classmethod
    def vec_to_dir(cls, vec):
        return cls.vecidx2dir[cls.vec_idx(vec)]

    @classmethod
    def dir_to_vec(cls, dir_):
        return cls.basis[cls.dir2vecidx[dir_]]

    @classmethod
    def model_supported(cls, model):
        if model == 'mrt':
            return hasattr(cls, 'mrt_matrix')
        return model == 'bgk'


class D2Q9(DxQy):
    dim = 2
    Q = 9
    slf_id = 0
    basis = [(0, 0), (1, 0), (0, 1), (-1, 0), (0, -1), (1, 1), (-1, 1), (-1, -1), (1, -1)]
    weights = [Fraction(4, 9)] + [Fraction(1, 9)] * 4 + [Fraction(1, 36)] * 4
    mrt_names = ['rho', 'en', 'ens', 'mx', 'ex', 'my', 'ey', 'pxx', 'pxy']
    # 0 = conserved; None = 1/tau (shear) -- reference sym.py:78-84,110-114
    mrt_collision = [0.0, 1.63, 1.14, 0.0, 1.9, 0.0, 1.9, None, None]

    @classmethod
    def _mrt_basis(cls):
        b = cls.basis
        sq = [x * x + y * y for x, y in b]
        return [[1] * 9, sq, [s * s for s in sq], [x for x, y in b], [x * s for (x, y), s in zip(b, sq)],
                [y for x, y in b], [y * s for (x, y), s in zip(b, sq)], [x * x - y * y for x, y in b],
                [x * y for x, y in b]]


class D3Q19(DxQy):
    dim = 3
    Q = 19
    slf_id = 1
    basis = [(0, 0, 0),
             (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1),
             (1, 1, 0), (-1, 1, 0), (1, -1, 0), (-1, -1, 0),
             (0, 1, 1), (0, -1, 1), (0, 1, -1), (0, -1, -1),
             (1, 0, 1), (-1, 0, 1), (1, 0, -1), (-1, 0, -1)]
    weights = [Fraction(1, 3)] + [Fraction(1, 18)] * 6 + [Fraction(1, 36)] * 12
    mrt_names = ['rho', 'en', 'eps', 'mx', 'ex', 'my', 'ey', 'mz', 'ez',
                 'pxx3', 'pixx3', 'pww', 'piww', 'pxy', 'pyz', 'pzx', 'm3x', 'm3y', 'm3z']
    mrt_collision = [0.0, 1.19, 1.4, 0.0, 1.2, 0.0, 1.2, 0.0, 1.2,
                     None, 1.4, None, 1.4, None, None, None, 1.98, 1.98, 1.98]

    @classmethod
    def _mrt_basis(cls):
        b = cls.basis
        sq = [x * x + y * y + z * z for x, y, z in b]
        return [[1] * 19, sq, [s * s for s in sq],
                [x for x, y, z in b], [x * s for (x, y, z), s in zip(b, sq)],
                [y for x, y, z in b], [y * s for (x, y, z), s in zip(b, sq)],
                [z for x, y, z in b], [z * s for (x, y, z), s in zip(b, sq)],
                [3 * x * x - s for (x, y, z), s in zip(b, sq)],
                [(3 * s - 5) * (3 * x * x - s) for (x, y, z), s in zip(b, sq)],
                [y * y - z * z for x, y, z in b],
                [(3 * s - 5) * (y * y - z * z) for (x, y, z), s in zip(b, sq)],
                [x * y for x, y, z in b], [y * z for x, y, z in b], [x * z for x, y, z in b],
                [(y * y - z * z) * x for x, y, z in b],
                [(z * z - x * x) * y for x, y, z in b],
                [(x * x - y * y) * z for x, y, z in b]]


KNOWN_GRIDS = (D2Q9, D3Q19)

_NAME_MAP_2D = {(0, 0): 'C', (1, 0): 'E', (-1, 0): 'W', (0, 1): 'N', (0, -1): 'S'}


def _dist_name(vec):
    """Population names, reference sym.py:990-1012: vertical component first (T/B), then N/S, then E/W."""
    if not any(vec):
        return 'fC'
    name = 'f'
    if len(vec) == 3 and vec[2]:
        name += 'T' if vec[2] > 0 else 'B'
    if vec[1]:
        name += 'N' if vec[1] > 0 else 'S'
    if vec[0]:
        name += 'E' if vec[0] > 0 else 'W'
    return name


def _gram_schmidt_int(rows):
    """Integer Gram-Schmidt orthogonalisation (reference sym.py:944-960 uses
    sympy.GramSchmidt and clears denominators)."""
    out = []
    for r in rows:
        v = [Fraction(x) for x in r]
        for u in out:
            num = sum(a * b for a, b in zip(v, u))
            den = sum(b * b for b in u)
            c = Fraction(num, den) if den else 0
            v = [a - c * b for a, b in zip(v, u)]
        # clear denominators, keep sign
        lcm = 1
        for a in v:
            d = a.denominator
            g = np.gcd(lcm, d)
            lcm = lcm * d // g
        v = [int(a * lcm) for a in v]
        g = 0
        for a in v:
            g = int(np.gcd(g, abs(a)))
        if g > 1:
            v = [a // g for a in v]
        out.append([Fraction(a) for a in v])
    return [[int(a) for a in r] for r in out]


def _prepare_grids():
    for grid in KNOWN_GRIDS:
        grid.idx_name = [_dist_name(v) for v in grid.basis]
        grid.idx_opposite = [grid.basis.index(tuple(-c for c in v)) for v in grid.basis]
        # orientation codes: 1..2*dim <-> primary directions in basis order (sym.py:1013-1018)
        grid.dir2vecidx = {}
        grid.vecidx2dir = {}
        d = 1
        for i, v in enumerate(grid.basis):
            if sum(abs(c) for c in v) == 1:
                grid.dir2vecidx[d] = i
                grid.vecidx2dir[i] = d
                d += 1
        grid.mrt_matrix = np.array(_gram_schmidt_int(grid._mrt_basis()), dtype=np.int64)
        grid.mrt_norms = (grid.mrt_matrix * grid.mrt_matrix).sum(axis=1)
        grid.weights_float = np.array([float(w) for w in grid.weights])
        grid.basis_array = np.array(grid.basis, dtype=np.int64)


_prepare_grids()


def relaxation_time(viscosity):
    """tau = (6 nu + 1) / 2, reference sym.py:847-848."""
    return (6.0 * viscosity + 1.0) / 2.0


def mrt_rates(grid, visc):
    """Per-moment relaxation rates with the viscosity-dependent ones filled in
    (inv_tau = 1 / (0.5 + 3 visc), reference sym.py:110,371)."""
    inv_tau = 1.0 / (0.5 + 3.0 * visc)
    return [inv_tau if c is None else float(c) for c in grid.mrt_collision]


def bb_swap_pairs(grid):
    """reference sym.py:468-479"""
    return set(min(i, j) for i, j in enumerate(grid.idx_opposite) if i != j)


def get_prop_dists(grid, dir_, axis=0):
    """Populations whose `axis` component equals dir_ (rest vector excluded), reference sym.py:819-827."""
    return [i for i, e in enumerate(grid.basis) if e[axis] == dir_ and i > 0]


def get_interblock_dists(grid, direction, opposite=False):
    """Populations transferred to a neighbour subdomain lying in `direction`, reference sym.py:829-845."""
    d = tuple(direction)
    dd = sum(c * c for c in d)
    ret = [i for i, e in enumerate(grid.basis) if sum(a * b for a, b in zip(e, d)) >= dd]
    if opposite:
        return [grid.idx_opposite[i] for i in ret]
    return ret


def get_missing_dists(grid, orientation):
    """Unknown populations at a node whose inward normal is `orientation`, reference sym.py:534-543, 737-747."""
    n = grid.dir_to_vec(orientation)
    return [i for i, e in enumerate(grid.basis) if sum(a * b for a, b in zip(e, n)) > 0]


def missing_dirs_from_tag(grid, tag_code):
    """reference sym.py:518-532"""
    ret = []
    for i, name in enumerate(grid.idx_name[1:]):
        if (tag_code & 1) == 0:
            ret.append(grid.idx_name[grid.idx_opposite[i + 1]])
        tag_code >>= 1
    return ret


def lookup_grid(name):
    for g in KNOWN_GRIDS:
        if g.__name__ == name:
            return g
    raise ValueError('unsupported grid %s (the HIP backend implements D2Q9 and D3Q19)' % name)


class _Symbols(object):
    """`sym.S`: the symbols user code writes time- and space-dependent values with (reference sym.py:1049-1149:
    S.gx / S.gy / S.gz -- node location in the global coordinate system, S.time -- time in physical units).  sympy objects,
    created on first use: nothing else on the host side needs sympy.  Values built from them (node_type.DynamicValue) are
    evaluated on the HOST -- per node when the geometry is encoded, per step where they depend on time -- and reach the
    kernels as ordinary entries of the node-parameter table (the reference renders them into device code)."""
    _names = {'gx': 'gx', 'gy': 'gy', 'gz': 'gz', 'time': 'phys_time'}

    def __getattr__(self, name):
        if name not in self._names:
            raise AttributeError(name)
        import sympy
        sym = sympy.Symbol(self._names[name], real=True)
        setattr(self, name, sym)
        return sym


S = _Symbols()
