"""Global geometry and its partition into subdomains (reference sailfish/geo.py)."""
from sailfish_amd.subdomain import SubdomainSpec2D, SubdomainSpec3D


class LBGeometry(object):
    def __init__(self, config):
        self.config = config
        self.gx = config.lat_nx
        self.gy = config.lat_ny
        self.gsize = [self.gx, self.gy]


class LBGeometry2D(LBGeometry):
    @classmethod
    def add_options(cls, group):
        group.add_argument('--lat_nx', help='lattice width', type=int, default=0)
        group.add_argument('--lat_ny', help='lattice height', type=int, default=0)
        group.add_argument('--periodic_x', dest='periodic_x', action='store_true', default=False,
                           help='make the lattice periodic in the X direction')
        group.add_argument('--periodic_y', dest='periodic_y', action='store_true', default=False,
                           help='make the lattice periodic in the Y direction')

    def subdomains(self):
        return [SubdomainSpec2D((0, 0), (self.config.lat_nx, self.config.lat_ny))]


class LBGeometry3D(LBGeometry):
    @classmethod
    def add_options(cls, group):
        LBGeometry2D.add_options(group)
        group.add_argument('--lat_nz', help='lattice depth', type=int, default=0)
        group.add_argument('--periodic_z', dest='periodic_z', action='store_true', default=False,
                           help='make the lattice periodic in the Z direction')

    def __init__(self, config):
        LBGeometry.__init__(self, config)
        self.gz = config.lat_nz
        self.gsize = [self.gx, self.gy, self.gz]

    def subdomains(self):
        return [SubdomainSpec3D((0, 0, 0), (self.config.lat_nx, self.config.lat_ny, self.config.lat_nz))]


def _split(total, parts):
    """Equal pieces, remainder to the last one (reference geo.py:113-135)."""
    base, rest = total // parts, total % parts
    return [(i * base, base if i < parts - 1 else base + rest) for i in range(parts)]


def _split_rows(total, parts, align, tolerance=0.15):
    """Pieces along x -- the axis the rows of the arrays run along.  A whole-row workgroup stores whole 128-byte lines
    (32 single-precision values; x = 1 of every row sits on a line boundary), except for the last line of a row whose
    length is no multiple of 32: that one is a partial-line store, a read-modify-write per direction and row.  Rows of
    171 nodes (512 / 3) sweep at 29 GMLUPS where rows of 160 and 192 reach 37-38 (profiles/r06/slab_align_ab.txt), so
    the cuts go to multiples of `align` nodes wherever every piece stays within `tolerance` of the equal share (a GPU per
    slab must not wait for the widest one longer than the better rows save); otherwise, and with align = 0, the
    reference's equal pieces."""
    mean = total / float(parts)
    if align and align > 1 and parts > 1 and mean >= 2 * align:
        cuts = [0] + [int(round(i * mean / align)) * align for i in range(1, parts)] + [total]
        sizes = [b - a for a, b in zip(cuts, cuts[1:])]
        if min(sizes) > 0 and max(abs(n - mean) for n in sizes) <= tolerance * mean:
            return list(zip(cuts, sizes))
    if align and align > 32:        # whole waves do not fit: whole lines, at the tolerance of balanced slabs
        return _split_rows(total, parts, 32, min(tolerance, 0.15))
    return _split(total, parts)


def _add_split_options(group, axes):
    group.add_argument('--subdomains', help='number of subdomains', type=int, default=1)
    group.add_argument('--conn_axis', type=str, default='x', choices=axes,
                       help='axis along which the subdomains will be connected')
    group.add_argument('--slab_align', type=int, default=None,
                       help='cut along x at multiples of this many nodes where every slab stays within --slab_tolerance of '
                            'the equal share: rows then end on a 128-byte line (0: equal pieces, as the reference; default: '
                            '32, or 64 = whole waves when every subdomain runs on the one device of this process)')
    group.add_argument('--slab_tolerance', type=float, default=None,
                       help='how far a slab may deviate from the equal share for the sake of --slab_align (default 0.15; '
                            '0.3 on one device, where the slabs run one after the other and their balance does not matter)')


def _row_split_rule(config):
    """(align, tolerance) of the cuts along x.  Slabs that run on DIFFERENT devices must be balanced: a GPU per slab waits
    for the widest one.  Slabs of one process on one device run one after the other (controller.LocalGroup), so only the
    sum of their times counts -- and that is smallest when every row is made of whole waves: the force-driven pipe
    512 x 256 x 256 in three x-slabs, 170 / 170 / 172: 30.0 GMLUPS, 160 / 192 / 160: 33.6-34.1, 192 / 128 / 192: 34.7-35.5
    (profiles/r06/slab_align_ab.txt, slab_align64.txt).  One device = --gpus names one entry (several entries start a process
    each, even on one device) and this is not one rank of several (torchrun / the controller's own ranks: WORLD_SIZE > 1)."""
    import os
    gpus = getattr(config, 'gpus', None)
    if gpus is not None and not isinstance(gpus, (list, tuple)):
        gpus = [gpus]
    one_device = gpus is not None and len(gpus) == 1 and int(os.environ.get('WORLD_SIZE', '1')) <= 1
    align, tol = getattr(config, 'slab_align', None), getattr(config, 'slab_tolerance', None)
    if align is None:
        align = 64 if one_device else 32
    if tol is None:
        tol = 0.3 if one_device else 0.15
    return align, tol


class EqualSubdomainsGeometry2D(LBGeometry2D):
    """--subdomains equal pieces along --conn_axis."""

    @classmethod
    def add_options(cls, group):
        LBGeometry2D.add_options(group)
        _add_split_options(group, ['x', 'y'])

    def subdomains(self):
        s = self.config.subdomains
        if self.config.conn_axis == 'x':
            return [SubdomainSpec2D((o, 0), (n, self.gy)) for o, n in _split_rows(self.gx, s, *_row_split_rule(self.config))]
        return [SubdomainSpec2D((0, o), (self.gx, n)) for o, n in _split(self.gy, s)]


class EqualSubdomainsGeometry3D(LBGeometry3D):
    @classmethod
    def add_options(cls, group):
        LBGeometry3D.add_options(group)
        _add_split_options(group, ['x', 'y', 'z'])

    def subdomains(self):
        s = self.config.subdomains
        if self.config.conn_axis == 'x':
            return [SubdomainSpec3D((o, 0, 0), (n, self.gy, self.gz))
                    for o, n in _split_rows(self.gx, s, *_row_split_rule(self.config))]
        elif self.config.conn_axis == 'y':
            return [SubdomainSpec3D((0, o, 0), (self.gx, n, self.gz)) for o, n in _split(self.gy, s)]
        return [SubdomainSpec3D((0, 0, o), (self.gx, self.gy, n)) for o, n in _split(self.gz, s)]


class WeightedSubdomainsGeometry3D(EqualSubdomainsGeometry3D):
    """Slabs along --conn_axis holding (nearly) the same number of *active* nodes each, for sparse
    geometries where equal-size slabs would leave some GPUs idle (reference geo.py:137-176).

    --geometry_for_decomposition: .npy Boolean array [nz, ny, nx], True = inactive node.  The reference
    builds the node count profile over the wrong array axis for conn_axis != y (its array is z, y, x but
    it indexes the axes as x, y, z); here the profile is taken along the connection axis itself."""

    @classmethod
    def add_options(cls, group):
        EqualSubdomainsGeometry3D.add_options(group)
        group.add_argument('--geometry_for_decomposition', type=str, default='',
                           help='Numpy boolean array with True entries indicating inactive nodes to use to '
                                'decide where to split the domain.')

    def subdomains(self):
        src = getattr(self.config, 'geometry_for_decomposition', '')
        if src is None or (isinstance(src, str) and not src):
            return super(WeightedSubdomainsGeometry3D, self).subdomains()
        import numpy as np
        inactive = np.load(src) if isinstance(src, str) else np.asarray(src)
        assert inactive.shape == (self.gz, self.gy, self.gx), 'decomposition geometry does not match the lattice'
        axis = 'xyz'.index(self.config.conn_axis)
        other = tuple(a for a in range(3) if a != 2 - axis)          # array axes are z, y, x
        profile = np.cumsum(np.sum(np.logical_not(inactive), axis=other).astype(np.int64))
        n, total = int(self.config.subdomains), int(profile[-1])
        # cut after the first layer whose cumulative count reaches k / n of the total
        cuts = [0]
        for k in range(1, n):
            c = int(np.searchsorted(profile, (total * k + n - 1) // n)) + 1
            cuts.append(min(max(c, cuts[-1] + 1), len(profile) - (n - k)))
        cuts.append(len(profile))
        gsize = [self.gx, self.gy, self.gz]
        ret = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            start, size = [0, 0, 0], list(gsize)
            start[axis], size[axis] = lo, hi - lo
            ret.append(SubdomainSpec3D(tuple(start), tuple(size)))
        return ret
