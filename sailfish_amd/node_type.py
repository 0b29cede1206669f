"""Node (boundary condition) types.

Same public names, class attributes and id-assignment rule as the reference's
sailfish/node_type.py (ids are handed out in the alphabetical order of the class
names, `_NTFluid` is 0: node_type.py:406-421), so that user geometry code
(`self.set_node(where, NTFullBBWall)`, `NTRegularizedVelocity((0.1, 0.0))`, ...)
runs unchanged and un-encoded type maps agree with the reference's.

The gfx950 kernels implement the types the north-star configurations use
(`HIP_KIND`); any other type raises when a subdomain using it is built for the
HIP backend.
"""
from collections import namedtuple

import numpy as np

from sailfish_amd import hipabi

ScratchSize = namedtuple('ScratchSize', ('dim2', 'dim3'))


class LBNodeType(object):
    """A node type = a set of flags the host pipeline and the kernels look at, plus the values an instance carries
    (`params`: name -> number, tuple or per-node array) and optionally a fixed orientation."""
    id = None
    wet_node = False            # undergoes the normal relaxation
    excluded = False            # does not take part in the simulation
    propagation_only = False
    standard_macro = False      # macroscopic fields computed the standard way
    needs_orientation = False
    link_tags = False           # the orientation field holds per-direction link tags
    scratch_space = 0           # reals of per-node scratch memory (ScratchSize where it depends on the dimension)
    location = 0.0              # wall position relative to the node, along the normal
    allow_unused = False
    value_name = None           # name of the single positional value of the type ('density', 'velocity'), if any

    def __init__(self, *value, **params):
        self.orientation = params.pop('orientation', None)
        if self.value_name is not None:
            if len(value) > 2 or (not value and self.value_name not in params):
                raise TypeError('%s(%s[, orientation])' % (type(self).__name__, self.value_name))
            if value:
                params[self.value_name] = value[0]
            if len(value) == 2:
                self.orientation = value[1]
        elif value:
            raise TypeError('%s takes keyword parameters only' % type(self).__name__)
        if self.orientation is None and not (self.value_name and self.needs_orientation):
            del self.orientation            # "not given": set_node() then asks the geometry for it
        self.params = params

    @classmethod
    def scratch_space_size(cls, dim):
        sp = cls.scratch_space
        return sp if isinstance(sp, int) else (sp.dim2 if dim == 2 else sp.dim3)


# The node types: name, value parameter, flags, kernel kind of the gfx950 kernels (None = not implemented there).
# Names and flags are the reference's (sailfish/node_type.py:86-400: user code says `NTFullBBWall`,
# `NTRegularizedVelocity((0.1, 0.0))`, ...); ids follow from the names (see _number_node_types).
_W, _S, _O = dict(wet_node=True), dict(standard_macro=True), dict(needs_orientation=True)
_OUTFLOW = dict(wet_node=True, standard_macro=True, needs_orientation=True)
_TABLE = (
    ('_NTFluid', None, dict(_W, id=0, **_S), hipabi.SLF_NK_FLUID),
    ('_NTGhost', None, dict(excluded=True), hipabi.SLF_NK_GHOST),
    ('_NTUnused', None, dict(excluded=True), hipabi.SLF_NK_UNUSED),
    ('_NTPropagationOnly', None, dict(propagation_only=True), hipabi.SLF_NK_PROPAGATION_ONLY),
    # walls: half-way bounce-back reflects at the link mid-point (link tags), full-way at the node
    ('NTHalfBBWall', None, dict(_W, link_tags=True, location=-0.5, allow_unused=True, **dict(_S, **_O)), hipabi.SLF_NK_HALF_BB),
    ('NTFullBBWall', None, dict(_S, location=0.5, **_O), hipabi.SLF_NK_FULL_BB),
    ('NTWallTMS', None, dict(_W, link_tags=True, location=0.5, allow_unused=True, **dict(_S, **_O)), None),
    ('NTSlip', None, dict(_S), hipabi.SLF_NK_SLIP),             # dry, specular reflection; the orientation is given: NTSlip(orientation=...)
    # imposed density (pressure)
    ('NTEquilibriumDensity', 'density', dict(_W, **_O), hipabi.SLF_NK_EQUILIBRIUM_DENSITY),
    ('NTRegularizedDensity', 'density', dict(_W, **_O), hipabi.SLF_NK_REGULARIZED_DENSITY),
    ('NTZouHeDensity', 'density', dict(_W, **_O), hipabi.SLF_NK_ZOUHE_DENSITY),
    ('NTGuoDensity', 'density', dict(), None),
    # imposed velocity
    ('NTEquilibriumVelocity', 'velocity', dict(_W, **_O), hipabi.SLF_NK_EQUILIBRIUM_VELOCITY),
    ('NTZouHeVelocity', 'velocity', dict(_W, **_O), hipabi.SLF_NK_ZOUHE_VELOCITY),
    ('NTRegularizedVelocity', 'velocity', dict(_W, **_O), hipabi.SLF_NK_REGULARIZED_VELOCITY),   # Latt et al., PRE 77 056703
    # outflow and friends
    ('NTCopy', None, _OUTFLOW, hipabi.SLF_NK_COPY),                 # two-copy (AB) access pattern only, like the reference
    ('NTYuOutflow', None, _OUTFLOW, hipabi.SLF_NK_YU_OUTFLOW),      # AB only
    ('NTDoNothing', None, _OUTFLOW, hipabi.SLF_NK_DO_NOTHING),      # AB: a plain fluid node (hip_kind()); AA: keeps its unknown populations
    ('NTExtendedCopy', None, _OUTFLOW, None),
    ('NTNeumann', None, _OUTFLOW, None),
    ('NTLaminarize', None, _OUTFLOW, None),
    ('NTGradFreeflow', None, dict(_W, scratch_space=ScratchSize(dim2=3, dim3=6), **_S), None),
)

HIP_KIND = {}      # node type class -> canonical kernel kind (SLF_NK_*)
for _name, _value, _flags, _kind in _TABLE:
    _cls = type(_name, (LBNodeType,), dict(_flags, value_name=_value, __module__=__name__))
    globals()[_name] = _cls
    if _kind is not None:
        HIP_KIND[_cls] = _kind


def hip_kind(cls, access_pattern):
    """Kernel kind (SLF_NK_*) of a node type under an access pattern; None: the gfx950 kernels do not implement it.
    NTDoNothing exists for the in-place pattern only -- "in the AB memory layout, leaving the outflow nodes defined as
    NTFluid works just fine" (reference node_type.py:296-307), and that is what it is there."""
    kind = HIP_KIND.get(cls)
    if kind == hipabi.SLF_NK_DO_NOTHING and access_pattern != 'AA':
        return hipabi.SLF_NK_FLUID
    return kind


def _number_node_types():
    """Every LBNodeType subclass defined in this module gets the next free id, walking the module in name order
    (that order is the contract: ids end up in node maps, golden fixtures and checkpoints; classes that fix their
    own id keep it).  Returns {id: class}."""
    classes = [obj for _, obj in sorted(globals().items())
               if isinstance(obj, type) and issubclass(obj, LBNodeType) and obj is not LBNodeType]
    for position, cls in enumerate(classes, start=1):
        if cls.id is None:
            cls.id = position
    return {cls.id: cls for cls in classes}


_NODE_TYPES = _number_node_types()


def get_wet_node_type_ids(allow_unused=None):
    return [i for i, t in _NODE_TYPES.items() if t.wet_node and
            (allow_unused is None or t.allow_unused == allow_unused)]


def get_dry_node_type_ids():
    return [i for i, t in _NODE_TYPES.items() if not t.wet_node]


def get_orientation_node_type_ids():
    return [i for i, t in _NODE_TYPES.items() if t.needs_orientation]


def get_link_tag_node_type_ids():
    return [i for i, t in _NODE_TYPES.items() if t.link_tags]


def multifield(values, where=None):
    """One record array holding several per-node parameters (e.g. the components of a velocity): every entry of
    `values` is an array over the nodes or a scalar that is broadcast to them.  With `where` the records of the
    selected nodes are returned, otherwise all of them, flattened -- the form set_node() expects."""
    arrays = [v for v in values if isinstance(v, np.ndarray)]
    if not arrays:
        raise AssertionError('multifield() needs at least one array to take the node shape from')
    shape = arrays[0].shape
    if any(a.shape != shape for a in arrays):
        raise AssertionError('multifield(): arrays of different shapes')
    columns = [np.array(np.broadcast_to(v, shape), dtype=np.float64) for v in values]
    rec = np.rec.fromarrays(columns)
    return rec.flatten() if where is None else rec[where]


def timeseries_interpolate(data, step_size, iteration):
    """Value of a wrapped, linearly interpolated time series at LB iteration `iteration` (reference boundary.mako:52-76:
    position = iteration mod (step * size); neighbours idx, idx + 1 wrapped; pos * d1 + d0 * (1 - pos))."""
    data = np.asarray(data, dtype=np.float64)
    size = data.size
    pos = np.fmod(float(iteration), float(step_size) * size) / float(step_size)
    idx = int(np.floor(pos))
    w = pos - idx
    idx2 = idx + 1
    if idx2 >= size:
        idx2 -= size
    return w * data[idx2] + data[idx] * (1.0 - w)


_lits_class = None


def _lits():
    """LinearlyInterpolatedTimeSeries is a sympy Symbol (it takes part in expressions): defined on first use so that
    importing this module does not import sympy."""
    global _lits_class
    if _lits_class is not None:
        return _lits_class
    import hashlib
    from sympy import Symbol

    class LinearlyInterpolatedTimeSeries(Symbol):
        """A time-dependent scalar data source based on a discrete time series (reference node_type.py:572-626): the data
        points are `step_size` LB iterations apart, values in between are interpolated linearly, the series is wrapped.
        Two series are equal iff their data and step sizes are."""

        def __new__(cls, data, step_size=1.0):
            arr = np.ascontiguousarray(np.float64(data))
            return Symbol.__new__(cls, 'lits%s_%s' % (hashlib.sha1(arr).hexdigest(), step_size))

        def __init__(self, data, step_size=1.0):
            self._data = np.ascontiguousarray(np.float64(data)).copy()
            self._step_size = step_size

        def __hash__(self):
            return hash((hashlib.sha1(self._data).digest(), str(self._step_size)))

        def __eq__(self, other):
            return isinstance(other, LinearlyInterpolatedTimeSeries) and self._step_size == other._step_size and \
                self._data.shape == other._data.shape and bool(np.all(other._data == self._data))

        def __ne__(self, other):
            return not self.__eq__(other)

        def __str__(self):
            return 'LinearlyInterpolatedTimeSeries([%d items], %f)' % (self._data.size, self._step_size)

        def data_hash(self):
            return hashlib.sha1(self._data).digest()

        def at(self, iteration):
            return timeseries_interpolate(self._data, self._step_size, iteration)

    _lits_class = LinearlyInterpolatedTimeSeries
    return _lits_class


def __getattr__(name):          # PEP 562: `from sailfish.node_type import LinearlyInterpolatedTimeSeries`
    if name == 'LinearlyInterpolatedTimeSeries':
        return _lits()
    raise AttributeError(name)


class DynamicValue(object):
    """A node parameter given as expressions of the node location (sym.S.gx, S.gy, S.gz), the time (S.time = iteration x
    --dt_per_lattice_time_unit) and time series (LinearlyInterpolatedTimeSeries) -- reference node_type.py:471-570, where
    the expressions become device code.  Here they are evaluated on the host (numpy, through sympy.lambdify): once per
    node when the geometry is encoded, and once per step for the values that depend on time, which the runner writes
    into the kernels' node-parameter table before the step (SubdomainRunner._update_dynamic_params).  One expression per
    component (a density: one; a velocity: one per axis)."""

    def __init__(self, *params):
        self.params = tuple(params)
        self._fn = None

    def __hash__(self):
        return hash(self.params)

    def __eq__(self, other):
        return isinstance(other, DynamicValue) and self.params == other.params

    def __ne__(self, other):
        return not self.__eq__(other)

    def __iter__(self):
        return iter(self.params)

    def __len__(self):
        return len(self.params)

    def __getitem__(self, i):
        return self.params[i]

    def __str__(self):
        return 'DynamicValue(' + ', '.join(str(x) for x in self.params) + ')'

    def _symbols(self):
        out = set()
        for p in self.params:
            out |= getattr(p, 'free_symbols', set())
        return out

    def has_symbols(self, *args):
        """True if any expression depends on at least one of the given symbols."""
        mine = self._symbols()
        return any(a in mine for a in args)

    def get_timeseries(self):
        lits = _lits()
        return [x for x in self._symbols() if isinstance(x, lits)]

    def time_dependent(self):
        from sailfish_amd import sym
        return self.has_symbols(sym.S.time) or bool(self.get_timeseries())

    def space_dependent(self):
        from sailfish_amd import sym
        return self.has_symbols(sym.S.gx, sym.S.gy, sym.S.gz)

    def evaluate(self, coords, iteration, dt):
        """[nodes, components] values at the nodes with global coordinates coords = (gx, gy[, gz]) (equal-length 1-D arrays)
        at LB iteration `iteration`."""
        import sympy
        from sailfish_amd import sym
        series = self.get_timeseries()
        if self._fn is None:
            args = [sym.S.gx, sym.S.gy, sym.S.gz, sym.S.time] + series
            self._fn = [sympy.lambdify(args, sympy.sympify(p), 'numpy') for p in self.params]
        n = len(coords[0])
        g = [np.asarray(c, dtype=np.float64) for c in coords] + [np.zeros(n)] * (3 - len(coords))
        vals = [s_.at(iteration) for s_ in series]
        out = np.empty((n, len(self.params)), dtype=np.float64)
        for k, fn in enumerate(self._fn):
            out[:, k] = np.broadcast_to(np.asarray(fn(g[0], g[1], g[2], float(iteration) * float(dt), *vals), dtype=np.float64), (n,))
        return out
