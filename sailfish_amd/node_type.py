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


class DynamicValue(object):
    """Time / space dependent BC values are generated as device code by the reference
    (node_type.py:471-570); the pre-built gfx950 kernels take constant parameters only."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('DynamicValue boundary parameters are not supported by the HIP backend')
