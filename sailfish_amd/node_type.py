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
    """Base class for node types (reference node_type.py:18-80)."""
    id = None
    wet_node = False            # undergoes the normal relaxation
    excluded = False            # does not take part in the simulation
    propagation_only = False
    standard_macro = False      # macroscopic fields computed the standard way
    needs_orientation = False
    link_tags = False           # orientation field holds per-direction link tags
    scratch_space = 0
    location = 0.0              # wall position offset along the normal
    allow_unused = False

    def __init__(self, **params):
        if 'orientation' in params:
            self.orientation = params['orientation']
            del params['orientation']
        self.params = params

    @classmethod
    def scratch_space_size(cls, dim):
        if type(cls.scratch_space) is int:
            return cls.scratch_space
        return cls.scratch_space.dim2 if dim == 2 else cls.scratch_space.dim3


# -- special types ---------------------------------------------------------------
class _NTFluid(LBNodeType):
    wet_node = True
    standard_macro = True
    id = 0


class _NTGhost(LBNodeType):
    excluded = True


class _NTUnused(LBNodeType):
    excluded = True


class _NTPropagationOnly(LBNodeType):
    propagation_only = True


# -- walls -----------------------------------------------------------------------
class NTHalfBBWall(LBNodeType):
    """Half-way bounce-back: f_i(x, t+1) = f_opp(i)^post(x, t) (reference node_type.py:115-141)."""
    wet_node = True
    standard_macro = True
    needs_orientation = True
    link_tags = True
    location = -0.5
    allow_unused = True


class NTFullBBWall(LBNodeType):
    """Full-way bounce-back (reference node_type.py:144-168)."""
    standard_macro = True
    location = 0.5
    needs_orientation = True


class NTWallTMS(LBNodeType):
    wet_node = True
    needs_orientation = True
    link_tags = True
    location = 0.5
    allow_unused = True
    standard_macro = True


# -- density (pressure) nodes ------------------------------------------------------
class NTEquilibriumDensity(LBNodeType):
    """Density BC using the equilibrium distribution (reference node_type.py:198-205)."""
    needs_orientation = True
    wet_node = True

    def __init__(self, density, orientation=None):
        self.params = {'density': density}
        self.orientation = orientation


class NTRegularizedDensity(LBNodeType):
    needs_orientation = True
    wet_node = True

    def __init__(self, density, orientation=None):
        self.params = {'density': density}
        self.orientation = orientation


class NTGuoDensity(LBNodeType):
    def __init__(self, density):
        self.params = {'density': density}


class NTZouHeDensity(LBNodeType):
    needs_orientation = True
    wet_node = True

    def __init__(self, density, orientation=None):
        self.params = {'density': density}
        self.orientation = orientation


# -- velocity nodes ------------------------------------------------------------------
class NTEquilibriumVelocity(LBNodeType):
    needs_orientation = True
    wet_node = True

    def __init__(self, velocity, orientation=None):
        self.params = {'velocity': velocity}
        self.orientation = orientation


class NTZouHeVelocity(LBNodeType):
    needs_orientation = True
    wet_node = True

    def __init__(self, velocity, orientation=None):
        self.params = {'velocity': velocity}
        self.orientation = orientation


class NTRegularizedVelocity(LBNodeType):
    """Regularized velocity BC, Latt et al. PRE 77 056703 (reference node_type.py:269-283)."""
    needs_orientation = True
    wet_node = True

    def __init__(self, velocity, orientation=None):
        self.params = {'velocity': velocity}
        self.orientation = orientation


# -- outflow / misc (declared for API compatibility; not implemented by the HIP kernels) ---
class NTGradFreeflow(LBNodeType):
    wet_node = True
    standard_macro = True
    scratch_space = ScratchSize(dim2=3, dim3=6)


class NTDoNothing(LBNodeType):
    wet_node = True
    standard_macro = True
    needs_orientation = True


class NTCopy(LBNodeType):
    wet_node = True
    standard_macro = True
    needs_orientation = True


class NTExtendedCopy(LBNodeType):
    wet_node = True
    standard_macro = True
    needs_orientation = True


class NTYuOutflow(LBNodeType):
    wet_node = True
    standard_macro = True
    needs_orientation = True


class NTNeumann(LBNodeType):
    wet_node = True
    standard_macro = True
    needs_orientation = True


class NTLaminarize(LBNodeType):
    wet_node = True
    standard_macro = True
    needs_orientation = True


class NTSlip(LBNodeType):
    standard_macro = True


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


# node type -> canonical kernel kind
HIP_KIND = {
    _NTFluid: hipabi.SLF_NK_FLUID,
    _NTGhost: hipabi.SLF_NK_GHOST,
    _NTUnused: hipabi.SLF_NK_UNUSED,
    _NTPropagationOnly: hipabi.SLF_NK_PROPAGATION_ONLY,
    NTFullBBWall: hipabi.SLF_NK_FULL_BB,
    NTHalfBBWall: hipabi.SLF_NK_HALF_BB,
    NTRegularizedVelocity: hipabi.SLF_NK_REGULARIZED_VELOCITY,
    NTEquilibriumDensity: hipabi.SLF_NK_EQUILIBRIUM_DENSITY,
    NTEquilibriumVelocity: hipabi.SLF_NK_EQUILIBRIUM_VELOCITY,
    NTZouHeVelocity: hipabi.SLF_NK_ZOUHE_VELOCITY,
    NTZouHeDensity: hipabi.SLF_NK_ZOUHE_DENSITY,
    NTRegularizedDensity: hipabi.SLF_NK_REGULARIZED_DENSITY,
    NTCopy: hipabi.SLF_NK_COPY,                 # two-copy (AB) access pattern only, like the reference
    NTYuOutflow: hipabi.SLF_NK_YU_OUTFLOW,      # AB only
    NTDoNothing: hipabi.SLF_NK_FLUID,           # AB: a plain fluid node (reference node_type.py:296-307); AA: unsupported
}
