"""Subdomain geometry: where a subdomain sits (SubdomainSpec*) and what its nodes
are (Subdomain*).  Public surface = the reference's sailfish/subdomain.py:

  * user code subclasses Subdomain2D / Subdomain3D and implements
    ``boundary_conditions(hx, hy[, hz])`` (calling ``set_node``) and
    ``initial_conditions(sim, hx, hy[, hz])``;
  * ``reset()`` runs the same pipeline as the reference (subdomain.py:675-732):
    user BCs on the ghost-including index grid -> provisional ghosts ->
    unused / propagation-only detection -> link tags / orientation -> ghosts ->
    encode.

The neighbourhood sums of the post-processing step (scipy.ndimage.convolve in the
reference, subdomain.py:845-869) are restated with numpy shifts so that the GPU
box needs no scipy.  Pinned against node maps produced by the reference itself
(tests/golden/geometry_*.npz).
"""
import inspect
from collections import defaultdict

import numpy as np

from sailfish_amd import node_type as nt
from sailfish_amd import util


class SubdomainSpec(object):
    """Where a subdomain sits in the global lattice and which subdomains touch its faces (the role of reference
    subdomain.py:32-304; attribute names are the ones user code and the runner rely on: location, size, ox.., nx..,
    ex.., end_location, actual_size, envelope_size, id, runner)."""
    dim = None
    # face ids: 2 * axis + (0: low side, 1: high side)
    X_LOW, X_HIGH, Y_LOW, Y_HIGH, Z_LOW, Z_HIGH = range(6)

    def __init__(self, location, size, envelope_size=None, id_=None, *args, **kwargs):
        assert len(location) == len(size) == self.dim
        self.location, self.size = tuple(location), tuple(size)
        self.end_location = [o + n for o, n in zip(location, size)]
        for axis, name in enumerate('xyz'[:self.dim]):          # ox, nx, ex, ... per axis
            setattr(self, 'o' + name, location[axis])
            setattr(self, 'n' + name, size[axis])
            setattr(self, 'e' + name, self.end_location[axis])
        self.actual_size = self.envelope_size = None
        if envelope_size is not None:
            self.set_actual_size(envelope_size)
        self.id = id_
        self.runner = None
        self._periodicity = [False] * self.dim
        self._clear_connections()
        self._clear_connectors()

    def __repr__(self):
        return '%s(%s, %s, id_=%s)' % (type(self).__name__, self.location, self.size, self.id)

    # -- sizes
    def set_actual_size(self, envelope_size):
        """Size including the ghost envelope on both sides of every axis."""
        self.envelope_size = envelope_size
        self.actual_size = [n + 2 * envelope_size for n in self.size]

    @property
    def num_nodes(self):
        return int(np.prod(self.size, dtype=np.int64))

    @property
    def num_actual_nodes(self):
        return int(np.prod(self.actual_size, dtype=np.int64))

    @property
    def _nonghost_slice(self):
        """Index (numpy axis order, x last) of the real nodes inside a ghost-including array."""
        es = self.envelope_size
        return tuple(slice(es, es + n) for n in reversed(self.size))

    # -- periodicity handled inside this subdomain (it spans the whole periodic axis)
    def enable_local_periodicity(self, axis):
        assert 0 <= axis < self.dim
        self._periodicity[axis] = True

    periodic_x = property(lambda self: self._periodicity[0])
    periodic_y = property(lambda self: self._periodicity[1])
    periodic_z = property(lambda self: self.dim == 3 and self._periodicity[2])
    periodic = property(lambda self: any(self._periodicity))

    # -- connections: face -> ids of the subdomains behind it
    def _clear_connections(self):
        self._connections = defaultdict(list)

    def _clear_connectors(self):
        self._connectors = {}

    def _add_connection(self, face, neighbour_id):
        ids = self._connections[face]
        if neighbour_id not in ids:
            ids.append(neighbour_id)

    def add_connector(self, subdomain_id, connector):
        assert subdomain_id not in self._connectors
        self._connectors[subdomain_id] = connector

    def connecting_subdomains(self):
        """[(face, subdomain id)]"""
        return [(face, nid) for face, ids in self._connections.items() for nid in ids]

    def neighbour_ids(self):
        return sorted({nid for ids in self._connections.values() for nid in ids})

    def has_face_conn(self, face):
        return bool(self._connections.get(face))

    # -- face algebra
    @staticmethod
    def face_to_axis(face):
        return face >> 1

    @staticmethod
    def face_to_dir(face):
        return 1 if face & 1 else -1

    @staticmethod
    def axis_dir_to_face(axis, dir_):
        return 2 * axis + int(dir_ > 0)

    @staticmethod
    def opposite_face(face):
        return face ^ 1

    def face_to_normal(self, face):
        normal = [0] * self.dim
        normal[face >> 1] = 1 if face & 1 else -1
        return normal

    def contains_global(self, pos):
        """Is the global node position a real node of this subdomain?"""
        return all(o <= p < e for p, o, e in zip(pos, self.location, self.end_location))


class SubdomainSpec2D(SubdomainSpec):
    dim = 2


class SubdomainSpec3D(SubdomainSpec):
    dim = 3


def _neighbour_sum(arr, kernel_offsets, cval):
    """sum_k arr[x + k] over the given offsets with constant padding (the effect of
    scipy.ndimage.convolve(arr, symmetric_kernel, mode='constant', cval=cval))."""
    pad = np.pad(arr.astype(np.int32), 1, mode='constant', constant_values=cval)
    out = np.zeros(arr.shape, dtype=np.int32)
    core = tuple(slice(1, 1 + n) for n in arr.shape)
    for off in kernel_offsets:
        sl = tuple(slice(c.start + o, c.stop + o) for c, o in zip(core, off))
        out += pad[sl]
    return out


class Subdomain(object):
    """Field and geometry information of one subdomain (reference subdomain.py:350-869)."""

    @classmethod
    def add_options(cls, group):
        pass

    def __init__(self, grid_shape, spec, grid, *args, **kwargs):
        """grid_shape: global lattice size, x last; spec: SubdomainSpec; grid: lattice class."""
        self.spec, self.grid, self.grid_shape = spec, grid, grid_shape
        # state of the node-map pipeline (set_node -> reset -> encoder)
        self._params = {}                 # parameter key -> LBNodeType instance, in the order of the set_node calls
        self._seen_types = {0}
        self._needs_orientation = False
        self._type_map_encoded = False
        self._encoder = None
        self._type_vis_map = np.zeros(self.lat_shape, dtype=np.uint8)     # node types of the real nodes, for output
        # indirect addressing only: dense Boolean array over all nodes incl. ghosts, True = the node takes part in
        # the simulation and owns a slot in the distribution arrays
        self.active_node_mask = None
        if self.config.node_addressing == 'indirect':
            self.load_active_node_map(*self._get_mgrid_base(self.config))
            fill = 100.0 * self.active_nodes / self.spec.num_actual_nodes
            self.config.logger.info('Fill ratio is: %0.2f%%' % fill)

    def allocate(self):
        """Host arrays of the node map pipeline, in the runner's padded layout: node types (with a ghost-including
        view), parameter keys and orientation codes; `<name>_base` is the whole padded array behind a view."""
        runner = self.spec.runner
        for name, dtype, ghosts in (('_type_map_ghost', np.uint32, True), ('_param_map', np.int64, False),
                                    ('_orientation', np.uint32, False)):
            view, _ = runner.make_scalar_field(dtype, register=False, nonghost_view=not ghosts)
            setattr(self, name, view)
            setattr(self, (name[:-len('_ghost')] if ghosts else name) + '_base', runner.field_base(view))
        self._type_map = self._type_map_ghost[self.spec._nonghost_slice]

    @property
    def config(self):
        return self.spec.runner.config

    @property
    def lat_shape(self):
        return list(reversed(self.spec.size))

    @property
    def full_lat_shape(self):
        return list(reversed(self.spec.actual_size))

    def boundary_conditions(self, *args):
        raise NotImplementedError('boundary_conditions() not defined in a child class.')

    def initial_conditions(self, sim, *args):
        raise NotImplementedError('initial_conditions() not defined in a child class')

    def select_subdomain(self, array, hx, hy, *args):
        es = self.spec.envelope_size
        if self.dim == 2:
            return array[hy + es, hx + es]
        return array[args[0] + es, hy + es, hx + es]

    def load_active_node_map(self, *args):
        """Sets active_node_mask (indirect addressing); override in a subclass, typically through
        set_active_node_map_from_wall_map().  Default: every node is active (reference subdomain.py:425-435)."""
        self.active_node_mask = np.ones(self.full_lat_shape, dtype=bool)
        self.config.logger.warning('Using indirect addressing with all nodes active. Consider '
                                   '--node_addressing=direct for better performance.')

    def set_active_node_map_from_wall_map(self, wall_map):
        """wall_map: Boolean array over all nodes incl. ghosts, True = solid.  Fluid nodes and every node
        with a fluid neighbour (one layer of wall / ghost nodes) are active (reference subdomain.py:452-474;
        the neighbourhood wraps around the array like its scipy 'wrap' convolution)."""
        fluid = np.logical_not(np.asarray(wall_map, dtype=bool))
        near = np.zeros(fluid.shape, dtype=bool)
        for off in self._neighbour_offsets():
            if not any(off):
                continue
            near |= np.roll(fluid, shift=tuple(-o for o in off), axis=tuple(range(fluid.ndim)))
        self.active_node_mask = fluid | near

    @property
    def active_nodes(self):
        if self.active_node_mask is not None:
            return int(np.sum(self.active_node_mask))
        return int(np.prod(self.lat_shape, dtype=np.int64))

    @util.lazy_property
    def num_fluid_nodes(self):
        return int(np.sum(self.fluid_map()))

    # -- node assignment ---------------------------------------------------------------
    @staticmethod
    def _check_param(name, value, n_selected):
        """A node parameter is a number, a tuple of numbers, or one value per selected node (ndarray)."""
        if util.is_number(value):
            return
        if isinstance(value, np.ndarray):
            if value.size != n_selected:
                raise AssertionError("A per-node parameter array needs exactly as many entries as there are True "
                                     "values in the 'where' array; node_type.multifield() builds such arrays.")
            return
        if type(value) is tuple:
            if all(util.is_number(el) for el in value):
                return
            raise ValueError('Tuple elements have to be numbers.')
        if isinstance(value, nt.DynamicValue):      # expressions of position / time, evaluated on the host (node_type.py)
            return
        raise ValueError('Unrecognized node param: {0} (type {1})'.format(name, type(value)))

    @staticmethod
    def _param_key(node_type):
        """One integer per distinct (node type, parameter values) combination: what the parameter map stores for
        the selected nodes until the encoder turns it into indices of the parameter table."""
        items = frozenset((name, value.tobytes() if isinstance(value, np.ndarray) else value)
                          for name, value in node_type.params.items())
        return hash((node_type.id, items))

    def set_node(self, where, node_type):
        """Makes the nodes selected by the Boolean array `where` (ghost-including index grid) nodes of `node_type`
        (a node_type.LBNodeType class or instance).  A node can be set once."""
        if self._type_map_encoded:
            raise AssertionError('the node map has been encoded already')
        if inspect.isclass(node_type):
            node_type = node_type()
        if not isinstance(node_type, nt.LBNodeType):
            raise AssertionError('node_type must be an LBNodeType class or instance')
        mask = np.asarray(where)
        selected = np.nonzero(mask)
        n_selected = int(np.count_nonzero(mask))
        for name, value in node_type.params.items():
            self._check_param(name, value, n_selected)
            if isinstance(value, nt.DynamicValue):          # reference subdomain.py:509-515
                if value.time_dependent():
                    self.config.time_dependence = True
                if value.space_dependent():
                    self.config.space_dependence = True
        if np.any(self._param_map_base[selected]):
            raise AssertionError('Overriding previously set nodes is not allowed.')
        key = self._param_key(node_type)
        self._params[key] = node_type
        self._seen_types.add(node_type.id)
        self._type_map_base[selected] = node_type.id
        self._param_map_base[selected] = key
        fixed = getattr(node_type, 'orientation', None)
        if fixed is not None:
            self._orientation_base[selected] = fixed
        elif node_type.needs_orientation:
            self._needs_orientation = True

    # -- orientation -----------------------------------------------------------------------
    def _present(self, type_ids):
        """Those of `type_ids` that occur in the node map."""
        return sorted(set(type_ids) & set(int(t) for t in np.unique(self._type_map_base)))

    def _at_neighbour(self, arr, vec):
        """arr[x + vec] for every node x (vec in lattice order x, y[, z]); wraps around the array, like the
        reference's np.roll-based detection."""
        shift = tuple(-int(c) for c in reversed(vec))
        return np.roll(arr, shift, axis=tuple(range(arr.ndim)))

    def tag_directions(self):
        """Link tags for the node types that use them: bit i - 1 of the orientation field is set when lattice
        direction i points from the node to a wet node.  Axes that are not locally periodic include their ghost
        layers (a wall node next to a connected face sees the neighbour's nodes).  Returns False when no such
        node type is present."""
        tagged_types = self._present(nt.get_link_tag_node_type_ids())
        if not tagged_types:
            return False
        window = tuple(ng if periodic else slice(None)
                       for ng, periodic in zip(self.spec._nonghost_slice, reversed(self.spec._periodicity)))
        types = self._type_map_base[window]
        tags = self._orientation_base[window]            # a view: updated in place
        todo = util.in_anyd_fast(types, tagged_types) & (tags == 0)
        wet = util.in_anyd_fast(types, self._present(nt.get_wet_node_type_ids()))
        for bit, vec in enumerate(self.grid.basis[1:]):
            tags[todo & self._at_neighbour(wet, vec)] |= np.uint32(1 << bit)
        return True

    def detect_orientation(self, use_tags):
        """Orientation code of the remaining orientation-aware node types: the axis-aligned lattice direction
        that points to a plain fluid node (first match in basis order)."""
        wanted = set(nt.get_orientation_node_type_ids())
        if use_tags:
            wanted -= set(nt.get_link_tag_node_type_ids())
        wanted = self._present(wanted)
        if not wanted:
            return
        candidates = util.in_anyd_fast(self._type_map_base, wanted)
        fluid = self._type_map_base == 0
        for vec in self.grid.basis:
            if sum(abs(int(c)) for c in vec) != 1:
                continue
            hit = candidates & self._at_neighbour(fluid, vec) & (self._orientation_base == 0)
            self._orientation_base[hit] = self.grid.vec_to_dir(list(vec))

    # -- pipeline ----------------------------------------------------------------------------
    def reset(self, encode=True):
        """Builds the node map: user boundary conditions on the ghost-including index grid, provisional ghosts,
        unused / propagation-only nodes, orientation (link tags first where allowed), final ghosts, encoder."""
        cfg = self.config
        self._type_map_encoded = False
        self.boundary_conditions(*self._get_mgrid_base(cfg))
        self._define_ghosts(unset_only=True)
        self._postprocess_nodes()
        tagged = False
        if self._needs_orientation:
            tagged = bool(cfg.use_link_tags) and self.tag_directions()
            self.detect_orientation(cfg.use_link_tags)
        self._define_ghosts()
        np.copyto(self._type_vis_map, self._type_map)
        from sailfish_amd import geo_encoder
        self._encoder = geo_encoder.GeoEncoderConst(self)
        self._encoder.prepare_encode(self._type_map_base, self._param_map_base, self._params,
                                     self._orientation_base, tagged)
        if encode:
            self.encoded_map()
        cfg.logger.info('Fluid node fraction: %.1f%%' % (100.0 * self.num_fluid_nodes / self.spec.num_nodes))

    @property
    def scratch_space_size(self):
        return 0

    def init_fields(self, sim):
        self.initial_conditions(sim, *self._get_mgrid())

    def update_context(self, ctx):
        assert self._encoder is not None
        self._encoder.update_context(ctx)

    def encoded_map(self, indirect_address=None):
        if not self._type_map_encoded:
            self._encoder.encode(self._orientation_base)
            self._type_map_encoded = True
        return self._type_map_base

    def visualization_map(self):
        return self._type_vis_map

    def fluid_map(self, wet=True):
        fm = self.visualization_map()
        if wet:
            uniq_types = set(int(x) for x in np.unique(fm))
            return util.in_anyd_fast(fm, list(set(nt.get_wet_node_type_ids()) & uniq_types))
        return fm == 0

    def _fluid_map(self, wet=True, base=True, allow_unused=None):
        assert not self._type_map_encoded
        src = self._type_map_base if base else self._type_map
        if wet:
            uniq_types = set(int(x) for x in np.unique(src))
            return util.in_anyd_fast(src, list(set(nt.get_wet_node_type_ids(allow_unused=allow_unused)) & uniq_types))
        return src == 0

    def _neighbour_offsets(self):
        """Offsets (numpy axis order) of the lattice neighbourhood, centre included
        (reference _lattice_kernel, subdomain.py:930-935 / 1007-1012)."""
        return [tuple(reversed(e)) for e in self.grid.basis]

    def _postprocess_nodes(self):
        """Unused / propagation-only detection, reference subdomain.py:845-869."""
        fluid_map = self._fluid_map(wet=False, base=True)
        wet_map_for_unused = self._fluid_map(wet=True, allow_unused=True, base=True)
        wet_map = self._fluid_map(wet=True, base=True)
        offs = self._neighbour_offsets()
        # wet nodes (that allow it) without any fluid neighbour -> unused
        where = _neighbour_sum(fluid_map, offs, 1) == 0
        self._type_map_base[where & wet_map_for_unused] = nt._NTUnused.id
        # dry nodes without any wet neighbour -> unused
        where = _neighbour_sum(wet_map, offs, 0) == 0
        self._type_map_base[where & np.logical_not(wet_map)] = nt._NTUnused.id
        # unused nodes touching a used node -> propagation only
        used_map = self._type_map_base != nt._NTUnused.id
        where = _neighbour_sum(used_map, offs, 0) > 0
        self._type_map_base[where & (self._type_map_base == nt._NTUnused.id)] = nt._NTPropagationOnly.id

    # -- index grids -----------------------------------------------------------------------------
    def _index_grids(self, envelope):
        """Global coordinates (hx, hy[, hz]) of the nodes of this subdomain, `envelope` ghost layers included."""
        shape = [n + 2 * envelope for n in reversed(self.spec.size)]
        grids = np.indices(shape)
        return [grids[self.dim - 1 - axis] + (self.spec.location[axis] - envelope) for axis in range(self.dim)]

    def _get_mgrid(self):
        """Index arrays of the real nodes (what initial_conditions() receives)."""
        return self._index_grids(0)

    def _get_mgrid_base(self, config):
        """Index arrays including the ghost layers (what boundary_conditions() receives); along a globally
        periodic axis the ghost layers carry the coordinates of the nodes they mirror."""
        grids = self._index_grids(self.spec.envelope_size)
        periodic = [config.periodic_x, config.periodic_y] + ([config.periodic_z] if self.dim == 3 else [])
        for axis, g in enumerate(reversed(self.grid_shape)):
            if periodic[axis]:
                grids[axis] %= g
        return grids

    # -- ghosts --------------------------------------------------------------------------------
    def _embed(self, lat_mask):
        """Lattice-shaped (ghost-including) boolean array -> padded array shape."""
        out = np.zeros(self._type_map_base.shape, dtype=bool)
        out[tuple(slice(0, n) for n in lat_mask.shape)] = lat_mask
        return out

    def _ghost_owned_elsewhere(self):
        """Ghost nodes whose global position is a real node of a *neighbouring* subdomain (they carry
        halo data and, for orientation detection, stand for that neighbour's nodes -- reference
        _define_ghosts(unset_only=True), subdomain.py:901-929)."""
        runner = self.spec.runner
        owner = getattr(runner, 'ghost_owner_map', None)
        if owner is None:
            return np.zeros(self._type_map_base.shape, dtype=bool)
        return self._embed(owner(self.spec))

    def _define_ghosts(self, unset_only=False):
        """Marks the envelope (and the x padding) as ghost nodes.  With unset_only (first pass, before
        unused-node and orientation detection) the ghost layer of a face that is connected to another
        subdomain keeps (i) explicitly set boundary nodes and (ii) the nodes that stand for the
        neighbour's real nodes (reference subdomain.py:901-929 / 973-1006)."""
        assert not self._type_map_encoded
        es = self.spec.envelope_size
        if not es:
            return
        ghost = np.ones(self._type_map_base.shape, dtype=bool)
        ghost[self.spec._nonghost_slice] = False
        if unset_only:
            connected, unconnected = self._face_masks()
            halo = self._ghost_owned_elsewhere()
            keep = (halo | (self._type_map_base != 0)) & connected & ~unconnected
            ghost &= ~keep
            # The reference keeps whole rows of the x-padding "fluid" wherever the high-x ghost column
            # stands for a neighbour (its src_slice selects rows, subdomain.py:905-921); padding is never
            # computed on, this only reproduces the orientation bits it leaves on ghost / wall nodes.
            lat_nx = self.full_lat_shape[-1]
            if self._type_map_base.shape[-1] > lat_nx and self.spec.has_face_conn(self.spec.X_HIGH):
                rows = halo[..., lat_nx - 1:lat_nx]
                ghost[..., lat_nx:] &= ~np.broadcast_to(rows, ghost[..., lat_nx:].shape)
        self._type_map_base[ghost] = nt._NTGhost.id

    def _face_masks(self):
        """(ghost layers of faces connected to another subdomain, ghost layers of all other faces)."""
        shape = self._type_map_base.shape
        conn = np.zeros(shape, dtype=bool)
        unconn = np.zeros(shape, dtype=bool)
        es = self.spec.envelope_size
        lat = self.full_lat_shape
        for face in range(2 * self.dim):
            axis = self.spec.face_to_axis(face)
            ax = self.dim - 1 - axis
            sl = [slice(None)] * self.dim
            sl[ax] = slice(0, es) if self.spec.face_to_dir(face) < 0 else slice(lat[ax] - es, None)
            (conn if self.spec.has_face_conn(face) else unconn)[tuple(sl)] = True
        return conn, unconn


class Subdomain2D(Subdomain):
    dim = 2

    def __init__(self, grid_shape, spec, *args, **kwargs):
        self.gy, self.gx = grid_shape
        Subdomain.__init__(self, grid_shape, spec, *args, **kwargs)


class Subdomain3D(Subdomain):
    dim = 3

    def __init__(self, grid_shape, spec, *args, **kwargs):
        self.gz, self.gy, self.gx = grid_shape
        Subdomain.__init__(self, grid_shape, spec, *args, **kwargs)
