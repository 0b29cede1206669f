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
import operator
from collections import defaultdict
from functools import reduce

import numpy as np

from sailfish_amd import node_type as nt
from sailfish_amd import util


class SubdomainSpec(object):
    """Location of a subdomain and its links to other subdomains (reference subdomain.py:32-304)."""
    dim = None

    # Face IDs.
    X_LOW = 0
    X_HIGH = 1
    Y_LOW = 2
    Y_HIGH = 3
    Z_LOW = 4
    Z_HIGH = 5

    def __init__(self, location, size, envelope_size=None, id_=None, *args, **kwargs):
        self.location = tuple(location)
        self.size = tuple(size)
        if envelope_size is not None:
            self.set_actual_size(envelope_size)
        else:
            self.actual_size = None
            self.envelope_size = None
        self._runner = None
        self._id = id_
        self._clear_connections()
        self._clear_connectors()
        self._periodicity = [False] * self.dim

    def __repr__(self):
        return '{0}({1}, {2}, id_={3})'.format(self.__class__.__name__, self.location, self.size, self._id)

    @property
    def runner(self):
        return self._runner

    @runner.setter
    def runner(self, x):
        self._runner = x

    @property
    def id(self):
        return self._id

    @id.setter
    def id(self, x):
        self._id = x

    @property
    def num_nodes(self):
        return reduce(operator.mul, self.size)

    @property
    def num_actual_nodes(self):
        return reduce(operator.mul, self.actual_size)

    @property
    def periodic_x(self):
        return self._periodicity[0]

    @property
    def periodic_y(self):
        return self._periodicity[1]

    @property
    def periodic_z(self):
        return self.dim == 3 and self._periodicity[2]

    @property
    def periodic(self):
        return any(self._periodicity)

    def enable_local_periodicity(self, axis):
        """The subdomain spans the whole (periodic) axis: PBC is applied inside it."""
        assert axis <= self.dim - 1
        self._periodicity[axis] = True

    # -- connections: face -> set of neighbour subdomain ids -------------------------
    def _clear_connections(self):
        self._connections = defaultdict(list)

    def _clear_connectors(self):
        self._connectors = {}

    def _add_connection(self, face, neighbour_id):
        if neighbour_id not in self._connections[face]:
            self._connections[face].append(neighbour_id)

    def add_connector(self, subdomain_id, connector):
        assert subdomain_id not in self._connectors
        self._connectors[subdomain_id] = connector

    def connecting_subdomains(self):
        """List of (face, subdomain id) pairs."""
        return [(face, nid) for face, ids in self._connections.items() for nid in ids]

    def neighbour_ids(self):
        return sorted(set(nid for ids in self._connections.values() for nid in ids))

    def has_face_conn(self, face):
        return len(self._connections.get(face, ())) > 0

    def set_actual_size(self, envelope_size):
        self.actual_size = [x + 2 * envelope_size for x in self.size]
        self.envelope_size = envelope_size

    @classmethod
    def face_to_dir(cls, face):
        return -1 if face in (cls.X_LOW, cls.Y_LOW, cls.Z_LOW) else 1

    @classmethod
    def face_to_axis(cls, face):
        return face // 2

    def face_to_normal(self, face):
        direction = [0] * self.dim
        direction[self.face_to_axis(face)] = self.face_to_dir(face)
        return direction

    def opposite_face(self, face):
        return face ^ 1

    @classmethod
    def axis_dir_to_face(cls, axis, dir_):
        return 2 * axis + (1 if dir_ > 0 else 0)

    def contains_global(self, pos):
        """Is the global node position inside this subdomain's real region?"""
        return all(o <= p < o + n for p, o, n in zip(pos, self.location, self.size))


class SubdomainSpec2D(SubdomainSpec):
    dim = 2

    def __init__(self, location, size, envelope_size=None, *args, **kwargs):
        self.ox, self.oy = location
        self.nx, self.ny = size
        self.ex = self.ox + self.nx
        self.ey = self.oy + self.ny
        self.end_location = [self.ex, self.ey]
        SubdomainSpec.__init__(self, location, size, envelope_size, *args, **kwargs)

    @property
    def _nonghost_slice(self):
        es = self.envelope_size
        return (slice(es, es + self.ny), slice(es, es + self.nx))


class SubdomainSpec3D(SubdomainSpec):
    dim = 3

    def __init__(self, location, size, envelope_size=None, *args, **kwargs):
        self.ox, self.oy, self.oz = location
        self.nx, self.ny, self.nz = size
        self.ex = self.ox + self.nx
        self.ey = self.oy + self.ny
        self.ez = self.oz + self.nz
        self.end_location = [self.ex, self.ey, self.ez]
        SubdomainSpec.__init__(self, location, size, envelope_size, *args, **kwargs)

    @property
    def _nonghost_slice(self):
        es = self.envelope_size
        return (slice(es, es + self.nz), slice(es, es + self.ny), slice(es, es + self.nx))


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
        self.spec = spec
        self.grid_shape = grid_shape
        self.grid = grid
        self._type_vis_map = np.zeros(self.lat_shape, dtype=np.uint8)
        self._type_map_encoded = False
        self._params = {}
        self._encoder = None
        self._seen_types = set([0])
        self._needs_orientation = False
        # indirect addressing only: dense boolean array over all nodes incl. ghosts, True = the node takes
        # part in the simulation and owns a slot in the distribution arrays (reference subdomain.py:385-394)
        self.active_node_mask = None
        if self.config.node_addressing == 'indirect':
            self.load_active_node_map(*self._get_mgrid_base(self.config))
            self.config.logger.info('Fill ratio is: %0.2f%%' %
                                    (self.active_nodes / float(self.spec.num_actual_nodes) * 100))

    def allocate(self):
        runner = self.spec.runner
        self._type_map_ghost, _ = runner.make_scalar_field(np.uint32, register=False, nonghost_view=False)
        self._type_map = self._type_map_ghost[self.spec._nonghost_slice]
        self._type_map_base = runner.field_base(self._type_map_ghost)
        self._param_map, _ = runner.make_scalar_field(dtype=np.int64, register=False)
        self._param_map_base = runner.field_base(self._param_map)
        self._orientation, _ = runner.make_scalar_field(np.uint32, register=False)
        self._orientation_base = runner.field_base(self._orientation)

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
        return reduce(operator.mul, self.lat_shape)

    @util.lazy_property
    def num_fluid_nodes(self):
        return int(np.sum(self.fluid_map()))

    # -- node assignment ---------------------------------------------------------------
    def _verify_params(self, where, node_type):
        for name, param in node_type.params.items():
            if util.is_number(param):
                continue
            elif type(param) is tuple:
                for el in param:
                    if not util.is_number(el):
                        raise ValueError('Tuple elements have to be numbers.')
            elif isinstance(param, np.ndarray):
                assert param.size == np.sum(where), (
                    "Your array needs to have exactly as many nodes as there are True values in the "
                    "'where' array.  Use node_type.multifield() to generate the array in an easy way.")
            else:
                raise ValueError('Unrecognized node param: {0} (type {1})'.format(name, type(param)))

    @staticmethod
    def _hashable_params(param_dict):
        params = []
        for k, v in param_dict.items():
            params.append((k, v.tobytes()) if hasattr(v, 'tobytes') else (k, v))
        return frozenset(params)

    def set_node(self, where, node_type):
        """Set a boundary condition at the selected nodes (reference subdomain.py:532-559)."""
        where_array = where
        where = np.where(where)
        assert not self._type_map_encoded
        if inspect.isclass(node_type):
            assert issubclass(node_type, nt.LBNodeType)
            node_type = node_type()
        else:
            assert isinstance(node_type, nt.LBNodeType)
        self._verify_params(where_array, node_type)
        self._type_map_base[where] = node_type.id
        key = hash((node_type.id, self._hashable_params(node_type.params)))
        assert np.all(self._param_map_base[where] == 0), 'Overriding previously set nodes is not allowed.'
        self._param_map_base[where] = key
        self._params[key] = node_type
        self._seen_types.add(node_type.id)
        if getattr(node_type, 'orientation', None) is not None:
            self._orientation_base[where] = node_type.orientation
        elif node_type.needs_orientation:
            self._needs_orientation = True

    # -- orientation -----------------------------------------------------------------------
    def tag_directions(self):
        """Link tags: bit i-1 set <=> direction i points to a wet node (reference subdomain.py:593-642)."""
        ngs = list(self.spec._nonghost_slice)
        for i, periodic in enumerate(reversed(self.spec._periodicity)):
            if not periodic:
                ngs[i] = slice(None)
        ngs = tuple(ngs)
        uniq_types = set(int(x) for x in np.unique(self._type_map_base))
        wet_types = list(set(nt.get_wet_node_type_ids()) & uniq_types)
        orient_types = list(set(nt.get_link_tag_node_type_ids()) & uniq_types)
        if not orient_types:
            return False
        orient_map = (util.in_anyd_fast(self._type_map_base[ngs], orient_types) &
                      (self._orientation_base[ngs] == 0))
        l = self.grid.dim - 1
        for i, vec in enumerate(self.grid.basis[1:]):
            shifted_map = self._type_map_base[ngs]
            for j, shift in enumerate(vec):
                if shift == 0:
                    continue
                shifted_map = np.roll(shifted_map, int(-shift), axis=l - j)
            idx = orient_map & util.in_anyd_fast(shifted_map, wet_types)
            self._orientation_base[ngs][idx] |= np.uint32(1 << i)
        return True

    def detect_orientation(self, use_tags):
        """Primary-direction orientation = the axis-aligned vector pointing to a fluid node
        (reference subdomain.py:644-673)."""
        uniq_types = set(int(x) for x in np.unique(self._type_map_base))
        orient_types = list((set(nt.get_orientation_node_type_ids()) -
                             set(nt.get_link_tag_node_type_ids() if use_tags else [])) & uniq_types)
        if not orient_types:
            return
        orient_map = util.in_anyd_fast(self._type_map_base, orient_types)
        l = self.grid.dim - 1
        for vec in self.grid.basis:
            if sum(c * c for c in vec) != 1:
                continue
            shifted_map = self._type_map_base
            for j, shift in enumerate(vec):
                if shift == 0:
                    continue
                shifted_map = np.roll(shifted_map, int(-shift), axis=l - j)
            idx = orient_map & (shifted_map == 0) & (self._orientation_base == 0)
            self._orientation_base[idx] = self.grid.vec_to_dir(list(vec))

    # -- pipeline ----------------------------------------------------------------------------
    def reset(self, encode=True):
        self._type_map_encoded = False
        self.boundary_conditions(*self._get_mgrid_base(self.config))
        have_link_tags = False
        self._define_ghosts(unset_only=True)
        self._postprocess_nodes()
        if self._needs_orientation:
            if self.config.use_link_tags:
                have_link_tags = self.tag_directions()
            self.detect_orientation(self.config.use_link_tags)
        self._define_ghosts()
        self._type_vis_map[:] = self._type_map[:]
        from sailfish_amd import geo_encoder
        self._encoder = geo_encoder.GeoEncoderConst(self)
        self._encoder.prepare_encode(self._type_map_base, self._param_map_base, self._params,
                                     self._orientation_base, have_link_tags)
        if encode:
            self.encoded_map()
        self.config.logger.info('Fluid node fraction: %.1f%%' %
                                (self.num_fluid_nodes * 100.0 / self.spec.num_nodes))

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

    def _get_mgrid(self):
        """Index arrays (x, y) of the non-ghost nodes in global coordinates."""
        return reversed(np.mgrid[self.spec.oy:self.spec.oy + self.spec.ny,
                                 self.spec.ox:self.spec.ox + self.spec.nx])

    def _get_mgrid_base(self, config):
        """Index arrays including ghosts; wrapped for globally periodic axes (reference subdomain.py:885-899)."""
        es = self.spec.envelope_size
        ox, oy = self.spec.ox - es, self.spec.oy - es
        hx, hy = reversed(np.mgrid[oy:oy + self.spec.ny + 2 * es, ox:ox + self.spec.nx + 2 * es])
        if config.periodic_x:
            hx[hx < 0] += self.gx
            hx[hx >= self.gx] -= self.gx
        if config.periodic_y:
            hy[hy < 0] += self.gy
            hy[hy >= self.gy] -= self.gy
        return hx, hy


class Subdomain3D(Subdomain):
    dim = 3

    def __init__(self, grid_shape, spec, *args, **kwargs):
        self.gz, self.gy, self.gx = grid_shape
        Subdomain.__init__(self, grid_shape, spec, *args, **kwargs)

    def _get_mgrid(self):
        return reversed(np.mgrid[self.spec.oz:self.spec.oz + self.spec.nz,
                                 self.spec.oy:self.spec.oy + self.spec.ny,
                                 self.spec.ox:self.spec.ox + self.spec.nx])

    def _get_mgrid_base(self, config):
        es = self.spec.envelope_size
        ox, oy, oz = self.spec.ox - es, self.spec.oy - es, self.spec.oz - es
        hx, hy, hz = reversed(np.mgrid[oz:oz + self.spec.nz + 2 * es,
                                       oy:oy + self.spec.ny + 2 * es,
                                       ox:ox + self.spec.nx + 2 * es])
        if config.periodic_x:
            hx[hx < 0] += self.gx
            hx[hx >= self.gx] -= self.gx
        if config.periodic_y:
            hy[hy < 0] += self.gy
            hy[hy >= self.gy] -= self.gy
        if config.periodic_z:
            hz[hz < 0] += self.gz
            hz[hz >= self.gz] -= self.gz
        return hx, hy, hz
