"""Packs node type, BC parameter index and orientation into one uint32 per node.

Bit layout and dense type-id renumbering are the reference's
(sailfish/geo_encoder.py:76-153, 341-382):

    orientation | scratch_index | param_index | node_type
                                               ^ nt_misc_shift bits
                                 ^ nt_param_shift bits
                 ^ nt_scratch_shift bits (always 0 here: no scratch-space node types)

so encoded maps are bit-identical to the reference's for the supported types
(pinned by tests/golden/geometry_*.npz).  Instead of Mako template variables the
encoder fills a module descriptor (`fill_desc`) for the pre-built kernels.
"""
import numpy as np

from sailfish_amd import node_type as nt
from sailfish_amd import util


class _ParamTable(object):
    """Flat table of boundary-condition parameter values; equal values (numbers or tuples) share one entry.
    The table index of an entry is what the node code stores."""

    def __init__(self):
        self.values = []
        self._where = {}

    def index_of(self, value):
        """value: a number or a tuple of numbers.  Returns the index of its first element in the table."""
        if value not in self._where:
            self._where[value] = len(self.values)
            self.values.extend(float(v) for v in (value if isinstance(value, tuple) else (value,)))
        return self._where[value]

    def append(self, rows):
        """Entries of their own for every row of `rows` [n, components] (values that change with time: equal values now
        need not stay equal).  Returns the index of the first one."""
        first = len(self.values)
        self.values.extend(float(v) for v in np.asarray(rows, dtype=np.float64).ravel())
        return first


class GeoEncoderConst(object):
    """Node map -> one uint32 per node, parameters -> one flat table (the role of reference geo_encoder.py:76-382;
    the bit layout, the dense renumbering of the types in use and the order of the parameter table are part of the
    contract with the reference's node maps, see the module docstring)."""

    def __init__(self, subdomain):
        self.subdomain = subdomain
        self.dim = subdomain.dim
        self.config = subdomain.config
        self._type_id_remap = {0: 0}
        self._node_types = set([nt._NTFluid])
        self._bits_type = self._bits_param = self._bits_scratch = 0
        self._type_map = None
        self._geo_params = []
        self.scratch_space_size = 0
        self._unused_tag_bits = 0
        self._have_link_tags = False
        self._dynamic = []        # (first table index, DynamicValue, node coordinates) of the time-dependent parameters

    def _type_id(self, node_type):
        return self._type_id_remap.get(node_type, 0xffffffff)

    # -- step 1: what occurs in the map decides the field widths
    def _renumber_types(self, type_map):
        """Types in use get the dense ids 1, 2, ... in ascending order of their global ids."""
        present = [int(t) for t in np.unique(type_map)]
        self._node_types.update(nt._NODE_TYPES[t] for t in present)
        self._type_id_remap.update((t, k + 1) for k, t in enumerate(present))
        self._bits_type = util.bit_len(len(present))
        for cls in self._node_types:
            if cls.scratch_space_size(self.dim) > 0:
                raise NotImplementedError('%s needs node scratch space, which the HIP backend does not provide'
                                          % cls.__name__)
        self._bits_scratch = 0

    def _index_params(self, param_map, param_dict):
        """Parameter-table index of every node (0 where a node has no parameter)."""
        table = _ParamTable()
        index_map = np.zeros(param_map.shape, dtype=np.uint32)
        for key, node_type in param_dict.items():              # in the order the nodes were set
            nodes = param_map == key
            for value in node_type.params.values():
                if util.is_number(value) or type(value) is tuple:
                    index_map[nodes] = table.index_of(value)
                elif isinstance(value, np.ndarray):            # one value (row) per selected node, in C order
                    where = np.argwhere(nodes)
                    for v in np.unique(value):
                        entry = tuple(v) if hasattr(v, '__len__') else (float(v),)
                        sel = where[value == v]
                        index_map[tuple(sel[:, k] for k in range(sel.shape[1]))] = table.index_of(entry)
                elif isinstance(value, nt.DynamicValue):
                    # expressions of position / time (reference: device code, node_type.py:471-570): evaluated here.
                    # Constant in time: one table entry per distinct value, like a per-node array.  Time-dependent: an
                    # entry of its own per node (per value, if it does not depend on position) that the runner rewrites
                    # before every step (dynamic_updates()).
                    where = np.argwhere(nodes)
                    coords = self._global_coords(where)
                    dt = getattr(self.config, 'dt_per_lattice_time_unit', 1.0)
                    ncomp = len(value)
                    if not value.time_dependent():
                        vals = value.evaluate(coords, 0, dt)
                        uniq, inv = np.unique(vals, axis=0, return_inverse=True)
                        inv = np.asarray(inv).ravel()
                        for u, row in enumerate(uniq):
                            sel = where[inv == u]
                            index_map[tuple(sel[:, k] for k in range(sel.shape[1]))] = table.index_of(tuple(float(x) for x in row))
                    elif value.space_dependent():
                        first = table.append(value.evaluate(coords, 0, dt))
                        index_map[tuple(where[:, k] for k in range(where.shape[1]))] = first + ncomp * np.arange(len(where), dtype=np.uint32)
                        self._dynamic.append((first, value, coords))
                    else:
                        one = tuple(c[:1] for c in coords)
                        first = table.append(value.evaluate(one, 0, dt))
                        index_map[nodes] = first
                        self._dynamic.append((first, value, one))
                else:
                    raise ValueError('unsupported node parameter type for the HIP backend: %r' % type(value))
        self._geo_params = table.values
        self._bits_param = util.bit_len(len(table.values))
        return index_map

    def _global_coords(self, where):
        """(gx, gy[, gz]) of the nodes at the array indices `where` ([n, dim], C order, ghost layers included)."""
        spec = self.subdomain.spec
        env = spec.envelope_size
        return tuple(where[:, self.dim - 1 - axis].astype(np.float64) + (spec.location[axis] - env) for axis in range(self.dim))

    def dynamic_updates(self, iteration):
        """[(first table index, values)] of the parameters that depend on time, at LB iteration `iteration`."""
        dt = getattr(self.config, 'dt_per_lattice_time_unit', 1.0)
        return [(first, value.evaluate(coords, iteration, dt).ravel()) for first, value, coords in self._dynamic]

    @property
    def time_dependent(self):
        return bool(self._dynamic)

    def prepare_encode(self, type_map, param_map, param_dict, orientation, have_link_tags):
        """type_map: node type ids (overwritten by encode()); param_map: per-node keys into param_dict
        ({key: LBNodeType instance}); orientation: orientation codes or link tags."""
        self._renumber_types(type_map)
        self._type_map = type_map
        self._encoded_param_map = self._index_params(param_map, param_dict)
        self._have_link_tags = bool(have_link_tags)
        if have_link_tags:
            tags = orientation[orientation > 0]
            self._unused_tag_bits = int(np.bitwise_and.reduce(tags)) if tags.size else 0
            if self.config.use_link_tags and 32 - self._bits_type - self._bits_param - self._bits_scratch < \
                    self.subdomain.grid.Q - 1:
                raise ValueError('Not enough bits available to tag neighbor nodes.')

    # -- step 2: pack  orientation | scratch | param | type
    def _pack(self, orientation, param, dense_type, scratch=0):
        code = np.asarray(orientation, dtype=np.uint32)
        for width, field in ((self._bits_scratch, scratch), (self._bits_param, param), (self._bits_type, dense_type)):
            code = (code << np.uint32(width)) | np.asarray(field, dtype=np.uint32)
        return code

    def _dense_lut(self):
        lut = np.zeros(max(self._type_id_remap) + 1, dtype=np.uint32)
        for orig, dense in self._type_id_remap.items():
            lut[orig] = dense
        return lut

    def encode(self, orientation):
        """Replaces the type ids in the map handed to prepare_encode() by the packed node codes."""
        assert self._type_map is not None, 'prepare_encode() first'
        self._type_choice_map = self._dense_lut()
        self._type_map[:] = self._pack(orientation, self._encoded_param_map,
                                       self._type_choice_map[self._type_map.astype(np.int64)])
        self._type_map = None

    def _subdomain_encode_node(self, orientation, node_type, param):
        return self._pack(orientation, param, self._type_choice_map[np.int64(node_type)])

    def get_param(self, location, values=1):
        idx = self._encoded_param_map[tuple(reversed(location))]
        return self._geo_params[idx:idx + values]

    def update_context(self, ctx):
        """The reference's template-context keys (geo_encoder.py:341-363)."""
        ctx.update({
            'use_link_tags': self.config.use_link_tags,
            'node_types': self._node_types,
            'type_id_remap': self._type_id_remap,
            'nt_id_fluid': self._type_id(0),
            'nt_misc_shift': self._bits_type,
            'nt_type_mask': (1 << self._bits_type) - 1,
            'nt_param_shift': self._bits_param,
            'nt_scratch_shift': self._bits_scratch,
            'nt_dir_other': 0,
            'node_params': self._geo_params,
            'scratch_space': False,
            'unused_tag_bits': self._unused_tag_bits,
        })

    def desc_fields(self):
        """Decode information for slf_module_desc."""
        n = max(self._type_id_remap.values()) + 1
        kinds = [0] * n          # dense id 0 is never produced (ids start at 1)
        for orig, new in self._type_id_remap.items():
            cls = nt._NODE_TYPES[orig]
            kind = nt.hip_kind(cls, getattr(self.config, 'access_pattern', 'AB'))
            if kind is None:
                raise NotImplementedError('node type %s is not implemented by the HIP backend' % cls.__name__)
            kinds[new] = kind
        return dict(nt_type_mask=(1 << self._bits_type) - 1, nt_misc_shift=self._bits_type,
                    nt_param_shift=self._bits_param, nt_scratch_shift=self._bits_scratch,
                    type_kind=kinds, node_params=list(self._geo_params),
                    use_link_tags=int(bool(self.config.use_link_tags and self._have_link_tags)))
