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


class GeoEncoderConst(object):
    def __init__(self, subdomain):
        self.subdomain = subdomain
        self.dim = subdomain.dim
        self.config = subdomain.config
        self._type_id_remap = {0: 0}
        self._node_types = set([nt._NTFluid])
        self._bits_type = 0
        self._bits_param = 0
        self._bits_scratch = 0
        self._type_map = None
        self._geo_params = []
        self.scratch_space_size = 0
        self._unused_tag_bits = 0
        self._have_link_tags = False

    def _type_id(self, node_type):
        return self._type_id_remap.get(node_type, 0xffffffff)

    def prepare_encode(self, type_map, param_map, param_dict, orientation, have_link_tags):
        """type_map: node type ids; param_map: keys into param_dict (LBNodeType instances)."""
        uniq_types = [int(x) for x in np.unique(type_map)]
        for nt_id in uniq_types:
            self._node_types.add(nt._NODE_TYPES[nt_id])
        # dense renumbering, starting at 1 (reference geo_encoder.py:83-91)
        for i, node_type in enumerate(uniq_types):
            self._type_id_remap[node_type] = i + 1
        self._bits_type = util.bit_len(len(uniq_types))
        self._type_map = type_map
        self._encoded_param_map = np.zeros_like(type_map)
        self._scratch_map = np.zeros_like(type_map)

        param_to_idx = {}
        param_items = 0
        for node_key, node_type in param_dict.items():
            for param in node_type.params.values():
                if util.is_number(param):
                    if param not in param_to_idx:
                        self._geo_params.append(float(param))
                        param_to_idx[param] = param_items
                        param_items += 1
                    self._encoded_param_map[param_map == node_key] = param_to_idx[param]
                elif type(param) is tuple:
                    if param not in param_to_idx:
                        self._geo_params.extend(float(p) for p in param)
                        param_to_idx[param] = param_items
                        param_items += len(param)
                    self._encoded_param_map[param_map == node_key] = param_to_idx[param]
                elif isinstance(param, np.ndarray):
                    nodes_idx = np.argwhere(param_map == node_key)
                    for value in np.unique(param):
                        key = tuple(value) if hasattr(value, '__len__') else (float(value),)
                        if key not in param_to_idx:
                            self._geo_params.extend(float(p) for p in key)
                            param_to_idx[key] = param_items
                            param_items += len(key)
                        idxs = nodes_idx[param == value]
                        self._encoded_param_map[tuple(idxs[:, k] for k in range(idxs.shape[1]))] = param_to_idx[key]
                else:
                    raise ValueError('unsupported node parameter type for the HIP backend: %r' % type(param))
        self._bits_param = util.bit_len(param_items)
        for node_type in self._node_types:
            if node_type.scratch_space_size(self.dim) > 0:
                raise NotImplementedError('%s needs node scratch space, which the HIP backend does not provide'
                                          % node_type.__name__)
        self._bits_scratch = 0
        self._have_link_tags = have_link_tags
        if have_link_tags:
            tags = orientation[orientation > 0]
            if tags.size:
                self._unused_tag_bits = int(np.bitwise_and.reduce(tags))

    def _encode_node(self, orientation, param, node_type, scratch_id=0):
        if (32 - self._bits_scratch < self.subdomain.grid.Q - 1 and self.config.use_link_tags and
                self._have_link_tags):
            raise ValueError('Not enough bits available to tag neighbor nodes.')
        misc_data = (orientation << self._bits_scratch) | scratch_id
        misc_data = (misc_data << self._bits_param) | param
        return (misc_data << self._bits_type) | node_type

    def encode(self, orientation):
        assert self._type_map is not None
        max_type_code = max(self._type_id_remap.keys())
        self._type_choice_map = np.zeros(max_type_code + 1, dtype=np.uint32)
        for orig_code, new_code in self._type_id_remap.items():
            self._type_choice_map[orig_code] = new_code
        self._type_map[:] = self._encode_node(orientation.astype(np.uint32),
                                              self._encoded_param_map.astype(np.uint32),
                                              self._type_choice_map[self._type_map.astype(np.int64)],
                                              self._scratch_map.astype(np.uint32))
        self._type_map = None

    def _subdomain_encode_node(self, orientation, node_type, param):
        return self._encode_node(np.uint32(orientation), param, self._type_choice_map[np.int64(node_type)])

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
            if cls not in nt.HIP_KIND:
                raise NotImplementedError('node type %s is not implemented by the HIP backend' % cls.__name__)
            kinds[new] = nt.HIP_KIND[cls]
        return dict(nt_type_mask=(1 << self._bits_type) - 1, nt_misc_shift=self._bits_type,
                    nt_param_shift=self._bits_param, nt_scratch_shift=self._bits_scratch,
                    type_kind=kinds, node_params=list(self._geo_params),
                    use_link_tags=int(bool(self.config.use_link_tags and self._have_link_tags)))
