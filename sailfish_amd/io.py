"""Simulation output: what the regression tests and the result tools of the reference rely on (sailfish/io.py) --
`<base>.<subdomain>.<iteration>.npz` with the fields `rho`, `v`, ... of one subdomain, non-wet nodes NaN, plus the
names of the distribution dumps, node-type maps, the subdomain list and checkpoints."""
import glob
import math
import re

import numpy as np


# ---- file names (the contract with utils/merge_subdomains.py, utils/compare_results.py and the reference's scripts)
def _zero_padded(number, digits):
    return '%0*d' % (int(digits), int(number))


def filename_iter_digits(max_iters=0):
    """Width of the iteration field: enough for max_iters, 7 for open-ended runs."""
    return str(1 + int(math.log10(max_iters))) if max_iters else '7'


def filename(base, digits, subdomain_id, it, suffix='.npz'):
    return '.'.join((base, str(subdomain_id), _zero_padded(it, digits))) + suffix


def merged_filename(base, digits, it, suffix='.npz'):
    return base + '.' + _zero_padded(it, digits) + suffix


def dists_filename(base, digits, subdomain_id, it, suffix='.npz'):
    return filename(base + '_dists', digits, subdomain_id, it, suffix)


def node_type_filename(base, subdomain_id, suffix='.npy'):
    return filename(base + '_node_type_map', 1, subdomain_id, 0, suffix)


def subdomains_filename(base):
    return base + '.subdomains'


def checkpoint_filename(base, digits, subdomain_id, it):
    """Without the '.npz' numpy appends."""
    return '.'.join((base, _zero_padded(it, digits), str(subdomain_id), 'cpoint'))


def subdomain_checkpoint(base, subdomain_id):
    """File to restore subdomain `subdomain_id` from; '<base>.last' picks the newest checkpoint of that base."""
    tail = '.%s.cpoint.npz' % subdomain_id
    if not base.endswith('.last'):
        return base + tail
    found = sorted(glob.glob(base[:-len('.last')] + '.*' + tail))
    return found[-1] if found else None


def iter_from_filename(fname):
    return re.search(r'([0-9]+)\.npz', fname).group(1)


# ---- output objects
class LBOutput(object):
    """Keeps references to the host arrays of the registered fields; subclasses write them."""
    format_name = 'none'

    def __init__(self, config, subdomain_id, *args, **kwargs):
        self.basename = config.output
        self.subdomain_id = subdomain_id
        self.num_subdomains = getattr(config, 'subdomains', 1)
        self._scalar_fields, self._vector_fields, self._visualization_fields = {}, {}, {}
        self._fluid_map = None

    def register_field(self, field, name, visualization=False):
        """field: an array, or a list of arrays for a vector field."""
        if visualization:
            target = self._visualization_fields
        else:
            target = self._vector_fields if isinstance(field, list) else self._scalar_fields
        target[name] = field

    def _components(self):
        for f in self._scalar_fields.values():
            yield f
        for comps in self._vector_fields.values():
            for f in comps:
                yield f

    def set_fluid_map(self, fluid_map):
        self._fluid_map = fluid_map

    def mask_nonfluid_nodes(self):
        """Nodes that carry no fluid show up as NaN in the output."""
        dry = ~self._fluid_map
        for f in self._components():
            f[dry] = np.nan

    def verify(self):
        """True when every wet node of every field holds a finite value."""
        wet = self._fluid_map
        return all(bool(np.isfinite(f[wet]).all()) for f in self._components())

    def save(self, i):
        pass

    def dump_dists(self, dists, i):
        pass

    def dump_node_type(self, node_type):
        pass

    def wait(self):
        pass


class NPYOutput(LBOutput):
    """One .npz per subdomain and output step (compressed unless --nooutput_compress)."""
    format_name = 'npy'

    def __init__(self, config, subdomain_id):
        LBOutput.__init__(self, config, subdomain_id)
        self.digits = filename_iter_digits(config.max_iters)
        self._write = np.savez_compressed if getattr(config, 'output_compress', True) else np.savez

    def save(self, i):
        self.mask_nonfluid_nodes()
        arrays = dict(self._scalar_fields)
        for name, comps in self._vector_fields.items():
            arrays[name] = np.array(comps)
        self._write(filename(self.basename, self.digits, self.subdomain_id, i), **arrays)

    def dump_dists(self, dists, i):
        self._write(dists_filename(self.basename, self.digits, self.subdomain_id, i), *dists)

    def dump_node_type(self, node_type_map):
        np.save(node_type_filename(self.basename, self.subdomain_id), node_type_map)


format_name_to_cls = {'npy': NPYOutput}
