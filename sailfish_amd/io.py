"""Simulation output (reference sailfish/io.py: LBOutput, NPYOutput and the file-name helpers
its regression tests rely on: <base>.<subdomain>.<iter>.npz with fields rho, v)."""
import glob
import math
import re

import numpy as np


class LBOutput(object):
    format_name = 'none'

    def __init__(self, config, subdomain_id, *args, **kwargs):
        self._scalar_fields = {}
        self._vector_fields = {}
        self._fluid_map = None
        self._visualization_fields = {}
        self.basename = config.output
        self.subdomain_id = subdomain_id
        self.num_subdomains = config.subdomains if hasattr(config, 'subdomains') else 1

    def register_field(self, field, name, visualization=False):
        if visualization:
            self._visualization_fields[name] = field
        elif type(field) is list:
            self._vector_fields[name] = field
        else:
            self._scalar_fields[name] = field

    def mask_nonfluid_nodes(self):
        """Non-wet nodes carry NaN in the output (reference io.py:53-59)."""
        nonfluid = np.logical_not(self._fluid_map)
        for f in self._scalar_fields.values():
            f[nonfluid] = np.nan
        for fv in self._vector_fields.values():
            for f in fv:
                f[nonfluid] = np.nan

    def save(self, i):
        pass

    def dump_dists(self, dists, i):
        pass

    def dump_node_type(self, node_type):
        pass

    def set_fluid_map(self, fluid_map):
        self._fluid_map = fluid_map

    def verify(self):
        fm = self._fluid_map
        return (all(np.all(np.isfinite(f[fm])) for f in self._scalar_fields.values()) and
                all(np.all(np.isfinite(fc[fm])) for f in self._vector_fields.values() for fc in f))

    def wait(self):
        pass


def filename_iter_digits(max_iters=0):
    return str(int(math.log10(max_iters)) + 1) if max_iters else str(7)


def filename(base, digits, subdomain_id, it, suffix='.npz'):
    return ('{0}.{1}.{2:0' + str(digits) + 'd}{3}').format(base, subdomain_id, it, suffix)


def merged_filename(base, digits, it, suffix='.npz'):
    return ('{0}.{1:0' + str(digits) + 'd}{2}').format(base, it, suffix)


def dists_filename(base, digits, subdomain_id, it, suffix='.npz'):
    return filename(base + '_dists', digits, subdomain_id, it, suffix=suffix)


def node_type_filename(base, subdomain_id, suffix='.npy'):
    return filename(base + '_node_type_map', 1, subdomain_id, 0, suffix=suffix)


def subdomains_filename(base):
    return base + '.subdomains'


def checkpoint_filename(base, digits, subdomain_id, it):
    return ('{0}.{1:0' + str(digits) + 'd}.{2}.cpoint').format(base, it, subdomain_id)


def subdomain_checkpoint(base, subdomain_id):
    if base.endswith('.last'):
        base = base[:-5]
        files = glob.glob('{0}.*.{1}.cpoint.npz'.format(base, subdomain_id))
        if not files:
            return None
        files.sort()
        return files[-1]
    return '{0}.{1}.cpoint.npz'.format(base, subdomain_id)


def iter_from_filename(fname):
    return re.findall(r'([0-9]+)\.npz', fname)[0]


class NPYOutput(LBOutput):
    """np.savez[_compressed] of the registered fields (reference io.py:301-347)."""
    format_name = 'npy'

    def __init__(self, config, subdomain_id):
        LBOutput.__init__(self, config, subdomain_id)
        self.digits = filename_iter_digits(config.max_iters)
        self._do_save = np.savez_compressed if getattr(config, 'output_compress', True) else np.savez

    def save(self, i):
        self.mask_nonfluid_nodes()
        fname = filename(self.basename, self.digits, self.subdomain_id, i, suffix='.npz')
        data = {}
        data.update(self._scalar_fields)
        data.update(dict((k, np.array(v)) for k, v in self._vector_fields.items()))
        self._do_save(fname, **data)

    def dump_dists(self, dists, i):
        fname = dists_filename(self.basename, self.digits, self.subdomain_id, i)
        self._do_save(fname, *dists)

    def dump_node_type(self, node_type_map):
        np.save(node_type_filename(self.basename, self.subdomain_id), node_type_map)


format_name_to_cls = {'npy': NPYOutput}
