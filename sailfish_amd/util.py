"""Small host helpers (reference sailfish/util.py, the parts the hot path uses)."""
import importlib
import logging
import sys
from collections import namedtuple

import numpy as np

from sailfish_amd import sym

TimingInfo = namedtuple('TimingInfo', 'comp bulk bnd coll net_wait recv send total total_sq subdomain_id')


class GridError(Exception):
    pass


def get_grid_from_config(config):
    for x in sym.KNOWN_GRIDS:
        if x.__name__ == config.grid:
            return x
    return None


def get_backends(backends=('hip',)):
    """Yields the `backend` class of every importable sailfish.backend_<name> module
    (reference util.py:52-59).  Only the HIP backend exists here; a dotted name is the full path of a module that
    provides a `backend` class (out-of-tree backends: the CPU test backend of tests/)."""
    for backend in backends:
        try:
            module = importlib.import_module(backend if '.' in backend else 'sailfish_amd.backend_{0}'.format(backend))
            yield module.backend
        except ImportError:
            pass


def is_number(param):
    return isinstance(param, (int, float, np.integer, np.floating))


def in_anyd_fast(arr1, values):
    """Boolean array: arr1 element is one of `values` (reference util.py:130-145)."""
    values = np.asarray(values).ravel()
    if values.size == 0:
        return np.zeros(arr1.shape, dtype=bool)
    return np.isin(arr1, values)


def bit_len(num):
    """Bits needed to represent num distinct-from-zero values (reference util.py bit_len)."""
    length = 0
    while num:
        num >>= 1
        length += 1
    return max(length, 1)


class lazy_property(object):
    def __init__(self, fget):
        self.fget = fget
        self.func_name = fget.__name__

    def __get__(self, obj, cls):
        if obj is None:
            return None
        value = self.fget(obj)
        setattr(obj, self.func_name, value)
        return value


def setup_logger(config, name='sailfish'):
    logger = logging.getLogger(name)
    if not logger.handlers:
        handler = logging.StreamHandler(sys.stderr)
        handler.setFormatter(logging.Formatter('[%(relativeCreated)6d %(levelname)5s %(processName)s] %(message)s'))
        logger.addHandler(handler)
        logger.propagate = False
    if getattr(config, 'verbose', False):
        logger.setLevel(logging.DEBUG)
    elif getattr(config, 'quiet', False):
        logger.setLevel(logging.WARNING)
    else:
        logger.setLevel(logging.INFO)
    if getattr(config, 'log', ''):
        fh = logging.FileHandler(config.log)
        fh.setLevel(getattr(config, 'loglevel', logging.INFO))
        logger.addHandler(fh)
    return logger
