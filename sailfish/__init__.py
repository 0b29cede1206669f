"""Import alias: `from sailfish.<module> import ...` resolves to sailfish_amd.<module>, so that
simulation scripts written for sailfish-team/sailfish (examples/ldc_2d.py, ldc_3d.py,
poiseuille.py, ...) run unchanged on the MI355X backend."""
import importlib
import sys

import sailfish_amd

_MODULES = ['sym', 'util', 'config', 'node_type', 'geo_encoder', 'subdomain_connection', 'subdomain', 'geo',
            'io', 'lb_base', 'subdomain_runner', 'lb_single', 'lb_binary', 'connector', 'controller', 'backend_hip', 'hipabi']

__version__ = sailfish_amd.__version__

for _name in _MODULES:
    _mod = importlib.import_module('sailfish_amd.' + _name)
    sys.modules[__name__ + '.' + _name] = _mod
    globals()[_name] = _mod
