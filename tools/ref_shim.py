"""Import shim that lets the reference's *host/symbolic* modules be imported in
the authoring container (never on the GPU box).  Used only by
tools/capture_goldens.py to generate the fixtures in tests/golden/.

The reference needs mako/zmq/skimage/... which are absent here; only the host
algebra (sympy expressions, node encoders, connection tables) is exercised, so
empty stand-in *modules* are enough -- no reference code is modified or copied.
"""
import sys, types, warnings
warnings.filterwarnings('ignore')
import sympy.printing.c as _c
sys.modules['sympy.printing.ccode'] = _c
for n in ['zmq', 'skimage', 'skimage.morphology', 'mako', 'mako.runtime', 'mako.exceptions',
          'mako.lookup', 'mako.template', 'blosc', 'netifaces']:
    sys.modules[n] = types.ModuleType(n)
sys.modules['mako.runtime'].Undefined = type('Undefined', (), {})
sys.modules['mako.lookup'].TemplateLookup = object
sys.modules['mako.template'].Template = object
sys.modules['skimage'].morphology = sys.modules['skimage.morphology']


class _Ctx:
    def socket(self, *a, **k):
        raise RuntimeError('zmq stub')

    def destroy(self):
        pass


z = sys.modules['zmq']
z.Context = _Ctx
z.PAIR, z.REQ, z.REP = 0, 1, 2
import numpy as np
np.bool = np.bool_
np.float = np.float64
np.int = int
REF = '/root/reference'
if REF not in sys.path:
    sys.path.insert(0, REF)
    sys.path.insert(1, REF + '/tests')
