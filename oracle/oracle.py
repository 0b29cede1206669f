"""ctypes front-end of the CPU oracle (oracle/lbm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by sailfish_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

from sailfish_amd.hipabi import SlfModuleDesc, dist_stride  # the problem-description struct (interface only)

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def lib(precision=4):
    """precision: 4 -> liboracle_f32.so, 8 -> liboracle_f64.so"""
    if precision in _libs:
        return _libs[precision]
    name = os.path.join(HERE, 'liboracle_f32.so' if precision == 4 else 'liboracle_f64.so')
    src = os.path.join(HERE, 'lbm_oracle.c')
    if not os.path.exists(name) or os.path.getmtime(name) < os.path.getmtime(src):
        build()
    L = ctypes.CDLL(name)
    assert L.orc_real_size() == precision
    P = ctypes.POINTER
    dp = P(ctypes.c_double)
    L.orc_node_feq.argtypes = [ctypes.c_int, ctypes.c_double, dp, ctypes.c_int, dp]
    L.orc_node_macro.argtypes = [ctypes.c_int, dp, ctypes.c_int, dp, dp]
    L.orc_node_update.argtypes = [P(SlfModuleDesc), ctypes.c_int, ctypes.c_int, dp, dp, dp, dp]
    vp = ctypes.c_void_p
    L.orc_init.argtypes = [P(SlfModuleDesc), vp, vp, vp, vp, vp]
    L.orc_step.argtypes = [P(SlfModuleDesc), ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32,
                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.orc_pbc.argtypes = [P(SlfModuleDesc), vp, ctypes.c_int, ctypes.c_int]
    L.orc_macro_pbc.argtypes = [P(SlfModuleDesc), vp, ctypes.c_int]
    L.orc_sparse.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int]
    L.orc_compute_macro.argtypes = [P(SlfModuleDesc), ctypes.c_int, vp, vp, vp, vp, vp, vp]
    L.orc_sc_init.argtypes = [P(SlfModuleDesc), vp, vp, vp, vp, vp, vp, vp]
    L.orc_sc_macro.argtypes = [P(SlfModuleDesc), ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_sc_step.argtypes = [P(SlfModuleDesc), ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_sc_force_node.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp, dp]
    L.orc_scs_macro.argtypes = [P(SlfModuleDesc), ctypes.c_int, vp, vp, vp]
    L.orc_scs_step.argtypes = [P(SlfModuleDesc), ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32]
    for fn in (L.orc_sc_init, L.orc_sc_macro, L.orc_sc_step, L.orc_sc_force_node, L.orc_scs_macro, L.orc_scs_step):
        fn.restype = None
    L.orc_set_nodes.argtypes = [vp]
    L.orc_set_nodes.restype = None
    L.orc_set_mrt_form.argtypes = [ctypes.c_int]
    L.orc_set_mrt_form.restype = None
    for fn in (L.orc_node_feq, L.orc_node_macro, L.orc_node_update, L.orc_init, L.orc_step, L.orc_pbc,
               L.orc_macro_pbc, L.orc_sparse, L.orc_compute_macro):
        fn.restype = None
    _libs[precision] = L
    return L


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _vp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def node_feq(lattice, rho, v, incompressible=False, precision=8):
    Q = 9 if lattice == 0 else 19
    out = np.zeros(Q)
    vv = np.zeros(3)
    vv[:len(v)] = v
    lib(precision).orc_node_feq(lattice, float(rho), _dp(vv), int(incompressible), _dp(out))
    return out


def node_macro(lattice, f, incompressible=False, precision=8):
    f = np.ascontiguousarray(f, dtype=np.float64)
    rho = np.zeros(1)
    v = np.zeros(3)
    lib(precision).orc_node_macro(lattice, _dp(f), int(incompressible), _dp(rho), _dp(v))
    return rho[0], v


def node_update(desc, kind, orientation, par, f, precision=8):
    f = np.array(f, dtype=np.float64)
    rho = np.zeros(1)
    v = np.zeros(3)
    pp = np.zeros(3)
    if par is not None:
        pp[:len(par)] = par
    lib(precision).orc_node_update(ctypes.byref(desc), kind, orientation, _dp(pp), _dp(f), _dp(rho), _dp(v))
    return f, rho[0], v


def set_mrt_form(matrix_form, precision=8):
    """D3Q19 MRT: False = products through the pairs of opposite directions (the arithmetic of the HIP kernels, the
    default), True = plain row-by-row products with the integer matrix (the definition; tests compare the two)."""
    lib(precision).orc_set_mrt_form(int(bool(matrix_form)))


def sc_force_node(lattice, potential, cc, rho_local, neigh, precision=8):
    out = np.zeros(3)
    n = np.ascontiguousarray(neigh, dtype=np.float64)
    lib(precision).orc_sc_force_node(lattice, potential, float(cc), float(rho_local), _dp(n), _dp(out))
    return out


class OracleSim(object):
    """Whole-subdomain oracle operating on numpy arrays in the reference's memory layout."""

    def __init__(self, desc):
        self.desc = desc
        self.precision = desc.precision
        self.dtype = np.float32 if desc.precision == 4 else np.float64
        self.L = lib(desc.precision)
        self.Q = 9 if desc.lattice == 0 else 19
        self.dim = 2 if desc.lattice == 0 else 3
        self.shape = (desc.arr_nz, desc.arr_ny, desc.arr_nx)
        self.n = desc.arr_nz * desc.arr_ny * desc.arr_nx
        self.stride = dist_stride(desc)

    def new_dist(self):
        """Returns a [Q, nz, ny, nx] view into a buffer with the descriptor's direction stride."""
        raw = np.full((self.Q, self.stride), np.nan, dtype=self.dtype)
        view = raw[:, :self.n].reshape((self.Q,) + self.shape)
        assert view.base is not None
        return view

    def new_field(self, fill=0.0):
        return np.full(self.shape, fill, dtype=self.dtype)

    # -- indirect node addressing: distributions are [Q, stride] arrays of active-node slots ----------
    nodes = None

    def set_nodes(self, nodes):
        """nodes: dense uint32 array (slot or 0xffffffff per node) or None for direct addressing."""
        self.nodes = None if nodes is None else np.ascontiguousarray(nodes, dtype=np.uint32)

    def new_sparse_dist(self):
        return np.full((self.Q, self.stride), np.nan, dtype=self.dtype)

    def _install_nodes(self):
        self.L.orc_set_nodes(_vp(self.nodes) if self.nodes is not None else None)

    def init(self, dist, rho, vx, vy, vz=None):
        self._install_nodes()
        self.L.orc_init(ctypes.byref(self.desc), _vp(dist), _vp(rho), _vp(vx), _vp(vy), _vp(vz))
        self.L.orc_set_nodes(None)

    def step(self, prop, nmap, din, dout, rho, vx, vy, vz, options=0, region=None):
        d = self.desc
        y0, y1, z0, z1 = 1, d.lat_ny - 1, 1, d.lat_nz - 1
        if region is not None:
            y0, y1, z0, z1 = region
        self._install_nodes()
        self.L.orc_step(ctypes.byref(d), prop, _vp(nmap), _vp(din), _vp(dout), _vp(rho), _vp(vx), _vp(vy),
                        _vp(vz), options, y0, y1, z0, z1)
        self.L.orc_set_nodes(None)

    def pbc(self, dist, axis, with_swap=False):
        self.L.orc_pbc(ctypes.byref(self.desc), _vp(dist), axis, int(with_swap))

    def macro_pbc(self, field, axis):
        self.L.orc_macro_pbc(ctypes.byref(self.desc), _vp(field), axis)

    def sparse(self, collect, idx, dist, buf):
        self.L.orc_sparse(int(collect), _vp(idx), _vp(dist), _vp(buf), len(idx))

    def sc_init(self, d1, d2, rho, phi, vx, vy, vz):
        self._install_nodes()
        self.L.orc_sc_init(ctypes.byref(self.desc), _vp(d1), _vp(d2), _vp(rho), _vp(phi), _vp(vx), _vp(vy), _vp(vz))
        self.L.orc_set_nodes(None)

    def sc_macro(self, prop, nmap, d1, d2, rho, phi, vx, vy, vz):
        self._install_nodes()
        self.L.orc_sc_macro(ctypes.byref(self.desc), prop, _vp(nmap), _vp(d1), _vp(d2), _vp(rho), _vp(phi), _vp(vx),
                            _vp(vy), _vp(vz))
        self.L.orc_set_nodes(None)

    def sc_step(self, grid_idx, prop, nmap, din, dout, rho, phi, vx, vy, vz):
        self._install_nodes()
        self.L.orc_sc_step(ctypes.byref(self.desc), grid_idx, prop, _vp(nmap), _vp(din), _vp(dout), _vp(rho),
                           _vp(phi), _vp(vx), _vp(vy), _vp(vz))
        self.L.orc_set_nodes(None)

    def scs_macro(self, prop, nmap, din, rho):
        self.L.orc_scs_macro(ctypes.byref(self.desc), prop, _vp(nmap), _vp(din), _vp(rho))

    def scs_step(self, prop, nmap, din, dout, rho, vx, vy, vz, options=0):
        self.L.orc_scs_step(ctypes.byref(self.desc), prop, _vp(nmap), _vp(din), _vp(dout), _vp(rho), _vp(vx),
                            _vp(vy), _vp(vz), options)

    def compute_macro(self, prop, nmap, din, rho, vx, vy, vz):
        self.L.orc_compute_macro(ctypes.byref(self.desc), prop, _vp(nmap), _vp(din), _vp(rho), _vp(vx), _vp(vy),
                                 _vp(vz))
