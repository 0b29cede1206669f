"""A CPU stand-in for the HIP backend, for tests only: the backend interface of sailfish_amd/backend_hip.py with the
kernels executed by the CPU oracle on host memory.  It lets the PRODUCT's host stack -- LBSimulationController,
SubdomainRunner.step / halo_messages / regions / events, TorchDistConnector -- run unchanged on a machine without a
GPU (world_size-2 gloo tests); "device addresses" are addresses of numpy arrays, so the runner's pointer arithmetic
on halo buffers works as on the device.  Nothing in sailfish_amd/ imports this."""
import ctypes

import numpy as np

from oracle import oracle as orc
from sailfish_amd import hipabi

VP = ctypes.c_void_p


class _Event(object):
    def synchronize(self):
        pass

    def record(self, stream=None):
        pass

    def time_since(self, other):
        return 0.0


class _Stream(object):
    native = 0

    def synchronize(self):
        pass

    def wait_for_event(self, ev):
        pass


class _Module(object):
    block_size = 64

    def __init__(self, desc):
        self.desc = desc
        self.sim = orc.OracleSim(desc)


class _Kernel(object):
    def __init__(self, module, name, args, fmt, needs_iteration):
        self.module, self.name, self.args, self.fmt = module, name, [int(a) if a is not None else 0 for a in args], fmt
        self.needs_iteration = needs_iteration
        self.iteration = 0


class OracleBackend(object):
    name = 'oracle_test'
    tensor_device = 'cpu'

    class FatalError(RuntimeError):
        pass

    @classmethod
    def add_options(cls, group):
        group.add_argument('--nohip_graphs', dest='hip_graphs', action='store_false', default=False)
        group.add_argument('--nohip_fused_periodic', dest='hip_fused_periodic', action='store_false', default=True)
        return 1

    def __init__(self, options=None, gpu_id=0):
        self.options, self.gpu_id = options, gpu_id
        self.buffers, self._keep, self._kernels = {}, {}, []
        self.total_memory = 1 << 40
        self.info = 'CPU oracle (test backend)'

    # -- memory: addresses of numpy arrays
    def alloc_buf(self, size=None, like=None, wrap_in_array=False, align_offset=0):
        if like is not None:
            host = like.base if (like.base is not None and isinstance(like.base, np.ndarray)) else like
            dev = host.copy()
        else:
            dev = np.zeros(int(size) + 256, dtype=np.uint8)
        addr = dev.ctypes.data + (int(align_offset) if like is None else 0)
        self._keep[addr] = dev
        if like is not None:
            self.buffers[addr] = host
        return addr

    def free_buf(self, addr):
        self._keep.pop(addr, None)
        self.buffers.pop(addr, None)

    @staticmethod
    def dist_align_offset(itemsize, envelope=1):
        return 0

    def alloc_async_host_buf(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)

    def _host(self, buf, other):
        if other is None:
            return self.buffers[buf]
        return other.base if (other.base is not None and isinstance(other.base, np.ndarray)) else other

    def to_buf(self, buf, source=None):
        host = self._host(buf, source)
        ctypes.memmove(buf, host.ctypes.data, host.nbytes)

    def from_buf(self, buf, target=None):
        host = self._host(buf, target)
        ctypes.memmove(host.ctypes.data, buf, host.nbytes)

    def to_buf_async(self, buf, stream=None):
        self.to_buf(buf)

    def from_buf_async(self, buf, stream=None):
        self.from_buf(buf)

    def copy_buf_async(self, dst, src, nbytes, stream=None):
        ctypes.memmove(dst, src, int(nbytes))

    def memset_buf(self, buf, value, nbytes, stream=None):
        ctypes.memset(buf, int(value), int(nbytes))

    # -- modules / kernels
    def build(self, source):
        assert isinstance(source, hipabi.SlfModuleDesc)
        return _Module(source)

    def get_kernel(self, prog, name, block, args, args_format, shared=0, needs_iteration=False, more_shared=False):
        k = _Kernel(prog, name, args, args_format, needs_iteration)
        if needs_iteration:
            self._kernels.append(k)
        return k

    def set_iteration(self, it):
        for k in self._kernels:
            k.iteration = int(it)

    def set_body_force(self, module, accel, lattice=0):
        for i, a in enumerate(accel):
            module.desc.accel[i] = float(a)

    def update_node_params(self, module, first, values, stream=None):
        for i, v in enumerate(np.asarray(values, dtype=np.float64).ravel()):
            module.desc.node_params[int(first) + i] = float(v)

    def run_kernel(self, k, grid_size=None, stream=None):
        m, d, L = k.module, k.module.desc, k.module.sim.L
        dim = 2 if d.lattice == hipabi.SLF_D2Q9 else 3
        a = k.args
        aa = d.access_pattern == hipabi.SLF_AA
        ref = ctypes.byref(d)
        with np.errstate(all='ignore'):
            if k.name in ('CollideAndPropagate', 'ComputeMacroFields'):
                nmap, din, dout, rho, vx, vy = a[0], a[1], a[2], a[3], a[4], a[5]
                vz = a[6] if dim == 3 else 0
                options = a[4 + dim]
                prop = (2 if (k.iteration & 1) else 1) if aa else 0
                if k.name == 'ComputeMacroFields':
                    L.orc_compute_macro(ref, prop, VP(nmap), VP(din), VP(rho), VP(vx), VP(vy), VP(vz))
                    return
                y0, y1, z0, z1 = 1, d.lat_ny - 1, 1, d.lat_nz - 1
                if grid_size is not None:
                    y0, y1, z0, z1 = [int(v) for v in grid_size]
                L.orc_step(ref, prop, VP(nmap), VP(din), VP(dout), VP(rho), VP(vx), VP(vy), VP(vz), options, y0, y1, z0, z1)
            elif k.name == 'SetInitialConditions':
                dist, v, rho = a[0], a[1:1 + dim], a[1 + dim]
                L.orc_init(ref, VP(dist), VP(rho), VP(v[0]), VP(v[1]), VP(v[2] if dim == 3 else 0))
            elif k.name in ('ApplyPeriodicBoundaryConditions', 'ApplyPeriodicBoundaryConditionsWithSwap'):
                L.orc_pbc(ref, VP(a[0]), a[1], int(k.name.endswith('WithSwap')))
            elif k.name == 'ApplyMacroPeriodicBoundaryConditions':
                L.orc_macro_pbc(ref, VP(a[0]), a[1])
            elif k.name in ('CollectSparseData', 'DistributeSparseData'):
                L.orc_sparse(int(k.name.startswith('Collect')), VP(a[0]), VP(a[1]), VP(a[2]), a[3])
            else:
                raise NotImplementedError('test backend: kernel %s' % k.name)

    # -- streams / events: everything is synchronous
    def make_stream(self):
        return _Stream()

    def make_event(self, stream, timing=False):
        return _Event()

    def sync(self):
        pass

    def sync_stream(self, *streams):
        pass

    def poll_invalid(self, module, stream):
        return None

    def capture_graph(self, stream, enqueue):
        raise self.FatalError('no graphs on the CPU test backend')

    def get_defines(self):
        return {'warp_size': 64, 'backend': 'oracle_test'}

    def close(self, free_pinned=False):
        self._keep.clear()


backend = OracleBackend
