"""Sailfish HIP backend for MI355X (gfx950).

Drop-in for the reference's backend modules (sailfish/backend_cuda.py,
backend_opencl.py): same class-level contract -- ``name``, ``FatalError``,
``add_options``, ``alloc_buf``, ``alloc_async_host_buf``, ``to_buf``,
``from_buf``, ``to_buf_async``, ``from_buf_async``, ``build``, ``get_kernel``,
``run_kernel``, ``set_iteration``, ``make_stream``, ``make_event``,
``sync_stream``, ``sync``, ``get_defines``, ``info``, ``total_memory`` -- and a
module-level ``backend`` symbol, which is what ``util.get_backends`` looks up
(reference sailfish/util.py:52-59).  Instantiated as ``backend(config, gpu_id)``
inside each subdomain process (reference sailfish/master.py:46-49).

Differences forced by the design (pre-built gfx950 kernels instead of run-time
generated CUDA/OpenCL text):
  * ``build(source)`` takes a *module descriptor* (hipabi.SlfModuleDesc) instead
    of C source and returns a module handle;
  * ``run_kernel(kernel, grid_size, stream)``: ``grid_size`` is either ``None``
    (whole subdomain) or a region ``(y0, y1, z0, z1)`` for the bulk/boundary
    split -- launch geometry is chosen inside the library for 64-wide wavefronts.

Everything calls libsailfish_hip.so through ctypes; there is no CPU fallback.
"""
import ctypes
import gc

import numpy as np

from sailfish_amd import hipabi, placement
from sailfish_amd.stepqueue import DirectQueue, NotPlannable  # noqa: F401  (re-exported)


class HIPFatalError(RuntimeError):
    """Raised on any failed HIP call / kernel launch (reference: pycuda.driver.LaunchError)."""


def _check(lib, status, what=''):
    if status != 0:
        msg = lib.slf_last_error()
        raise HIPFatalError('%s failed (status %d): %s' % (what, status, msg.decode() if msg else '?'))


class HIPStream(object):
    def __init__(self, backend, high_priority=False):
        self._backend = backend
        self._lib = backend._lib
        h = ctypes.c_void_p()
        create = self._lib.slf_stream_create_high_priority if high_priority else self._lib.slf_stream_create
        _check(self._lib, create(backend._ctx, ctypes.byref(h)), 'slf_stream_create')
        self.handle = h

    def synchronize(self):
        _check(self._lib, self._lib.slf_stream_sync(self.handle), 'slf_stream_sync')

    def wait_for_event(self, event):
        _check(self._lib, self._lib.slf_stream_wait_event(self.handle, event.handle), 'slf_stream_wait_event')

    @property
    def native(self):
        """Raw hipStream_t (for torch.cuda.ExternalStream)."""
        p = ctypes.c_void_p()
        _check(self._lib, self._lib.slf_stream_native(self.handle, ctypes.byref(p)), 'slf_stream_native')
        return p.value or 0

    def __del__(self):
        try:
            self._lib.slf_stream_destroy(self.handle)
        except Exception:
            pass


class HIPGraph(object):
    """A recorded stretch of stream work that replays with one runtime call (slf_graph_*)."""

    def __init__(self, backend, handle):
        self._lib = backend._lib
        self.handle = handle

    def launch(self, stream):
        _check(self._lib, self._lib.slf_graph_launch(self.handle, stream.handle if stream is not None else None),
               'slf_graph_launch')

    def __del__(self):
        try:
            self._lib.slf_graph_destroy(self.handle)
        except Exception:
            pass


class HIPEvent(object):
    def __init__(self, backend, timing=False):
        self._lib = backend._lib
        h = ctypes.c_void_p()
        _check(self._lib, self._lib.slf_event_create(backend._ctx, int(bool(timing)), ctypes.byref(h)),
               'slf_event_create')
        self.handle = h

    def record(self, stream=None):
        _check(self._lib, self._lib.slf_event_record(self.handle, stream.handle if stream else None),
               'slf_event_record')

    def synchronize(self):
        _check(self._lib, self._lib.slf_event_sync(self.handle), 'slf_event_sync')

    def time_since(self, other):
        """Milliseconds elapsed since `other` (pycuda Event.time_since semantics)."""
        ms = ctypes.c_float()
        _check(self._lib, self._lib.slf_event_elapsed_ms(other.handle, self.handle, ctypes.byref(ms)),
               'slf_event_elapsed_ms')
        return ms.value

    def __del__(self):
        try:
            self._lib.slf_event_destroy(self.handle)
        except Exception:
            pass


class HIPModule(object):
    def __init__(self, backend, desc):
        self._lib = backend._lib
        self.desc = desc
        h = ctypes.c_void_p()
        _check(self._lib, self._lib.slf_module_create(backend._ctx, ctypes.byref(desc), ctypes.byref(h)),
               'slf_module_create')
        self.handle = h

    @property
    def block_size(self):
        n = ctypes.c_int()
        _check(self._lib, self._lib.slf_module_block_size(self.handle, ctypes.byref(n)), 'slf_module_block_size')
        return n.value

    def __del__(self):
        try:
            self._lib.slf_module_destroy(self.handle)
        except Exception:
            pass


class HIPKernel(object):
    def __init__(self, lib, module, name):
        self._lib = lib
        self.module = module
        self.name = name
        h = ctypes.c_void_p()
        _check(lib, lib.slf_kernel_get(module.handle, name.encode(), ctypes.byref(h)), 'slf_kernel_get(%s)' % name)
        self.handle = h
        self.args = None
        self.needs_iteration = False

    def set_args(self, args, args_format, needs_iteration):
        n = len(args)
        if n != len(args_format):
            raise ValueError('args / args_format length mismatch for kernel %s' % self.name)
        vals = []
        ptrs = (ctypes.c_void_p * max(1, n))()
        for i, (a, f) in enumerate(zip(args, args_format)):
            if f == 'P':
                v = ctypes.c_uint64(int(a) if a is not None else 0)
            elif f == 'i':
                v = ctypes.c_int32(int(a))
            else:
                raise ValueError('unsupported argument format %r' % f)
            vals.append(v)
            ptrs[i] = ctypes.cast(ctypes.pointer(v), ctypes.c_void_p)
        _check(self._lib, self._lib.slf_kernel_set_args(self.handle, args_format.encode(), ptrs, n,
                                                       int(bool(needs_iteration))),
               'slf_kernel_set_args(%s)' % self.name)
        self.args = list(args)
        self.needs_iteration = bool(needs_iteration)

    def __del__(self):
        try:
            self._lib.slf_kernel_destroy(self.handle)
        except Exception:
            pass


class HIPPlan(object):
    """The launch list of one time step, built once and enqueued with ONE C-ABI call per step (slf_plan_*,
    include/sailfish_hip.h "step plans"): same methods as stepqueue.DirectQueue, but every entry is appended to the plan;
    run(iteration) performs them in order inside the library.  Keeps the Python owners of everything it names alive."""
    planned = True

    def __init__(self, backend):
        self.backend = backend
        self._lib = backend._lib
        h = ctypes.c_void_p()
        _check(self._lib, self._lib.slf_plan_create(backend._ctx, ctypes.byref(h)), 'slf_plan_create')
        self.handle = h
        self._keep = []

    def _h(self, stream):
        return stream.handle if stream is not None else None

    def launch(self, kernel, region, stream):
        reg = None
        if region is not None:
            y0, y1, z0, z1 = region
            reg = ctypes.byref(hipabi.SlfRegion(int(y0), int(y1), int(z0), int(z1)))
        _check(self._lib, self._lib.slf_plan_add_launch(self.handle, kernel.handle, reg, self._h(stream)),
               'slf_plan_add_launch(%s)' % kernel.name)
        self._keep += [kernel, stream]

    def record(self, event, stream):
        _check(self._lib, self._lib.slf_plan_add_record(self.handle, event.handle, self._h(stream)), 'slf_plan_add_record')
        self._keep += [event, stream]

    def wait(self, stream, event):
        _check(self._lib, self._lib.slf_plan_add_wait(self.handle, self._h(stream), event.handle), 'slf_plan_add_wait')
        self._keep += [event, stream]

    def exchange(self, rccl, batch, stream):
        arr, n = batch
        _check(self._lib, self._lib.slf_plan_add_exchange(self.handle, rccl.comm, arr, n, self._h(stream)),
               'slf_plan_add_exchange')
        self._keep += [rccl, stream]

    def peer_signal(self, peer, ranks, channel, stream):
        arr, n = peer.ranks_array(ranks)
        _check(self._lib, self._lib.slf_plan_add_peer_signal(self.handle, peer.handle, arr, n, int(channel), self._h(stream)),
               'slf_plan_add_peer_signal')
        self._keep += [peer, stream]

    def peer_wait(self, peer, ranks, channel, stream, count=1):
        arr, n = peer.ranks_array(ranks)
        _check(self._lib, self._lib.slf_plan_add_peer_wait(self.handle, peer.handle, arr, n, int(channel), int(count), self._h(stream)),
               'slf_plan_add_peer_wait')
        self._keep += [peer, stream]

    def memset(self, addr, value, nbytes, stream):
        for a, _, n in self.backend._segments(addr, nbytes):
            _check(self._lib, self._lib.slf_plan_add_memset(self.handle, ctypes.c_void_p(a), int(value), n, self._h(stream)),
                   'slf_plan_add_memset')
        self._keep.append(stream)

    def copy(self, dst, src, nbytes, stream):
        _check(self._lib, self._lib.slf_plan_add_copy(self.handle, ctypes.c_void_p(dst), ctypes.c_void_p(src), int(nbytes),
                                                      self._h(stream)), 'slf_plan_add_copy')
        self._keep.append(stream)

    def xface(self, module, send_low, send_high, recv_low, recv_high):
        _check(self._lib, self._lib.slf_plan_add_xface_buffers(self.handle, module.handle,
                                                               *[ctypes.c_void_p(a or None) for a in
                                                                 (send_low, send_high, recv_low, recv_high)]),
               'slf_plan_add_xface_buffers')
        self._keep.append(module)

    def xface_planes(self, module, which, send_low, send_high, recv_low, recv_high):
        _check(self._lib, self._lib.slf_plan_add_xface_planes(self.handle, module.handle, int(which),
                                                              *[ctypes.c_void_p(a or None) for a in
                                                                (send_low, send_high, recv_low, recv_high)]),
               'slf_plan_add_xface_planes')
        self._keep.append(module)

    def call(self, fn):
        raise NotPlannable('this step needs Python between its launches')

    def __len__(self):
        n = ctypes.c_int()
        _check(self._lib, self._lib.slf_plan_size(self.handle, ctypes.byref(n)), 'slf_plan_size')
        return n.value

    def run(self, iteration):
        _check(self._lib, self._lib.slf_plan_run(self.handle, int(iteration) & 0xFFFFFFFF), 'slf_plan_run')

    def __del__(self):
        try:
            self._lib.slf_plan_destroy(self.handle)
        except Exception:
            pass


class HIPBackend(placement.VmmMixin):
    name = 'hip'
    supports_xface = True      # slf_module_set_xface_buffers (sailfish_amd/xface.py)
    supports_stream_priority = True
    supports_fused_shan_chen = True   # kernel "ShanChenCollideAndPropagateFused"
    supports_fused_shan_chen_local_velocity = True   # "ShanChenPrepareDensities" + "ShanChenCollideAndPropagateFusedV"
    FatalError = HIPFatalError

    @classmethod
    def devices_count(cls):
        lib = hipabi.load()
        n = ctypes.c_int()
        lib.slf_device_count(ctypes.byref(n))
        return n.value

    @classmethod
    def pci_bus_id(cls, device):
        """PCI address of HIP device `device` ('0000:05:00.0'), '' if it cannot be had."""
        lib = hipabi.load()
        buf = ctypes.create_string_buffer(32)
        if lib.slf_device_pci_bus_id(int(device), buf, 32) != 0:
            return ''
        return buf.value.decode().lower()

    @classmethod
    def add_options(cls, group):
        group.add_argument('--hip-kernel-stats', dest='hip_kernel_stats', action='store_true', default=False,
                           help='print the workgroup shape chosen for the sweep kernels')
        group.add_argument('--nohip_graphs', dest='hip_graphs', action='store_false', default=True,
                           help='do not replay stretches of steps without host interaction as HIP graphs')
        group.add_argument('--nohip_resident', dest='hip_resident', action='store_false', default=True,
                           help='small 2-D subdomains: one launch per step instead of launches that perform several steps '
                                'on LDS-resident windows (CollideAndPropagateResident)')
        group.add_argument('--nohip_step_plans', dest='hip_step_plans', action='store_false', default=True,
                           help='enqueue every kernel, event and halo exchange of a step from Python instead of replaying '
                                'the step from a C-ABI step plan (one runtime call per step)')
        group.add_argument('--nohip_xface', dest='hip_xface', action='store_false', default=True,
                           help='1-D decompositions along x: exchange the x faces through ghost columns and pack / '
                                'unpack kernels (the reference\'s scheme) instead of the face buffers the sweep writes '
                                'and reads itself (sailfish_amd/xface.py)')
        group.add_argument('--nohip_sc_fused', dest='hip_sc_fused', action='store_false', default=True,
                           help='binary Shan-Chen: one kernel per lattice (the reference\'s ShanChenCollideAndPropagate0 / 1) '
                                'instead of the fused sweep of both')
        group.add_argument('--nohip_row_classes', dest='hip_row_classes', action='store_false', default=True,
                           help='do not classify the rows of the node map: every wavefront reads the map and the whole '
                                'subdomain runs the kernel instantiation for the module\'s node-type table')
        group.add_argument('--nohip_placement', dest='hip_placement', action='store_false', default=True,
                           help='plain allocations for the distribution arrays instead of spreading their physical '
                                'backing over HBM (sailfish_amd/placement.py)')
        group.add_argument('--nohip_placement_tune', dest='hip_placement_tune', action='store_false', default=True,
                           help='keep the first placement of the distribution arrays instead of timing a few steps on '
                                'it, placing again and keeping the better one (runs of 200 steps and more)')
        group.add_argument('--nohip_fused_periodic', dest='hip_fused_periodic', action='store_false',
                           default=True,
                           help='apply periodic boundary conditions with separate ghost-layer kernels '
                                '(the reference\'s scheme) instead of wrapping inside the sweep')
        return 1

    def __init__(self, options, gpu_id):
        """:param options: LBConfig-like object; :param gpu_id: HIP device ordinal"""
        self._lib = hipabi.load()
        self.options = options
        self.gpu_id = gpu_id
        self.buffers = {}   # device address -> host mirror
        self._sizes = {}
        self._raw = {}
        self._placed = {}
        self._pinned = []
        self._total_memory_bytes = 0
        self._iteration = 0
        ctx = ctypes.c_void_p()
        _check(self._lib, self._lib.slf_ctx_create(int(gpu_id), ctypes.byref(ctx)), 'slf_ctx_create')
        self._ctx = ctx
        name = ctypes.create_string_buffer(256)
        mem = ctypes.c_size_t()
        cus = ctypes.c_int()
        wave = ctypes.c_int()
        _check(self._lib, self._lib.slf_ctx_info(ctx, name, 256, ctypes.byref(mem), ctypes.byref(cus),
                                                 ctypes.byref(wave)), 'slf_ctx_info')
        self._dev_name = name.value.decode()
        self._total_memory = mem.value
        self._cu_count = cus.value
        self._wavefront = wave.value

    def close(self, free_pinned=False):
        """Releases every device buffer this backend handed out (the reference leaves that to the PyCUDA context
        going away with the subdomain process; here one process may run many simulations).  Called by
        SubdomainRunner.release() and when the backend is collected.  Pinned host arrays are only freed on
        request: numpy views of them (sim.rho, sim.v of asynchronous fields) may outlive the backend."""
        lib = getattr(self, '_lib', None)
        if lib is None or getattr(self, '_ctx', None) is None:
            return
        try:
            lib.slf_ctx_sync(self._ctx)
        except Exception:
            pass
        for addr in list(self._placed):
            try:
                self.free_buf(addr)
            except Exception:
                pass
        for addr in list(self._raw):
            try:
                self.free_buf(addr)
            except Exception:
                pass
        if free_pinned:
            for ptr in self._pinned:
                lib.slf_host_free(ctypes.c_void_p(ptr))
            self._pinned = []

    def __del__(self):
        try:
            self.close()
            self._lib.slf_ctx_destroy(self._ctx)
            self._ctx = None
        except Exception:
            pass

    # -- diagnostics ------------------------------------------------------
    @property
    def supports_printf(self):
        return True

    @property
    def info(self):
        return '{0} / {1} CUs / MEM {2}'.format(self._dev_name, self._cu_count, self.total_memory)

    @property
    def total_memory(self):
        return self._total_memory

    def get_defines(self):
        return {'warp_size': self._wavefront, 'supports_shuffle': True, 'supports_printf': True,
                'backend': 'hip'}

    # -- memory -----------------------------------------------------------
    @staticmethod
    def _host_base(arr):
        return arr.base if (arr.base is not None and isinstance(arr.base, np.ndarray)) else arr

    def alloc_buf(self, size=None, like=None, wrap_in_array=False, align_offset=0):
        """Allocates a device buffer; with ``like`` the host array (or its base)
        becomes the buffer's mirror and is copied to the device immediately
        (reference backend_cuda.py:132-154).

        align_offset (extension): the returned address is a 256-byte aligned address plus this many
        bytes.  The runner uses it for the distribution arrays so that the first *real* node of every
        row (x = 1, behind the ghost column) starts a 128-byte line."""
        if like is not None:
            host = self._host_base(like)
            if not host.flags['C_CONTIGUOUS']:
                raise ValueError('host mirror must be C-contiguous')
            size = host.nbytes
        ptr = ctypes.c_void_p()
        pad = 256 if align_offset else 0
        _check(self._lib, self._lib.slf_malloc(self._ctx, int(size) + pad, ctypes.byref(ptr)), 'slf_malloc')
        raw = ptr.value
        addr = raw + int(align_offset)
        self._total_memory_bytes += int(size) + pad
        self._sizes[addr] = int(size) + pad
        self._raw[addr] = raw
        if like is not None:
            self.buffers[addr] = self._host_base(like)
            self.to_buf(addr)
        else:
            # padding columns / strides are never written by the kernels: start from zeros so that raw
            # dumps (checkpoints, _debug_get_dist) are reproducible whatever the allocator hands back
            _check(self._lib, self._lib.slf_memset(self._ctx, ctypes.c_void_p(raw), 0, int(size) + pad, None),
                   'slf_memset')
        return addr

    def alloc_placed(self, sizes, align_offset=0):
        """Distribution arrays whose physical backing is spread over HBM (sailfish_amd/placement.py): one
        PlacedBuffer per entry of `sizes`, placed together; use their .addr like any device address.  Where the
        virtual-memory calls are not available or the spacers do not fit, the arrays are plain allocations (same
        interface: objects with .addr)."""
        bufs = []
        try:
            for n in sizes:        # one by one: a failure half-way leaves the earlier ones in `bufs` for the clean-up below
                bufs.append(placement.PlacedBuffer(self, n, align_offset))
            info = placement.place(self, bufs)
        except HIPFatalError as e:
            for buf in bufs:
                try:
                    buf.release()
                except HIPFatalError:
                    pass
            self.last_placement = {'fallback': 'plain allocations (%s)' % str(e)[:120]}

            class _Plain(object):
                def __init__(self, addr):
                    self.addr = addr
            return [_Plain(self.alloc_buf(size=n, align_offset=align_offset)) for n in sizes]
        for buf in bufs:
            self._placed[buf.addr] = buf
            self._total_memory_bytes += buf.total
        self.last_placement = info
        for buf in bufs:       # padding columns / strides start from zeros, as in alloc_buf()
            self.memset_buf(buf.va, 0, buf.total)
        return bufs

    def allocated_bytes(self):
        return self._total_memory_bytes

    def free_buf(self, addr):
        if addr in self._placed:
            buf = self._placed.pop(addr)
            self._total_memory_bytes -= buf.total
            buf.release()
            return
        raw = self._raw.pop(addr, addr)
        _check(self._lib, self._lib.slf_free(self._ctx, ctypes.c_void_p(raw)), 'slf_free')
        self.buffers.pop(addr, None)
        self._total_memory_bytes -= self._sizes.pop(addr, 0)

    @staticmethod
    def dist_align_offset(itemsize, envelope=1):
        """Byte offset that puts node index `envelope` (the first real x of row 0) on a 128-byte line."""
        return (128 - envelope * itemsize) % 128

    def alloc_async_host_buf(self, shape, dtype):
        """Page-locked host array (reference backend_cuda.py:156-159)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = ctypes.c_void_p()
        _check(self._lib, self._lib.slf_host_alloc_pinned(max(1, n), ctypes.byref(ptr)), 'slf_host_alloc_pinned')
        self._pinned.append(ptr.value)      # freed by close(); the array must not be used after that
        buf = (ctypes.c_char * max(1, n)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        arr[...] = 0
        return arr

    def _resolve(self, buf, other):
        if other is None:
            if buf not in self.buffers:
                raise ValueError('Unknown compute buffer and source/target not specified.')
            return self.buffers[buf]
        return self._host_base(other)

    def _segments(self, addr, nbytes):
        """[(device address, offset, bytes)] covering [addr, addr + nbytes) without crossing the boundary between two
        physical chunks of a placed buffer: the runtime's copy / fill calls work per allocation."""
        for pb in self._placed.values():
            if pb.va <= addr < pb.va + pb.total:
                out, a, end = [], int(addr), int(addr) + int(nbytes)
                while a < end:
                    stop = min(end, pb.va + ((a - pb.va) // pb.part_bytes + 1) * pb.part_bytes)
                    out.append((a, a - int(addr), stop - a))
                    a = stop
                return out
        return [(int(addr), 0, int(nbytes))]

    def to_buf(self, buf, source=None):
        host = self._resolve(buf, source)
        for a, off, n in self._segments(buf, host.nbytes):
            _check(self._lib, self._lib.slf_memcpy_h2d(self._ctx, ctypes.c_void_p(a), host.ctypes.data + off, n),
                   'slf_memcpy_h2d')

    def from_buf(self, buf, target=None):
        host = self._resolve(buf, target)
        for a, off, n in self._segments(buf, host.nbytes):
            _check(self._lib, self._lib.slf_memcpy_d2h(self._ctx, host.ctypes.data + off, ctypes.c_void_p(a), n),
                   'slf_memcpy_d2h')

    def to_buf_async(self, buf, stream=None):
        host = self.buffers[buf]
        _check(self._lib, self._lib.slf_memcpy_h2d_async(self._ctx, ctypes.c_void_p(buf), host.ctypes.data,
                                                        host.nbytes, stream.handle if stream else None),
               'slf_memcpy_h2d_async')

    def from_buf_async(self, buf, stream=None):
        host = self.buffers[buf]
        _check(self._lib, self._lib.slf_memcpy_d2h_async(self._ctx, host.ctypes.data, ctypes.c_void_p(buf),
                                                        host.nbytes, stream.handle if stream else None),
               'slf_memcpy_d2h_async')

    def copy_buf_async(self, dst, src, nbytes, stream=None):
        _check(self._lib, self._lib.slf_memcpy_d2d_async(self._ctx, ctypes.c_void_p(dst), ctypes.c_void_p(src),
                                                        int(nbytes), stream.handle if stream else None),
               'slf_memcpy_d2d_async')

    def copy_peer_async(self, dst, dst_device, src, src_device, nbytes, stream=None):
        _check(self._lib, self._lib.slf_memcpy_peer_async(self._ctx, ctypes.c_void_p(dst), int(dst_device),
                                                         ctypes.c_void_p(src), int(src_device), int(nbytes),
                                                         stream.handle if stream else None),
               'slf_memcpy_peer_async')

    def memset_buf(self, buf, value, nbytes, stream=None):
        for a, off, n in self._segments(buf, nbytes):
            _check(self._lib, self._lib.slf_memset(self._ctx, ctypes.c_void_p(a), int(value), n,
                                                  stream.handle if stream else None), 'slf_memset')

    # -- modules / kernels --------------------------------------------------
    def build(self, source):
        """``source`` is a hipabi.SlfModuleDesc (see module docstring)."""
        if not isinstance(source, hipabi.SlfModuleDesc):
            raise TypeError('the HIP backend builds from a module descriptor, not from source text')
        return HIPModule(self, source)

    def get_kernel(self, prog, name, block, args, args_format, shared=0, needs_iteration=False,
                   more_shared=False):
        """Same arguments as the reference (backend_cuda.py:220-251); ``block`` and
        ``shared`` are accepted for compatibility and ignored."""
        kern = HIPKernel(self._lib, prog, name)
        kern.set_args(args, args_format, needs_iteration)
        return kern

    def set_iteration(self, it):
        """The iteration number the AA kernels see (reference backend_cuda.py:128-130 rewrites the trailing kernel
        argument of every registered kernel); applied when a kernel that takes it is launched."""
        self._iteration = int(it) & 0xFFFFFFFF

    def run_kernel(self, kernel, grid_size=None, stream=None):
        region = None
        if grid_size is not None:
            y0, y1, z0, z1 = grid_size
            region = ctypes.byref(hipabi.SlfRegion(int(y0), int(y1), int(z0), int(z1)))
        if kernel.needs_iteration:
            self._lib.slf_kernel_set_iteration(kernel.handle, self._iteration)
        _check(self._lib, self._lib.slf_kernel_launch(kernel.handle, region, stream.handle if stream else None),
               'slf_kernel_launch(%s)' % kernel.name)

    def classify_rows(self, module, gpu_map, stream=None):
        """Row classes of the node map at device address `gpu_map` (C ABI slf_module_classify_rows): waves whose 64 nodes
        are plain fluid skip the map, rows without boundary-condition nodes run the small instantiation.  Returns
        {'rows', 'bc_rows', 'segments', 'fluid_segments'}; gpu_map = 0 drops the tables."""
        out = (ctypes.c_int32 * 4)()
        _check(self._lib, self._lib.slf_module_classify_rows(module.handle, ctypes.c_void_p(gpu_map or None),
                                                             stream.handle if stream else None, ctypes.byref(out)),
               'slf_module_classify_rows')
        return {'rows': out[0], 'bc_rows': out[1], 'segments': out[2], 'fluid_segments': out[3]}

    def update_node_params(self, module, first, values, stream=None):
        """Entries first .. of the module's boundary-condition parameter table take `values` for the launches enqueued on
        `stream` from now on (time-dependent boundary values: C ABI slf_module_update_node_params)."""
        vals = np.ascontiguousarray(values, dtype=np.float64)
        _check(self._lib, self._lib.slf_module_update_node_params(module.handle, int(first), vals.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                                  int(vals.size), stream.handle if stream else None),
               'slf_module_update_node_params')

    def set_body_force(self, module, accel, lattice=0):
        """The acceleration the sweeps launched from now on apply (time-dependent body forces)."""
        a = (ctypes.c_double * 3)(*([float(x) for x in accel] + [0.0] * (3 - len(accel))))
        _check(self._lib, self._lib.slf_module_set_body_force(module.handle, int(lattice), a), 'slf_module_set_body_force')

    def set_x_ghost_unused(self, module, low, high):
        """Nothing reads the ghost column x = 0 (low) / x = nx + 1 (high): the sweeps stop storing into it."""
        _check(self._lib, self._lib.slf_module_set_x_ghost_unused(module.handle, int(bool(low)), int(bool(high))),
               'slf_module_set_x_ghost_unused')

    @staticmethod
    def supports_row_classes(desc):
        """Modules the row classes exist for: D3Q19 single-fluid with a node map, direct addressing."""
        return (desc.lattice == hipabi.SLF_D3Q19 and not desc.fluid_only and not int(desc.node_addressing) and
                not int(desc.simtype) and int(desc.incompressible) != hipabi.SLF_DENSITY_ROUNDOFF and
                not int(desc.regularized) and not int(desc.subgrid))      # (those run the per-node kernels)

    # -- streams / events -----------------------------------------------------
    def poll_invalid(self, module, stream):
        """(x, y, z) of a wet node with a non-finite density seen by a sweep since the last poll, or None
        (on-GPU invalid value check; waits for `stream`)."""
        out = (ctypes.c_int32 * 4)()
        _check(self._lib, self._lib.slf_module_poll_invalid(module.handle, stream.handle if stream else None,
                                                          ctypes.byref(out)), 'slf_module_poll_invalid')
        return (out[1], out[2], out[3]) if out[0] else None

    def capture_graph(self, stream, enqueue):
        """Records everything `enqueue()` puts on `stream` (nothing executes) and returns a HIPGraph."""
        # no garbage collection while recording: finalisers of dead HIP objects (events, modules, buffers of
        # earlier simulations) would issue runtime calls in the middle of the capture
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            _check(self._lib, self._lib.slf_graph_capture_begin(stream.handle), 'slf_graph_capture_begin')
            try:
                enqueue()
            finally:
                h = ctypes.c_void_p()
                rc = self._lib.slf_graph_capture_end(stream.handle, ctypes.byref(h))
        finally:
            if gc_was_on:
                gc.enable()
        _check(self._lib, rc, 'slf_graph_capture_end')
        return HIPGraph(self, h)

    def make_plan(self):
        """An empty step plan (HIPPlan)."""
        return HIPPlan(self)

    def set_xface_buffers(self, module, send_low, send_high, recv_low, recv_high):
        """x-face buffers the CollideAndPropagate kernels of `module` use from now on (0 / None: face not connected)."""
        _check(self._lib, self._lib.slf_module_set_xface_buffers(module.handle, *[ctypes.c_void_p(a or None) for a in
                                                                                   (send_low, send_high, recv_low, recv_high)]),
               'slf_module_set_xface_buffers')

    supports_xface_planes = True      # slf_module_set_xface_planes: binary Shan-Chen over connected x faces

    def set_xface_planes(self, module, which, send_low, send_high, recv_low, recv_high):
        """x-face planes of a binary Shan-Chen module: which = 0 / 1 populations of lattice 0 / 1, 2 = densities."""
        _check(self._lib, self._lib.slf_module_set_xface_planes(module.handle, int(which),
                                                                *[ctypes.c_void_p(a or None) for a in
                                                                  (send_low, send_high, recv_low, recv_high)]),
               'slf_module_set_xface_planes')

    supports_step_plans = True

    def make_stream(self, high_priority=False):
        """high_priority (no counterpart in backend_cuda.py:291-296): for the halo stream, see slf_api.hip."""
        return HIPStream(self, high_priority)

    def make_event(self, stream, timing=False):
        """Creates an event *and records it* on `stream` (reference backend_cuda.py:298-305)."""
        ev = HIPEvent(self, timing)
        ev.record(stream)
        return ev

    def sync(self):
        _check(self._lib, self._lib.slf_ctx_sync(self._ctx), 'slf_ctx_sync')

    def sync_stream(self, *streams):
        for s in streams:
            s.synchronize()


backend = HIPBackend
