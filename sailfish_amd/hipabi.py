"""ctypes binding of include/sailfish_hip.h (the C ABI of libsailfish_hip.so).

There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
import ctypes
import os

from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint32,
                    c_void_p)

SLF_MAX_NODE_TYPES = 16
SLF_MAX_Q = 27
SLF_D2Q9, SLF_D3Q19 = 0, 1
SLF_BGK, SLF_MRT = 0, 1
SLF_AB, SLF_AA = 0, 1
SLF_SIM_LBM, SLF_SIM_SHAN_CHEN_BINARY, SLF_SIM_SHAN_CHEN_SINGLE = 0, 1, 2
(SLF_NK_FLUID, SLF_NK_GHOST, SLF_NK_UNUSED, SLF_NK_PROPAGATION_ONLY, SLF_NK_FULL_BB, SLF_NK_HALF_BB,
 SLF_NK_REGULARIZED_VELOCITY, SLF_NK_EQUILIBRIUM_DENSITY, SLF_NK_EQUILIBRIUM_VELOCITY, SLF_NK_ZOUHE_VELOCITY,
 SLF_NK_ZOUHE_DENSITY, SLF_NK_REGULARIZED_DENSITY, SLF_NK_COPY, SLF_NK_YU_OUTFLOW, SLF_NK_DO_NOTHING,
 SLF_NK_SLIP) = range(16)

# SLF_LIBRARY: another build of the library (A/B runs of kernel changes on the same GPU box)
LIB_PATH = os.environ.get('SLF_LIBRARY') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib',
                                                         'libsailfish_hip.so')


class SlfModuleDesc(Structure):
    """Mirror of slf_module_desc."""
    _fields_ = [
        ('struct_size', c_uint32),
        ('lattice', c_int32), ('model', c_int32), ('precision', c_int32), ('access_pattern', c_int32),
        ('lat_nx', c_int32), ('lat_ny', c_int32), ('lat_nz', c_int32),
        ('arr_nx', c_int32), ('arr_ny', c_int32), ('arr_nz', c_int32),
        ('envelope', c_int32),
        ('periodic_fused', c_int32 * 3),
        ('incompressible', c_int32), ('relaxation_enabled', c_int32), ('has_force', c_int32),
        ('fluid_only', c_int32),
        ('tau', c_double), ('visc', c_double),
        ('accel', c_double * 3),
        ('mrt_rates', c_double * SLF_MAX_Q),
        ('nt_type_mask', c_uint32), ('nt_misc_shift', c_uint32), ('nt_param_shift', c_uint32),
        ('nt_scratch_shift', c_uint32),
        ('n_types', c_int32),
        ('type_kind', c_int32 * SLF_MAX_NODE_TYPES),
        ('use_link_tags', c_int32),
        ('n_node_params', c_int32),
        ('node_params', POINTER(c_double)),
        ('dist_stride', ctypes.c_uint64),
        ('periodic_local', c_int32 * 3),
        ('simtype', c_int32),
        ('tau_phi', c_double),
        ('sc_G', c_double * 4),
        ('sc_potential', c_int32),
        ('node_addressing', c_int32),
        ('accel1', c_double * 3),
        ('force_implementation', c_int32),
        ('sparse_geometry', c_int32),
        ('regularized', c_int32),
        ('subgrid', c_int32),
        ('smagorinsky_const', c_double),
    ]


def dist_stride(desc):
    """Elements between consecutive direction arrays of a distribution buffer."""
    return int(desc.dist_stride) or desc.arr_nx * desc.arr_ny * desc.arr_nz


class SlfCommOp(Structure):
    """Mirror of slf_comm_op."""
    _fields_ = [('kind', c_int32), ('peer', c_int32), ('dptr', c_void_p), ('count', ctypes.c_uint64),
                ('elem_bytes', c_int32), ('reserved', c_int32)]


class SlfRegion(Structure):
    _fields_ = [('y0', c_int32), ('y1', c_int32), ('z0', c_int32), ('z1', c_int32)]


# name -> (restype, argtypes); every symbol include/sailfish_hip.h declares.
SIGNATURES = {
    'slf_abi_version': (c_int, []),
    'slf_device_count': (c_int, [POINTER(c_int)]),
    'slf_device_pci_bus_id': (c_int, [c_int, c_char_p, c_size_t]),
    'slf_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
    'slf_ctx_destroy': (c_int, [c_void_p]),
    'slf_ctx_sync': (c_int, [c_void_p]),
    'slf_ctx_info': (c_int, [c_void_p, c_char_p, c_size_t, POINTER(c_size_t), POINTER(c_int), POINTER(c_int)]),
    'slf_ctx_free_memory': (c_int, [c_void_p, POINTER(c_size_t)]),
    'slf_malloc': (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    'slf_free': (c_int, [c_void_p, c_void_p]),
    'slf_memset': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'slf_vmm_granularity': (c_int, [c_void_p, POINTER(c_size_t)]),
    'slf_vmm_reserve': (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    'slf_vmm_release_range': (c_int, [c_void_p, c_void_p, c_size_t]),
    'slf_vmm_chunk_create': (c_int, [c_void_p, c_size_t, POINTER(ctypes.c_uint64)]),
    'slf_vmm_chunk_release': (c_int, [c_void_p, ctypes.c_uint64]),
    'slf_vmm_map': (c_int, [c_void_p, c_void_p, c_size_t, ctypes.c_uint64]),
    'slf_vmm_unmap': (c_int, [c_void_p, c_void_p, c_size_t]),
    'slf_comm_unique_id': (c_int, [c_void_p]),
    'slf_comm_init': (c_int, [c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    'slf_comm_destroy': (c_int, [c_void_p]),
    'slf_comm_count': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    'slf_comm_group_begin': (c_int, []),
    'slf_comm_group_end': (c_int, []),
    'slf_comm_sendrecv': (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p]),
    'slf_comm_exchange': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'slf_host_alloc_pinned': (c_int, [c_size_t, POINTER(c_void_p)]),
    'slf_host_free': (c_int, [c_void_p]),
    'slf_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'slf_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'slf_memcpy_h2d_async': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'slf_memcpy_d2h_async': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'slf_memcpy_d2d_async': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'slf_memcpy_peer_async': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_size_t, c_void_p]),
    'slf_stream_create': (c_int, [c_void_p, POINTER(c_void_p)]),
    'slf_stream_create_high_priority': (c_int, [c_void_p, POINTER(c_void_p)]),
    'slf_stream_destroy': (c_int, [c_void_p]),
    'slf_stream_sync': (c_int, [c_void_p]),
    'slf_stream_native': (c_int, [c_void_p, POINTER(c_void_p)]),
    'slf_stream_wait_event': (c_int, [c_void_p, c_void_p]),
    'slf_event_create': (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    'slf_event_destroy': (c_int, [c_void_p]),
    'slf_event_record': (c_int, [c_void_p, c_void_p]),
    'slf_event_sync': (c_int, [c_void_p]),
    'slf_event_elapsed_ms': (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    'slf_module_create': (c_int, [c_void_p, POINTER(SlfModuleDesc), POINTER(c_void_p)]),
    'slf_module_destroy': (c_int, [c_void_p]),
    'slf_module_set_xface_buffers': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'slf_module_set_xface_planes': (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'slf_module_set_x_ghost_unused': (c_int, [c_void_p, c_int, c_int]),
    'slf_module_block_size': (c_int, [c_void_p, POINTER(c_int)]),
    'slf_kernel_get': (c_int, [c_void_p, c_char_p, POINTER(c_void_p)]),
    'slf_kernel_destroy': (c_int, [c_void_p]),
    'slf_kernel_set_args': (c_int, [c_void_p, c_char_p, POINTER(c_void_p), c_int, c_int]),
    'slf_kernel_set_iteration': (c_int, [c_void_p, c_uint32]),
    'slf_kernel_launch': (c_int, [c_void_p, POINTER(SlfRegion), c_void_p]),
    'slf_module_poll_invalid': (c_int, [c_void_p, c_void_p, POINTER(c_int32 * 4)]),
    'slf_module_classify_rows': (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int32 * 4)]),
    'slf_module_update_node_params': (c_int, [c_void_p, c_int, POINTER(c_double), c_int, c_void_p]),
    'slf_module_set_body_force': (c_int, [c_void_p, c_int, POINTER(c_double)]),
    'slf_graph_capture_begin': (c_int, [c_void_p]),
    'slf_graph_capture_end': (c_int, [c_void_p, POINTER(c_void_p)]),
    'slf_graph_launch': (c_int, [c_void_p, c_void_p]),
    'slf_graph_destroy': (c_int, [c_void_p]),
    'slf_plan_create': (c_int, [c_void_p, POINTER(c_void_p)]),
    'slf_plan_destroy': (c_int, [c_void_p]),
    'slf_plan_size': (c_int, [c_void_p, POINTER(c_int)]),
    'slf_plan_add_launch': (c_int, [c_void_p, c_void_p, POINTER(SlfRegion), c_void_p]),
    'slf_plan_add_record': (c_int, [c_void_p, c_void_p, c_void_p]),
    'slf_plan_add_wait': (c_int, [c_void_p, c_void_p, c_void_p]),
    'slf_plan_add_exchange': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'slf_plan_add_memset': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'slf_plan_add_copy': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'slf_plan_add_xface_buffers': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'slf_plan_add_xface_planes': (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'slf_plan_add_peer_signal': (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_int, c_int, c_void_p]),
    'slf_plan_add_peer_wait': (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_int, c_int, c_int, c_void_p]),
    'slf_plan_run': (c_int, [c_void_p, c_uint32]),
    'slf_peer_create': (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    'slf_peer_destroy': (c_int, [c_void_p]),
    'slf_peer_flags_handle': (c_int, [c_void_p, c_void_p]),
    'slf_peer_connect': (c_int, [c_void_p, c_int, c_void_p]),
    'slf_peer_alloc': (c_int, [c_void_p, c_size_t, POINTER(c_void_p), c_void_p]),
    'slf_peer_free': (c_int, [c_void_p, c_void_p]),
    'slf_peer_open': (c_int, [c_void_p, c_void_p, POINTER(c_void_p)]),
    'slf_peer_close': (c_int, [c_void_p, c_void_p]),
    'slf_peer_signal': (c_int, [c_void_p, POINTER(c_int32), c_int, c_int, c_void_p]),
    'slf_peer_wait': (c_int, [c_void_p, POINTER(c_int32), c_int, c_int, c_int, c_void_p]),
    'slf_peer_set_timeout': (c_int, [c_void_p, c_double]),
    'slf_peer_status': (c_int, [c_void_p, POINTER(ctypes.c_int64 * 8)]),
    'slf_peer_progress': (c_int, [c_void_p, c_int, c_int, POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64),
                                  POINTER(ctypes.c_uint64)]),
    'slf_peer_selftest_fill': (c_int, [c_void_p, c_void_p, c_size_t, c_uint32, c_int, c_void_p]),
    'slf_peer_selftest_check': (c_int, [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p, POINTER(c_uint32)]),
    'slf_last_error': (c_char_p, []),
}

SLF_ADDR_DIRECT, SLF_ADDR_INDIRECT = 0, 1
SLF_DENSITY_COMPRESSIBLE, SLF_DENSITY_INCOMPRESSIBLE, SLF_DENSITY_ROUNDOFF = 0, 1, 2
SLF_FORCE_GUO, SLF_FORCE_EDM = 0, 1
SLF_SUBGRID_NONE, SLF_SUBGRID_LES_SMAGORINSKY = 0, 1
SLF_INVALID_NODE = 0xffffffff
SLF_PEER_HANDLE_BYTES, SLF_PEER_CHANNELS = 64, 4

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (+ HSA runtime)
    and always load that copy (NEEDED "libamdhip64.so", RPATH $ORIGIN); libsailfish_hip.so asks for the
    soname libamdhip64.so.7 and would pull in /opt/rocm's copy if it is loaded first.  Two runtimes in
    one process cannot share streams (TorchDistConnector hands our halo stream to torch / RCCL) and the
    second one does not even see the GPU.  So, if torch is installed, map its copy first: the dynamic
    loader then resolves our dependency to the already-loaded soname.  torch itself is not imported."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if not os.path.exists(cand):
        return None
    return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load(path=None):
    """Load libsailfish_hip.so and attach the signatures.  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipLibraryMissing(
            'libsailfish_hip.so not found at %s -- run `python -m sailfish_amd.build` '
            '(there is no CPU fallback for the HIP backend)' % p)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.slf_abi_version() != 1:
        raise RuntimeError('libsailfish_hip.so ABI version mismatch')
    if path is None:
        _lib = lib
    return lib


def make_desc(**kw):
    d = SlfModuleDesc()
    d.struct_size = ctypes.sizeof(SlfModuleDesc)
    d.envelope = 1
    d.relaxation_enabled = 1
    d.lat_nz = 1
    d.arr_nz = 1
    keep = []
    for k, v in kw.items():
        if k in ('periodic_fused', 'accel', 'periodic_local', 'sc_G', 'accel1'):
            for i, x in enumerate(v):
                getattr(d, k)[i] = x
        elif k == 'mrt_rates':
            for i, x in enumerate(v):
                d.mrt_rates[i] = float(x)
        elif k == 'type_kind':
            d.n_types = len(v)
            for i, x in enumerate(v):
                d.type_kind[i] = int(x)
        elif k == 'node_params':
            arr = (c_double * max(1, len(v)))(*[float(x) for x in v])
            keep.append(arr)
            d.node_params = ctypes.cast(arr, POINTER(c_double))
            d.n_node_params = len(v)
        else:
            setattr(d, k, v)
    d._keepalive = keep
    return d
