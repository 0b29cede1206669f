"""The RCCL entry points of the C ABI (slf_comm_*): a communicator of one rank on this GPU, send / receive to self on
a stream of the library.  What a host without torch would bind for the device-to-device halo exchange
(include/sailfish_hip.h; reference subdomain_runner.py:1064-1139 moves halos through host memory and zmq)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_sendrecv_to_self():
    from sailfish_amd.backend_hip import HIPBackend, _check

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    lib = b._lib
    uid = ctypes.create_string_buffer(128)
    _check(lib, lib.slf_comm_unique_id(uid), 'slf_comm_unique_id')
    comm = ctypes.c_void_p()
    _check(lib, lib.slf_comm_init(b._ctx, 1, 0, uid, ctypes.byref(comm)), 'slf_comm_init')
    n = 5 * 512 * 512                      # one x-face of the 8-GPU layout
    src = np.arange(n, dtype=np.float32)
    g_src = b.alloc_buf(like=src)
    g_dst = b.alloc_buf(size=n * 4)
    stream = b.make_stream()
    for _ in range(3):
        _check(lib, lib.slf_comm_group_begin(), 'slf_comm_group_begin')
        _check(lib, lib.slf_comm_sendrecv(comm, 0, ctypes.c_void_p(g_src), n, ctypes.c_void_p(g_dst), n, 4, stream.handle),
               'slf_comm_sendrecv')
        _check(lib, lib.slf_comm_group_end(), 'slf_comm_group_end')
    stream.synchronize()
    out = np.zeros(n, dtype=np.float32)
    b.from_buf(g_dst, out)
    assert np.array_equal(out, src)
    _check(lib, lib.slf_comm_destroy(comm), 'slf_comm_destroy')
